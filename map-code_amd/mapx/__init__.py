"""mapx: MI355X-native DCNv2 + MFP/RFD pretraining hot path of MAP (CHIANGEL/MAP-CODE).

Host-side mirror of the reference's module surface over hand-written gfx950 kernels
(csrc/*.hip, C ABI in include/mapx_hip.h).  No CPU fallback: importing `mapx.native`
without the built library raises, and every op requires device tensors.
"""
__version__ = "0.1.0"

import os as _os

# The step graphs are laid out for the runtime's default of 4 hardware queues per process: with 6
# the same capture replays in 2.2 ms instead of 1.22 (more branches run at once and the
# one-block-per-CU GEMMs lose CUs to them; 2-4 measure the same).  Pin the default unless the
# user chose otherwise; read by the HIP runtime when it initialises, i.e. after this import.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
