"""mapx: MI355X-native DCNv2 + MFP/RFD pretraining hot path of MAP (CHIANGEL/MAP-CODE).

Host-side mirror of the reference's module surface over hand-written gfx950 kernels
(csrc/*.hip, C ABI in include/mapx_hip.h).  No CPU fallback: importing `mapx.native`
without the built library raises, and every op requires device tensors.
"""
__version__ = "0.1.0"
