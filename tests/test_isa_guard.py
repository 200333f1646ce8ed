"""tools/isa_guard.py: no shipped kernel holds a packed VALU instruction whose op_sel takes the high
half of a register that an LDS read delivered — the instruction form behind round 3's timing-dependent wrong result
in skinny_dw_tall_kernel.  CPU test: reads the device code of the objects the library was linked from."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_guard  # noqa: E402

BAD = """
0000000000001000 <_ZN4mapx9some_kernelEv>:
	ds_read_b128 v[80:83], v12                                 // 000000001000: D9FE0000 50000000
	s_waitcnt lgkmcnt(0)                                       // 000000001008: BF8CC07F
	v_pk_fma_f32 v[4:5], v[20:21], v[80:81], v[4:5] op_sel:[0,1,0] op_sel_hi:[1,1,1] // 00000000100C: D3B04004 1C12A114
	v_mov_b32_e32 v81, v3                                      // 000000001014: 7EA20303
	v_pk_fma_f32 v[6:7], v[20:21], v[80:81], v[6:7] op_sel:[0,1,0] op_sel_hi:[1,1,1] // 000000001018: D3B04006 1C1AA114
	v_pk_mul_f32 v[8:9], v[82:83], v[30:31] op_sel_hi:[1,1] // 000000001020: D3B14008 18023D52
"""


def test_the_detector_fires_on_the_round_3_form_and_only_on_it():
    found = isa_guard.scan(BAD, "synthetic")
    assert len(found) == 1                       # the first v_pk_fma: v81 came from the LDS read; the second: v81 rewritten
    where, func, addr, ins, pos, reg = found[0]
    assert reg == "v81" and pos == 1 and "some_kernel" in func and ins.startswith("v_pk_fma_f32 v[4:5]")
    assert isa_guard.regs("v[4:7]") == ["v4", "v5", "v6", "v7"] and isa_guard.regs("v9") == ["v9"] and isa_guard.regs("s3") == []


def test_no_shipped_kernel_has_the_form():
    build = os.path.join(ROOT, "map-code_amd", "csrc", "build")
    if not (os.path.isdir(build) and any(f.endswith(".o") for f in os.listdir(build))) or not os.path.exists(isa_guard.OBJDUMP):
        pytest.skip("library objects not built here")
    found, scanned = isa_guard.guard([build])
    assert scanned >= 10, scanned
    assert not found, "\n".join(f"{w}: {f[:60]} `{i}` ({r})" for w, f, a, i, p, r in found)
