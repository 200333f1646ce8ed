"""The C-ABI library loads on a CPU-only box and exports exactly what include/mapx_hip.h
declares (no compute calls here)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "mapx_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mapx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from mapx import native
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(native.lib, n), f"{n} declared in mapx_hip.h but not exported"
        assert n in native.SIGNATURES, f"{n} has no ctypes signature in mapx/native.py"
    assert sorted(native.SIGNATURES) == names, "binding lists symbols the header does not declare"
    assert native.lib.mapx_abi_version() == native.MAPX_ABI_VERSION


def test_host_only_entry_points_work_without_gpu():
    import torch
    from mapx import ops
    probs = torch.tensor([0.5, 0.25, 0.125, 0.125])
    prob, alias = ops.alias_build(probs)
    # table encodes the distribution exactly for dyadic probabilities
    V = 4
    dist = prob.double() / V
    dist.index_add_(0, alias, (1 - prob.double()) / V)
    assert torch.allclose(dist, probs.double())


def test_cpu_tensors_are_rejected_loudly():
    import pytest
    import torch
    from mapx import ops
    from mapx.native import MapxError
    with pytest.raises(MapxError):
        ops.emb_gather(torch.zeros(3, dtype=torch.int64), torch.zeros(4, 16))
