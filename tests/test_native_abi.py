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


def test_every_binding_has_the_argument_count_and_kinds_of_its_prototype():
    """A ctypes signature one argument short of the C prototype is a segmentation fault at the first call (round 4: a
    trailing pointer added to mapx_emb_gather_fwd).  Every prototype of the header is parsed and compared with the
    binding table: number of parameters, and per parameter pointer / 64-bit integer / size / int / float / double."""
    import ctypes as C
    from mapx import native
    text = open(os.path.join(ROOT, "include", "mapx_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = re.findall(r"\b(?:int|size_t|void|const char\*)\s+(mapx_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S)
    assert len(protos) >= 90

    def kind(param):
        param = " ".join(param.split())
        if "*" in param or "hipStream_t" in param:
            return C.c_void_p
        base = param.rsplit(" ", 1)[0].replace("const ", "").strip()
        return {"int": C.c_int, "int64_t": C.c_int64, "uint64_t": C.c_uint64, "size_t": C.c_size_t, "float": C.c_float,
                "double": C.c_double, "int32_t": C.c_int32, "unsigned": C.c_uint}.get(base, None)

    for name, params in protos:
        want = [] if params.strip() in ("", "void") else [kind(x) for x in params.split(",")]
        got = list(native.SIGNATURES[name][1])
        assert len(got) == len(want), f"{name}: binding has {len(got)} arguments, the prototype {len(want)}"
        for i, (g, w) in enumerate(zip(got, want)):
            assert w is not None, f"{name}: parameter {i} of the prototype not understood"
            assert C.sizeof(g) == C.sizeof(w) and (g in (C.c_float, C.c_double)) == (w in (C.c_float, C.c_double)), \
                f"{name}: argument {i} is {g.__name__} in the binding, {w.__name__} in the prototype"


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
