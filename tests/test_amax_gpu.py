"""Magnitude records and the two-piece fp16 arithmetic of the fp32 GEMMs (csrc/amax.h, gemm_h2.hip, gemm_h2w.hip):
what the records hold after every kernel that leaves one, what the arithmetic guarantees away from randn operands,
and that the weight planes reproduce the in-kernel cut bit for bit."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from mapx import ops as _ops
    return _ops


def _cpu(x):
    return x.detach().cpu()


def test_records_of_every_producer_equal_the_true_maximum(ops):
    """gather, GEMM epilogues (plain, bias + ReLU, cross layer, fused backward: dZ columns and t), the elementwise
    backward kernels, the NCE scatter, the narrow head's join kernel: record == max |tensor| exactly (the maximum
    is order-independent), non-finite elements left out."""
    g = torch.Generator().manual_seed(0)
    V, E, n = 500, 16, 4096 * 23
    table = torch.randn(V, E, generator=g).to(DEV) * 3
    ids = torch.randint(0, V, (n,), generator=g).to(DEV)
    out = ops.emb_gather(ids, table)
    assert ops.amax_value(ops.amax_of(out)) == float(out.abs().max())
    M, N, K = 300, 200, 136
    x, w, b = torch.randn(M, K, generator=g).to(DEV), torch.randn(N, K, generator=g).to(DEV), torch.randn(N, generator=g).to(DEV)
    for relu in (False, True):
        y = ops.linear_fwd(x, w, b, relu=relu)
        assert ops.amax_value(ops.amax_of(y)) == float(y.abs().max())
    x0, xi = torch.randn(M, N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV)
    wc, bc = torch.randn(N, N, generator=g).to(DEV) / 14, torch.randn(N, generator=g).to(DEV)
    y, u = ops.cross_layer_fwd(x0, xi, wc, bc)
    assert ops.amax_value(ops.amax_of(y)) == float(y.abs().max()) and ops.amax_of(u) is None
    # two kernels raising ONE record (the towers' concatenated output)
    buf = torch.empty(M, 2 * N, device=DEV)
    ops.tag(buf, ops.amax_record(buf.device))
    ops.cross_layer_fwd(x0, xi, wc, bc, out=ops.alias_cols(buf, 0, N))
    ops.linear_fwd(x, w, b, relu=True, out=ops.alias_cols(buf, N, N))
    assert ops.amax_value(ops.amax_of(buf)) == float(buf.abs().max())
    # fused backward epilogue: dZ right of c0, t left of it
    dy, w2 = torch.randn(M, K, generator=g).to(DEV), torch.randn(K, 2 * N, generator=g).to(DEV)
    mask, u2 = torch.randn(M, 2 * N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV)
    C, t, dx0, part = ops.gemm_bwd_fused(dy, w2, N, mask=mask, x0=x0, u=u2)
    assert ops.amax_value(ops.amax_of(t)) == float(t.abs().max())
    assert ops.amax_value(C._amax_dz) == float(C[:, N:].abs().max())
    dz, _ = ops.relu_mask_colsum(dy, torch.randn(M, K, generator=g).to(DEV))
    assert ops.amax_value(ops.amax_of(dz)) == float(dz.abs().max())
    tt, _, _ = ops.cross_bwd_pre_colsum(x0, xi, u2)
    assert ops.amax_value(ops.amax_of(tt)) == float(tt.abs().max())
    B, L, F, P = 64, 6, 23, 32
    dh = torch.randn(B * L, P, generator=g).to(DEV) * 1e-5
    mi = torch.randint(0, F, (B, L), generator=g).to(DEV)
    denc = ops.nce_scatter_dh(dh, mi, F, P)
    assert ops.amax_value(ops.amax_of(denc)) == float(denc.abs().max())
    # non-finite elements are left out of a record
    bad = torch.randn(64, 64, generator=g)
    bad[3, 5], bad[7, 7] = float("inf"), float("nan")
    fin = bad[torch.isfinite(bad)].abs().max()
    assert ops.amax_value(ops.amax(bad.to(DEV))) == float(fin)
    assert ops.amax_value(ops.amax(torch.zeros(5, 8, device=DEV))) == 0.0


def test_a_later_epoch_outranks_an_earlier_one(ops):
    """A record is never reset inside a captured step: the epoch tag (advanced by mapx_step_advance) makes what the
    next step's kernels write outrank what this step's left, also when it is smaller."""
    x = torch.full((256, 64), 8.0, device=DEV)
    rec = ops.amax(x)
    assert ops.amax_value(rec) == 8.0
    ops.amax(x * 0.25, rec=rec, reset=False)              # same epoch: the maximum stays
    assert ops.amax_value(rec) == 8.0
    done = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.step_advance(done)                                # epoch + 1
    ops.amax(x * 0.25, rec=rec, reset=False)
    assert ops.amax_value(rec) == 2.0 and int(done.item()) == 1


def test_adamw_keeps_the_weights_records(ops):
    """MapxOptimizer: every dense parameter's record == max |p| after a step (written by the AdamW kernel for the next
    epoch), after load_state_dict-style writes (version counter) and after refresh."""
    from mapx.arguments import TrainingArguments
    from mapx.models import BaseModel
    from mapx.optim import MapxOptimizer
    from util import make_config
    torch.manual_seed(0)
    cfg = dict(F=23, V=300, E=16, H=72, NL=2, NC=2, P=32, K=5)
    model = BaseModel.from_config(make_config(cfg, "CTR", np.ones(cfg["V"], np.float32))).to(DEV)
    args = TrainingArguments(output_dir="/tmp/x", learning_rate=1e-2, weight_decay=0.1, lr_sched="const")
    opt = MapxOptimizer(model, args, 100, 0)

    def check():
        for p in opt.dense_params:
            assert ops.amax_value(p._amax) == float(p.detach().abs().max()), p.shape
    check()
    for _ in range(3):
        for p in opt.dense_params:                        # (the slots' padding keeps its zero gradient, as in a step)
            p._mapx_grad.normal_()
        opt.step()
        check()
    w = model.parallel_dnn.dnn["0"].weight
    with torch.no_grad():
        w.mul_(40.0)                                      # written through torch: the version counter moves
    assert ops.amax_value(ops.amax_of(w)) == float(w.detach().abs().max())


def test_gemm_h2_operand_magnitudes(ops):
    """The two-piece fp16 arithmetic away from randn operands (csrc/gemm_h2.hip's header): per-tensor power-of-two
    scales from the records, so tensors of any common magnitude — 2^+-100 — multiply as randn does; inside a tensor,
    elements down to 2^-29 of its maximum keep 22 bits, smaller ones lose relative (not absolute) precision: with
    exponents spread over 2^-30 .. 2^30 in every row the error stays below 1e-6 of sum_k |a_k b_k| (the bound of the
    six-product arithmetic's test); values up to fp32's largest are fine (the six-product kernels' pieces overflow
    bf16 above 3.39e38); an infinity or NaN makes ITS rows non-finite and leaves the others exact."""
    g = torch.Generator().manual_seed(0)
    M, N, K = 256, 128, 512

    def spread(shape, lo, hi):
        mant = 1 + torch.rand(shape, generator=g)
        e = torch.randint(lo, hi + 1, shape, generator=g).float()
        return torch.where(torch.rand(shape, generator=g) < 0.5, -1.0, 1.0) * mant * torch.exp2(e)

    def worst(A, B):
        Ad, Bd = A.to(DEV), B.to(DEV)
        out = _cpu(ops.gemm(Ad, Bd, True, True, M, N, K, amax_a=ops.amax(Ad), amax_b=ops.amax(Bd)))
        ref, mag = A.double() @ B.double().t(), A.abs().double() @ B.abs().double().t()
        return out, float(((out.double() - ref).abs() / mag).max())

    assert worst(torch.randn(M, K, generator=g), torch.randn(N, K, generator=g))[1] <= 2e-7
    assert worst(spread((M, K), -30, 30), spread((N, K), -30, 30))[1] <= 1e-6
    assert worst(spread((M, K), -3, 3) * 2.0 ** 100, spread((N, K), -3, 3) * 2.0 ** -100)[1] <= 2e-7
    assert worst(spread((M, K), -100, -100), spread((N, K), 60, 60))[1] <= 2e-7
    out, err = worst(spread((M, K), 0, 0) * 2.0 ** -125, spread((N, K), 20, 20))       # scale capped at 2^126
    assert bool(torch.isfinite(out).all()) and err <= 2e-7
    # ONE scale per tensor: an outlier up to 2^28 times the rest costs the rest nothing (tools/micro/h2_range_probe.py:
    # the other rows stay below 1.2e-7 up to 2^28, 7e-7 at 2^32, 2e-4 at 2^40; the outlier's own row sees what any fp32
    # accumulator does to small terms added to a large sum) ...
    A = torch.randn(M, K, generator=g)
    A[3, 7] = 2.0 ** 24
    Bm = torch.randn(N, K, generator=g) * 1e-3
    out, err = worst(A, Bm)
    others = torch.ones(M, dtype=torch.bool)
    others[3] = False
    e2 = (out.double() - A.double() @ Bm.double().t()).abs() / (A.abs().double() @ Bm.abs().double().t())
    assert err <= 2e-6 and float(e2[others].max()) <= 2e-7
    # ... beyond that the small elements lose bits, and 2^50 below the maximum they are gone: with an element near
    # fp32's largest (above the largest bf16: the six-product kernels give NaN there) the result is finite, the
    # outlier's row exact, the other rows zero — per-tensor scaling trades this corner for half the MFMAs
    A = torch.randn(M, K, generator=g)
    A[3, 7] = 3.40e38
    Bm = torch.randn(N, K, generator=g) * 1e-3
    out, _ = worst(A, Bm)
    ref3 = A[3].double() @ Bm.double().t()
    assert bool(torch.isfinite(out).all())
    assert bool(((out[3].double() - ref3).abs() <= 1e-6 * (A[3].abs().double() @ Bm.abs().double().t())).all())
    assert float(out[:3].abs().max()) == 0.0 and float(out[4:].abs().max()) == 0.0
    A = torch.randn(M, K, generator=g)
    A[5, 1], A[6, 2] = float("inf"), float("nan")
    B = torch.randn(N, K, generator=g)
    out, _ = worst(A, B)
    assert bool(torch.isnan(out[5]).all()) and bool(torch.isnan(out[6]).all())
    keep = torch.ones(M, dtype=torch.bool)
    keep[5] = keep[6] = False
    ref = A[keep].double() @ B.double().t()
    assert bool(((out[keep].double() - ref).abs() <= 2e-7 * (A[keep].abs().double() @ B.abs().double().t())).all())


@pytest.mark.parametrize("b_kc,M,N,K", [(1, 4096, 1000, 1024), (0, 4096, 1000, 1000), (1, 4100, 736, 368), (0, 8192, 1368, 200)])
def test_eight_wave_weight_planes_kernel_equals_the_four_wave_one(ops, monkeypatch, b_kc, M, N, K):
    """gemm_f32h2w8_kernel (64 x 256 tiles by eight waves, opt-in: MAPX_GEMM_H2W8=1) and the 4-wave kernel with the
    four-buffer LDS pipeline (opt-in: MAPX_GEMM_H2W_DEEP=1) form the same sums in the same order as the 4-wave kernel: bit-identical C through the plain, the bias + ReLU, the ReLU-mask + column-sum and the
    fused-backward epilogues; its partial rows (one per 64 rows) and the 4-wave kernel's (sum + zero row per 128)
    hold the same column sums grouped differently."""
    from mapx.native import EPI_BIAS_RELU, EPI_RELU_MASK_COLSUM
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(DEV) * 0.3
    B = (torch.randn((N, K) if b_kc else (K, N), generator=g) / math.sqrt(K)).to(DEV)
    bias, y = torch.randn(N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV)
    ra, rb = ops.amax(A), ops.amax(B)
    pl = ops.h2_weight_planes(B, b_kc, rb)
    kw = dict(amax_a=ra, amax_b=rb, b_planes=pl)
    c0 = 368 if N > 368 else 0
    x0, u = torch.randn(M, max(c0, 4), generator=g).to(DEV), torch.randn(M, max(c0, 4), generator=g).to(DEV)
    out = {}
    for arm in ("0", "1", "deep"):          # 4-wave kernel, 8-wave kernel, 4-wave kernel with the four-buffer pipeline
        monkeypatch.setenv("MAPX_GEMM_H2W8", "1" if arm == "1" else "0")
        monkeypatch.setenv("MAPX_GEMM_H2W_DEEP", "1" if arm == "deep" else "0")
        part = torch.full((ops.part_rows(M), N), 7.0, device=DEV)
        res = [ops.gemm(A, B, True, bool(b_kc), M, N, K, **kw),
               ops.gemm(A, B, True, bool(b_kc), M, N, K, epi=EPI_BIAS_RELU, bias=bias, **kw),
               ops.gemm(A, B, True, bool(b_kc), M, N, K, epi=EPI_RELU_MASK_COLSUM, aux1=y, out2=part, **kw), part]
        if not b_kc:
            monkeypatch.setattr(ops, "AUTO_AMAX", True)          # (records and planes computed by the wrapper)
            C, t, dx0, p2 = ops.gemm_bwd_fused(A, B, c0, mask=y, x0=x0 if c0 else None, u=u if c0 else None, plus_v=True)
            res += [C, t if c0 else C, dx0 if c0 else C, p2]
        out[arm] = res
    for k, (a, b) in enumerate(zip(out["0"], out["deep"])):
        assert torch.equal(a, b), k
    for k, (a, b) in enumerate(zip(out["0"], out["1"])):
        if a.shape[0] == ops.part_rows(M) and a.shape[0] != M:          # partial rows
            assert bool((a[1::2] == 0).all()) and not bool((b[1::2] == 0).all())
            sa, sb = a.double().sum(0), b.double().sum(0)
            assert bool(((sa - sb).abs() <= 1e-5 * sa.abs() + 1e-4).all()), k
        else:
            assert torch.equal(a, b), k
    ref = A.double() @ (B.double().t() if b_kc else B.double())
    assert bool(((out["1"][0].double() - ref).abs() <= 2e-6 * (A.abs().double() @ (B.abs().double().t() if b_kc else B.abs().double())) + 1e-6).all())


@pytest.mark.parametrize("b_kc,M,N,K", [(1, 4096, 1000, 1024), (0, 4096, 1000, 736), (1, 4096, 1000, 1000), (1, 4096, 1000, 368),
                                        (0, 4097, 1368, 200), (1, 16384, 130, 72),
                                        (1, 4096, 368, 736), (0, 4096, 368, 1000), (1, 4100, 368, 368)])        # 128 x 64 tiles
def test_weight_planes_products_equal_the_in_kernel_cut(ops, b_kc, M, N, K):
    """gemm_h2w.hip (operand B read from mapx_h2_weight_planes' fragment-ordered fp16 pieces) forms the very
    products gemm_h2.hip forms from the fp32 weight: same pieces, same scale, the same sums per k16 block in the same
    order — bit-identical results when K is whole K-steps (with a remainder gemm_h2.hip takes the partial K-step
    first, this kernel last: the same terms grouped differently, compared against fp64 instead) — on whole tiles,
    edge tiles (N = 1000, 1368, 130; M = 4097) and K remainders."""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(DEV) * 0.3
    B = (torch.randn((N, K) if b_kc else (K, N), generator=g) / math.sqrt(K)).to(DEV)
    ra, rb = ops.amax(A), ops.amax(B)
    assert ops.planes_wanted(M, N, K)
    pl = ops.h2_weight_planes(B, b_kc, rb)
    bias = torch.randn(N, generator=g).to(DEV)
    from mapx.native import EPI_BIAS_RELU
    got = ops.gemm(A, B, True, bool(b_kc), M, N, K, amax_a=ra, amax_b=rb, b_planes=pl, epi=EPI_BIAS_RELU, bias=bias)
    wide = math.ceil(M / 128) * math.ceil(N / 128) >= 128          # else the 128 x 64 form of both kernels
    want = ops.gemm(A, B, True, bool(b_kc), M, N, K, amax_a=ra, amax_b=rb, tile=3 if wide else 1, epi=EPI_BIAS_RELU, bias=bias)
    ref = torch.relu(A.double() @ (B.double().t() if b_kc else B.double()) + bias.double())
    bound = 2e-6 * (A.abs().double() @ (B.abs().double().t() if b_kc else B.abs().double())) + 1e-6
    assert bool(((got.double() - ref).abs() <= bound).all())
    assert bool(((want.double() - ref).abs() <= bound).all())
    if K % 32 == 0:
        assert torch.equal(got, want)
    # a column slice of a wider weight (the heads' input gradient per tower)
    if not b_kc:
        wide = torch.randn(K, N + 40, generator=g).to(DEV)
        sl = wide[:, 40:]
        rs = ops.amax(sl.contiguous())
        pl2 = ops.h2_weight_planes(sl, False, rs)
        g2 = ops.gemm(A, sl, True, False, M, N, K, amax_a=ra, amax_b=rs, b_planes=pl2)
        ref2 = A.double() @ sl.double()
        assert bool(((g2.double() - ref2).abs() <= 2e-6 * (A.abs().double() @ sl.abs().double()) + 1e-6).all())
