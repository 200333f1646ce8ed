"""Kernel-level parity: every C-ABI entry point against the oracle / plain torch-CPU fp32
on the same seeded inputs.  Bit-exact for index work; fp32 tolerance written per test."""
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


# --------------------------------------------------------------------------- gather
@pytest.mark.parametrize("n,E", [(0, 16), (1, 16), (7 * 23, 16), (4096 * 23, 16), (33, 8), (5, 6)])
def test_emb_gather_bit_exact(ops, n, E):
    g = torch.Generator().manual_seed(n + E)
    V = 1000
    table = torch.randn(V, E, generator=g)
    ids = torch.randint(0, V, (n,), generator=g)
    out = ops.emb_gather(ids.to(DEV), table.to(DEV), validate=True)
    assert torch.equal(_cpu(out), table[ids])


def test_emb_gather_out_of_range_raises(ops):
    table = torch.randn(10, 16, device=DEV)
    with pytest.raises(IndexError):
        ops.emb_gather(torch.tensor([1, 10], device=DEV), table, validate=True)
    with pytest.raises(IndexError):
        ops.emb_gather(torch.tensor([-1], device=DEV), table, validate=True)


# --------------------------------------------------------------------------- seg plan / reduce
def _skewed_keys(n, V, seed):
    g = torch.Generator().manual_seed(seed)
    k = (torch.rand(n, generator=g) ** 6 * V).long().clamp_(0, V - 1)
    k[: n // 4] = 3                       # a '<mask>'-like hot key spanning many chunks
    return k[torch.randperm(n, generator=g)]


@pytest.mark.parametrize("n,V,W", [(1, 50, 16), (31, 50, 16), (32, 50, 16), (33, 7, 16),
                                   (5000, 1000, 16), (94208, 9449445, 16), (70001, 300, 32),
                                   (4097, 5, 64), (2000, 100, 4), (131072, 70000, 4), (131073, 70000, 4),
                                   (638976, 9449445, 4)])
def test_seg_plan_and_reduce_rows(ops, n, V, W):
    """The plan must be THE stable sort of the keys (perm == torch's stable argsort), with runs,
    ranks and unique keys consistent with it."""
    keys = _skewed_keys(n, V, n)
    src = torch.randn(n, W, generator=torch.Generator().manual_seed(1))
    plan = ops.SegPlan(keys.to(torch.int32).to(DEV), V)
    U = plan.count()
    uniq_ref, inv = torch.unique(keys, return_inverse=True)
    assert U == uniq_ref.numel()
    assert torch.equal(_cpu(plan.uniq[:U]).long(), uniq_ref)
    sk = _cpu(plan.sorted_keys[:n]).long()
    assert torch.equal(sk, keys[_cpu(plan.perm[:n]).long()]) and bool((sk[1:] >= sk[:-1]).all())
    assert torch.equal(_cpu(plan.perm[:n]).long(), torch.sort(keys, stable=True).indices)
    starts = _cpu(plan.seg_start[:U + 1]).long()
    assert int(starts[U]) == n and torch.equal(sk[starts[:U]], uniq_ref)
    assert torch.equal(_cpu(plan.rank[:n]).long() - 1, inv[_cpu(plan.perm[:n]).long()])
    out = ops.seg_reduce_rows(plan, src.to(DEV), W)
    ref = torch.zeros(U, W, dtype=torch.float64).index_add_(0, inv, src.double())
    scale = torch.zeros(U, W, dtype=torch.float64).index_add_(0, inv, src.double().abs())
    err = (_cpu(out[:U]).double() - ref).abs()
    assert bool((err <= 1e-6 * scale + 1e-6).all()), float(err.max())
    # bit-reproducible: same inputs -> same bits
    out2 = ops.seg_reduce_rows(plan, src.to(DEV), W)
    assert torch.equal(out[:U], out2[:U])
    # two sources summed inside the kernel == the reduction of their elementwise sum, bit for bit
    # (DCNv2: the towers' two dL/dX0, layers._X0Link)
    src_b = torch.randn(n, W, generator=torch.Generator().manual_seed(2))
    both = ops.seg_reduce_rows(plan, src.to(DEV), W, src2=src_b.to(DEV))
    assert torch.equal(both[:U], ops.seg_reduce_rows(plan, src.to(DEV) + src_b.to(DEV), W)[:U])
    if W % 4 == 0 and n > 1:
        ha, hb = src.to(DEV).bfloat16(), src_b.to(DEV).bfloat16()
        assert torch.equal(ops.seg_reduce_rows(plan, ha, W, src2=hb)[:U], ops.seg_reduce_rows(plan, ha + hb, W)[:U])


@pytest.mark.parametrize("lists,length,V", [(1, 1000, 5000), (2, 700, 900), (4, 5000, 100000), (8, 20000, 9_449_445),
                                            (3, 64, 50)])
def test_seg_plan_merge_equals_sort(ops, lists, length, V):
    """The data-parallel merge plan (one ranking launch over `lists` sorted, -1-padded lists) must
    give exactly what the stable radix sort of the concatenation gives — every output array."""
    g = torch.Generator().manual_seed(lists * 7919 + length)
    keys = torch.full((lists, length), -1, dtype=torch.int32)
    for r in range(lists):
        n_r = int(torch.randint(0 if r == 1 else length // 2, length + 1, (1,), generator=g))
        n_r = min(n_r, V)
        pick = torch.randperm(V, generator=g)[:n_r].sort().values
        keys[r, :n_r] = pick.to(torch.int32)
    flat = keys.reshape(-1).to(DEV)
    a = ops.SegPlan(flat, V + 1, sorted_lists=lists)
    b = ops.SegPlan(flat, V + 1)
    U = b.count()
    assert a.count() == U
    n = lists * length
    for name in ("sorted_keys", "perm", "rank"):
        assert torch.equal(getattr(a, name)[:n], getattr(b, name)[:n]), name
    assert torch.equal(a.uniq[:U], b.uniq[:U]) and torch.equal(a.seg_start[:U + 1], b.seg_start[:U + 1])
    real = keys.reshape(-1)[keys.reshape(-1) >= 0].long()
    assert torch.equal(_cpu(a.uniq[:U]).long()[_cpu(a.uniq[:U]) >= 0], torch.unique(real))
    assert int(a.uniq[U - 1]) == -1 or bool((keys >= 0).all())      # the padding run sorts last


@pytest.mark.parametrize("sizes", [((94208, 9449445), (638976, 9449445)), ((159744, 33762577), (1171456, 33762577)),
                                   ((31, 50), (5000, 70000)), ((4097, 5), (1, 9449445)), ((0, 100), (70001, 300)),
                                   ((131073, 70000),), ((638976, 9449445), (94208, 200))])
def test_seg_plan_build_many_equals_single_plans(ops, sizes):
    """The multi-problem sort (both tables' plans from one chain of launches) must give, array by array, what
    one mapx_seg_plan per list gives — including lists of different key widths (3 and 4 passes in
    one launch sequence), an empty list, and a list of one key."""
    keys = [_skewed_keys(n, V, n + q).to(torch.int32).to(DEV) if n else torch.empty(0, dtype=torch.int32, device=DEV)
            for q, (n, V) in enumerate(sizes)]
    many = ops.SegPlan.build_many(keys, [V for _, V in sizes])
    again = ops.SegPlan.build_many(keys, [V for _, V in sizes])
    for k, (n, V), a, a2 in zip(keys, sizes, many, again):
        b = ops.SegPlan(k, V)
        U = b.count()
        assert a.count() == U == a2.count()
        for name in ("sorted_keys", "perm", "rank"):
            assert torch.equal(getattr(a, name)[:n], getattr(b, name)[:n]), (name, n, V)
            assert torch.equal(getattr(a, name)[:n], getattr(a2, name)[:n]), (name, n, V)
        assert torch.equal(a.uniq[:U], b.uniq[:U]) and torch.equal(a.seg_start[:U + 1], b.seg_start[:U + 1])
        if n:
            assert torch.equal(a.perm[:n].long().cpu(), torch.sort(k.long().cpu(), stable=True).indices)
            assert int(a.n_uniq[1]) == 0


def test_seg_plan_empty(ops):
    plan = ops.SegPlan(torch.empty(0, dtype=torch.int32, device=DEV), 10)
    assert plan.count() == 0


# --------------------------------------------------------------------------- alias
def test_alias_build_matches_oracle(ops):
    from oracle import ref_model as R
    g = np.random.default_rng(0)
    cnt = np.floor(g.pareto(1.2, 5000) * 5).astype(np.float32)
    cnt[:10] = 0
    _, _, q = R.nce_buffers(cnt)
    prob, alias = ops.alias_build(q)
    p_ref, a_ref = R.alias_build(q.numpy())
    assert np.array_equal(alias.numpy(), a_ref) and np.array_equal(prob.numpy(), p_ref)


def test_alias_draw_distribution(ops):
    """chi-square of 2.5 M draws against the renormalised noise distribution."""
    from oracle import ref_model as R
    g = np.random.default_rng(1)
    V, T, K = 200, 100000, 25
    cnt = np.floor(g.pareto(1.0, V) * 20).astype(np.float32) + (g.random(V) < 0.8)
    _, _, q = R.nce_buffers(cnt)
    prob, alias = ops.alias_build(q)
    packed = ops.alias_pack(prob.to(DEV), alias.to(DEV))
    targets = torch.randint(0, V, (T,), device=DEV)
    idx = ops.alias_draw(packed, targets, K, seed=1234, offset=7)
    assert torch.equal(idx[:, 0].long(), targets)
    noise = _cpu(idx[:, 1:]).reshape(-1).long()
    assert int(noise.min()) >= 0 and int(noise.max()) < V
    obs = torch.bincount(noise, minlength=V).double().numpy()
    exp = R.alias_distribution(prob.numpy(), alias.numpy()) * noise.numel()
    big = exp > 20
    chi2 = float((((obs - exp) ** 2) / np.maximum(exp, 1e-9))[big].sum())
    dof = int(big.sum()) - 1
    assert chi2 < dof + 6 * math.sqrt(2 * dof), (chi2, dof)
    assert obs[~big].sum() <= exp[~big].sum() + 6 * math.sqrt(exp[~big].sum() + 1) + 5
    # a different offset gives a different stream; the same (seed, offset) repeats exactly
    idx2 = ops.alias_draw(packed, targets, K, seed=1234, offset=8)
    idx3 = ops.alias_draw(packed, targets, K, seed=1234, offset=7)
    assert not torch.equal(idx, idx2) and torch.equal(idx, idx3)


# --------------------------------------------------------------------------- NCE
@pytest.mark.parametrize("B,F,P,K,V", [(7, 23, 32, 25, 1000), (64, 39, 32, 25, 1000),
                                       (33, 25, 16, 4, 300), (256, 23, 32, 25, 50000)])
def test_nce_fwd_and_grads_vs_oracle(ops, B, F, P, K, V):
    from oracle import ref_model as R
    g = torch.Generator().manual_seed(B * F)
    L = int(F * 0.3)
    enc = torch.randn(B, F * P, generator=g).requires_grad_(True)
    mi = torch.randint(0, F, (B, L), generator=g)
    mi[0, 1] = mi[0, 0]
    labels = torch.randint(0, V, (B, L), generator=g)
    noise = (torch.rand(B, L, K, generator=g) ** 3 * V).long().clamp_(0, V - 1)
    noise[0, 0, :2] = labels[0, 0]
    emb = (torch.rand(V, P, generator=g) * 2 - 1) / math.sqrt(P)
    emb.requires_grad_(True)
    cnt = torch.floor(torch.rand(V, generator=g) ** 4 * 100)
    logq, lnV, _ = R.nce_buffers(cnt)
    bias = (logq + lnV + 0.1 * torch.randn(V, generator=g)).unsqueeze(1).requires_grad_(True)
    params = {"feat_encoder.weight": torch.eye(F * P), "feat_encoder.bias": torch.zeros(F * P),
              "mfp_criterion.emb.weight": emb, "mfp_criterion.bias.weight": bias}
    loss_ref, logits_ref, acc_ref = R.mfp_head(params, enc, labels, mi, noise, logq, F, P, K)
    loss_ref.backward()

    idx = ops.nce_pack_idx(labels.to(DEV), noise.to(DEV), V, validate=True)
    o = ops.nce_fwd(enc.detach().to(DEV), mi.to(DEV), idx, emb.detach().to(DEV),
                    bias.detach().view(-1).to(DEV), logq.to(DEV), F, P, want_logits=True)
    # tolerance: fp32 1e-5 relative (north star), logits have |s| ~ 1..10
    np.testing.assert_allclose(_cpu(o["logits"]).view(B, L, K + 1).numpy(),
                               logits_ref.detach().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(float(o["loss"][0]), float(loss_ref), rtol=1e-5)
    assert int(o["acc"]) == acc_ref
    np.testing.assert_allclose(float(o["loss"][1]), acc_ref / (B * L), rtol=1e-6)   # accuracy, same launch
    denc = ops.nce_scatter_dh(o["dh"], mi.to(DEV), F, P)
    np.testing.assert_allclose(_cpu(denc).numpy(), enc.grad.numpy(), rtol=1e-4, atol=1e-7)
    # loss totals left to the scatter launch (a training step's form): same bits as the forward's own finalize step
    o2 = ops.nce_fwd(enc.detach().to(DEV), mi.to(DEV), idx, emb.detach().to(DEV),
                     bias.detach().view(-1).to(DEV), logq.to(DEV), F, P, totals_later=True)
    assert o2["totals"] is not None
    denc2 = ops.nce_scatter_dh(o2["dh"], mi.to(DEV), F, P, totals=o2["totals"])
    assert torch.equal(o2["loss"], o["loss"]) and torch.equal(o2["acc"], o["acc"]) and torch.equal(denc2, denc)
    plan = ops.SegPlan(idx.view(-1), V)
    ge, gb = ops.nce_table_grad(plan, o["dlogit"], o["h"], K, P)
    U = plan.count()
    uniq = _cpu(plan.uniq[:U]).long()
    dense_e = torch.zeros(V, P).index_copy_(0, uniq, _cpu(ge[:U]))
    dense_b = torch.zeros(V).index_copy_(0, uniq, _cpu(gb[:U]))
    np.testing.assert_allclose(dense_e.numpy(), emb.grad.numpy(), rtol=1e-4, atol=2e-7)
    np.testing.assert_allclose(dense_b.numpy(), bias.grad.view(-1).numpy(), rtol=1e-4, atol=2e-7)
    # rows never sampled get no gradient at all (sparse semantics == dense zeros)
    touched = torch.zeros(V, dtype=torch.bool).index_fill_(0, uniq, True)
    assert float(emb.grad[~touched].abs().sum()) == 0.0


# --------------------------------------------------------------------------- GEMM family
def _ref_mm(a, b):
    return (a.double() @ b.double())


@pytest.fixture(params=["x3", "h2"])
def arith(request, monkeypatch, ops):
    """Every GEMM test runs twice: through the six-product bf16 arithmetic (csrc/gemm_x3.hip: operands without
    magnitude records) and through the two-piece fp16 one (csrc/gemm_h2.hip, and gemm_h2w.hip where operand B is
    weight-like and the product large: ops.AUTO_AMAX computes the records — and planes — the step's own kernels
    would have left).  Same bounds for both."""
    monkeypatch.setattr(ops, "AUTO_AMAX", request.param == "h2")
    return request.param


@pytest.mark.parametrize("M,N,K", [(7, 368, 368), (64, 1000, 400), (4096, 1000, 368), (256, 736, 1368),
                                   (100, 23, 736), (129, 1, 1368), (4096, 368, 368), (65, 130, 17)])
def test_gemm_linear_forward(ops, arith, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    ref = _ref_mm(x, w.t()) + b.double()
    y = ops.linear_fwd(x.to(DEV), w.to(DEV), b.to(DEV))
    # fp32 k-ordered fma chain: |err| <= ~1e-6 * sum|a*b|
    bound = 2e-6 * (x.abs().double() @ w.abs().double().t()) + 1e-6
    assert bool(((_cpu(y).double() - ref).abs() <= bound).all())
    yr = ops.linear_fwd(x.to(DEV), w.to(DEV), b.to(DEV), relu=True)
    assert bool(((_cpu(yr).double() - ref.clamp(min=0)).abs() <= bound).all())


@pytest.mark.parametrize("M,N,K", [(7, 368, 400), (4096, 1000, 368), (64, 23, 736), (300, 1, 1368),
                                   (4096, 368, 368), (33, 65, 129)])
def test_gemm_backward_products(ops, arith, M, N, K):
    """dX = dY W (a_kc, b_nc) and dW = dY^T X (a_mc, b_nc), incl. deterministic split-K."""
    g = torch.Generator().manual_seed(M * N + K)
    dy, w, x = torch.randn(M, N, generator=g), torch.randn(N, K, generator=g), torch.randn(M, K, generator=g)
    dx = ops.linear_bwd_input(dy.to(DEV), w.to(DEV))
    ref = _ref_mm(dy, w)
    bound = 2e-6 * (dy.abs().double() @ w.abs().double()) + 1e-6
    assert bool(((_cpu(dx).double() - ref).abs() <= bound).all())
    dw = ops.linear_bwd_weight(dy.to(DEV), x.to(DEV))
    refw = _ref_mm(dy.t(), x)
    boundw = 4e-6 * (dy.abs().double().t() @ x.abs().double()) + 1e-6
    assert bool(((_cpu(dw).double() - refw).abs() <= boundw).all())
    assert torch.equal(dw, ops.linear_bwd_weight(dy.to(DEV), x.to(DEV)))   # reproducible
    add = torch.randn(M, K, generator=g)
    dxa = ops.linear_bwd_input(dy.to(DEV), w.to(DEV), add=add.to(DEV))
    assert bool(((_cpu(dxa).double() - ref - add.double()).abs() <= bound).all())
    act = torch.randn(M, K, generator=g)
    dxm = ops.linear_bwd_input(dy.to(DEV), w.to(DEV), relu_of=act.to(DEV))
    assert bool(((_cpu(dxm).double() - ref * (act > 0)).abs() <= bound).all())


@pytest.mark.parametrize("M,N,K,ns", [(128, 192, 512, 1), (128, 192, 512, 2), (1000, 1000, 256, 1), (1000, 1000, 4096, 1),
                                      (64, 64, 384, 1), (100, 68, 256, 1), (1000, 368, 2048, 2)])
def test_gemm_weight_gradient_shapes(ops, arith, M, N, K, ns):
    """Weight-gradient products (both operands k-strided) with K a multiple of 128: edge tiles
    (1000 = 7 x 128 + 104, 100 x 68), split-K slabs, padded leading dimensions, an output with a row stride of its
    own, and bit-reproducibility."""
    g = torch.Generator().manual_seed(M + N + K + ns)
    a, b = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    ad, bd = a.to(DEV), b.to(DEV)
    out = ops.gemm(ad, bd, False, False, M, N, K, nsplit=ns)
    ref = a.double().t() @ b.double()
    bound = 4e-6 * (a.abs().double().t() @ b.abs().double()) + 1e-6
    assert bool(((_cpu(out).double() - ref).abs() <= bound).all())
    assert torch.equal(out, ops.gemm(ad, bd, False, False, M, N, K, nsplit=ns))
    # a padded leading dimension and an output with its own row stride
    ap, bp = torch.zeros(K, M + 12, device=DEV), torch.zeros(K, N + 4, device=DEV)
    ap[:, :M], bp[:, :N] = ad, bd
    if ns == 1:
        big = torch.full((M, N + 8), 3.0, device=DEV)
        ops.gemm(ap, bp, False, False, M, N, K, out=big[:, :N], lda=M + 12, ldb=N + 4)
        assert torch.equal(big[:, :N], out) and bool((big[:, N:] == 3.0).all())


@pytest.mark.parametrize("a_kc,b_kc,M,N,K", [(1, 1, 4096, 1000, 368), (1, 1, 4096, 1000, 1000), (1, 0, 4096, 1368, 736),
                                             (0, 0, 1000, 1000, 4096), (1, 1, 300, 200, 200), (0, 0, 300, 200, 520),
                                             (1, 0, 777, 1624, 1248), (1, 1, 256, 128, 64), (1, 0, 130, 72, 40)])
def test_gemm_every_tile_layout(ops, arith, a_kc, b_kc, M, N, K):
    """Every tile layout of the fp32 GEMM (hints 0-3: 64x64, 128x64, 128x128 by 8 waves, 128x128 by 4
    waves with the hand-woven K-step) on the step's shapes and on the woven loop's corner cases: a K
    remainder (the partial tile goes first), edge tiles in both dimensions (out-of-range rows are read from
    row 0, not masked), K of exactly two K-steps, and K too short for the woven loop (falls back)."""
    g = torch.Generator().manual_seed(M * 131 + N * 17 + K)
    A = torch.randn((M, K) if a_kc else (K, M), generator=g)
    B = torch.randn((N, K) if b_kc else (K, N), generator=g)
    Am, Bm = (A if a_kc else A.t()), (B.t() if b_kc else B)
    ref = Am.double() @ Bm.double()
    bound = 2e-6 * (Am.abs().double() @ Bm.abs().double()) + 1e-6
    Ad, Bd = A.to(DEV), B.to(DEV)
    for tile in (-1, 0, 1, 2, 3):
        out = ops.gemm(Ad, Bd, bool(a_kc), bool(b_kc), M, N, K, tile=tile)
        assert bool(((_cpu(out).double() - ref).abs() <= bound).all()), tile
        assert torch.equal(out, ops.gemm(Ad, Bd, bool(a_kc), bool(b_kc), M, N, K, tile=tile)), tile


@pytest.mark.parametrize("M,N,K,c0", [(4096, 1368, 736, 368), (300, 432, 96, 368), (129, 368, 368, 368),
                                      (260, 1000, 200, 0), (64, 72, 40, 40)])
@pytest.mark.parametrize("add,accumulate,plus_v", [(False, False, False), (True, True, False), (True, True, True)])
def test_gemm_bwd_fused_epilogue(ops, arith, M, N, K, c0, add, accumulate, plus_v):
    """mapx_gemm_f32_bwd_fused: the dX GEMM whose epilogue does the ReLU backward right of column c0 and the
    cross layer's backward (t = v x0, dx0 (+)= v u (+ v)) left of it, plus one partial row of the bias
    gradients per 128-row tile — against fp64 on the step's shapes (the heads' concatenated input: N = D + H,
    c0 = D), a cross layer alone (c0 = N), a ReLU layer alone (c0 = 0), ragged rows and edge tiles."""
    g = torch.Generator().manual_seed(M + N + K + c0)
    rn = lambda *sh: torch.randn(*sh, generator=g)
    dy, w = rn(M, K), rn(K, N)
    addv = rn(M, N) if add else None
    mask, x0, u, dx0_in = rn(M, N), rn(M, max(c0, 4)), rn(M, max(c0, 4)), rn(M, max(c0, 4))
    d = lambda x: None if x is None else x.to(DEV)
    dx0_dev = dx0_in[:, :c0].contiguous().to(DEV) if (accumulate and c0 > 0) else None
    C, t, dx0, part = ops.gemm_bwd_fused(d(dy), d(w), c0, add=d(addv), mask=d(mask) if c0 < N else None,
                                         x0=d(x0[:, :c0].contiguous()) if c0 else None,
                                         u=d(u[:, :c0].contiguous()) if c0 else None, dx0=dx0_dev, plus_v=plus_v)
    v = dy.double() @ w.double() + (addv.double() if add else 0)
    bound = 2e-6 * (dy.abs().double() @ w.abs().double() + (addv.abs().double() if add else 0)) + 1e-6
    vr = v.clone()
    vr[:, c0:] = v[:, c0:] * (mask[:, c0:] > 0)
    assert bool(((_cpu(C).double() - vr).abs() <= bound).all())
    col_ref = torch.zeros(N, dtype=torch.float64)
    col_ref[c0:] = _cpu(C).double()[:, c0:].sum(0)
    if c0:
        Cc = _cpu(C).double()[:, :c0]
        # t and dx0 are formed from the stored v: exact products up to one fp32 rounding each
        assert bool(((_cpu(t).double() - Cc * x0[:, :c0].double()).abs() <= 1e-6 * (Cc * x0[:, :c0].double()).abs() + 1e-12).all())
        want = Cc * u[:, :c0].double() + (dx0_in[:, :c0].double() if accumulate else 0) + (Cc if plus_v else 0)
        mag = (Cc * u[:, :c0].double()).abs() + (dx0_in[:, :c0].abs().double() if accumulate else 0) + (Cc.abs() if plus_v else 0)
        assert bool(((_cpu(dx0).double() - want).abs() <= 3e-7 * mag + 1e-12).all())
        col_ref[:c0] = _cpu(t).double().sum(0)
    assert part.shape == ((M + 63) // 64, N)
    got = _cpu(part).double().sum(0)
    absum = torch.zeros(N, dtype=torch.float64)
    absum[c0:] = _cpu(C).double()[:, c0:].abs().sum(0)
    if c0:
        absum[:c0] = _cpu(t).double().abs().sum(0)
    assert bool(((got - col_ref).abs() <= 1e-6 * absum + 1e-6).all())
    # reproducible, and the partial rows feed mapx_sum_tasks as the bias gradients
    C2, t2, dx02, part2 = ops.gemm_bwd_fused(d(dy), d(w), c0, add=d(addv), mask=d(mask) if c0 < N else None,
                                            x0=d(x0[:, :c0].contiguous()) if c0 else None,
                                            u=d(u[:, :c0].contiguous()) if c0 else None,
                                            dx0=dx0_in[:, :c0].contiguous().to(DEV) if (accumulate and c0 > 0) else None,
                                            plus_v=plus_v)
    assert torch.equal(C, C2) and torch.equal(part, part2) and (c0 == 0 or (torch.equal(t, t2) and torch.equal(dx0, dx02)))
    if c0 and c0 < N:
        dst_a, dst_b = torch.zeros(c0, device=DEV), torch.zeros(N - c0, device=DEV)
        ops.defer_part_rows(dst_a, part, 0, c0)
        ops.defer_part_rows(dst_b, part, c0, N - c0)
        ops.flush_deferred()
        assert bool(((_cpu(dst_a).double() - col_ref[:c0]).abs() <= 1e-6 * absum[:c0] + 1e-6).all())
        assert bool(((_cpu(dst_b).double() - col_ref[c0:]).abs() <= 1e-6 * absum[c0:] + 1e-6).all())


@pytest.mark.parametrize("cnt,Bn,Nn,K", [(3, 4096, 368, 368), (2, 777, 72, 40), (4, 512, 128, 136), (1, 300, 64, 64)])
def test_linear_bwd_weight_batched(ops, arith, cnt, Bn, Nn, K):
    """mapx_gemm_f32_batched: the cross layers' weight gradients (equal shapes) from one launch + one slab sum."""
    g = torch.Generator().manual_seed(cnt + Bn + Nn + K)
    dys = [torch.randn(Bn, Nn, generator=g) for _ in range(cnt)]
    xs = [torch.randn(Bn, K, generator=g) for _ in range(cnt)]
    outs = [torch.full((Nn, K), 7.0, device=DEV) for _ in range(cnt)]
    ops.linear_bwd_weight_batched([t.to(DEV) for t in dys], [t.to(DEV) for t in xs], outs)
    for dy, x, o in zip(dys, xs, outs):
        ref = dy.double().t() @ x.double()
        bound = 4e-6 * (dy.abs().double().t() @ x.abs().double()) + 1e-6
        assert bool(((_cpu(o).double() - ref).abs() <= bound).all())
    again = [torch.empty(Nn, K, device=DEV) for _ in range(cnt)]
    ops.linear_bwd_weight_batched([t.to(DEV) for t in dys], [t.to(DEV) for t in xs], again)
    assert all(torch.equal(a, b) for a, b in zip(outs, again))


def test_gemm_x3_operand_magnitudes(ops):
    """The six-product arithmetic of the fp32 GEMM (three bf16 pieces per operand, csrc/gemm_x3.hip) away from
    randn operands: what it guarantees and where it stops, as measured by
    tools/micro/x3_magnitude_probe.py and stated in DESIGN 4.1:
    * exponents spread over 2^-30 .. 2^30 inside a row, rows scaled to 2^+-100: the error stays below 1e-6 of
      sum_k |a_k b_k| (randn: 8e-8) — the pieces carry the fp32 exponent range, nothing is scaled per tensor;
    * |x| < 2^-110: the lower pieces are bf16 subnormals and are lost, the precision falls off towards bf16's 8 bits
      at 2^-126 and below (bounded by 2^-7 here) — fp32 training values are 25 orders of magnitude above that;
    * |x| > 3.3895e38 (the largest bf16): the leading piece rounds to infinity and the residual is NaN, so that
      output ROW is NaN where fp32 arithmetic would be finite; infinities and NaNs make their rows non-finite
      (an infinity gives NaN, not +-inf); other rows are untouched."""
    g = torch.Generator().manual_seed(0)
    M, N, K = 256, 128, 512

    def spread(shape, lo, hi):
        mant = 1 + torch.rand(shape, generator=g)
        e = torch.randint(lo, hi + 1, shape, generator=g).float()
        return torch.where(torch.rand(shape, generator=g) < 0.5, -1.0, 1.0) * mant * torch.exp2(e)

    def worst(A, B):
        out = _cpu(ops.gemm(A.to(DEV), B.to(DEV), True, True, M, N, K))
        ref, mag = A.double() @ B.double().t(), A.abs().double() @ B.abs().double().t()
        return out, float(((out.double() - ref).abs() / mag).max())

    assert worst(spread((M, K), -30, 30), spread((N, K), -30, 30))[1] <= 1e-6
    assert worst(spread((M, K), -3, 3) * 2.0 ** 100, spread((N, K), -3, 3) * 2.0 ** -100)[1] <= 1e-6
    assert worst(spread((M, K), -100, -100), spread((N, K), 60, 60))[1] <= 1e-6
    out, err = worst(spread((M, K), 0, 0) * 2.0 ** -125, spread((N, K), 20, 20))       # lower pieces subnormal in bf16
    assert bool(torch.isfinite(out).all()) and err <= 2.0 ** -7
    A = torch.randn(M, K, generator=g)
    A[3, 7] = 3.40e38                                      # between the largest bf16 and the largest fp32
    out, _ = worst(A, torch.randn(N, K, generator=g) * 1e-3)
    assert bool(torch.isnan(out[3]).all()) and bool(torch.isfinite(out[:3]).all()) and bool(torch.isfinite(out[4:]).all())
    A = torch.randn(M, K, generator=g)
    A[5, 1], A[6, 2] = float("inf"), float("nan")
    B = torch.randn(N, K, generator=g)
    out = _cpu(ops.gemm(A.to(DEV), B.to(DEV), True, True, M, N, K))
    assert bool(torch.isnan(out[5]).all()) and bool(torch.isnan(out[6]).all())
    keep = torch.ones(M, dtype=torch.bool)
    keep[5] = keep[6] = False
    ref = A[keep].double() @ B.double().t()
    assert bool(((out[keep].double() - ref).abs() <= 1e-6 * (A[keep].abs().double() @ B.abs().double().t())).all())


@pytest.mark.parametrize("M,N,K", [(4096, 1000, 1000), (300, 368, 200), (129, 72, 1000)])
def test_gemm_relu_mask_colsum_epilogue(ops, arith, M, N, K):
    """EPI_RELU_MASK_COLSUM (an MLP layer's dX GEMM doing the upstream layer's ReLU backward): the masked
    product, and partial rows of its column sums (one per 64 rows of the product, every one written whatever the
    kernel's tile height) that add up to the bias gradient."""
    from mapx.native import EPI_RELU_MASK_COLSUM
    g = torch.Generator().manual_seed(M + N + K)
    dy, w, y = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g), torch.randn(M, N, generator=g)
    part = torch.full(((M + 63) // 64, N), 7.0, device=DEV)
    dz = ops.gemm(dy.to(DEV), w.to(DEV), True, False, M, N, K, epi=EPI_RELU_MASK_COLSUM, aux1=y.to(DEV), out2=part)
    ref = (dy.double() @ w.double()) * (y > 0)
    bound = 2e-6 * (dy.abs().double() @ w.abs().double()) + 1e-6
    assert bool(((_cpu(dz).double() - ref).abs() <= bound).all())
    # the partial rows are exactly the fp32 sums of the stored tile rows, tile by tile, in a fixed order
    got = _cpu(part).double().sum(0)
    want = _cpu(dz).double().sum(0)
    assert bool(((got - want).abs() <= 1e-6 * _cpu(dz).double().abs().sum(0) + 1e-6).all())
    again = torch.empty_like(part)
    ops.gemm(dy.to(DEV), w.to(DEV), True, False, M, N, K, epi=EPI_RELU_MASK_COLSUM, aux1=y.to(DEV), out2=again)
    assert torch.equal(part, again)


def test_mlp_backward_link_equals_unlinked(ops, monkeypatch):
    """layers._ReluLink: the MLP's gradients with the ReLU mask + bias gradient folded into the consumer's dX
    epilogue agree with the plain chain (same math, different summation order of the bias gradient)."""
    from mapx import layers
    torch.manual_seed(3)
    x = torch.randn(512, 368, device=DEV)
    outs = []
    for link in (True, False):
        monkeypatch.setattr(ops, "RELU_LINK", link)
        torch.manual_seed(7)
        mlp = layers.MLPBlock(368, hidden_size=1000, num_hidden_layers=3, hidden_dropout_rate=0.0).to(DEV)
        slots = {id(p): torch.zeros_like(p) for p in mlp.parameters()}
        monkeypatch.setattr(layers, "_grad_slot", lambda p: slots.get(id(p)))
        xin = x.clone().requires_grad_(True)
        mlp(xin).square().sum().backward()
        ops.flush_deferred()
        outs.append(([slots[id(p)].clone() for p in mlp.parameters()], xin.grad.clone()))
    for a, b in zip(outs[0][0], outs[1][0]):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-6
    assert float((outs[0][1] - outs[1][1]).abs().max()) <= 2e-5 * float(outs[1][1].abs().max())


def test_mask_rows_with_device_cursor(ops):
    """mapx_dynamic_mask_mfp_rows with a device-side cursor (a captured step walking an epoch's permutation)
    == the same call on the slice of row numbers."""
    g = torch.Generator().manual_seed(0)
    N_, F, B, L = 5000, 23, 256, 6
    split = torch.randint(10, 1000, (N_, F), generator=g).to(DEV)
    order = torch.randperm(N_, generator=g).to(DEV)
    mi = torch.stack([torch.randperm(F, generator=g)[:L] for _ in range(B)]).to(DEV)
    for start in (0, 256, 4096):
        cur = torch.tensor([start], dtype=torch.int64, device=DEV)
        a = ops.dynamic_mask_mfp(split, L, masked_index=mi, sel=order, sel_cursor=cur, batch=B)
        b = ops.dynamic_mask_mfp(split, L, masked_index=mi, sel=order[start:start + B].contiguous())
        for x_, y_ in zip(a, b):
            assert torch.equal(x_, y_)


def test_gemm_writes_into_column_slice(ops, arith):
    x, w, b = torch.randn(50, 64, device=DEV), torch.randn(40, 64, device=DEV), torch.randn(40, device=DEV)
    final = torch.full((50, 100), 7.0, device=DEV)
    ops.linear_fwd(x, w, b, out=final[:, 60:])
    assert bool((final[:, :60] == 7.0).all())
    np.testing.assert_allclose(_cpu(final[:, 60:]).numpy(), (_cpu(x) @ _cpu(w).t() + _cpu(b)).numpy(),
                               rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("M,N", [(1, 5), (4096, 1000), (77, 368), (5000, 23)])
def test_colsum(ops, M, N):
    x = torch.randn(M, N, generator=torch.Generator().manual_seed(M))
    out = ops.colsum(x.to(DEV))
    np.testing.assert_allclose(_cpu(out).double().numpy(), x.double().sum(0).numpy(),
                               rtol=0, atol=2e-6 * float(x.abs().sum(0).max()) + 1e-6)


@pytest.mark.parametrize("B,D,NC", [(7, 368, 3), (64, 400, 3), (4096, 368, 3), (33, 624, 2), (100, 368, 3), (4, 368, 1)])
def test_cross_network_fwd_bwd_vs_oracle(ops, B, D, NC):
    """CrossNetV2 forward + hand-written backward chain vs autograd on the oracle.  B = 100, 33, 4:
    row counts whose last 32-row band ends 1-4 rows in, where the upper half-wave of the GEMM
    epilogue (rows +4) has no valid row at all (its auxiliary loads must be clamped)."""
    from oracle import ref_model as R
    g = torch.Generator().manual_seed(B + D)
    x0 = (0.3 * torch.randn(B, D, generator=g)).requires_grad_(True)
    P = {}
    for i in range(NC):
        P[f"cross_net.cross_layers.{i}.weight"] = (torch.randn(D, D, generator=g) / math.sqrt(D)).requires_grad_(True)
        P[f"cross_net.cross_layers.{i}.bias"] = (0.1 * torch.randn(D, generator=g)).requires_grad_(True)
    y_ref = R.cross(P, x0, NC)
    gout = torch.randn(B, D, generator=g)
    y_ref.backward(gout)

    x0d = x0.detach().to(DEV)
    xs, us = [x0d], []
    for i in range(NC):
        y, u = ops.cross_layer_fwd(x0d, xs[-1], P[f"cross_net.cross_layers.{i}.weight"].detach().to(DEV),
                                   P[f"cross_net.cross_layers.{i}.bias"].detach().to(DEV))
        xs.append(y)
        us.append(u)
    np.testing.assert_allclose(_cpu(xs[-1]).numpy(), y_ref.detach().numpy(), rtol=1e-5, atol=1e-5)
    gcur, dx0 = gout.to(DEV), None
    for i in reversed(range(NC)):
        w = P[f"cross_net.cross_layers.{i}.weight"]
        t, dx0 = ops.cross_bwd_pre(gcur, x0d, us[i], dx0)
        dw = ops.linear_bwd_weight(t, xs[i])
        db = ops.colsum(t)
        gcur = ops.linear_bwd_input(t, w.detach().to(DEV), add=gcur)
        np.testing.assert_allclose(_cpu(dw).numpy(), w.grad.numpy(), rtol=2e-4,
                                   atol=2e-5 * float(w.grad.abs().max()))
        np.testing.assert_allclose(_cpu(db).numpy(), P[f"cross_net.cross_layers.{i}.bias"].grad.numpy(),
                                   rtol=2e-4, atol=2e-5 * float(P[f"cross_net.cross_layers.{i}.bias"].grad.abs().max()))
    dx0_total = _cpu(dx0 + gcur)
    np.testing.assert_allclose(dx0_total.numpy(), x0.grad.numpy(), rtol=2e-4,
                               atol=2e-5 * float(x0.grad.abs().max()))


# --------------------------------------------------------------------------- heads / masks
@pytest.mark.parametrize("n", [1, 7 * 23, 4096 * 23, 4096])
def test_bce_with_logits(ops, n):
    g = torch.Generator().manual_seed(n)
    x = 3 * torch.randn(n, generator=g)
    y = (torch.rand(n, generator=g) < 0.3).float()
    xr = x.clone().requires_grad_(True)
    ref = torch.nn.functional.binary_cross_entropy_with_logits(xr, y)
    ref.backward()
    out3, dl = ops.bce_with_logits(x.to(DEV), y.to(DEV))
    out3 = _cpu(out3)
    np.testing.assert_allclose(float(out3[0]), float(ref), rtol=1e-5)
    acc = ((torch.sigmoid(x) > 0.5).float() == y).sum() / n
    np.testing.assert_allclose(float(out3[1]), float(acc), rtol=1e-6)
    np.testing.assert_allclose(float(out3[2]), float(y.mean()), rtol=1e-6)
    np.testing.assert_allclose(_cpu(dl).numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-9)


def test_dynamic_mask_injected_matches_oracle(ops):
    from oracle import ref_model as R
    g = torch.Generator().manual_seed(5)
    B, F, L = 300, 23, 6
    ids = torch.randint(10, 1000, (B, F), generator=g)
    mi = torch.randint(0, F, (B, L), generator=g)
    mi[:, 1] = mi[:, 0]                                          # duplicates everywhere
    rep = torch.randint(10, 1000, (B, L), generator=g)
    m_ref, l_ref = R.dynamic_mask_mfp(ids, mi)
    m, l, mo = ops.dynamic_mask_mfp(ids.to(DEV), L, masked_index=mi.to(DEV))
    assert torch.equal(_cpu(m), m_ref) and torch.equal(_cpu(l), l_ref) and torch.equal(_cpu(mo), mi)
    r_ref, y_ref = R.dynamic_mask_rfd(ids, mi, rep)
    r, y, _ = ops.dynamic_mask_rfd(ids.to(DEV), L, masked_index=mi.to(DEV), replace_feat=rep.to(DEV))
    assert torch.equal(_cpu(r), r_ref) and torch.equal(_cpu(y), y_ref)


def test_dynamic_mask_generated_properties(ops):
    B, F, L, Ntrain = 4096, 23, 6, 20000
    g = torch.Generator().manual_seed(6)
    lo = torch.arange(F) * 100 + 10
    x_train = (lo[None, :] + torch.randint(0, 100, (Ntrain, F), generator=g)).to(DEV)
    ids = x_train[:B].clone()
    m, labels, mi = ops.dynamic_mask_mfp(ids, L, seed=42, offset=3)
    mi_c, m_c, ids_c = _cpu(mi), _cpu(m), _cpu(ids)
    assert int(mi_c.min()) >= 0 and int(mi_c.max()) < F
    assert torch.equal(_cpu(labels), torch.gather(ids_c, 1, mi_c))
    expect = torch.scatter(ids_c, 1, mi_c, torch.full_like(mi_c, 3))
    assert torch.equal(m_c, expect)
    counts = torch.bincount(mi_c.view(-1), minlength=F).double()
    chi2 = float(((counts - B * L / F) ** 2 / (B * L / F)).sum())
    assert chi2 < (F - 1) + 6 * math.sqrt(2 * (F - 1))
    # RFD/Unigram: replaced fields stay inside their own field's id range
    r, y, mi2 = ops.dynamic_mask_rfd(ids, L, x_train=x_train, seed=42, offset=4)
    r_c, y_c = _cpu(r), _cpu(y)
    fld = (r_c - 10) // 100
    assert torch.equal(fld, torch.arange(F).expand(B, F))
    assert torch.equal(y_c, (r_c != ids_c).float())
    untouched = torch.ones(B, F, dtype=torch.bool).scatter_(1, _cpu(mi2), False)
    assert bool((r_c[untouched] == ids_c[untouched]).all())
    assert 0.15 < float(y_c.mean()) < 0.3          # ~ (1-(1-1/F)^L) * P(different value)


# --------------------------------------------------------------------------- optimizer
def _sched(ops, T, lr0=1e-3, kind="cosine", warm=0, b1=0.9, b2=0.999):
    from oracle import ref_model as R
    lambdas = [R.lr_lambda(kind, s, T, warm) for s in range(T)]
    return ops.make_sched(lr0, lambdas, b1, b2).to(DEV), lambdas


@pytest.mark.parametrize("wd", [0.0, 0.05])
def test_adamw_dense_trajectory_vs_oracle(ops, wd):
    from oracle import ref_model as R
    T, n = 12, 1003
    sched, lambdas = _sched(ops, T)
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(n, generator=g)
    p_ref, m_ref, v_ref = p0.clone(), torch.zeros(n), torch.zeros(n)
    p = p0.to(DEV).clone()
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    done = torch.zeros(1, dtype=torch.int32, device=DEV)
    for s in range(1, T + 1):
        grad = torch.randn(n, generator=g) * (0.1 if s % 3 else 0.0)
        R.hf_adamw_step(p_ref, grad, m_ref, v_ref, s, 1e-3 * lambdas[s - 1], wd=wd)
        ops.adamw_dense(p, grad.to(DEV), m, v, sched, done, 0.9, 0.999, 1e-8, wd)
        ops.step_advance(done)
    assert int(done) == T
    cur = torch.tensor([5], dtype=torch.int64, device=DEV)       # the batch cursor rides on the same launch
    done2 = done.clone()
    ops.step_advance(done2, cur, 4096)
    assert int(done2) == T + 1 and int(cur) == 4101
    np.testing.assert_allclose(_cpu(p).numpy(), p_ref.numpy(), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(_cpu(m).numpy(), m_ref.numpy(), rtol=2e-6, atol=2e-8)
    np.testing.assert_allclose(_cpu(v).numpy(), v_ref.numpy(), rtol=2e-6, atol=1e-10)


@pytest.mark.parametrize("W,with_bias", [(16, False), (32, True)])
def test_lazy_table_adam_equals_dense_reference(ops, W, with_bias):
    """Row-sparse lazy AdamW == the reference's dense every-row-every-step AdamW (rows that
    receive no gradient still decay and move by momentum), after a final flush."""
    from oracle import ref_model as R
    T, V = 40, 257
    sched, lambdas = _sched(ops, T, kind="cosine")
    aux = ops.make_replay_aux(1e-3, lambdas, 0.9, 0.999, 0.05).to(DEV)
    g = torch.Generator().manual_seed(W)
    p0, b0 = torch.randn(V, W, generator=g), torch.randn(V, generator=g)
    pr, mr, vr = p0.clone(), torch.zeros(V, W), torch.zeros(V, W)
    br, bmr, bvr = b0.clone(), torch.zeros(V), torch.zeros(V)
    p, m, v = p0.to(DEV).clone(), torch.zeros(V, W, device=DEV), torch.zeros(V, W, device=DEV)
    b, bm, bv = b0.to(DEV).clone(), torch.zeros(V, device=DEV), torch.zeros(V, device=DEV)
    last = torch.zeros(V, dtype=torch.int32, device=DEV)
    done = torch.zeros(1, dtype=torch.int32, device=DEV)
    kw = dict(p1=b, m1=bm, v1=bv, wd1=0.0) if with_bias else {}
    for s in range(1, T + 1):
        nrow = int(torch.randint(1, 40, (1,), generator=g))
        rows = torch.randperm(V, generator=g)[:nrow].sort().values
        grad = 0.1 * torch.randn(nrow, W, generator=g)
        gb = 0.1 * torch.randn(nrow, generator=g)
        dense_g = torch.zeros(V, W).index_copy_(0, rows, grad)
        dense_gb = torch.zeros(V).index_copy_(0, rows, gb)
        lr = 1e-3 * lambdas[s - 1]
        R.hf_adamw_step(pr, dense_g, mr, vr, s, lr, wd=0.05)
        if with_bias:
            R.hf_adamw_step(br, dense_gb, bmr, bvr, s, lr, wd=0.0)
        rows_d = rows.to(torch.int32).to(DEV)
        nrd = torch.tensor([nrow], dtype=torch.int32, device=DEV)
        # catch-up before "forward": touched rows must already equal the dense reference
        ops.table_adam(p, m, v, 0.05, last, sched, done, aux, 0.9, 0.999, 1e-8, rows=rows_d,
                       n_rows_dev=nrd, **kw)
        ops.table_adam(p, m, v, 0.05, last, sched, done, aux, 0.9, 0.999, 1e-8, rows=rows_d,
                       n_rows_dev=nrd, grad0=grad.to(DEV), grad1=gb.to(DEV) if with_bias else None, **kw)
        ops.step_advance(done)
        np.testing.assert_allclose(_cpu(p)[rows].numpy(), pr[rows].numpy(), rtol=5e-6, atol=1e-7)
    ops.table_adam(p, m, v, 0.05, last, sched, done, aux, 0.9, 0.999, 1e-8, **kw)       # flush all rows
    assert bool((_cpu(last) == T).all())
    np.testing.assert_allclose(_cpu(p).numpy(), pr.numpy(), rtol=5e-6, atol=1e-7)
    np.testing.assert_allclose(_cpu(m).numpy(), mr.numpy(), rtol=5e-6, atol=2e-8)
    np.testing.assert_allclose(_cpu(v).numpy(), vr.numpy(), rtol=5e-6, atol=1e-10)
    if with_bias:
        np.testing.assert_allclose(_cpu(b).numpy(), br.numpy(), rtol=5e-6, atol=1e-7)


def test_lazy_replay_long_gap_closed_form_tail(ops):
    """Rows left untouched for thousands of steps: the step-by-step replay hands over to the
    closed-form tail (fp64 prefix products); the result must stay within fp32 rounding noise of
    the reference's dense step-by-step AdamW."""
    from oracle import ref_model as R
    T, V, W = 3000, 12, 16
    sched, lambdas = _sched(ops, T, kind="cosine", warm=100)
    aux = ops.make_replay_aux(1e-3, lambdas, 0.9, 0.999, 0.05).to(DEV)
    g = torch.Generator().manual_seed(9)
    p0 = torch.randn(V, W, generator=g)
    p0[0, :4] = 0.0                                        # exact zeros never satisfy |term| < c|p|
    pr, mr, vr = p0.clone(), torch.zeros(V, W), torch.zeros(V, W)
    p, m, v = p0.to(DEV).clone(), torch.zeros(V, W, device=DEV), torch.zeros(V, W, device=DEV)
    last = torch.zeros(V, dtype=torch.int32, device=DEV)
    done = torch.zeros(1, dtype=torch.int32, device=DEV)
    touch = {1: [0, 1, 2, 3], 2: [2, 3, 4], 700: [3, 5], 2999: [1]}
    for s in range(1, T + 1):
        dense_g = torch.zeros(V, W)
        if s in touch:
            rows = torch.tensor(touch[s])
            grad = 0.05 * torch.randn(len(rows), W, generator=g)
            dense_g.index_copy_(0, rows, grad)
            ops.table_adam(p, m, v, 0.05, last, sched, done, aux, 0.9, 0.999, 1e-8,
                           rows=rows.to(torch.int32).to(DEV), grad0=grad.to(DEV))
        R.hf_adamw_step(pr, dense_g, mr, vr, s, 1e-3 * lambdas[s - 1], wd=0.05)
        ops.step_advance(done)
    ops.table_adam(p, m, v, 0.05, last, sched, done, aux, 0.9, 0.999, 1e-8)         # flush
    assert bool((_cpu(last) == T).all())
    np.testing.assert_allclose(_cpu(p).numpy(), pr.numpy(), rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(_cpu(v).numpy(), vr.numpy(), rtol=1e-4, atol=1e-12)
    np.testing.assert_allclose(_cpu(m).numpy(), mr.numpy(), rtol=1e-4, atol=1e-12)


@pytest.mark.parametrize("grouped", [False, True])
def test_nce_forward_reads_rows_through_their_pending_updates(ops, grouped):
    """mapx_nce_fwd(lazy=): the loss kernel reads every sampled row of the NCE table through its missing
    zero-gradient updates (last[row], the m | v record, the closed form with this step's tabulated coefficients) and
    writes nothing.  Bit-identical — loss, logits, dlogit, dh — to the catch-up pass followed by the plain kernel, on a
    table whose rows are 0 .. 40 updates behind (hot ids current, the bias table included), and the table itself is
    untouched.  Then the gradient update: from the stale table it leaves the very rows the catch-up + update pair
    leaves (a row's one read-modify-write of the step)."""
    from mapx import native as N
    V, P, K, B, L, F = 3000, 32, 25, 64, 3, 5
    T_done = 57
    sched, lambdas = _sched(ops, 200, kind="cosine", warm=10)
    aux = ops.make_replay_aux(1e-3, lambdas, 0.9, 0.999, 0.05).to(DEV)
    g = torch.Generator().manual_seed(11)
    emb = (0.3 * torch.randn(V, P, generator=g)).to(DEV)
    bias = torch.randn(V, generator=g).to(DEV)
    mv0 = torch.cat([0.01 * torch.randn(V, P, generator=g), 1e-4 * torch.rand(V, P, generator=g)], 1).to(DEV)
    mv1 = torch.stack([0.01 * torch.randn(V, generator=g), 1e-4 * torch.rand(V, generator=g)], 1).to(DEV)
    m0, v0, m1, v1 = mv0[:, :P], mv0[:, P:], mv1[:, 0], mv1[:, 1]
    last = (T_done - torch.randint(0, 41, (V,), generator=g)).to(torch.int32).to(DEV)
    last[:50] = T_done                                            # hot ids: current
    done = torch.full((1,), T_done, dtype=torch.int32, device=DEV)
    logq = torch.log_softmax(torch.randn(V, generator=g), 0).to(DEV)
    masked_index = torch.stack([torch.randperm(F, generator=g)[:L] for _ in range(B)]).to(DEV)
    idx = torch.randint(0, V, (B * L, K + 1), generator=g).to(torch.int32)
    idx[:, 3] = idx[:, 3] % 50
    idx = idx.to(DEV)
    enc = torch.randn(B, F * P, generator=g).to(DEV)
    kw = {}
    if grouped:
        groups = ops.EncGroups(masked_index, F)
        h_slots = torch.zeros(groups.cap, P, device=DEV)
        hsel = enc.view(B, F, P)[torch.arange(B, device=DEV)[:, None], masked_index].reshape(B * L, P)
        h_slots[groups.hpos.long()] = hsel
        kw = dict(hpos=groups.hpos)
        src = h_slots
    else:
        src = enc

    def fwd(e, b, lazy=None):
        dh_slots = torch.zeros(groups.cap, P, device=DEV) if grouped else None
        o = ops.nce_fwd(src, masked_index, idx, e, b, logq, F, P, want_logits=True, lazy=lazy,
                        dh_slots=dh_slots, **kw)
        return [o["loss"], o["acc"], o["h"], o["dlogit"], o["dh"], o["logits"]] + ([dh_slots] if grouped else [])

    coef = ops.replay_coef_table(aux, 0.9, 0.999, done)
    lz = N.LazyRows()
    lz.m0, lz.v0, lz.ld_mv0, lz.wd0 = m0.data_ptr(), v0.data_ptr(), mv0.stride(0), 0.05
    lz.m1, lz.v1, lz.ld_mv1, lz.wd1 = m1.data_ptr(), v1.data_ptr(), mv1.stride(0), 0.0
    lz.last, lz.sched, lz.sched_len, lz.done = last.data_ptr(), sched.data_ptr(), sched.shape[0], done.data_ptr()
    lz.aux, lz.aux_len, lz.aux_rows = aux.data_ptr(), aux.shape[1], aux.shape[0]
    lz.beta1, lz.beta2, lz.eps, lz.coef_opt = 0.9, 0.999, 1e-8, coef.data_ptr()
    before = [t.clone() for t in (emb, bias, mv0, mv1, last)]
    got = fwd(emb, bias, lazy=lz)
    for a, b in zip(before, (emb, bias, mv0, mv1, last)):
        assert torch.equal(a, b)                                  # nothing written
    # the catch-up pass on copies, then the plain kernel
    e2, b2, mv0b, mv1b, last2 = [t.clone() for t in before]
    ta = dict(p1=b2, m1=mv1b[:, 0], v1=mv1b[:, 1], wd1=0.0)
    ops.table_adam(e2, mv0b[:, :P], mv0b[:, P:], 0.05, last2, sched, done, aux, 0.9, 0.999, 1e-8,
                   rows=idx.view(-1), rows_may_repeat=True, **ta)
    assert not torch.equal(e2, emb)
    want = fwd(e2, b2)
    for k, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), k
    # the update: stale table + gradient == caught-up table + gradient
    uniq = torch.unique(idx.view(-1).long()).to(torch.int32)
    g0, g1 = 0.05 * torch.randn(uniq.numel(), P, generator=g).to(DEV), 0.05 * torch.randn(uniq.numel(), generator=g).to(DEV)
    ops.table_adam(emb, m0, v0, 0.05, last, sched, done, aux, 0.9, 0.999, 1e-8, rows=uniq, grad0=g0, grad1=g1,
                   p1=bias, m1=m1, v1=v1, wd1=0.0)
    ops.table_adam(e2, mv0b[:, :P], mv0b[:, P:], 0.05, last2, sched, done, aux, 0.9, 0.999, 1e-8, rows=uniq,
                   grad0=g0, grad1=g1, **ta)
    for a, b in zip((emb, bias, mv0, mv1, last), (e2, b2, mv0b, mv1b, last2)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("half", [False, True])
def test_gather_reads_rows_through_their_pending_updates(ops, half):
    """mapx_emb_gather_fwd(lazy=) / _bf16: the gather of a training step replays a stale row's missing zero-gradient
    updates in registers and writes nothing; bit-identical to the catch-up pass followed by the plain gather (fp32
    rows, and the bf16 rows of the bf16 compute mode), hot ids current, repeats in the id list."""
    from mapx import native as N
    V, E, n, T_done = 5000, 16, 4096 * 3, 41
    sched, lambdas = _sched(ops, 120, kind="cosine", warm=10)
    aux = ops.make_replay_aux(1e-3, lambdas, 0.9, 0.999, 0.05).to(DEV)
    g = torch.Generator().manual_seed(4)
    emb = (0.3 * torch.randn(V, E, generator=g)).to(DEV)
    mv0 = torch.cat([0.01 * torch.randn(V, E, generator=g), 1e-4 * torch.rand(V, E, generator=g)], 1).to(DEV)
    last = (T_done - torch.randint(0, 30, (V,), generator=g)).clamp(min=0).to(torch.int32).to(DEV)
    last[:64] = T_done
    done = torch.full((1,), T_done, dtype=torch.int32, device=DEV)
    ids = torch.randint(0, V, (n,), generator=g)
    ids[::5] = ids[::5] % 64
    ids = ids.to(DEV)
    coef = ops.replay_coef_table(aux, 0.9, 0.999, done)
    lz = N.LazyRows()
    lz.m0, lz.v0, lz.ld_mv0, lz.wd0 = mv0[:, :E].data_ptr(), mv0[:, E:].data_ptr(), mv0.stride(0), 0.05
    lz.m1, lz.v1, lz.ld_mv1, lz.wd1 = None, None, 1, 0.0
    lz.last, lz.sched, lz.sched_len, lz.done = last.data_ptr(), sched.data_ptr(), sched.shape[0], done.data_ptr()
    lz.aux, lz.aux_len, lz.aux_rows = aux.data_ptr(), aux.shape[1], aux.shape[0]
    lz.beta1, lz.beta2, lz.eps, lz.coef_opt = 0.9, 0.999, 1e-8, coef.data_ptr()
    dt = torch.bfloat16 if half else torch.float32
    before = [t.clone() for t in (emb, mv0, last)]
    got = ops.emb_gather(ids, emb, out_dtype=dt, lazy=lz)
    for a, b in zip(before, (emb, mv0, last)):
        assert torch.equal(a, b)
    e2, mvb, last2 = [t.clone() for t in before]
    ops.table_adam(e2, mvb[:, :E], mvb[:, E:], 0.05, last2, sched, done, aux, 0.9, 0.999, 1e-8,
                   rows=ids.to(torch.int32), rows_may_repeat=True)
    assert not torch.equal(e2, emb)
    want = ops.emb_gather(ids, e2, out_dtype=dt)
    assert torch.equal(got, want)
    if not half:
        assert ops.amax_value(ops.amax_of(got)) == ops.amax_value(ops.amax_of(want)) == float(want.abs().max())


def test_errors_are_loud(ops):
    from mapx.native import MapxError
    with pytest.raises(MapxError):
        ops.emb_gather(torch.zeros(3, dtype=torch.int64), torch.zeros(4, 16))      # CPU tensors
    with pytest.raises(MapxError):
        ops.nce_fwd(torch.zeros(2, 23 * 12, device=DEV), torch.zeros(2, 6, dtype=torch.int64, device=DEV),
                    torch.zeros(12, 26, dtype=torch.int32, device=DEV), torch.zeros(10, 12, device=DEV),
                    torch.zeros(10, device=DEV), torch.zeros(10, device=DEV), 23, 12)   # P = 12 unsupported


def test_catch_up_from_raw_repeating_ids(ops):
    """table_adam with rows_may_repeat: the raw batch id list (heavy repeats, hot id) brings every
    distinct stale row up to date exactly once."""
    from oracle import ref_model as R
    T, V, W = 30, 500, 16
    sched, lambdas = _sched(ops, T + 5)
    aux = ops.make_replay_aux(1e-3, lambdas, 0.9, 0.999, 0.05).to(DEV)
    g = torch.Generator().manual_seed(2)
    p0 = torch.randn(V, W, generator=g)
    m0, v0 = 0.01 * torch.randn(V, W, generator=g), 1e-4 * torch.rand(V, W, generator=g)
    pr, mr, vr = p0.clone(), m0.clone(), v0.clone()
    for s in range(1, T + 1):
        R.hf_adamw_step(pr, torch.zeros(V, W), mr, vr, s, 1e-3 * lambdas[s - 1], wd=0.05)
    p, m, v = p0.to(DEV).clone(), m0.to(DEV).clone(), v0.to(DEV).clone()
    last = torch.zeros(V, dtype=torch.int32, device=DEV)
    done = torch.full((1,), T, dtype=torch.int32, device=DEV)
    ids = torch.randint(0, 200, (50000,), generator=g)
    ids[:20000] = 3
    ops.table_adam(p, m, v, 0.05, last, sched, done, aux, 0.9, 0.999, 1e-8,
                   rows=ids.to(torch.int32).to(DEV), rows_may_repeat=True)
    touched = torch.zeros(V, dtype=torch.bool).index_fill_(0, ids.unique(), True)
    lc = _cpu(last)
    assert bool((lc[touched] == T).all()) and bool((lc[~touched] == 0).all())
    np.testing.assert_allclose(_cpu(p)[touched].numpy(), pr[touched].numpy(), rtol=1e-5, atol=1e-7)
    assert torch.equal(_cpu(p)[~touched], p0[~touched])
    # second call: everything current -> no change at all
    before = p.clone()
    ops.table_adam(p, m, v, 0.05, last, sched, done, aux, 0.9, 0.999, 1e-8,
                   rows=ids.to(torch.int32).to(DEV), rows_may_repeat=True)
    assert torch.equal(p, before)


@pytest.mark.parametrize("M,N", [(7, 368), (4096, 1000), (333, 68), (70, 4)])
def test_fused_elementwise_colsum(ops, M, N):
    g = torch.Generator().manual_seed(M + N)
    dy, y = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    dz, db = ops.relu_mask_colsum(dy.to(DEV), y.to(DEV))
    ref = dy * (y > 0)
    assert torch.equal(_cpu(dz), ref)
    np.testing.assert_allclose(_cpu(db).double().numpy(), ref.double().sum(0).numpy(), rtol=0,
                               atol=2e-6 * float(ref.abs().sum(0).max()) + 1e-6)
    x0, u, prev = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    t, dx0, dbc = ops.cross_bwd_pre_colsum(dy.to(DEV), x0.to(DEV), u.to(DEV), dx0=prev.to(DEV).clone())
    assert torch.equal(_cpu(t), dy * x0)
    np.testing.assert_allclose(_cpu(dx0).numpy(), (dy * u + prev).numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(_cpu(dbc).double().numpy(), (dy * x0).double().sum(0).numpy(), rtol=0,
                               atol=2e-6 * float((dy * x0).abs().sum(0).max()) + 1e-6)
    from mapx.native import MapxError
    with pytest.raises(MapxError):                       # float4 kernels: N % 4 == 0 only
        ops.relu_mask_colsum(torch.zeros(3, 65, device=DEV), torch.zeros(3, 65, device=DEV))


def test_rfd_replacement_generators(ops):
    """The four RFD_replace generators of trainer.py:233-262 (statistical properties)."""
    B, F, L, Ntrain, V = 2048, 23, 6, 5000, 10 + 23 * 100
    g = torch.Generator().manual_seed(8)
    lo = torch.arange(F) * 100 + 10
    x_train = (lo[None, :] + torch.randint(0, 100, (Ntrain, F), generator=g)).to(DEV)
    ids = x_train[:B].clone()
    ids_c = ids.cpu()
    field_of = lambda t: (t - 10) // 100
    # Uniform: inside the masked field's own [idx_low, idx_high)
    r, y, mi = ops.dynamic_mask_rfd(ids, L, seed=1, offset=1, mode="Uniform", idx_low=lo.to(DEV),
                                    idx_high=(lo + 100).to(DEV))
    assert torch.equal(field_of(r.cpu()), torch.arange(F).expand(B, F))
    assert torch.equal(y.cpu(), (r.cpu() != ids_c).float())
    # Whole-Uniform: anything in [10, V) -> mostly a foreign field
    r, y, mi = ops.dynamic_mask_rfd(ids, L, seed=1, offset=2, mode="Whole-Uniform", vocab=V)
    rc = r.cpu()
    changed = rc != ids_c
    assert int(rc.min()) >= 10 and int(rc.max()) < V
    foreign = (field_of(rc) != torch.arange(F).expand(B, F)) & changed
    assert 0.9 < float(foreign.sum()) / float(changed.sum()) <= 1.0
    # Whole-Unigram: ids that occur in the training matrix, from a random column
    r, y, mi = ops.dynamic_mask_rfd(ids, L, x_train=x_train, seed=1, offset=3, mode="Whole-Unigram")
    rc = r.cpu()
    changed = rc != ids_c
    foreign = (field_of(rc) != torch.arange(F).expand(B, F)) & changed
    assert 0.85 < float(foreign.sum()) / float(changed.sum()) <= 1.0
    with pytest.raises(NotImplementedError):
        ops.dynamic_mask_rfd(ids, L, x_train=x_train, mode="Bogus")
    from mapx.native import MapxError
    with pytest.raises(MapxError):
        ops.dynamic_mask_rfd(ids, L, mode="Uniform")          # idx_low / idx_high missing


@pytest.mark.parametrize("B,F,D", [(64, 23, 432), (4096, 23, 1368), (100, 39, 688)])
def test_grouped_feat_encoder_matches_dense(ops, arith, B, F, D):
    """Grouped forward / dW of feat_encoder == dense GEMM + field gather (models.py:74-75)."""
    g = torch.Generator().manual_seed(B + F)
    P, L = 32, int(F * 0.3)
    final = torch.randn(B, D, generator=g)
    w = torch.randn(F * P, D, generator=g) / math.sqrt(D)
    b = torch.randn(F * P, generator=g)
    mi = torch.randint(0, F, (B, L), generator=g)
    mi[0, 1] = mi[0, 0]
    if F > 30:
        mi[mi == 5] = 6                                   # a field nobody masks
    groups = ops.EncGroups(mi.to(DEV), F)
    h_slots = ops.enc_grouped_fwd(final.to(DEV), w.to(DEV), b.to(DEV), groups)
    enc = (final.double() @ w.double().t() + b.double()).view(B, F, P)
    want = torch.gather(enc, 1, mi.unsqueeze(-1).expand(-1, -1, P)).view(B * L, P)
    hpos = _cpu(groups.hpos).long()
    assert len(set(hpos.tolist())) == B * L               # every target has its own slot
    got = _cpu(h_slots)[hpos].double()
    bound = 2e-6 * (final.abs().double() @ w.abs().double().t()).max() + 1e-6
    assert float((got - want).abs().max()) <= float(bound)
    rowmap = _cpu(groups.rowmap)
    assert torch.equal(rowmap[hpos].long(), torch.arange(B).repeat_interleave(L))
    assert int((rowmap >= 0).sum()) == B * L
    # THE layout: slots ordered by (field, target index), every field's group padded to 128
    flat = mi.reshape(-1)
    counts = torch.bincount(flat, minlength=F)
    starts = torch.cumsum(torch.cat([torch.zeros(1, dtype=torch.long), (counts + 127) // 128 * 128]), 0)
    order = torch.sort(flat, stable=True).indices
    want_pos = torch.empty(B * L, dtype=torch.long)
    within = torch.arange(B * L) - torch.repeat_interleave(torch.cumsum(counts, 0) - counts, counts)
    want_pos[order] = starts[flat[order]] + within
    assert torch.equal(hpos, want_pos)
    assert torch.equal(_cpu(groups.group_start).long(), starts)
    tg = _cpu(groups.tile_group).long()
    want_tg = torch.full_like(tg, -1)
    for f_ in range(F):
        want_tg[starts[f_] // 128:starts[f_ + 1] // 128] = f_
    assert torch.equal(tg, want_tg)
    # dW from slot-ordered dh
    dh = torch.randn(B * L, P, generator=g)
    dh_slots = torch.zeros(groups.cap, P)
    dh_slots[hpos] = dh
    dw = ops.enc_grouped_dw(dh_slots.to(DEV), final.to(DEV), groups)
    denc = torch.zeros(B, F, P, dtype=torch.float64)
    denc.scatter_add_(1, mi.unsqueeze(-1).expand(-1, -1, P), dh.view(B, L, P).double())
    want_dw = denc.view(B, F * P).t() @ final.double()
    np.testing.assert_allclose(_cpu(dw).double().numpy(), want_dw.numpy(), rtol=0,
                               atol=4e-6 * float((denc.view(B, -1).abs().t() @ final.abs().double()).max()) + 1e-6)
    if F > 30:
        assert float(_cpu(dw)[5 * P:6 * P].abs().max()) == 0.0       # unmasked field: exact zeros


# ----------------------------------------------------------------------------- eval metrics
@pytest.mark.parametrize("n,kind", [(1000, "smooth"), (50000, "ties"), (200001, "saturated"), (2, "smooth")])
def test_eval_metrics_vs_sklearn(ops, n, kind):
    """Device AUC / log-loss == sklearn on float64 copies of the fp32 sigmoid (reference
    trainer.py:189-194), including heavy ties and saturated probabilities (p32 == 0 or 1)."""
    from sklearn.metrics import log_loss, roc_auc_score
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g) * 2.0
    if kind == "ties":
        x = (x * 4).round() / 4                                 # ~60 distinct scores
    if kind == "saturated":
        # far tails only (p32 == 1.0 exactly, or ~1e-22): in the band 14 < |x| < 17 one ulp of
        # expf moves -log(1 - p32) by O(1), so log-loss parity there would test expf, not us
        x = torch.where(x.abs() > 3.0, torch.sign(x) * 50.0, x)
    y = (torch.rand(n, generator=g) < torch.sigmoid(0.5 * x)).float()
    if kind == "saturated":                                     # confidently wrong rows: the clip matters
        y = torch.where(torch.rand(n, generator=g) < 0.05, 1.0 - y, y)
    if n == 2:
        y = torch.tensor([0.0, 1.0])
    got = ops.eval_metrics(x.to(DEV), y.to(DEV))
    probs = torch.sigmoid(x).numpy().astype("float64")
    assert got["positives"] == int(y.sum()) and got["negatives"] == n - int(y.sum())
    np.testing.assert_allclose(got["auc"], roc_auc_score(y.numpy(), probs), rtol=0, atol=2e-7 if kind == "smooth" else 1e-6)
    np.testing.assert_allclose(got["logloss"], log_loss(y.numpy(), probs), rtol=1e-5)
    np.testing.assert_allclose(got["avg_logits"], x.double().mean().item(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(got["avg_probs"], probs.mean(), rtol=1e-6)
    # exact self-consistency: AUC from the device's own fp32 probabilities, by definition
    p_dev = (1.0 / (1.0 + torch.exp(-x.to(DEV)))).cpu().double()
    if n <= 50000:
        pos, neg = p_dev[y > 0.5], p_dev[y <= 0.5]
        u = (pos[:, None] > neg[None, :]).double().sum() + 0.5 * (pos[:, None] == neg[None, :]).double().sum()
        want = float(u / (pos.numel() * neg.numel()))
        # identical unless torch's GPU expf and the kernel's differ in the last bit somewhere
        assert abs(got["auc"] - want) < 1e-7


def test_eval_metrics_one_class_raises(ops):
    x = torch.randn(100).to(DEV)
    with pytest.raises(ValueError, match="Only one class"):
        ops.eval_metrics(x, torch.ones(100).to(DEV))
    with pytest.raises(ValueError):
        ops.eval_metrics(x, torch.ones(99).to(DEV))


@pytest.mark.parametrize("B,F,E,H", [(3, 5, 4, 7), (64, 25, 16, 12), (17, 39, 16, 70)])
def test_cin_pieces(ops, B, F, E, H):
    """xDeepFM's CIN glue kernels against their one-line definitions (layers.py:714-718)."""
    g = torch.Generator().manual_seed(B * F + H)
    x3 = torch.randn(B, F, E, generator=g)
    xt = ops.transpose_batched(x3.to(DEV))
    assert torch.equal(_cpu(xt), x3.transpose(1, 2).contiguous())
    x0t = xt.view(B * E, F)
    xi = torch.randn(B * E, H, generator=g).to(DEV)
    had = ops.cin_outer_fwd(x0t, xi)
    ref = (_cpu(x0t)[:, :, None] * _cpu(xi)[:, None, :]).reshape(B * E, F * H)
    assert torch.equal(_cpu(had), ref)
    dhad = torch.randn(B * E, F * H, generator=g)
    dx0t = torch.full((B * E, F), 0.5, device=DEV)
    dxi = ops.cin_outer_bwd(dhad.to(DEV), x0t, xi, dx0t, accumulate_x0=True)
    d3 = dhad.view(B * E, F, H).double()
    np.testing.assert_allclose(_cpu(dxi).double().numpy(), (d3 * _cpu(x0t).double()[:, :, None]).sum(1).numpy(),
                               rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(_cpu(dx0t).double().numpy(),
                               0.5 + (d3 * _cpu(xi).double()[:, None, :]).sum(2).numpy(), rtol=1e-5, atol=1e-5)
    out = torch.full((B, H + 3), -1.0, device=DEV)
    ops.cin_pool_fwd(xi, B, E, out[:, 2:2 + H])
    np.testing.assert_allclose(_cpu(out[:, 2:2 + H]).numpy(), _cpu(xi).view(B, E, H).sum(1).numpy(), rtol=1e-5, atol=1e-5)
    assert bool((out[:, :2] == -1).all()) and bool((out[:, 2 + H:] == -1).all())
    gp = torch.randn(B, H + 3, generator=g).to(DEV)
    dxt = torch.ones(B * E, H, device=DEV)
    ops.cin_pool_bwd(gp[:, 1:1 + H], B, E, dxt, accumulate=True)
    assert torch.equal(_cpu(dxt).view(B, E, H), 1.0 + _cpu(gp[:, 1:1 + H])[:, None, :].expand(B, E, H))


def test_publish_i32_mailbox(ops):
    """mapx_publish_i32 into coherent host memory: values, stamp, checksum; stream-ordered and
    usable from a captured graph (the data-parallel step reads its segment counts this way)."""
    box = ops.HostMailbox(8)
    stamp = torch.zeros(1, dtype=torch.int32, device=DEV)
    src = torch.tensor([123456, 7, -3], dtype=torch.int32, device=DEV)
    ops.publish_i32(src, 3, stamp, box, at=1)
    torch.cuda.synchronize()
    a = box.np
    assert list(a[1:4]) == [123456, 7, -3] and int(a[4]) == 1 and int(a[5]) == 123456 + 7 - 3 + 1 and int(a[0]) == 0
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        src.add_(1)
        ops.publish_i32(src, 3, stamp, box, at=1)
    for k in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert list(a[1:4]) == [123459, 10, 0] and int(a[4]) == 4 and int(a[5]) == 123459 + 10 + 0 + 4
    with pytest.raises(ValueError):
        ops.publish_i32(src, 3, stamp, box, at=5)


# --------------------------------------------------------------------------- dropout / LayerNorm options
def test_dropout_mask_is_regenerated_not_stored(ops):
    """nn.Dropout semantics (layers.py:95,183): keep rate 1 - p, survivors scaled by 1 / (1 - p); the
    backward pass regenerates the very mask of the forward pass from (seed, offset); another offset
    gives another mask; the device-side offset adds to the host one (graph replay)."""
    n, p = 1_000_003, 0.3
    gen = torch.Generator(device=DEV).manual_seed(5)
    sign = lambda: torch.where(torch.rand(n, device=DEV, generator=gen) < 0.5, -1.0, 1.0)
    x = (torch.rand(n, device=DEV, generator=gen) + 0.25) * sign()        # never 0: "kept" is visible as y != 0
    y = ops.dropout(x, p, 42, 7)
    keep = y != 0
    assert abs(float(keep.float().mean()) - (1 - p)) < 3e-3
    assert torch.allclose(y[keep], x[keep] / (1 - p), rtol=1e-6)
    g = (torch.rand(n, device=DEV, generator=gen) + 0.25) * sign()
    assert torch.equal(ops.dropout(g, p, 42, 7) != 0, keep)                       # same mask in backward
    assert not torch.equal(ops.dropout(x, p, 42, 8) != 0, keep)
    dev = torch.tensor([3], dtype=torch.int32, device=DEV)
    assert torch.equal(ops.dropout(x, p, 42, 4, dev), y)                          # 4 + 3 == 7
    assert torch.equal(ops.dropout(x, 0.0, 1, 1), x)
    # independence of neighbouring elements / streams: correlation of two masks ~ 0
    a, b = keep.float() - (1 - p), (ops.dropout(x, p, 43, 7) != 0).float() - (1 - p)
    assert abs(float((a * b).mean())) < 2e-3


@pytest.mark.parametrize("R,E", [(94208, 16), (7, 16), (1000, 32), (33, 64), (5, 4)])
def test_layernorm_vs_torch(ops, R, E):
    g = torch.Generator().manual_seed(R + E)
    x = torch.randn(R, E, generator=g) * 2 + 0.5
    w, b = torch.randn(E, generator=g), torch.randn(E, generator=g)
    dy = torch.randn(R, E, generator=g)
    xr = x.clone().double().requires_grad_(True)
    wr, br = w.clone().double().requires_grad_(True), b.clone().double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (E,), wr, br, 1e-12)
    yr.backward(dy.double())
    y, stats = ops.layernorm_fwd(x.to(DEV), w.to(DEV), b.to(DEV), 1e-12)
    np.testing.assert_allclose(_cpu(y).numpy(), yr.detach().numpy(), rtol=2e-5, atol=2e-5)
    dx, dyx = ops.layernorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), stats)
    np.testing.assert_allclose(_cpu(dx).numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(_cpu(dyx).double().sum(0).numpy(), wr.grad.numpy(), rtol=1e-4, atol=1e-3)


@pytest.mark.gpu
def test_take_rows_cuts_a_batch_behind_a_device_cursor(ops):
    """mapx_take_rows_i64: rows sel[c : c + B] of a resident int64 matrix / vector (the RFD / finetune steps'
    collate); row numbers out of range are clamped, not read."""
    g = torch.Generator().manual_seed(3)
    X = torch.randint(0, 1 << 40, (1000, 23), generator=g).to(DEV)
    Y = torch.randint(0, 2, (1000,), generator=g).to(DEV)
    sel = torch.randperm(1000, generator=g).to(DEV)
    assert torch.equal(ops.take_rows(X, sel[:64].contiguous()), X[sel[:64]])
    cur = torch.tensor([128], dtype=torch.int64, device=DEV)
    assert torch.equal(ops.take_rows(X, sel, cur, 100), X[sel[128:228]])
    assert torch.equal(ops.take_rows(Y, sel, cur, 100), Y[sel[128:228]])
    yf = ops.take_rows(Y, sel, cur, 100, as_f32=True)
    assert yf.dtype == torch.float32 and torch.equal(yf, Y[sel[128:228]].float())
    assert ops.take_rows(X, sel[:0].contiguous()).shape == (0, 23)
    bad = torch.tensor([-5, 2000, 7], dtype=torch.int64, device=DEV)
    assert torch.equal(ops.take_rows(X, bad), X[torch.tensor([0, 999, 7], device=DEV)])
    with pytest.raises(TypeError):
        ops.take_rows(X.int(), sel)
    with pytest.raises(IndexError):
        ops.take_rows(X, sel[:10].contiguous(), batch=11)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(4096, 1, 1368), (4096, 23, 736), (777, 32, 64), (5, 7, 4), (64, 24, 2004),
                                   (8500, 23, 40), (94208, 40, 16), (9000, 64, 64), (10000, 12, 40), (4096, 39, 1248),
                                   (20000, 48, 200)])
def test_skinny_linear_kernels_vs_fp64(ops, monkeypatch, M, N, K):
    """Narrow layers (the finetune head's single output; the kernels take up to 32) run as fp32 streaming
    kernels (csrc/skinny.hip): forward (+ bias, ReLU, strided destination), weight gradient (row-chunk partials
    + mapx_sum_tasks, immediate and deferred), input gradient — against fp64.  (The dispatch keeps layers wider
    than ops.SKINNY_MAX = 8 on the GEMM in forward, where they measured faster; lifted here to test every width.
    M = 8192 rows per chunk > 64: the untiled weight-gradient kernel.)"""
    monkeypatch.setattr(ops, "SKINNY_MAX", 32)
    monkeypatch.setattr(ops, "SKINNY_MAX_BWD", 32)
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    dy = torch.randn(M, N, generator=g).to(DEV)
    if N > 32:                      # 33..64 outputs: weight and input gradients only
        monkeypatch.setattr(ops, "SKINNY_MAX_BWD", 64)
        monkeypatch.setattr(ops, "SKINNY_TALL", True)        # (the tall weight-gradient kernel is opt-in since round 4)
        dw_ref = dy.double().T @ x.double()
        dw = ops.linear_bwd_weight(dy, x)
        assert float((dw.double() - dw_ref).abs().max()) <= 3e-6 * float(dw_ref.abs().max())
        dx_ref = dy.double() @ w.double()
        dx = ops.linear_bwd_input(dy, w)
        assert float((dx.double() - dx_ref).abs().max()) <= 2e-6 * float(dx_ref.abs().max())
        return
    assert ops._skinny(N, K, x, w)
    ref = x.double() @ w.double().T + b.double()
    y = ops.linear_fwd(x, w, b)
    assert float((y.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    wide = torch.full((M, N + 9), -7.0, device=DEV)
    yr = ops.linear_fwd(x, w, b, relu=True, out=wide[:, 4:4 + N])
    assert yr.data_ptr() == wide[:, 4:].data_ptr() and torch.equal(yr, torch.relu(y))
    assert float(wide[:, :4].max()) == -7.0 and float(wide[:, 4 + N:].max()) == -7.0
    dw_ref = dy.double().T @ x.double()
    dw = ops.linear_bwd_weight(dy, x)
    assert float((dw.double() - dw_ref).abs().max()) <= 3e-6 * float(dw_ref.abs().max())
    slot = torch.zeros(N, K, device=DEV)
    ops.linear_bwd_weight(dy, x, out=slot, defer=True)
    ops.flush_deferred()
    assert torch.equal(slot, dw)                          # same partials, same sum
    dx_ref = dy.double() @ w.double()
    dx = ops.linear_bwd_input(dy, w)
    assert float((dx.double() - dx_ref).abs().max()) <= 2e-6 * float(dx_ref.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,D,H,plus_v", [(4096, 1, 368, 1000, False), (300, 3, 8, 12, True), (129, 8, 64, 4, False)])
def test_skinny_join_bwd_vs_fp64(ops, M, N, D, H, plus_v):
    """The narrow heads' input gradient over DCNv2's towers with both towers' first backward step in one streaming
    pass (mapx_skinny_join_bwd): every output and the per-tile partial sums against fp64."""
    g_ = torch.Generator().manual_seed(M + N + D)
    dz = torch.randn(M, N, generator=g_).to(DEV)
    w = torch.randn(N, D + H, generator=g_).to(DEV)
    final = torch.randn(M, D + H, generator=g_).to(DEV)
    x0 = torch.randn(M, D, generator=g_).to(DEV)
    u = torch.randn(M, D, generator=g_).to(DEV)
    g, t, dx0, dzr, pc, pd = ops.skinny_join_bwd(dz, w, final, D, x0, u, plus_v)
    v = dz.double() @ w.double()
    vc, vd = v[:, :D], v[:, D:]
    ref = dict(g=vc, t=vc * x0.double(), dx0=vc * u.double() + (vc if plus_v else 0.0),
               dzr=torch.where(final[:, D:] > 0, vd, torch.zeros_like(vd)))
    for name, got in (("g", g), ("t", t), ("dx0", dx0), ("dzr", dzr)):
        assert float((got.double() - ref[name]).abs().max()) <= 2e-6 * max(1.0, float(ref[name].abs().max())), name
    assert pc.shape == ((M + 127) // 128, D) and pd.shape == ((M + 127) // 128, H)
    assert float((pc.double().sum(0) - ref["t"].sum(0)).abs().max()) <= 1e-4 * max(1.0, float(ref["t"].sum(0).abs().max()))
    assert float((pd.double().sum(0) - ref["dzr"].sum(0)).abs().max()) <= 1e-4 * max(1.0, float(ref["dzr"].sum(0).abs().max()))
