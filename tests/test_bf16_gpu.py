"""bf16 compute mode (BASELINE configs[2], north star tolerance 1e-2) on the GPU, through the C ABI:
the bf16 MFMA GEMM family and its elementwise companions against fp64 products of the SAME bf16
operands (so the only differences are fp32 accumulation order and the bf16 rounding of the result),
exact-integer checks of every operand layout (a wrong fragment map cannot hide behind a tolerance),
then the model: golden fixtures of the real reference, the fp32 oracle at the BASELINE batch size,
and an 8-step AdamW trajectory, all within 1e-2."""
import numpy as np
import pytest
import torch

import paramgen as pg
from util import build_model, check_pattern as _check_pattern, hook_relu_pattern as _hook_relu_pattern, load_case, \
    oracle_case_grads as _oracle_grads, t

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16
EPS_BF = 2.0 ** -7          # bf16: 8 significant bits, one ulp = 2^-7 relative; round to nearest errs by <= half of it


def _operands(a_kc, b_kc, M, N, K, seed, ints=False):
    g = torch.Generator().manual_seed(seed)
    if ints:
        A = torch.randint(-3, 4, (M, K) if a_kc else (K, M), generator=g).float()
        B = torch.randint(-3, 4, (N, K) if b_kc else (K, N), generator=g).float()
    else:
        A = torch.randn((M, K) if a_kc else (K, M), generator=g)
        B = torch.randn((N, K) if b_kc else (K, N), generator=g)
    A, B = A.to(BF), B.to(BF)
    Am = (A if a_kc else A.t()).double()
    Bm = (B.t() if b_kc else B).double()
    return A, B, Am, Bm


LAYOUTS = [(True, True), (True, False), (False, False)]


@pytest.mark.parametrize("a_kc,b_kc", LAYOUTS)
@pytest.mark.parametrize("tile", [0, 1, 2])
def test_gemm_bf16_exact_on_integers(a_kc, b_kc, tile):
    """Small-integer operands: every product and partial sum is exact in fp32, so the fp32 result is
    bit-exact whatever the contraction order — but only if every fragment lane holds the element
    the MFMA thinks it holds (asymmetric operands: a transposed map cannot pass)."""
    from mapx import ops
    for (M, N, K) in [(128, 128, 64), (200, 136, 192), (64, 72, 72), (333, 40, 24)]:
        A, B, Am, Bm = _operands(a_kc, b_kc, M, N, K, seed=M + N + K, ints=True)
        ref = (Am @ Bm).float()
        out = ops.gemm_bf16(A.to(DEV), B.to(DEV), a_kc, b_kc, M, N, K, out_dtype=torch.float32, tile=tile)
        assert torch.equal(out.cpu(), ref), (a_kc, b_kc, tile, M, N, K, float((out.cpu() - ref).abs().max()))


@pytest.mark.parametrize("a_kc,b_kc", LAYOUTS)
def test_gemm_bf16_shapes_vs_fp64(a_kc, b_kc):
    """The step's shapes (Avazu / Criteo widths, B = 4096), ragged rows, unaligned extents (the
    scalar-load path), K tails, every tile choice; fp32 and bf16 outputs."""
    from mapx import ops
    shapes = [(4096, 1000, 368), (4096, 368, 368), (4096, 736, 1368), (777, 1248, 1624), (4096, 1000, 1000),
              (7, 1000, 368), (64, 39, 1248), (777, 1248, 39), (4096, 1, 1368), (130, 23, 736), (9, 5, 3),
              (257, 129, 65), (1000, 1000, 4096) if not a_kc else (512, 1000, 1000)]
    for (M, N, K) in shapes:
        A, B, Am, Bm = _operands(a_kc, b_kc, M, N, K, seed=M * 7 + N * 3 + K)
        ref = Am @ Bm
        bound = Am.abs() @ Bm.abs()
        for tile in (-1, 0, 2):
            o32 = ops.gemm_bf16(A.to(DEV), B.to(DEV), a_kc, b_kc, M, N, K, out_dtype=torch.float32, tile=tile)
            e = ((o32.double().cpu() - ref).abs() / bound.clamp(min=1e-30)).max().item()
            assert e < 2e-6, (M, N, K, tile, e)                        # fp32 accumulation of exact bf16 products
            o16 = ops.gemm_bf16(A.to(DEV), B.to(DEV), a_kc, b_kc, M, N, K, out_dtype=BF, tile=tile)
            assert o16.dtype == BF
            d = (o16.double().cpu() - ref).abs()
            assert bool((d <= 0.5 * EPS_BF * ref.abs() + 2e-6 * bound + 1e-30).all()), (M, N, K, tile, float(d.max()))


def test_gemm_bf16_split_k_weight_gradient():
    from mapx import ops
    for (M, N, K, ns) in [(1000, 368, 4096, 4), (368, 368, 4096, 16), (736, 1368, 4096, 2), (23, 736, 4096, 8),
                          (64, 64, 300, 4)]:
        A, B, Am, Bm = _operands(False, False, M, N, K, seed=M + N)
        ref, bound = Am @ Bm, Am.abs() @ Bm.abs()
        out = ops.gemm_bf16(A.to(DEV), B.to(DEV), False, False, M, N, K, out_dtype=torch.float32, nsplit=ns)
        e = ((out.double().cpu() - ref).abs() / bound.clamp(min=1e-30)).max().item()
        assert e < 2e-6, (M, N, K, ns, e)
        again = ops.gemm_bf16(A.to(DEV), B.to(DEV), False, False, M, N, K, out_dtype=torch.float32, nsplit=ns)
        assert torch.equal(out, again)                # slab order is fixed: bit-reproducible


@pytest.mark.parametrize("M,N,K", [(4096, 368, 368), (100, 400, 400), (33, 624, 624)])
def test_gemm_bf16_epilogues(M, N, K):
    """bias / bias+ReLU / cross (Xi + X0 * (acc + b), u = acc + b kept) / add (fp32 and bf16 addend) /
    ReLU mask, on bf16 and fp32 results, into strided destinations."""
    from mapx import ops
    from mapx.native import EPI_ADD, EPI_BIAS, EPI_BIAS_CROSS, EPI_BIAS_RELU, EPI_RELU_MASK
    g = torch.Generator().manual_seed(M + N)
    A, B, Am, Bm = _operands(True, True, M, N, K, seed=5)
    acc = Am @ Bm
    bound = Am.abs() @ Bm.abs()
    bias = torch.randn(N, generator=g)
    Ad, Bd, bd = A.to(DEV), B.to(DEV), bias.to(DEV)

    def close(got, want, slack=0.0):
        d = (got.double().cpu() - want).abs()
        tol = (0.5 * EPS_BF * want.abs() if got.dtype == BF else 0.0) + 4e-6 * (bound + bias.abs().double()) + slack
        assert bool((d <= tol + 1e-30).all()), float((d - tol).max())

    for dt in (BF, torch.float32):
        close(ops.gemm_bf16(Ad, Bd, True, True, M, N, K, out_dtype=dt, epi=EPI_BIAS, bias=bd), acc + bias.double())
        close(ops.gemm_bf16(Ad, Bd, True, True, M, N, K, out_dtype=dt, epi=EPI_BIAS_RELU, bias=bd),
              (acc + bias.double()).clamp(min=0))
    # strided destination (a column slice of a wider buffer, as the towers write the concat buffer)
    wide = torch.full((M, N + 40), 7.0, dtype=BF, device=DEV)
    ops.gemm_bf16(Ad, Bd, True, True, M, N, K, out=ops.alias_cols(wide, 24, N), epi=EPI_BIAS_RELU, bias=bd)
    close(wide[:, 24:24 + N], (acc + bias.double()).clamp(min=0))
    assert bool((wide[:, :24] == 7).all()) and bool((wide[:, 24 + N:] == 7).all())
    if N == K:          # cross layer: A is Xi [M, D], B is W [D, D]
        x0 = torch.randn(M, N, generator=g).to(BF)
        u_want = acc + bias.double()
        u = torch.empty(M, N, dtype=BF, device=DEV)
        y = ops.gemm_bf16(Ad, Bd, True, True, M, N, K, epi=EPI_BIAS_CROSS, bias=bd, aux1=Ad, aux2=x0.to(DEV), out2=u)
        close(u, u_want)
        d = (y.double().cpu() - (A.double() + x0.double() * u_want)).abs()
        tol = 0.5 * EPS_BF * (A.double() + x0.double() * u_want).abs() + 4e-6 * (1 + x0.double().abs()) * (bound + 1)
        assert bool((d <= tol).all())
    add32 = torch.randn(M, N, generator=g)
    add16 = add32.to(BF)
    close(ops.gemm_bf16(Ad, Bd, True, True, M, N, K, epi=EPI_ADD, aux1=add32.to(DEV)), acc + add32.double(),
          slack=4e-6 * add32.abs().double())
    close(ops.gemm_bf16(Ad, Bd, True, True, M, N, K, epi=EPI_ADD, aux1=add16.to(DEV)), acc + add16.double(),
          slack=4e-6 * add16.abs().double())
    ymask = torch.randn(M, N, generator=g).clamp(min=0).to(BF)
    close(ops.gemm_bf16(Ad, Bd, True, True, M, N, K, epi=EPI_RELU_MASK, aux1=ymask.to(DEV)),
          torch.where(ymask.double() > 0, acc, torch.zeros_like(acc)))


def test_bf16_elementwise_backward_kernels():
    """relu-mask + column sum, cross-layer pre-products + running fp32 dL/dX0 + column sum, plain
    column sum, casts: bf16 in, fp32 sums, any width / leading dimension."""
    from mapx import ops
    g = torch.Generator().manual_seed(3)
    for (M, N, pad) in [(4096, 1000, 0), (777, 368, 24), (64, 23, 0), (5, 1, 3), (130, 401, 7)]:
        dy_w = torch.randn(M, N + pad, generator=g).to(BF).to(DEV)
        dy = dy_w[:, pad // 2:pad // 2 + N] if pad else dy_w
        y = torch.randn(M, N, generator=g).clamp(min=0).to(BF).to(DEV)
        dz, db = ops.relu_mask_colsum(dy, y)
        want = torch.where(y > 0, dy, torch.zeros_like(dy))
        assert torch.equal(dz, want)
        np.testing.assert_allclose(db.cpu().numpy(), want.double().sum(0).cpu().numpy(), rtol=1e-5,
                                   atol=1e-5 * float(want.double().abs().sum(0).max()) + 1e-30)
        assert torch.equal(ops.relu_mask(dy.contiguous(), y), want)
        np.testing.assert_allclose(ops.colsum(dy).cpu().numpy(), dy.double().sum(0).cpu().numpy(), rtol=1e-5,
                                   atol=1e-5 * float(dy.double().abs().sum(0).max()))
        x0 = torch.randn(M, N, generator=g).to(BF).to(DEV)
        u = torch.randn(M, N, generator=g).to(BF).to(DEV)
        tt, dx0, dbc = ops.cross_bwd_pre_colsum(dy, x0, u, plus_g=True)
        t_want = (dy.float() * x0.float()).to(BF)
        assert torch.equal(tt, t_want) and dx0.dtype == torch.float32
        np.testing.assert_allclose(dx0.cpu().numpy(), (dy.float() * u.float() + dy.float()).cpu().numpy(), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(dbc.cpu().numpy(), t_want.double().sum(0).cpu().numpy(), rtol=1e-5,
                                   atol=1e-5 * float(t_want.double().abs().sum(0).max()) + 1e-30)
        first = dx0.clone()
        _, dx0b, _ = ops.cross_bwd_pre_colsum(dy, x0, u, dx0=dx0)             # accumulate into the running sum
        np.testing.assert_allclose(dx0b.cpu().numpy(), (first + dy.float() * u.float()).cpu().numpy(), rtol=1e-6, atol=1e-6)
    x = torch.randn(100003, generator=g).to(DEV)
    assert torch.equal(ops.cast_bf16(x), x.to(BF))                            # round to nearest even, like torch
    assert torch.equal(ops.cast_f32(x.to(BF)), x.to(BF).float())
    nan = torch.tensor([float("nan"), float("inf"), -float("inf"), 0.0] * 2, device=DEV)
    out = ops.cast_bf16(nan)
    assert bool(torch.isnan(out[0])) and bool(torch.isinf(out[1])) and bool(torch.isinf(out[2]))


def test_bf16_gather_and_row_gradient():
    """Embedding rows leave the gather as bf16 (the fp32 row rounded once); bf16 gradient rows are
    summed per id in fp32, in the plan's fixed order (bit-reproducible)."""
    from mapx import ops
    g = torch.Generator().manual_seed(1)
    V, E, B, F = 5000, 16, 512, 23
    table = torch.randn(V, E, generator=g).to(DEV)
    ids = torch.randint(0, V, (B, F), generator=g)
    ids[:, 0] = 3                                             # a hot id (<mask>)
    x = ops.emb_gather(ids.to(DEV), table, out_dtype=BF)
    assert x.dtype == BF and torch.equal(x, table[ids.to(DEV)].to(BF))
    grad = torch.randn(B * F, E, generator=g).to(BF).to(DEV)
    plan = ops.SegPlan(ops.ids_to_i32(ids.to(DEV), V), V)
    rows = ops.seg_reduce_rows(plan, grad, E)
    U = plan.count()
    want = torch.zeros(V, E, dtype=torch.float64, device=DEV).index_add_(0, ids.view(-1).to(DEV), grad.double())
    got = torch.zeros(V, E, dtype=torch.float64, device=DEV).index_copy_(0, plan.uniq[:U].long(), rows[:U].double())
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=2e-5)
    plan2 = ops.SegPlan(ops.ids_to_i32(ids.to(DEV), V), V)
    assert torch.equal(ops.seg_reduce_rows(plan2, grad, E)[:U], rows[:U])


# ------------------------------------------------------------------------------------------- model level
def _rel(got, want):
    """max |got - want| relative to the tensor's scale."""
    want = np.asarray(want, dtype=np.float64)
    return float(np.abs(np.asarray(got, dtype=np.float64) - want).max() / max(np.abs(want).max(), 1e-30))


def _grads_vs(model, ref_grads, tol, what):
    tab = model.table_parameter_ids()
    names = {id(p): n for n, p in model.named_parameters()}
    worst = ("", 0.0)
    for n, p in model.named_parameters():
        if id(p) in tab:
            continue
        e = _rel(p.grad.float().cpu().numpy(), ref_grads[n])
        worst = max(worst, (n, e), key=lambda x: x[1])
        assert e <= tol, f"{what}: {n} differs by {e:.3e} of its scale"
    for table in model.row_tables():
        g0, g1 = table.dense_grad()
        for gt, p in ((g0, table.p0), (g1, table.p1)):
            if gt is not None:
                e = _rel(gt.cpu().numpy(), ref_grads[names[id(p)]])
                worst = max(worst, (names[id(p)], e), key=lambda x: x[1])
                assert e <= tol, f"{what}: {names[id(p)]} differs by {e:.3e} of its scale"
    return worst


@pytest.mark.parametrize("case", list(pg.CASES))
@pytest.mark.parametrize("mode", ["MFP", "RFD", "CTR"])
def test_bf16_model_vs_golden_and_oracle(case, mode):
    """All nine DCNv2 fixture cases of the real reference in bf16 compute mode: loss and logits within
    1e-2 of the golden values (north star: 1e-2 bf16), every gradient within 1e-2 of its scale of the
    fp32 oracle's (which the fp32 tests pin to the same fixtures at 2e-5) on the step's own ReLU
    pattern — bf16 rounding moves a pre-activation that is zero to 2^-8 across the kink, and with 7
    or 64 rows one such unit is a visible share of a gradient row; the pattern may differ from the
    fp32 one only at such units (checked)."""
    from mapx import ops
    cfg, z, inp, params = load_case(case, mode)
    model = build_model(cfg, mode, params, inp["feat_count"] if mode == "MFP" else None, compute_dtype="bf16")
    masks = _hook_relu_pattern(model)
    ids, mi = t(inp["input_ids"], DEV), t(inp["masked_index"], DEV)
    model.train()
    if mode == "MFP":
        model.mfp_criterion.return_logits = True
        masked, labels, _ = ops.dynamic_mask_mfp(ids, mi.shape[1], masked_index=mi)
        assert np.array_equal(masked.cpu().numpy(), z["in/input_ids_masked"])
        feat = model.embed(masked).flatten(1)
        assert feat.dtype == BF
        final = torch.cat([model.cross_net(feat), model.parallel_dnn(feat)], -1)
        enc = model.feat_encoder(final)
        assert final.dtype == BF and enc.dtype == torch.float32
        loss, logits, _ = model.mfp_criterion(labels, enc, masked_index=mi, noise_samples=t(inp["noise"], DEV))
        assert _rel(logits.detach().cpu().numpy(), z["out/logits"]) <= 1e-2
    elif mode == "RFD":
        replaced, labels, _ = ops.dynamic_mask_rfd(ids, mi.shape[1], masked_index=mi,
                                                   replace_feat=t(inp["replace_feat"], DEV))
        loss, count, acc, pos = model(input_ids=replaced, labels=labels)
        assert abs(float(acc) - float(z["out/acc"])) <= 0.02
    else:
        loss, logits = model(input_ids=ids, labels=t(inp["y"], DEV))
        assert logits.dtype == torch.float32
        assert _rel(logits.detach().cpu().numpy(), z["out/logits"]) <= 1e-2
    assert abs(float(loss.detach()) - float(z["out/loss"])) <= 1e-2 * abs(float(z["out/loss"]))
    loss.backward()
    pre = {}
    loss_ref, _, _ = _oracle_grads(mode, cfg, params, inp, preacts=pre)
    assert abs(loss_ref - float(z["out/loss"])) <= 1e-5 * abs(loss_ref)       # the oracle IS the golden path
    flips = _check_pattern(masks, pre, f"{case}/{mode}")
    from util import note_golden
    note_golden(f"bf16/DCNv2/{mode}/{case}", "bf16-pattern", flips)      # bf16 mode is always compared on its own pattern
    _, _, ref_grads = _oracle_grads(mode, cfg, params, inp, relu_masks=masks)
    worst = _grads_vs(model, ref_grads, 1e-2, f"{case}/{mode}")
    print(f"[bf16 {case} {mode}] loss {float(loss.detach()):.6f} vs {float(z['out/loss']):.6f}; {flips} of "
          f"{sum(v.numel() for v in pre.values())} ReLU units on the other side of the kink; "
          f"worst gradient {worst[1]:.2e} of scale at {worst[0]}")


@pytest.mark.parametrize("mode,B,F", [("MFP", 4096, 23), ("RFD", 4096, 23), ("CTR", 4096, 23), ("MFP", 777, 39)])
def test_bf16_full_batch_vs_fp32_oracle(mode, B, F):
    """BASELINE batch (4096 x 23, K = 25, P = 32, H = 1000; and a ragged 777 x 39 Criteo-width batch) in bf16
    mode against the fp32 oracle on the step's own masks / negatives / replacements: loss, logits, every
    gradient within 1e-2."""
    from mapx import ops
    from mapx.dataset import synth_table
    from mapx.models import BaseModel
    from oracle import ref_model as R
    from util import make_config
    cfg = dict(F=F, V=60000, E=16, H=1000, NL=3, NC=3, P=32, K=25)
    ids_np, y_np, _, _ = synth_table(B, F, cfg["V"], seed=1)
    cnt = np.bincount(ids_np.reshape(-1), minlength=cfg["V"]).astype(np.float32)
    torch.manual_seed(0)
    model = BaseModel.from_config(make_config(cfg, mode, cnt if mode == "MFP" else None, compute_dtype="bf16")).to(DEV)
    L = int(F * 0.3)
    ids = torch.from_numpy(ids_np).to(DEV)
    masks = _hook_relu_pattern(model)
    model.train()
    P = {k: (v.detach().cpu().clone().requires_grad_(True) if v.dtype.is_floating_point and "alias" not in k
             and "logprob" not in k else v.detach().cpu()) for k, v in model.state_dict().items()}
    pre = {}
    if mode == "MFP":
        model.mfp_criterion.return_logits = True
        masked, labels, mi = ops.dynamic_mask_mfp(ids, L, seed=7, offset=1)
        feat = model.embed(masked).flatten(1)
        final = torch.cat([model.cross_net(feat), model.parallel_dnn(feat)], -1)
        loss, out, idx = model.mfp_criterion(labels, model.feat_encoder(final), masked_index=mi)
        logq = R.nce_buffers(cnt)[0]
        head = lambda **kw: R.mfp_head(P, R.trunk(P, masked.cpu(), 3, 3, **kw), labels.cpu(), mi.cpu(),
                                       idx[..., 1:].long().cpu(), logq, F, 32, 25)[:2]
    elif mode == "RFD":
        x_train = torch.from_numpy(synth_table(3 * B, F, cfg["V"], seed=2)[0]).to(DEV)
        replaced, labels, _ = ops.dynamic_mask_rfd(ids, L, x_train=x_train, seed=7, offset=1, mode="Unigram")
        model.pred_rfd["2"].register_forward_hook(lambda m, i, o: setattr(model, "_rfd_logits", o.detach()))
        loss = model(input_ids=replaced, labels=labels)[0]
        out = model._rfd_logits

        def head(**kw):
            r = R.rfd_head(P, R.trunk(P, replaced.cpu(), 3, 3, **kw), labels.cpu(), **kw)
            return r[0], r[4]
    else:
        y = torch.from_numpy(y_np).to(DEV)
        loss, out = model(input_ids=ids, labels=y)
        head = lambda **kw: R.ctr_head(P, R.trunk(P, ids.cpu(), 3, 3, **kw), y.cpu())
    loss.backward()
    with torch.no_grad():
        head(preacts=pre)
    flips = _check_pattern(masks, pre, f"B=4096 {mode}")
    loss_r, out_r = head(relu_masks=masks)           # the fp32 oracle on the step's own ReLU pattern
    loss_r.backward()
    assert abs(float(loss.detach()) - float(loss_r)) <= 1e-2 * abs(float(loss_r))
    assert _rel(out.detach().float().cpu().numpy().reshape(out_r.shape), out_r.detach().numpy()) <= 1e-2
    worst = _grads_vs(model, {k: v.grad.numpy() for k, v in P.items() if torch.is_tensor(v) and v.requires_grad},
                      1e-2, f"B=4096 {mode}")
    print(f"[bf16 B=4096 {mode}] loss {float(loss.detach()):.6f} vs fp32 oracle {float(loss_r):.6f}; {flips} of "
          f"{sum(v.numel() for v in pre.values())} ReLU units on the other side of the kink; "
          f"worst gradient {worst[1]:.2e} of scale at {worst[0]}")


@pytest.mark.parametrize("mode,kind", [("MFP", "cosine"), ("RFD", "cosine"), ("CTR", "const")])
def test_bf16_training_trajectory(mode, kind):
    """8 AdamW steps in bf16 mode against the fp32 oracle running the reference's dense optimizer:
    losses within 1e-2.  Parameters: an Adam step moves a weight by ~lr = 1e-3 whatever the gradient's
    size, so where a gradient is rounding noise the two runs walk apart by up to lr per step — the
    worst element is bounded by steps x lr (8e-3) and says nothing more; what the mode must keep is
    the UPDATE as a whole: per tensor, the 8-step displacement agrees with the oracle's in direction
    (cosine >= 0.98) and size (relative L2 error <= 0.2).  The bf16 weight shadows equal the rounded
    fp32 master weights after every step."""
    from test_trainer_gpu import _oracle_loop, _step_inputs, _targs
    from mapx import ops
    from mapx.optim import MapxOptimizer
    case = "B_f25_b64"
    cfg = pg.CASES[case]
    _, _, inp, params = load_case(case, mode)
    steps, total, lr0, wd = 8, 10, 1e-3, 5e-2
    ref_losses, ref_params = _oracle_loop(mode, cfg, params, steps, case, total, 0, kind, lr0, wd)
    model = build_model(cfg, mode, params, inp["feat_count"] if mode == "MFP" else None, compute_dtype="bf16")
    opt = MapxOptimizer(model, _targs(lr_sched=kind, learning_rate=lr0, weight_decay=wd), num_training_steps=total,
                        num_warmup_steps=0)
    assert opt.bf16
    model.train()
    L = int(cfg["F"] * cfg["mask_ratio"])
    losses = []
    for s in range(steps):
        si = _step_inputs(case, cfg, s)
        ids = t(si["ids"], DEV)
        if mode == "MFP":
            masked, labels, mi = ops.dynamic_mask_mfp(ids, L, masked_index=t(si["mi"], DEV))
            loss = model(input_ids=masked, labels=labels, masked_index=mi, noise_samples=t(si["noise"], DEV))[0]
        elif mode == "RFD":
            rep, labels, _ = ops.dynamic_mask_rfd(ids, L, masked_index=t(si["mi"], DEV), replace_feat=t(si["repl"], DEV))
            loss = model(input_ids=rep, labels=labels)[0]
        else:
            loss = model(input_ids=ids, labels=t(si["y"], DEV))[0]
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        for p in opt.dense_params:
            assert torch.equal(p._mapx_bf16, p.detach().to(BF))
    np.testing.assert_allclose(losses, ref_losses, rtol=1e-2)
    opt.flush()
    sd = model.state_dict()
    worst = max(float((sd[k].cpu() - ref).abs().max()) for k, ref in ref_params.items())
    assert worst <= steps * lr0 * 1.1, worst
    stats = {}
    for k, ref in ref_params.items():
        d_ref = (ref - t(params[k])).double().view(-1)
        d_got = (sd[k].cpu() - t(params[k])).double().view(-1)
        if float(d_ref.norm()) == 0:
            continue
        cos = float(d_got @ d_ref / (d_got.norm() * d_ref.norm()))
        rel = float((d_got - d_ref).norm() / d_ref.norm())
        stats[k] = (cos, rel)
        assert cos >= 0.98 and rel <= 0.2, f"{k}: update cosine {cos:.4f}, relative L2 error {rel:.3f}"
    lo = min(stats, key=lambda k: stats[k][0])
    print(f"[bf16 trajectory {mode}] losses {losses[0]:.5f}..{losses[-1]:.5f} (oracle {ref_losses[0]:.5f}..{ref_losses[-1]:.5f}); "
          f"largest parameter drift {worst:.2e}; least aligned update {lo}: cosine {stats[lo][0]:.4f}, rel L2 {stats[lo][1]:.3f}")


def test_bf16_graph_replay_equals_eager_bitwise_and_checkpoint_is_fp32():
    """The captured step in bf16 mode == the eager step, bit for bit; the checkpoint holds the fp32
    master weights under the reference's keys (no bf16 tensor leaks into {step}.model)."""
    import os
    from mapx.arguments import TrainingArguments
    from mapx.dataset import OurDataset, synth_table
    from mapx.models import BaseModel
    from mapx.trainer import Trainer
    from util import make_config
    cfg = dict(F=23, V=3000, E=16, H=64, NL=3, NC=3, P=32, K=25)
    ids, labels, _, _ = synth_table(512 * 5 + 100, 23, cfg["V"], seed=3)
    cnt = np.bincount(ids.reshape(-1), minlength=cfg["V"]).astype(np.float32)
    out = []
    for use_graph in (True, False):
        torch.manual_seed(5)
        config = make_config(cfg, "MFP", cnt, compute_dtype="bf16")
        model = BaseModel.from_config(config)
        targs = TrainingArguments(output_dir="/tmp/mapx_bf16_graph_test", per_gpu_train_batch_size=512,
                                  per_gpu_eval_batch_size=512, learning_rate=1e-3, lr_sched="cosine",
                                  weight_decay=5e-2, num_train_epochs=2, pretrain=True, pt_type="MFP",
                                  sampling_method="randint", mask_ratio=0.3, logging_steps=7, seed=11)
        targs._device = torch.device(DEV)
        os.makedirs(targs.output_dir, exist_ok=True)
        tr = Trainer(model, config, targs, OurDataset(ids, labels), OurDataset(ids[:600], labels[:600]))
        tr.use_graph = use_graph
        tr.MFP_pretrain()
        assert (len(tr._graphs) == 1 and not isinstance(next(iter(tr._graphs.values())), int)) == use_graph
        out.append({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
        ck = torch.load(os.path.join(targs.output_dir, f"{tr.global_step}.model"))
        assert all(v.dtype != BF for v in ck.values()) and set(ck) == set(out[-1])
    for k in out[0]:
        assert torch.equal(out[0][k], out[1][k]), k


@pytest.mark.parametrize("M,N,K", [(4096, 1000, 1000), (300, 368, 200)])
def test_gemm_bf16_relu_mask_colsum_epilogue(M, N, K):
    """EPI_RELU_MASK_COLSUM on the bf16 family: masked bf16 result + fp32 partial rows (one per 128-row tile)
    whose sum is the column sum of the bf16 values stored; and the woven K loop against hipcc's own order."""
    from mapx import ops
    from mapx.native import EPI_RELU_MASK_COLSUM
    g = torch.Generator().manual_seed(M + N)
    dy, w = torch.randn(M, K, generator=g).to(BF), torch.randn(K, N, generator=g).to(BF)
    y = torch.randn(M, N, generator=g).to(BF)
    part = torch.full(((M + 127) // 128, N), 7.0, dtype=torch.float32, device=DEV)
    dz = ops.gemm_bf16(dy.to(DEV), w.to(DEV), True, False, M, N, K, epi=EPI_RELU_MASK_COLSUM, aux1=y.to(DEV), out2=part)
    ref = (dy.double() @ w.double()) * (y.double() > 0)
    bound = dy.abs().double() @ w.abs().double()
    d = (dz.double().cpu() - ref).abs()
    assert bool((d <= 0.5 * EPS_BF * ref.abs() + 4e-6 * bound + 1e-30).all())
    got, want = part.double().cpu().sum(0), dz.double().cpu().sum(0)
    assert bool(((got - want).abs() <= 1e-6 * dz.double().cpu().abs().sum(0) + 1e-6).all())



@pytest.mark.parametrize("M,N,K", [(4096, 1, 1368), (777, 3, 1368), (4096, 8, 736), (5, 1, 8), (130, 4, 2024)])
def test_streaming_narrow_layers_in_bf16_mode(M, N, K, monkeypatch):
    """Layers of at most 8 outputs over bf16 activations (the finetune head of the bf16 trunk, models.py:304,319) on
    the streaming kernels' bf16 forms (csrc/skinny.hip: mapx_skinny_linear_{fwd,dw,dx}_bf16): forward (fp32 logits),
    weight gradient (fp32) and input gradient (bf16) against fp64 products of the same bf16 operands, and against the
    bf16 MFMA GEMM path they replace (MAPX_SKINNY_BF16=0) — the same products, another summation order."""
    from mapx import ops
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(BF)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF)
    b = torch.randn(N, generator=g)
    dy = torch.randn(M, N, generator=g).to(BF)
    xd, wd, bd, dyd = x.to(DEV), w.to(DEV), b.to(DEV), dy.to(DEV)
    used = []
    real = {n: getattr(ops.lib, n) for n in ("mapx_skinny_linear_fwd_bf16", "mapx_skinny_linear_dw_bf16",
                                              "mapx_skinny_linear_dx_bf16")}
    out = {}
    for arm in (True, False):
        monkeypatch.setattr(ops, "SKINNY_BF16", arm)
        y = ops.linear_fwd(xd, wd, bd, out_dtype=torch.float32)
        dw = ops.linear_bwd_weight(dyd, xd)
        dx = ops.linear_bwd_input(dyd, wd)
        assert y.dtype == torch.float32 and dw.dtype == torch.float32 and dx.dtype == BF
        out[arm] = (y.cpu(), dw.cpu(), dx.cpu())
    yr = x.double() @ w.double().t() + b.double()
    dwr = dy.double().t() @ x.double()
    dxr = dy.double() @ w.double()
    for arm in (True, False):
        y, dw, dx = out[arm]
        assert bool(((y.double() - yr).abs() <= 2e-6 * (x.double().abs() @ w.double().abs().t()) + 1e-6).all()), arm
        assert bool(((dw.double() - dwr).abs() <= 4e-6 * (dy.double().abs().t() @ x.double().abs()) + 1e-6).all()), arm
        bound = EPS_BF * dxr.abs() + 4e-6 * (dy.double().abs() @ w.double().abs()) + 1e-6
        assert bool(((dx.double() - dxr).abs() <= bound).all()), arm
    # the streaming forms really ran: their results are fp32 sums in another order than the MFMA's
    for n in real:
        assert hasattr(ops.lib, n)


def test_bf16_streaming_kernels_reject_what_they_cannot_take():
    from mapx import ops
    from mapx.native import MapxError
    x = torch.zeros(8, 16, dtype=BF, device=DEV)
    w = torch.zeros(9, 16, dtype=BF, device=DEV)
    y = torch.zeros(8, 9, device=DEV)
    with pytest.raises(MapxError):
        ops.check(ops.lib.mapx_skinny_linear_fwd_bf16(x.data_ptr(), 16, w.data_ptr(), 16, None, 8, 9, 16, 0, y.data_ptr(), 9,
                                                      None))
