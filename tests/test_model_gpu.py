"""Model-level parity on the GPU: mapx.DCNV2 (HIP kernels through the C ABI) against
(1) the committed golden vectors captured from the real reference and (2) the oracle on
larger seeded inputs, for the three modes of the hot path.  fp32 tolerance = north star 1e-5
relative on logits/loss; gradients 2e-5 relative to their scale."""
import numpy as np
import pytest
import torch

import paramgen as pg
from util import assert_digest, build_model, load_case, t

pytestmark = pytest.mark.gpu
DEV = "cuda"
CASES = list(pg.CASES)
OTHER = [("B_f25_b64", b) for b in pg.BACKBONES[1:]]      # DNN, DeepFM


def _dense_named_grads(model):
    tab = model.table_parameter_ids()
    return {n: p.grad for n, p in model.named_parameters() if id(p) not in tab}


def _check_grads(z, model):
    for n, g in _dense_named_grads(model).items():
        assert g is not None, n
        assert_digest(z, "grad", n, g.cpu().numpy())
    names = {id(p): n for n, p in model.named_parameters()}
    for table in model.row_tables():
        g0, g1 = table.dense_grad()
        assert_digest(z, "grad", names[id(table.p0)], g0.cpu().numpy())
        if g1 is not None:
            assert_digest(z, "grad", names[id(table.p1)], g1.cpu().numpy())


@pytest.mark.parametrize("case", CASES)
def test_mfp_golden(case):
    cfg, z, inp, params = load_case(case, "MFP")
    model = build_model(cfg, "MFP", params, inp["feat_count"])
    model.mfp_criterion.return_logits = True
    np.testing.assert_allclose(model.mfp_criterion.logprob_noise.cpu().numpy(), z["nce/logprob_noise"],
                               rtol=1e-6, atol=1e-6)
    assert np.array_equal(model.mfp_criterion.alias.alias.cpu().numpy(), z["nce/alias_alias"])
    assert np.array_equal(model.mfp_criterion.alias.prob.cpu().numpy(), z["nce/alias_prob"])
    from mapx import ops
    ids = t(inp["input_ids"], DEV)
    mi = t(inp["masked_index"], DEV)
    masked, labels, _ = ops.dynamic_mask_mfp(ids, mi.shape[1], masked_index=mi)
    assert np.array_equal(masked.cpu().numpy(), z["in/input_ids_masked"])
    assert np.array_equal(labels.cpu().numpy(), z["in/labels"])
    model.train()
    loss, count, acc = model(input_ids=masked, labels=labels, masked_index=mi,
                             noise_samples=t(inp["noise"], DEV))
    np.testing.assert_allclose(float(loss.detach()), float(z["out/loss"]), rtol=1e-5)
    assert count == int(z["out/count"]) and int(acc) == int(z["out/total_acc"])
    loss.backward()
    _check_grads(z, model)


@pytest.mark.parametrize("case", CASES)
def test_mfp_logits_golden(case):
    cfg, z, inp, params = load_case(case, "MFP")
    model = build_model(cfg, "MFP", params, inp["feat_count"])
    model.mfp_criterion.return_logits = True
    model.eval()
    with torch.no_grad():
        feat = model.embed(t(z["in/input_ids_masked"], DEV)).flatten(1)
        final = torch.cat([model.cross_net(feat), model.parallel_dnn(feat)], -1)
        enc = model.feat_encoder(final)
        _, logits, idx = model.mfp_criterion(t(z["in/labels"], DEV), enc,
                                             masked_index=t(inp["masked_index"], DEV),
                                             noise_samples=t(inp["noise"], DEV))
    np.testing.assert_allclose(logits.cpu().numpy(), z["out/logits"], rtol=1e-5, atol=1e-5)
    assert np.array_equal(idx.cpu().numpy(), z["out/indices"])


@pytest.mark.parametrize("case", CASES)
def test_rfd_golden(case):
    cfg, z, inp, params = load_case(case, "RFD")
    model = build_model(cfg, "RFD", params, None)
    from mapx import ops
    ids, mi = t(inp["input_ids"], DEV), t(inp["masked_index"], DEV)
    replaced, labels, _ = ops.dynamic_mask_rfd(ids, mi.shape[1], masked_index=mi,
                                               replace_feat=t(inp["replace_feat"], DEV))
    assert np.array_equal(replaced.cpu().numpy(), z["in/input_ids_replaced"])
    assert np.array_equal(labels.cpu().numpy(), z["in/labels"])
    model.train()
    loss, count, acc, pos = model(input_ids=replaced, labels=labels)
    np.testing.assert_allclose(float(loss.detach()), float(z["out/loss"]), rtol=1e-5)
    assert count == int(z["out/count"])
    np.testing.assert_allclose(float(acc), float(z["out/acc"]), rtol=1e-6)
    np.testing.assert_allclose(float(pos), float(z["out/pos_ratio"]), rtol=1e-6)
    loss.backward()
    _check_grads(z, model)


@pytest.mark.parametrize("case", CASES)
def test_ctr_golden(case):
    cfg, z, inp, params = load_case(case, "CTR")
    model = build_model(cfg, "CTR", params, None)
    ids = t(inp["input_ids"], DEV)
    model.train()
    feat = model.embed(ids).flatten(1)
    assert np.array_equal(feat.detach().cpu().numpy(), z["mid/embed_flat"])           # bit-exact gather
    np.testing.assert_allclose(model.cross_net(feat).detach().cpu().numpy(), z["mid/cross_out"],
                               rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(model.parallel_dnn(feat).detach().cpu().numpy(), z["mid/dnn_out"],
                               rtol=1e-5, atol=1e-5)
    loss, logits = model(input_ids=ids, labels=t(inp["y"], DEV))
    np.testing.assert_allclose(float(loss.detach()), float(z["out/loss"]), rtol=1e-5)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), z["out/logits"], rtol=1e-5, atol=1e-5)
    loss.backward()
    _check_grads(z, model)
    (logits_only,) = model(input_ids=ids)
    np.testing.assert_allclose(logits_only.detach().cpu().numpy(), z["out/logits"], rtol=1e-5, atol=1e-5)


def _oracle_params(model):
    return {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point and "alias" not in k
                                                      and "logprob" not in k)
            for k, v in model.state_dict().items()}


@pytest.mark.parametrize("B,F", [(4096, 23), (777, 39)])
def test_mfp_full_batch_vs_oracle(B, F):
    """BASELINE batch size (4096 x 23, K=25, P=32, H=1000) with generated masks and negatives:
    the GPU step's sampled indices are fed to the oracle, outputs must agree."""
    from mapx import ops
    from mapx.dataset import synth_table
    from oracle import ref_model as R
    from util import make_config
    from mapx.models import BaseModel
    cfg = dict(F=F, V=60000, E=16, H=1000, NL=3, NC=3, P=32, K=25)
    ids_np, _, _, _ = synth_table(B, F, cfg["V"], seed=1)
    cnt = np.bincount(ids_np.reshape(-1), minlength=cfg["V"]).astype(np.float32)
    torch.manual_seed(0)
    model = BaseModel.from_config(make_config(cfg, "MFP", cnt)).to(DEV)
    model.mfp_criterion.return_logits = True
    L = int(F * 0.3)
    ids = torch.from_numpy(ids_np).to(DEV)
    masked, labels, mi = ops.dynamic_mask_mfp(ids, L, seed=7, offset=1)
    model.train()
    feat = model.embed(masked).flatten(1)
    final = torch.cat([model.cross_net(feat), model.parallel_dnn(feat)], -1)
    enc = model.feat_encoder(final)
    loss, logits, idx = model.mfp_criterion(labels, enc, masked_index=mi)
    acc = int(model.mfp_criterion.last_acc)
    loss.backward()

    P = _oracle_params(model)
    noise = idx[..., 1:].long().cpu()
    assert torch.equal(idx[..., 0].long().cpu(), labels.cpu())
    logq, _, _ = R.nce_buffers(cnt)
    fin = R.trunk(P, masked.cpu(), cfg["NC"], cfg["NL"])
    loss_r, logits_r, acc_r = R.mfp_head(P, fin, labels.cpu(), mi.cpu(), noise, logq, F, cfg["P"], cfg["K"])
    loss_r.backward()
    np.testing.assert_allclose(float(loss), float(loss_r), rtol=1e-5)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), logits_r.detach().numpy(), rtol=1e-5, atol=2e-5)
    assert abs(acc - acc_r) <= 2                      # exact ties aside
    names = {id(p): n for n, p in model.named_parameters()}
    for n, g in _dense_named_grads(model).items():
        ref = P[n].grad
        np.testing.assert_allclose(g.cpu().numpy(), ref.numpy(), rtol=1e-3, atol=1e-4 * float(ref.abs().max()),
                                   err_msg=n)
    for table in model.row_tables():
        g0, g1 = table.dense_grad()
        ref0 = P[names[id(table.p0)]].grad
        np.testing.assert_allclose(g0.cpu().numpy(), ref0.numpy(), rtol=1e-3, atol=1e-4 * float(ref0.abs().max()))
        if g1 is not None:
            ref1 = P[names[id(table.p1)]].grad
            np.testing.assert_allclose(g1.cpu().numpy(), ref1.numpy(), rtol=1e-3,
                                       atol=1e-4 * float(ref1.abs().max()))


def test_eval_mode_needs_no_plan_and_matches_train_forward():
    cfg, z, inp, params = load_case("A_f23_b7", "CTR")
    model = build_model(cfg, "CTR", params, None)
    ids = t(inp["input_ids"], DEV)
    model.eval()
    with torch.no_grad():
        (a,) = model(input_ids=ids)
    assert model.embed.table.plan is None
    model.train()
    (b,) = model(input_ids=ids)
    assert torch.equal(a, b.detach())


def test_unknown_backbones_raise_like_the_reference():
    from mapx.models import BaseModel
    from util import make_config
    cfg = pg.CASES["A_f23_b7"]
    for name in ("fignn", "nonsense"):
        c = make_config(cfg, "CTR")
        c.model_name = name
        with pytest.raises(NotImplementedError):
            BaseModel.from_config(c)


@pytest.mark.parametrize("E,NC", [(4, 2), (16, 1), (16, 0)])
def test_cross_tower_depths_vs_oracle(E, NC):
    """Narrow rows, a hidden width that is not a multiple of 4, and 0 / 1 / 2 cross layers:
    forward and every gradient of the DCNv2 trunk + CTR head against the oracle."""
    from mapx.models import BaseModel
    from oracle import ref_model as R
    from util import make_config
    cfg = dict(F=5, V=300, E=E, H=10, NL=2, NC=NC, P=32, K=5)
    torch.manual_seed(3)
    model = BaseModel.from_config(make_config(cfg, "CTR", None)).to(DEV)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, cfg["V"], (9, cfg["F"]), generator=g)
    y = torch.randint(0, 2, (9,), generator=g)
    loss, logits = model(input_ids=ids.to(DEV), labels=y.to(DEV))
    loss.backward()
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()
              if v.dtype.is_floating_point}
    fin = R.trunk(params, ids, NC, cfg["NL"])
    loss_ref, logits_ref = R.ctr_head(params, fin, y)
    loss_ref.backward()
    np.testing.assert_allclose(logits.detach().cpu().numpy().ravel(), logits_ref.detach().numpy().ravel(),
                               rtol=1e-5, atol=1e-6)
    from mapx.optim import decays  # noqa: F401  (import check only)
    for name, p in model.named_parameters():
        if name == "embed.embedding.weight":
            got = model.embed.table.dense_grad()[0].cpu()
        else:
            got = p.grad.cpu()
        np.testing.assert_allclose(got.numpy(), params[name].grad.numpy(), rtol=2e-5, atol=2e-6, err_msg=name)


# ----------------------------------------------------------------------------- other backbones (§8 f4)
@pytest.mark.parametrize("case,backbone", OTHER)
@pytest.mark.parametrize("mode", ["MFP", "RFD", "CTR"])
def test_other_backbones_golden(case, backbone, mode):
    """DNN / DeepFM through the same C-ABI kernels, heads and row tables vs golden vectors of the
    reference's own classes (models.py:164-233): loss, outputs and every gradient (for DeepFM
    including the LR weight rows, reduced together with the embedding rows)."""
    from mapx import ops
    cfg, z, inp, params = load_case(case, mode, backbone)
    model = build_model(cfg, mode, params, inp["feat_count"] if mode == "MFP" else None, backbone=backbone)
    ids, mi = t(inp["input_ids"], DEV), t(inp["masked_index"], DEV)
    model.train()
    if mode == "MFP":
        masked, labels, _ = ops.dynamic_mask_mfp(ids, mi.shape[1], masked_index=mi)
        loss, count, acc = model(input_ids=masked, labels=labels, masked_index=mi, noise_samples=t(inp["noise"], DEV))
        assert count == int(z["out/count"]) and int(acc) == int(z["out/total_acc"])
    elif mode == "RFD":
        replaced, labels, _ = ops.dynamic_mask_rfd(ids, mi.shape[1], masked_index=mi,
                                                   replace_feat=t(inp["replace_feat"], DEV))
        loss, count, acc, pos = model(input_ids=replaced, labels=labels)
        np.testing.assert_allclose(float(acc), float(z["out/acc"]), rtol=1e-6)
        np.testing.assert_allclose(float(pos), float(z["out/pos_ratio"]), rtol=1e-6)
    else:
        loss, logits = model(input_ids=ids, labels=t(inp["y"], DEV))
        np.testing.assert_allclose(logits.detach().cpu().numpy(), z["out/logits"], rtol=1e-5, atol=1e-5)
        if backbone == "DeepFM":
            x3, lr = model.embed.forward_with_linear(ids, model.lr_layer.embed_w.weight)
            np.testing.assert_allclose((lr.view(-1, 1) + model.lr_layer.bias).detach().cpu().numpy(), z["mid/lr"],
                                       rtol=1e-5, atol=1e-6)
            from mapx.layers import fm_product_sum
            # 0.5 * sum_e((sum_f x)^2 - sum_f x^2) cancels: absolute tolerance at the scale of its two terms
            np.testing.assert_allclose(fm_product_sum(x3).detach().cpu().numpy(), z["mid/fm"], rtol=1e-5, atol=5e-6)
        if backbone == "AutoInt":
            x = model.embed(ids)
            for li, layer in enumerate(model.self_attention):
                x = layer(x)
                np.testing.assert_allclose(x.detach().cpu().numpy(), z[f"mid/attn{li}"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(float(loss.detach()), float(z["out/loss"]), rtol=1e-5)
    loss.backward()
    _check_grads(z, model)
