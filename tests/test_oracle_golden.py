"""Pin the oracle (oracle/ref_model.py) against golden vectors captured from the real
reference modules (tests/golden/gen_golden.py).  CPU only."""
import math

import numpy as np
import pytest
import torch

import paramgen as pg
from oracle import ref_model as R
from util import assert_digest, load_case, t

CASES = [(c, "DCNv2") for c in pg.CASES] + [("B_f25_b64", b) for b in pg.BACKBONES[1:]]   # (case, backbone)


def _params(params, requires_grad=True):
    return {k: t(v).requires_grad_(requires_grad) for k, v in params.items()}


@pytest.mark.parametrize("case,backbone", CASES)
def test_mfp_matches_reference(case, backbone):
    cfg, z, inp, params = load_case(case, "MFP", backbone)
    P = _params(params)
    ids, mi = t(inp["input_ids"]), t(inp["masked_index"])
    masked, labels = R.dynamic_mask_mfp(ids, mi)
    assert torch.equal(masked, t(z["in/input_ids_masked"]))
    assert torch.equal(labels, t(z["in/labels"]))
    logq, lnV, _ = R.nce_buffers(inp["feat_count"])
    np.testing.assert_allclose(logq.numpy(), z["nce/logprob_noise"], rtol=1e-6, atol=1e-6)
    assert lnV == pytest.approx(float(z["nce/norm_term"]))
    final = R.final_of(backbone, P, masked, cfg["NC"], cfg["NL"], pg.extras_of(backbone))
    loss, logits, acc = R.mfp_head(P, final, labels, mi, t(inp["noise"]), logq,
                                   cfg["F"], cfg["P"], cfg["K"])
    np.testing.assert_allclose(loss.item(), float(z["out/loss"]), rtol=2e-6)
    np.testing.assert_allclose(logits.detach().numpy(), z["out/logits"], rtol=1e-5, atol=2e-6)
    assert acc == int(z["out/total_acc"])
    assert labels.numel() == int(z["out/count"])
    loss.backward()
    for k, p in P.items():
        assert_digest(z, "grad", k, p.grad.numpy())


@pytest.mark.parametrize("case,backbone", CASES)
def test_rfd_matches_reference(case, backbone):
    cfg, z, inp, params = load_case(case, "RFD", backbone)
    P = _params(params)
    ids, mi = t(inp["input_ids"]), t(inp["masked_index"])
    replaced, labels = R.dynamic_mask_rfd(ids, mi, t(inp["replace_feat"]))
    assert torch.equal(replaced, t(z["in/input_ids_replaced"]))
    assert torch.equal(labels, t(z["in/labels"]))
    final = R.final_of(backbone, P, replaced, cfg["NC"], cfg["NL"], pg.extras_of(backbone))
    loss, count, acc, pos, logits = R.rfd_head(P, final, labels)
    np.testing.assert_allclose(loss.item(), float(z["out/loss"]), rtol=2e-6)
    if "out/logits" in z.files:
        np.testing.assert_allclose(logits.detach().numpy(), z["out/logits"], rtol=1e-5, atol=2e-6)
    assert count == int(z["out/count"])
    np.testing.assert_allclose(acc.item(), float(z["out/acc"]), rtol=1e-6)
    np.testing.assert_allclose(pos.item(), float(z["out/pos_ratio"]), rtol=1e-6)
    loss.backward()
    for k, p in P.items():
        assert_digest(z, "grad", k, p.grad.numpy())


@pytest.mark.parametrize("case,backbone", CASES)
def test_ctr_matches_reference(case, backbone):
    cfg, z, inp, params = load_case(case, "CTR", backbone)
    P = _params(params)
    ids = t(inp["input_ids"])
    x0 = R.embed(P, ids)
    np.testing.assert_array_equal(x0.detach().numpy(), z["mid/embed_flat"])
    if backbone == "DCNv2":
        np.testing.assert_allclose(R.cross(P, x0, cfg["NC"]).detach().numpy(), z["mid/cross_out"],
                                   rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(R.dnn(P, x0, cfg["NL"]).detach().numpy(), z["mid/dnn_out"],
                                   rtol=1e-5, atol=2e-6)
    if backbone == "DeepFM":
        np.testing.assert_allclose(R.lr_logit(P, ids).detach().numpy(), z["mid/lr"], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(R.fm_product_sum(P["embed.embedding.weight"][ids]).detach().numpy(),
                                   z["mid/fm"], rtol=1e-5, atol=2e-6)
    if backbone == "xDeepFM":
        units = [int(c) for c in pg.XDEEPFM["cin_layer_units"].split(",")]
        np.testing.assert_allclose(R.cin(P, P["embed.embedding.weight"][ids], units).detach().numpy(),
                                   z["mid/cin_out"], rtol=1e-5, atol=2e-6)
    if backbone == "AutoInt":
        x = P["embed.embedding.weight"][ids]
        ai = pg.AUTOINT
        for li in range(ai["num_attn_layers"]):
            x = R.autoint_layer(P, x, li, ai["num_attn_heads"], ai["attn_size"], ai["res_conn"], ai["attn_scale"])
            np.testing.assert_allclose(x.detach().numpy(), z[f"mid/attn{li}"], rtol=1e-5, atol=2e-6)
    logits = R.ctr_logits_of(backbone, P, ids, cfg["NC"], cfg["NL"], pg.extras_of(backbone))
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits.view(-1), t(inp["y"]).float())
    np.testing.assert_allclose(loss.item(), float(z["out/loss"]), rtol=2e-6)
    np.testing.assert_allclose(logits.detach().numpy(), z["out/logits"], rtol=1e-5, atol=2e-6)
    loss.backward()
    for k, p in P.items():
        assert_digest(z, "grad", k, p.grad.numpy())


@pytest.mark.parametrize("case", list(pg.CASES))
def test_alias_table_and_bias_init(case):
    cfg, z, inp, _ = load_case(case, "MFP")
    logq, lnV, q = R.nce_buffers(inp["feat_count"])
    prob, alias = R.alias_build(q.numpy())
    np.testing.assert_array_equal(alias, z["nce/alias_alias"])
    np.testing.assert_allclose(prob, z["nce/alias_prob"], rtol=0, atol=0)
    # the table encodes the renormalised noise distribution
    np.testing.assert_allclose(R.alias_distribution(prob, alias), q.double().numpy(),
                               rtol=0, atol=5e-7)
    np.testing.assert_allclose((logq + lnV).numpy()[:, None], z["nce/bias_init"], rtol=1e-6,
                               atol=1e-6)
    assert float(z["nce/emb_init_absmax"]) <= 1.0 / math.sqrt(cfg["P"]) + 1e-7
    assert float(z["init/embed_std"]) == pytest.approx(math.sqrt(2.0 / (cfg["F"] + cfg["E"])),
                                                       rel=0.05)


def test_alias_draw_semantics():
    prob = torch.tensor([1.0, 0.25, 0.5])
    alias = torch.tensor([0, 0, 1])
    kk = torch.tensor([0, 1, 1, 2, 2])
    u = torch.tensor([0.99, 0.1, 0.3, 0.49, 0.5])
    assert R.alias_draw(prob, alias, kk, u).tolist() == [0, 1, 0, 2, 1]


def test_lr_schedules_match_transformers(golden_dir):
    import os
    z = np.load(os.path.join(golden_dir, "lr_schedules.npz"))
    for name in z.files:
        kind, T, W = name.split("_")
        T, W = int(T[1:]), int(W[1:])
        kind = "cosine" if kind == "cos" else "const"
        got = [R.lr_lambda(kind, s, T, W) for s in range(len(z[name]))]
        np.testing.assert_allclose(got, z[name], rtol=1e-12, atol=1e-15, err_msg=name)


def test_hf_adamw_known_answer():
    """Hand-computed (float64) two-step trajectory of transformers-4.26 AdamW semantics:
    m,v EMA; denom = sqrt(v)+eps (eps NOT bias-scaled); step = lr*sqrt(1-b2^t)/(1-b1^t);
    decay p -= lr*wd*p applied after the update on the already-updated p."""
    lr, b1, b2, eps, wd = 1e-2, 0.9, 0.999, 1e-8, 0.1
    p0, g1, g2 = 0.5, 0.2, -0.4
    m1, v1 = (1 - b1) * g1, (1 - b2) * g1 * g1
    s1 = lr * math.sqrt(1 - b2) / (1 - b1)
    p1 = p0 - s1 * m1 / (math.sqrt(v1) + eps)
    p1 = p1 - lr * wd * p1
    m2, v2 = b1 * m1 + (1 - b1) * g2, b2 * v1 + (1 - b2) * g2 * g2
    s2 = lr * math.sqrt(1 - b2 ** 2) / (1 - b1 ** 2)
    p2 = p1 - s2 * m2 / (math.sqrt(v2) + eps)
    p2 = p2 - lr * wd * p2
    p, m, v = torch.tensor([p0]), torch.zeros(1), torch.zeros(1)
    R.hf_adamw_step(p, torch.tensor([g1]), m, v, 1, lr, b1, b2, eps, wd)
    assert p.item() == pytest.approx(p1, rel=1e-6)
    R.hf_adamw_step(p, torch.tensor([g2]), m, v, 2, lr, b1, b2, eps, wd)
    assert p.item() == pytest.approx(p2, rel=1e-6)
    assert m.item() == pytest.approx(m2, rel=1e-6) and v.item() == pytest.approx(v2, rel=1e-6)
    # first step of Adam moves by ~lr regardless of gradient scale
    assert abs((p0 - p1 / (1 - lr * wd)) - lr) < 1e-6


def test_decay_groups():
    assert R.decays("embed.embedding.weight") and R.decays("mfp_criterion.emb.weight")
    assert not R.decays("mfp_criterion.bias.weight")       # name contains "bias"
    assert not R.decays("cross_net.cross_layers.0.bias")
    assert R.decays("parallel_dnn.dnn.3.weight")


# --------------------------------------------------------------------------- vocabulary builders
@pytest.mark.parametrize("name", ["avazu", "criteo"])
def test_vocab_oracle_matches_reference_preprocessing(golden_dir, name):
    """oracle/vocab.py against the feat_map that the reference's own generate_dataset()
    (data_preprocess/proc_avazu.py, proc_criteo.py) wrote for the fixture's columns: every key, every id, in order,
    and every row's translation."""
    import os
    from oracle import vocab as V
    z = np.load(os.path.join(golden_dir, f"vocab_{name}.npz"))
    cols = {str(n): z[f"col/{n}"].tolist() for n in z["names"]}
    feat_map, rows = V.build_feat_map(cols, int(z["n_core"]))
    assert list(feat_map.keys()) == [str(k) for k in z["feat_map_keys"]]
    assert list(feat_map.values()) == z["feat_map_ids"].tolist() == list(range(int(z["input_size"])))
    assert np.array_equal(np.array(rows, dtype=np.int64), z["feat_ids"])


# --------------------------------------------------------------------------- hidden_act other than relu
ACT_KINDS = ["tanh", "sigmoid", "none", "elu", "leu", "gelu", "gelu_new", "swish", "mish"]


@pytest.mark.parametrize("kind", ACT_KINDS)
def test_activation_restatement_matches_reference(golden_dir, kind):
    """oracle act() / dnn(hidden_act=) against outputs and gradients of the reference's own get_act classes and
    MLPBlock (tests/golden/activations.npz, from code/layers.py:13-80, 173-188)."""
    import os
    z = np.load(os.path.join(golden_dir, "activations.npz"))
    x = torch.from_numpy(z["x"]).requires_grad_(True)
    y = R.act(kind, x)
    y.backward(torch.ones_like(y))
    np.testing.assert_allclose(y.detach().numpy(), z[f"{kind}/y"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(x.grad.numpy(), z[f"{kind}/dy_dx"], rtol=1e-5, atol=1e-7)
    params = {f"parallel_dnn.{k.split('/p/')[1]}": torch.from_numpy(z[k]).requires_grad_(True)
              for k in z.files if k.startswith(f"mlp/{kind}/p/")}
    xin = torch.from_numpy(z["mlp/x"]).requires_grad_(True)
    out = R.dnn(params, xin, 2, hidden_act=kind)
    (out * torch.linspace(-1, 1, out.numel()).view_as(out)).sum().backward()
    np.testing.assert_allclose(out.detach().numpy(), z[f"mlp/{kind}/y"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(xin.grad.numpy(), z[f"mlp/{kind}/dx"], rtol=1e-4, atol=1e-6)
    for k, p in params.items():
        np.testing.assert_allclose(p.grad.numpy(), z[f"mlp/{kind}/g/{k.split('parallel_dnn.')[1]}"], rtol=1e-4, atol=1e-6)
