"""Deterministic inputs / parameters shared by the fixture generator and the tests.

Pure numpy, no torch RNG: both `gen_golden.py` (which runs the real reference, in the
build container only) and the parity tests (which run anywhere) rebuild bit-identical
parameters and inputs from a case description, so fixtures only need to hold OUTPUTS.
"""
import zlib

import numpy as np

RESERVED = 10          # ids 0..9 are <pad>,<cls>,<sep>,<mask>=3,<unused0-5>  
MASK_ID = 3
FULL_MAX = 20000       # tensors up to this many elements are stored whole in a fixture

# The fixture cases: canonical run-script shapes for A, reduced hidden size for B/C so
# that fixtures stay small.  F = 23 (paper Avazu), 25 (proc_avazu.py layout), 39 (Criteo).
CASES = {
    "A_f23_b7": dict(F=23, B=7, V=1000, E=16, H=1000, NL=3, NC=3, P=32, K=25, mask_ratio=0.3),
    "B_f25_b64": dict(F=25, B=64, V=1000, E=16, H=64, NL=3, NC=3, P=32, K=25, mask_ratio=0.3),
    "C_f39_b64": dict(F=39, B=64, V=1000, E=16, H=64, NL=2, NC=2, P=32, K=25, mask_ratio=0.3),
}


def _rng(*key):
    return np.random.default_rng(zlib.crc32("/".join(str(k) for k in key).encode()))


def field_ranges(F, V):
    """Contiguous per-field id ranges over [RESERVED, V) with a skewed size profile."""
    w = np.array([1.0 + (i % 5) ** 2 + (7.0 if i % 4 == 0 else 0.0) for i in range(F)])
    sizes = np.maximum(2, np.floor(w / w.sum() * (V - RESERVED)).astype(np.int64))
    sizes[-1] += (V - RESERVED) - sizes.sum()
    assert sizes.min() >= 2 and sizes.sum() == V - RESERVED
    lo = RESERVED + np.concatenate([[0], np.cumsum(sizes)[:-1]])
    return lo.astype(np.int64), (lo + sizes).astype(np.int64)


def make_inputs(case, cfg):
    """input_ids [B,F] (global ids, field f inside its range, low ids more frequent),
    masked_index [B,L] (with replacement -> duplicates, like `randint` masking),
    noise [B,L,K], ctr labels [B], feat_count [V] (contains zeros -> clamp path)."""
    F, B, V, K = cfg["F"], cfg["B"], cfg["V"], cfg["K"]
    L = int(F * cfg["mask_ratio"])
    r = _rng(case, "inputs")
    lo, hi = field_ranges(F, V)
    u = r.random((B, F))
    ids = (lo[None, :] + np.floor((u ** 2.5) * (hi - lo)[None, :])).astype(np.int64)
    masked_index = r.integers(0, F, size=(B, L)).astype(np.int64)
    masked_index[0, 1] = masked_index[0, 0]          # force at least one duplicate
    noise = r.integers(0, V, size=(B, L, K)).astype(np.int64)
    noise[0, 0, :3] = ids[0, masked_index[0, 0]]     # noise that collides with the target
    y = (r.random(B) < 0.3).astype(np.int64)
    cnt = np.floor(r.pareto(1.2, size=V) * 5.0).astype(np.float32)
    cnt[:RESERVED] = 0.0
    cnt[r.integers(RESERVED, V, size=V // 10)] = 0.0  # never-seen ids -> BACKOFF clamp
    cnt[ids.reshape(-1)] += 1.0
    # RFD replacement ids: a column drawn from other rows (Unigram flavour)
    repl = ids[r.integers(0, B, size=(B, L)), masked_index]
    return dict(input_ids=ids, masked_index=masked_index, noise=noise, y=y,
                feat_count=cnt, replace_feat=repl.astype(np.int64))


def make_param(case, name, shape, scale):
    return (_rng(case, "param", name).standard_normal(shape) * scale).astype(np.float32)


BACKBONES = ("DCNv2", "DNN", "DeepFM", "AutoInt", "xDeepFM")      # the others: SURVEY §8(f4), fixtures for case B only
# AutoInt settings of the fixtures (non-default on purpose: 2 heads, residual + scaling on, the
# second layer has no W_res because its input width equals heads * attn_size)
AUTOINT = dict(num_attn_layers=2, attn_size=12, num_attn_heads=2, res_conn=True, attn_scale=True,
               attn_probs_dropout_rate=0.0, use_lr=False, num_dnn_layers=0, dnn_size=1000, dnn_act="relu",
               dnn_drop=0.0)


# AutoInt's finetune-only options (models.py:463-471, 482-486): the LR term and an MLP tower over the embeddings
# (the reference sizes the tower's input as num_fields * heads * attn_size and feeds it the flattened EMBEDDINGS,
# models.py:466, 486: the option only runs with embed_size == heads * attn_size — 2 x 8 = 16 here)
AUTOINT_FULL = dict(AUTOINT, attn_size=8, use_lr=True, num_dnn_layers=2, dnn_size=40)
CTR_ONLY = ("AutoIntFull",)                       # fixture variants that exist for the CTR mode only
MODEL_NAME = {"AutoIntFull": "AutoInt"}          # fixture variant -> the reference's model_name


def model_name_of(backbone):
    return MODEL_NAME.get(backbone, backbone)


# xDeepFM settings of the fixtures: two CIN layers of different widths; the CTR model adds the LR term
XDEEPFM = dict(cin_layer_units="12,8", use_lr=True)
# xDeepFM without the MLP tower (models.py:246-255: final_dim = the CIN's output alone)
XDEEPFM_CIN = dict(XDEEPFM, num_hidden_layers=0)
MODEL_NAME["xDeepFMCin"] = "xDeepFM"
EXTRAS = {"AutoInt": AUTOINT, "AutoIntFull": AUTOINT_FULL, "xDeepFM": XDEEPFM, "xDeepFMCin": XDEEPFM_CIN}


def extras_of(backbone):
    """Backbone-specific config keys of the fixtures ({} for DCNv2 / DNN / DeepFM)."""
    return dict(EXTRAS.get(backbone, {}))


def param_shapes(cfg, mode, backbone="DCNv2"):
    """state_dict key of every TRAINABLE parameter -> (shape, init scale) for `backbone` in `mode`
    in {MFP, RFD, CTR}.  Key layout = SURVEY §8(b) checkpoint manifest (DCNv2) and the reference
    modules' attribute names (models.py:164-233) for DNN / DeepFM."""
    F, V, E, H, NL, NC, P = (cfg[k] for k in ("F", "V", "E", "H", "NL", "NC", "P"))
    D = F * E
    out = {"embed.embedding.weight": ((V, E), (2.0 / (F + E)) ** 0.5)}
    if backbone == "DCNv2":
        for i in range(NC):
            out[f"cross_net.cross_layers.{i}.weight"] = ((D, D), D ** -0.5)
            out[f"cross_net.cross_layers.{i}.bias"] = ((D,), 0.1)
    elif backbone == "DeepFM":
        out["lr_layer.embed_w.weight"] = ((V, 1), 0.3)
        out["lr_layer.bias"] = ((1,), 0.1)
    elif backbone in ("AutoInt", "AutoIntFull"):
        AI = EXTRAS[backbone]
        HA = AI["num_attn_heads"] * AI["attn_size"]
        d_in = E
        for i in range(AI["num_attn_layers"]):
            for nm in ("W_q", "W_k", "W_v") + (("W_res",) if d_in != HA else ()):
                out[f"self_attention.{i}.{nm}.weight"] = ((HA, d_in), d_in ** -0.5)
            d_in = HA
        Dfin = F * HA
        if mode == "CTR":
            out["attn_out.weight"] = ((1, Dfin), Dfin ** -0.5)
            out["attn_out.bias"] = ((1,), 0.1)
            if backbone == "AutoIntFull":
                out["lr_layer.embed_w.weight"] = ((V, 1), 0.3)
                out["lr_layer.bias"] = ((1,), 0.1)
                Hd, d_in = AUTOINT_FULL["dnn_size"], Dfin          # (models.py:466: input_dim = final_dim)
                for i in range(AUTOINT_FULL["num_dnn_layers"]):
                    out[f"dnn.dnn.{3 * i}.weight"] = ((Hd, d_in), d_in ** -0.5)
                    out[f"dnn.dnn.{3 * i}.bias"] = ((Hd,), 0.1)
                    d_in = Hd
                out["dnn_out.weight"] = ((1, Hd), Hd ** -0.5)
                out["dnn_out.bias"] = ((1,), 0.1)
            return out
        if backbone == "AutoIntFull":
            raise ValueError("AutoIntFull is a finetune (CTR) fixture")
        NL = 0                                       # no MLP tower
    elif backbone in ("xDeepFM", "xDeepFMCin"):
        if backbone == "xDeepFMCin":
            NL = 0
        units = [int(c) for c in XDEEPFM["cin_layer_units"].split(",")]
        h_in = F
        for i, u in enumerate(units):                # nn.Conv1d(F * h_in, u, kernel_size=1): layers.py:701-706
            out[f"cin.cin_layer.layer_{i + 1}.weight"] = ((u, F * h_in, 1), (F * h_in) ** -0.5)
            out[f"cin.cin_layer.layer_{i + 1}.bias"] = ((u,), 0.1)
            h_in = u
        if mode == "CTR" and XDEEPFM["use_lr"]:
            out["lr_layer.embed_w.weight"] = ((V, 1), 0.3)
            out["lr_layer.bias"] = ((1,), 0.1)
    elif backbone != "DNN":
        raise ValueError(backbone)
    tower = "parallel_dnn" if backbone == "DCNv2" else "dnn"
    d_in = D
    for i in range(NL):
        out[f"{tower}.dnn.{3 * i}.weight"] = ((H, d_in), d_in ** -0.5)
        out[f"{tower}.dnn.{3 * i}.bias"] = ((H,), 0.1)
        d_in = H
    if backbone != "AutoInt":
        cin_out = sum(int(c) for c in XDEEPFM["cin_layer_units"].split(","))
        Dfin = {"DCNv2": D + (H if NL > 0 else 0), "DNN": H, "DeepFM": H + 1, "xDeepFM": cin_out + H,
                "xDeepFMCin": cin_out}[backbone]
    if mode == "CTR" and backbone in ("xDeepFM", "xDeepFMCin"):       # models.py:261: nn.Linear(final_dim, 1)
        out["fc.weight"] = ((1, Dfin), Dfin ** -0.5)
        out["fc.bias"] = ((1,), 0.1)
        return out
    if mode == "CTR" and backbone != "DCNv2":
        head = "fc_out" if backbone == "DNN" else "dnn_fc_out"       # models.py:180, 212: Linear(H, 1)
        out[f"{head}.weight"] = ((1, H), H ** -0.5)
        out[f"{head}.bias"] = ((1,), 0.1)
        return out
    if mode == "MFP":
        out["feat_encoder.weight"] = ((F * P, Dfin), Dfin ** -0.5)
        out["feat_encoder.bias"] = ((F * P,), 0.1)
        out["mfp_criterion.emb.weight"] = ((V, P), P ** -0.5)
        out["mfp_criterion.bias.weight"] = ((V, 1), 0.5)
    elif mode == "RFD":
        out["pred_rfd.0.weight"] = ((F * P, Dfin), Dfin ** -0.5)
        out["pred_rfd.0.bias"] = ((F * P,), 0.1)
        out["pred_rfd.2.weight"] = ((F, F * P), (F * P) ** -0.5)
        out["pred_rfd.2.bias"] = ((F,), 0.1)
    elif mode == "CTR":
        out["fc_out.weight"] = ((1, Dfin), Dfin ** -0.5)
        out["fc_out.bias"] = ((1,), 0.1)
    else:
        raise ValueError(mode)
    return out


def make_params(case, cfg, mode, backbone="DCNv2"):
    return {k: make_param(case, k, shp, sc) for k, (shp, sc) in param_shapes(cfg, mode, backbone).items()}


def digest(name, g):
    """What a fixture keeps of a (gradient) tensor: everything when small, else linear
    functionals (random projections along each axis) + sums, all tolerance-comparable."""
    g = np.asarray(g, dtype=np.float32)
    if g.size <= FULL_MAX:
        return {"full": g}
    g2 = g.reshape(g.shape[0], -1).astype(np.float64)
    rr = _rng("proj", name, "r").standard_normal(g2.shape[1])
    rl = _rng("proj", name, "l").standard_normal(g2.shape[0])
    return {"sum": np.float64(g2.sum()), "abssum": np.float64(np.abs(g2).sum()),
            "proj_r": g2 @ rr, "proj_l": rl @ g2}
