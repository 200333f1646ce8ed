"""ORACLE — test infrastructure only, never the product path.

CPU (PyTorch fp32, autograd) restatement of the reference's DCNv2 pretraining hot path,
written functionally over a flat {state_dict key: tensor} parameter dict.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package.

Parity pin: every function here is checked against golden vectors produced by running
the real reference modules (tests/golden/gen_golden.py -> tests/golden/*.npz) in
tests/test_oracle_golden.py.  The optimizer (third-party transformers==4.26.1 AdamW,
absent from /root/reference and from this image) is pinned by a hand-computed
known-answer test instead: "parity unpinned upstream" for that one function.

Reference sites restated (paths under /root/reference/code):
  embed            layers.py:97-102     (nn.Embedding gather, flatten in models.py:308)
  cross            layers.py:197-201    Xi+1 = Xi + X0 * (W_i Xi + b_i)
  dnn              layers.py:173-188    [Linear, ReLU, Dropout(p=0)] x NL
  trunk            models.py:306-314    cat(cross, dnn)
  DNN / DeepFM     models.py:164-233, 129-143 (LR), layers.py:123-131 (FM product_sum)
  AutoInt          models.py:440-488, layers.py:724-744, 848-914 (self-attention, view head split)
  mfp head         models.py:71-78, nce/nce_loss.py:79-144,158-173,201-230,
                   nce/index_linear.py:68-106
  rfd head         models.py:79-85,119-124
  ctr head         models.py:88-93,304,319
  nce buffers      nce/nce_loss.py:55-71, nce/index_linear.py:41-48
  alias table      nce/alias_multinomial.py:39-72 ; draw :81-97
  dynamic_mask     trainer.py:217-240
  optimizer        trainer.py:60-85 + transformers 4.26.1 optimization.py AdamW.step
"""
import math

import numpy as np
import torch
import torch.nn.functional as Fn

BACKOFF_PROB = 1e-10   # nce_loss.py:10
MASK_ID = 3            # trainer.py:229


# ----------------------------------------------------------------------------- trunk
def embed(params, ids):
    return params["embed.embedding.weight"][ids].flatten(1)


def cross(params, x0, num_cross):
    xi = x0
    for i in range(num_cross):
        w = params[f"cross_net.cross_layers.{i}.weight"]
        b = params[f"cross_net.cross_layers.{i}.bias"]
        xi = xi + x0 * (xi @ w.t() + b)
    return xi


def _relu(z, key, relu_masks, preacts):
    """ReLU.  Test hooks (both None in normal use): `preacts` collects the pre-activations by layer
    key; `relu_masks[key]` (bool) imposes an activation pattern instead of z > 0 — comparing two
    arithmetics across a pre-activation that rounds to either side of zero compares ReLU branches,
    not arithmetic, so the parity tests re-run the exact (fp64) pass on the pattern under test."""
    if preacts is not None:
        preacts[key] = z.detach()
    if relu_masks is not None and key in relu_masks:
        return z * relu_masks[key].to(z.dtype)
    return torch.relu(z)


def act(kind, x):
    """The reference's get_act (layers.py:55-80) other than relu, restated with the torch ops its classes use
    (LEU :13-27 with alpha = 1, GELU :36-37, GELU_new :41-42, Swish :46-47, Mish :51-52)."""
    kind = kind.lower()
    if kind == "tanh":
        return torch.tanh(x)
    if kind == "sigmoid":
        return torch.sigmoid(x)
    if kind == "none":
        return x
    if kind == "elu":
        return Fn.elu(x)
    if kind == "leu":
        return torch.where(x > 0, torch.log(x.clamp(min=0) + 1), torch.exp(x.clamp(max=0)) - 1)
    if kind == "gelu":
        return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))
    if kind == "gelu_new":
        return 0.5 * x * (1 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * torch.pow(x, 3))))
    if kind == "swish":
        return x * torch.sigmoid(x)
    if kind == "mish":
        return x * torch.tanh(Fn.softplus(x))
    raise NotImplementedError(kind)


def dnn(params, x, num_hidden, tower="parallel_dnn", relu_masks=None, preacts=None, hidden_act="relu"):
    for i in range(num_hidden):
        w = params[f"{tower}.dnn.{3 * i}.weight"]
        b = params[f"{tower}.dnn.{3 * i}.bias"]
        z = x @ w.t() + b
        x = _relu(z, f"{tower}.dnn.{3 * i}", relu_masks, preacts) if hidden_act == "relu" else act(hidden_act, z)
    return x


def trunk(params, ids, num_cross, num_hidden, relu_masks=None, preacts=None, hidden_act="relu"):
    x0 = embed(params, ids)
    c = cross(params, x0, num_cross)
    if num_hidden > 0:
        return torch.cat([c, dnn(params, x0, num_hidden, relu_masks=relu_masks, preacts=preacts,
                                 hidden_act=hidden_act)], dim=-1)
    return c


# The other backbones of SURVEY §8(f4); their heads are the functions below, unchanged.
def trunk_dnn(params, ids, num_hidden):
    """DNN (models.py:164-193): the MLP over the flattened embeddings feeds the heads / fc_out."""
    return dnn(params, embed(params, ids), num_hidden, tower="dnn")


def lr_logit(params, ids):
    """LR (models.py:129-143): sum over the fields of a scalar weight per feature id, + bias -> [B,1]."""
    return params["lr_layer.embed_w.weight"][ids].sum(dim=1) + params["lr_layer.bias"]


def fm_product_sum(x3):
    """InnerProductLayer(output='product_sum') (layers.py:123-131) on [B,F,E] -> [B,1]."""
    return (0.5 * (x3.sum(dim=1) ** 2 - (x3 ** 2).sum(dim=1))).sum(dim=-1, keepdim=True)


def trunk_deepfm(params, ids, num_hidden):
    """DeepFM (models.py:196-233) -> (dnn_vec [B,H], lr + fm [B,1]).  Pretraining feeds
    cat([dnn_vec, lr_fm]) to the heads; CTR adds lr_fm to dnn_fc_out(dnn_vec)."""
    x3 = params["embed.embedding.weight"][ids]
    return dnn(params, x3.flatten(1), num_hidden, tower="dnn"), lr_logit(params, ids) + fm_product_sum(x3)


def autoint_layer(params, x, i, num_heads, attn_size, res_conn, attn_scale):
    """One MultiHeadSelfAttention (layers.py:848-914; AutoInt style, align_to="output") on [B,F,Din].
    Heads are split by a plain .view(B*H, -1, A) of the projected [B,F,H*A] tensor (layers.py:886-888),
    NOT by a transpose: each sample's F*H*A values are cut into H consecutive chunks of F*A."""
    B = x.shape[0]
    pre = f"self_attention.{i}."
    q = x @ params[pre + "W_q.weight"].t()
    k = x @ params[pre + "W_k.weight"].t()
    v = x @ params[pre + "W_v.weight"].t()
    HA = num_heads * attn_size
    qh, kh, vh = (t.reshape(B * num_heads, -1, attn_size) for t in (q, k, v))
    att = torch.bmm(qh, kh.transpose(1, 2))
    if attn_scale:
        att = att / (attn_size ** 0.5)
    out = torch.bmm(torch.softmax(att, dim=2), vh).reshape(B, -1, HA)
    res = x @ params[pre + "W_res.weight"].t() if (pre + "W_res.weight") in params else x
    if res_conn:
        out = out + res
    return torch.relu(out)


def trunk_autoint(params, ids, ai):
    """AutoInt (models.py:440-488): stacked self-attention over the field embeddings, flattened."""
    x = params["embed.embedding.weight"][ids]
    for i in range(ai["num_attn_layers"]):
        x = autoint_layer(params, x, i, ai["num_attn_heads"], ai["attn_size"], ai["res_conn"], ai["attn_scale"])
    return x.flatten(1)


def cin(params, x3, units):
    """Compressed Interaction Network (layers.py:708-721): X_{i+1}[b,o,:] = bias[o] +
    sum_{h,m} W[o, h*H_i + m] X_0[b,h,:] * X_i[b,m,:]  (a 1x1 Conv1d over the F*H_i Hadamard
    channels, no activation); every layer's output is sum-pooled over the embedding axis."""
    B, F, E = x3.shape
    xi, pooled = x3, []
    for i, u in enumerate(units):
        had = torch.einsum("bhd,bmd->bhmd", x3, xi).reshape(B, -1, E)
        w = params[f"cin.cin_layer.layer_{i + 1}.weight"][:, :, 0]
        xi = torch.einsum("ok,bkd->bod", w, had) + params[f"cin.cin_layer.layer_{i + 1}.bias"].view(1, -1, 1)
        pooled.append(xi.sum(-1))
    return torch.cat(pooled, dim=-1)


def trunk_xdeepfm(params, ids, num_hidden, units):
    """xDeepFM (models.py:263-269): cat([CIN(embed), MLP(embed.flatten)])."""
    x3 = params["embed.embedding.weight"][ids]
    return torch.cat([cin(params, x3, units), dnn(params, x3.flatten(1), num_hidden, "dnn")], dim=1)


def _units(extra):
    return [int(c) for c in extra["cin_layer_units"].split(",")]


def final_of(backbone, params, ids, num_cross, num_hidden, autoint=None):
    """The vector the pretraining heads see, per backbone (`autoint`: the backbone's extra config
    keys — AutoInt's attention settings or xDeepFM's cin_layer_units / use_lr)."""
    if backbone == "xDeepFM":
        return trunk_xdeepfm(params, ids, num_hidden, _units(autoint))
    if backbone == "DCNv2":
        return trunk(params, ids, num_cross, num_hidden)
    if backbone == "DNN":
        return trunk_dnn(params, ids, num_hidden)
    if backbone == "DeepFM":
        return torch.cat(trunk_deepfm(params, ids, num_hidden), dim=1)
    if backbone == "AutoInt":
        return trunk_autoint(params, ids, autoint)
    raise NotImplementedError(backbone)


def ctr_logits_of(backbone, params, ids, num_cross, num_hidden, autoint=None):
    """CTR logits [B,1] per backbone (models.py:189-190, 228-231, 274-277, 319)."""
    if backbone == "xDeepFM":
        logits = trunk_xdeepfm(params, ids, num_hidden, _units(autoint)) @ params["fc.weight"].t() + params["fc.bias"]
        if autoint.get("use_lr"):
            logits = logits + lr_logit(params, ids)
        return logits
    if backbone == "DCNv2":
        return ctr_head(params, trunk(params, ids, num_cross, num_hidden))[0]
    if backbone == "DNN":
        return ctr_head(params, trunk_dnn(params, ids, num_hidden))[0]
    if backbone == "DeepFM":
        vec, lr_fm = trunk_deepfm(params, ids, num_hidden)
        return vec @ params["dnn_fc_out.weight"].t() + params["dnn_fc_out.bias"] + lr_fm
    if backbone == "AutoInt":     # use_lr False, num_dnn_layers 0 (models.py:481-487)
        return trunk_autoint(params, ids, autoint) @ params["attn_out.weight"].t() + params["attn_out.bias"]
    raise NotImplementedError(backbone)


# ----------------------------------------------------------------------------- heads
def nce_buffers(feat_count):
    """-> (logprob_noise [V] f32, norm_term = ln V, renormalised noise probs [V] f32)."""
    fc = torch.as_tensor(feat_count, dtype=torch.float32)
    probs = (fc / fc.sum()).clamp(min=BACKOFF_PROB)
    probs = probs / probs.sum()
    return probs.log(), math.log(fc.numel()), probs


def softplus(x):
    return torch.clamp(x, min=0) + torch.log1p(torch.exp(-x.abs()))


def mfp_head(params, final, labels, masked_index, noise, logq, num_fields, proj, neg_num):
    """-> (loss scalar, logits [B,L,K+1] (= s, not noise-corrected), total_acc int)."""
    B = final.shape[0]
    enc = (final @ params["feat_encoder.weight"].t() + params["feat_encoder.bias"])
    enc = enc.view(B, num_fields, proj)
    h = torch.gather(enc, 1, masked_index.unsqueeze(-1).expand(-1, -1, proj))       # [B,L,P]
    idx = torch.cat([labels.unsqueeze(-1), noise], dim=-1)                          # [B,L,K+1]
    rows = params["mfp_criterion.emb.weight"][idx]                                  # [B,L,K+1,P]
    bias = params["mfp_criterion.bias.weight"][idx].squeeze(-1)
    s = (h.unsqueeze(2) * rows).sum(-1) + bias - math.log(logq.numel())
    lt = s - logq[idx] - math.log(neg_num)
    per_target = softplus(-lt[..., 0]) + softplus(lt[..., 1:]).sum(-1)
    loss = per_target.mean()
    total_acc = int((s[..., :1] >= s).all(dim=-1).sum())     # argmax==0, ties -> first index
    return loss, s, total_acc


def rfd_head(params, final, labels, relu_masks=None, preacts=None):
    """-> (loss, count, acc tensor, pos_ratio tensor, logits [B,F])."""
    z = _relu(final @ params["pred_rfd.0.weight"].t() + params["pred_rfd.0.bias"], "pred_rfd.0", relu_masks, preacts)
    logits = z @ params["pred_rfd.2.weight"].t() + params["pred_rfd.2.bias"]
    loss = Fn.binary_cross_entropy_with_logits(logits, labels)
    count = labels.numel()
    acc = ((torch.sigmoid(logits) > 0.5).float() == labels).sum() / count
    return loss, count, acc, labels.mean(), logits


def ctr_head(params, final, y=None):
    logits = final @ params["fc_out.weight"].t() + params["fc_out.bias"]
    if y is None:
        return (logits,)
    return Fn.binary_cross_entropy_with_logits(logits.view(-1), y.float()), logits


# ----------------------------------------------------------------------------- masking
def dynamic_mask_mfp(ids, masked_index):
    labels = torch.gather(ids, 1, masked_index)
    masked = torch.scatter(ids, 1, masked_index, torch.full_like(masked_index, MASK_ID))
    return masked, labels


def dynamic_mask_rfd(ids, masked_index, replace_feat):
    """Duplicate positions in masked_index: the LAST one written wins (CPU scatter order)."""
    out = ids.clone()
    B, L = masked_index.shape
    for b in range(B):
        for l in range(L):
            out[b, masked_index[b, l]] = replace_feat[b, l]
    return out, (ids != out).float()


# ----------------------------------------------------------------------------- alias
def alias_build(probs):
    """Walker table in the reference's exact visiting order, float32 arithmetic.
    -> (prob f32 [V], alias i64 [V])."""
    p = np.asarray(probs, dtype=np.float32)
    K = p.shape[0]
    q = np.zeros(K, dtype=np.float32)
    alias = np.zeros(K, dtype=np.int64)
    small, large = [], []
    kf = np.float32(K)
    for i in range(K):
        q[i] = kf * p[i]
        (small if q[i] < 1.0 else large).append(i)
    one = np.float32(1.0)
    while small and large:
        s, l = small.pop(), large.pop()
        alias[s] = l
        q[l] = (q[l] - one) + q[s]
        (small if q[l] < 1.0 else large).append(l)
    for i in small + large:
        q[i] = 1.0
    return q, alias


def alias_draw(prob, alias, kk, u):
    """Injected randomness: kk ~ U{0..V-1} i64, u ~ U[0,1).  b = u < prob[kk]."""
    b = (u < prob[kk])
    return torch.where(b, kk, alias[kk])


def alias_distribution(prob, alias):
    """The distribution a (prob, alias) table encodes (float64)."""
    prob = np.asarray(prob, dtype=np.float64)
    V = prob.shape[0]
    out = prob / V
    np.add.at(out, np.asarray(alias), (1.0 - prob) / V)
    return out


# ----------------------------------------------------------------------------- optimizer
def lr_lambda(kind, step, total, warmup):
    """transformers get_{cosine,constant}_schedule_with_warmup multipliers."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    if kind == "const":
        return 1.0
    prog = float(step - warmup) / float(max(1, total - warmup))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * 0.5 * 2.0 * prog)))


def hf_adamw_step(p, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.0):
    """One transformers-4.26 AdamW update, in place; t = 1-based step count.
    Differences from torch.optim.AdamW: eps added to sqrt(v) un-scaled; bias correction
    folded into the step size; decoupled decay applied AFTER the Adam update."""
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    denom = v.sqrt().add_(eps)
    step_size = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    p.addcdiv_(m, denom, value=-step_size)
    if wd > 0.0:
        p.add_(p, alpha=-lr * wd)


def decays(name):
    """trainer.py:61-63: no decay iff the parameter NAME contains 'bias' or
    'LayerNorm.weight' (so mfp_criterion.bias.weight is not decayed; tables are)."""
    return not any(nd in name for nd in ("bias", "LayerNorm.weight"))
