#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference modules (CPU, fp32).

Runs only where /root/reference exists (the build container); the GPU box and the test
suite only ever read the .npz files this writes next to itself.  Nothing from the
reference is copied: fixtures hold inputs-by-construction (tests/golden/paramgen.py) and
the reference's numerical OUTPUTS.

Import note: reference `arguments.py:10` imports
`transformers.utils.cached_property`, which the installed transformers no longer exports;
one attribute is set before import so that the module loads.  `trainer.py` (needs the
removed `transformers.AdamW`) and `dataset.py` (needs h5py) are not imported.

    python tests/golden/gen_golden.py
"""
import functools
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import paramgen as pg  # noqa: E402

REF = "/root/reference/code"


def import_reference():
    import transformers.utils as tu
    if not hasattr(tu, "cached_property"):
        tu.cached_property = functools.cached_property
    sys.path.insert(0, REF)
    import arguments  # noqa: F401
    import models
    return arguments, models


def ref_config(arguments, cfg, mode, feat_count, data_dir, backbone="DCNv2"):
    d = dict(
        model_name=pg.model_name_of(backbone), data_dir=data_dir, input_size=cfg["V"], num_fields=cfg["F"],
        embed_size=cfg["E"], embed_dropout_rate=0.0, embed_norm=False, layer_norm_eps=1e-12,
        hidden_size=cfg["H"], num_hidden_layers=cfg["NL"], hidden_act="relu",
        hidden_dropout_rate=0.0, num_cross_layers=cfg["NC"], pt_neg_num=cfg["K"],
        proj_size=cfg["P"], pretrain=(mode != "CTR"), pt_type=("RFD" if mode == "RFD" else "MFP"),
        RFD_replace="Unigram", feat_count=torch.from_numpy(feat_count), device=torch.device("cpu"),
        n_gpu=0, idx_low=None, idx_high=None, feat_num_per_field=None)
    d.update(pg.extras_of(backbone))
    return arguments.Config.from_dict(d)


def put(store, prefix, name, arr):
    for k, v in pg.digest(name, arr).items():
        store[f"{prefix}/{name}/{k}"] = v


def trunk_of(model, backbone, emb3):
    """The backbone's representation that feeds the heads, from the embedded input [B,F,E]."""
    flat = emb3.flatten(1)
    if backbone == "DCNv2":
        return torch.cat([model.cross_net(flat), model.parallel_dnn(flat)], -1)
    if backbone.startswith("AutoInt"):
        return model.self_attention(emb3).flatten(1)
    if backbone == "xDeepFMCin":
        return model.cin(emb3)
    if backbone == "xDeepFM":
        return torch.cat([model.cin(emb3), model.dnn(flat)], 1)
    return model.dnn(flat)      # DNN; DeepFM's pretrain vector also appends lr + fm (not needed below)


def run_case(arguments, models, case, cfg, mode, outdir, backbone="DCNv2"):
    torch.manual_seed(0)
    inp = pg.make_inputs(case, cfg)
    params = pg.make_params(case, cfg, mode, backbone)
    store = {}
    with tempfile.TemporaryDirectory() as tmp:      # alias_self_*.h5 cache goes here
        config = ref_config(arguments, cfg, mode, inp["feat_count"], tmp, backbone)
        model = models.BaseModel.from_config(config)
    sd = model.state_dict()
    manifest = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()}
    # buffers / init facts the NCE head derives from feat_count (before we overwrite params)
    if mode == "MFP":
        crit = model.mfp_criterion
        store["nce/logprob_noise"] = crit.logprob_noise.numpy().copy()
        store["nce/alias_prob"] = crit.alias.prob.numpy().copy()
        store["nce/alias_alias"] = crit.alias.alias.numpy().copy()
        store["nce/bias_init"] = crit.bias.weight.detach().numpy().copy()
        store["nce/norm_term"] = np.float64(crit.norm_term)
        store["nce/emb_init_absmax"] = np.float64(crit.emb.weight.detach().abs().max())
    store["init/embed_std"] = np.float64(model.embed.embedding.weight.detach().std())
    # overwrite every parameter with the reproducible numpy ones
    trainable = {k: p for k, p in model.named_parameters() if p.requires_grad}   # (DeepFM's ip_layer keeps
    with torch.no_grad():                                                        # index tensors as frozen Parameters)
        for k, p in trainable.items():
            assert tuple(p.shape) == params[k].shape, (k, p.shape, params[k].shape)
            p.copy_(torch.from_numpy(params[k]))
    assert set(params) == set(trainable)
    model.train()
    ids = torch.from_numpy(inp["input_ids"])
    mi = torch.from_numpy(inp["masked_index"])

    if mode == "MFP":
        # dynamic_mask MFP branch (trainer.py:227-232) restated with the injected index
        labels = torch.gather(ids, 1, mi)
        ids_in = torch.scatter(ids, 1, mi, torch.full_like(mi, pg.MASK_ID))
        noise = torch.from_numpy(inp["noise"])
        model.mfp_criterion.get_noise = lambda b, l: noise
        # capture logits through the criterion (get_outputs drops them)
        cap = {}
        orig_forward = model.mfp_criterion.forward

        def fwd(target, *a, **k):
            out = orig_forward(target, *a, **k)
            cap["logits"], cap["indices"] = out[1].detach(), out[2].detach()
            return out
        model.mfp_criterion.forward = fwd
        loss, count, total_acc = model(input_ids=ids_in, labels=labels, masked_index=mi)
        store["in/input_ids_masked"] = ids_in.numpy()
        store["in/labels"] = labels.numpy()
        store["out/count"] = np.int64(count)
        store["out/total_acc"] = np.int64(total_acc)
        store["out/logits"] = cap["logits"].numpy()
        store["out/indices"] = cap["indices"].numpy()
    elif mode == "RFD":
        repl = torch.from_numpy(inp["replace_feat"])
        ids_in = torch.scatter(ids, 1, mi, repl)          # trainer.py:239 (last duplicate wins on CPU)
        labels = (ids != ids_in).float()                  # trainer.py:240
        loss, count, acc, pos_ratio = model(input_ids=ids_in, labels=labels, masked_index=None)
        store["in/input_ids_replaced"] = ids_in.numpy()
        store["in/labels"] = labels.numpy()
        store["out/count"] = np.int64(count)
        store["out/acc"] = acc.detach().numpy()
        store["out/pos_ratio"] = pos_ratio.detach().numpy()
        if backbone != "DeepFM":
            with torch.no_grad():
                store["out/logits"] = model.pred_rfd(trunk_of(model, backbone, model.embed(ids_in))).numpy()
    else:
        y = torch.from_numpy(inp["y"])
        loss, logits = model(input_ids=ids, labels=y)
        store["out/logits"] = logits.detach().numpy()
        with torch.no_grad():   # intermediate activations for per-kernel unit tests
            emb = model.embed(ids).flatten(1)
            store["mid/embed_flat"] = emb.numpy()
            if backbone == "DCNv2":
                store["mid/cross_out"] = model.cross_net(emb).numpy()
                store["mid/dnn_out"] = model.parallel_dnn(emb).numpy()
            if backbone == "DeepFM":
                store["mid/lr"] = model.lr_layer(ids)[0].numpy()
                store["mid/fm"] = model.ip_layer(model.embed(ids)).numpy()
            if backbone.startswith("xDeepFM"):
                store["mid/cin_out"] = model.cin(model.embed(ids)).numpy()
            if backbone.startswith("AutoInt"):
                x = model.embed(ids)
                for li, layer in enumerate(model.self_attention):
                    x = layer(x)
                    store[f"mid/attn{li}"] = x.numpy()
    loss.backward()
    store["out/loss"] = loss.detach().numpy()
    for k, p in trainable.items():
        put(store, "grad", k, p.grad.numpy())
    suffix = "" if backbone == "DCNv2" else f"_{backbone}"
    np.savez_compressed(os.path.join(outdir, f"{case}_{mode}{suffix}.npz"), **store)
    return manifest


def lr_schedules(outdir):
    """Third-party arithmetic: the schedulers the trainer calls
    (trainer.py:78-82) still ship in the installed transformers; pin their multipliers."""
    from transformers import get_constant_schedule_with_warmup, get_cosine_schedule_with_warmup
    out = {}
    for name, T, W in (("cos_T50_W0", 50, 0), ("cos_T37_W5", 37, 5), ("const_T20_W4", 20, 4),
                       ("const_T8_W0", 8, 0)):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=1.0)
        sch = (get_cosine_schedule_with_warmup(opt, W, T) if name.startswith("cos")
               else get_constant_schedule_with_warmup(opt, W))
        lrs = []
        for _ in range(T + 3):
            lrs.append(sch.get_last_lr()[0])
            opt.step()
            sch.step()
        out[name] = np.array(lrs, dtype=np.float64)
    np.savez_compressed(os.path.join(outdir, "lr_schedules.npz"), **out)


def main():
    arguments, models = import_reference()
    torch.set_num_threads(1)
    manifests = {}
    if "--only" in sys.argv:                  # one fixture variant (CTR-only ones): the other files stay as they are
        backbone = sys.argv[sys.argv.index("--only") + 1]
        case = "B_f25_b64"
        path = os.path.join(HERE, "state_dict_manifest.json")
        manifests = json.load(open(path))
        modes = ("CTR",) if backbone in pg.CTR_ONLY else ("MFP", "RFD", "CTR")
        for mode in modes:
            manifests[f"{case}_{mode}_{backbone}"] = run_case(arguments, models, case, pg.CASES[case], mode, HERE, backbone)
            print("wrote", case, mode, backbone)
        json.dump(manifests, open(path, "w"), indent=1, sort_keys=True)
        return
    for case, cfg in pg.CASES.items():
        for mode in ("MFP", "RFD", "CTR"):
            manifests[f"{case}_{mode}"] = run_case(arguments, models, case, cfg, mode, HERE)
            print("wrote", case, mode)
    case = "B_f25_b64"                       # the other backbones: one case each
    for backbone in pg.BACKBONES[1:]:
        for mode in ("MFP", "RFD", "CTR"):
            manifests[f"{case}_{mode}_{backbone}"] = run_case(arguments, models, case, pg.CASES[case], mode, HERE,
                                                              backbone)
            print("wrote", case, mode, backbone)
    with open(os.path.join(HERE, "state_dict_manifest.json"), "w") as f:
        json.dump(manifests, f, indent=1, sort_keys=True)
    # flag surface (names + defaults) of the two argument dataclasses -> json
    import dataclasses
    flags = {}
    for cls in (arguments.ModelArguments, arguments.TrainingArguments):
        for fld in dataclasses.fields(cls):
            default = None if fld.default is dataclasses.MISSING else fld.default
            typ = fld.type if isinstance(fld.type, str) else getattr(fld.type, "__name__", str(fld.type))
            flags[fld.name] = {"cls": cls.__name__, "default": default, "type": typ,
                               "required": fld.default is dataclasses.MISSING}
    with open(os.path.join(HERE, "flag_surface.json"), "w") as f:
        json.dump(flags, f, indent=1, sort_keys=True)
    lr_schedules(HERE)
    print("done")


if __name__ == "__main__":
    main()
