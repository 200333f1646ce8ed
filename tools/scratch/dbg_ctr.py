import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/map-code_amd", "/root/repo/tests", "/root/repo/tests/golden"]
import numpy as np, torch
import paramgen as pg
from util import build_model, load_case, t, oracle_case_grads
cfg, z, inp, params = load_case("A_f23_b7", "CTR")
model = build_model(cfg, "CTR", params, None)
ids = t(inp["input_ids"], "cuda")
model.train()
loss, logits = model(input_ids=ids, labels=t(inp["y"], "cuda"))
loss.backward()
name = "parallel_dnn.dnn.3.weight"
g = dict(model.named_parameters())[name].grad.double().cpu()
# fp64 oracle
from oracle import ref_model as R
P = {k: t(v).double().clone().requires_grad_(True) for k, v in params.items()}
fin = R.trunk(P, t(inp["input_ids"]), cfg["NC"], cfg["NL"])
l64, _ = R.ctr_head(P, fin, t(inp["y"]))
l64.backward()
r = P[name].grad
print(os.environ.get("MAPX_GEMM"), "sum got", float(g.sum()), "ref64", float(r.sum()), "golden", float(z[f"grad/{name}/sum"]), "abssum", float(z[f"grad/{name}/abssum"]))
d = (g - r).abs()
print(" max abs err", float(d.max()), "scale", float(r.abs().max()), "rel", float(d.max() / r.abs().max()), "sum of errs", float((g - r).sum()))
for n2 in ["parallel_dnn.dnn.6.weight", "parallel_dnn.dnn.0.weight", "fc_out.weight"]:
    g2 = dict(model.named_parameters())[n2].grad.double().cpu(); r2 = P[n2].grad
    print(" ", n2, "rel max err", float((g2 - r2).abs().max() / r2.abs().max()), "sum err", float((g2 - r2).sum()), "sum", float(r2.sum()))
