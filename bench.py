#!/usr/bin/env python3
"""Headline benchmark: DCNv2 + MFP pretraining samples/s on Avazu-shaped synthetic data
(BASELINE.json configs[1]: F=23, V=9 449 445, E=16, H=1000x3, 3 cross layers, P=32, K=25,
batch 4096 per GPU, fp32), one process per GPU.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = dynamic_mask (on device) -> forward -> backward -> gradient exchange (N>1) ->
AdamW (dense groups fused; tables row-sparse with exact lazy replay) -> schedule -> zero_grad,
over one batch already resident in HBM.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")     # before the HIP runtime starts; see mapx/__init__.py

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "map-code_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = {"f32": 157.3,     # v_mfma_f32_32x32x2_f32, dense fp32 matrix peak
                    "bf16": 2500.0,   # v_mfma_f32_32x32x16_bf16, dense bf16 peak (no 2:1 sparsity)
                    # fp32 products formed from six bf16 MFMAs (csrc/gemm_x3.hip: operands cut into three bf16
                    # pieces): the instruction-level ceiling of that algorithm in fp32-equivalent flop/s
                    "f32x3": 2500.0 / 6,
                    # ... from three fp16 MFMAs (csrc/gemm_h2.hip, gemm_h2w.hip: operands scaled by their magnitude
                    # records and cut into two fp16 pieces); the fp16 forms issue at the bf16 rate
                    "f32h2": 2500.0 / 3}
ROUND = "r04"                # profiles/<ROUND>_pmc_hbm_traffic[_bf16].json is this round's PMC summary

WORKLOADS = {
    "avazu": dict(F=23, V=9449445),
    "criteo": dict(F=39, V=33762577),
    "small": dict(F=23, V=200000),          # smoke-sized
}


# kernel class (ops.Timers name) -> substrings of the rocprof kernel names it launches
PMC_MAP = {
    "gemm_fwd_nt": [["gemm_f32x3_kernel<", ", true, true, true>"], ["gemm_f32h2_kernel<", ", true, true>"],
                    ["gemm_bf16_kernel<", ", true, true, "]],
    "gemm_dx_nn": [["gemm_f32x3_kernel<", ", true, false, true>"], ["gemm_f32h2_kernel<", ", true, false>"],
                   ["gemm_bf16_kernel<", ", true, false, "]],
    "gemm_dw_tn": [["gemm_f32x3_kernel<", ", false, false, true>"], ["gemm_f32h2_kernel<", ", false, false>"],
                   ["gemm_bf16_kernel<", ", false, false, "]],
    "gemm_weight_planes": [["gemm_f32h2w_kernel"]],      # (forward and input-gradient products alike: one kernel)
    "gemm_enc_grouped_fwd": [["gemm_grouped_x3_kernel<false>"], ["gemm_grouped_h2_kernel<false>"]],
    "gemm_enc_grouped_dw": [["gemm_grouped_x3_kernel<true>"], ["gemm_grouped_h2_kernel<true>"]],
    "nce_fwd": ["nce_fwd_"],
    "nce_table_grad": ["seg_reduce_pass_a<8, true"],
    "seg_reduce_rows": ["seg_reduce_pass_a<4, false"],
    "table_adam_catchup": ["table_adam_raw_kernel"],
    "table_adam_update": ["table_adam_kernel"],
    "emb_gather": ["emb_gather_kernel"],
    "adamw_dense": ["adamw_dense_kernel"],
}


def csrc_hash():
    """sha1 over the kernel sources: stamps a PMC summary with the code it was measured on."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "map-code_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


_PMC = {}


def pmc_file(dtype):
    return os.path.join(ROOT, "profiles", f"{ROUND}_pmc_hbm_traffic{'' if dtype == 'f32' else '_' + dtype}.json")


def pmc_data(dtype):
    """The committed PMC summary for this dtype (tools/pmc_summary.py) or {}; `_meta.csrc_hash` says
    which kernel sources it was taken on."""
    if dtype not in _PMC:
        path = pmc_file(dtype)
        _PMC[dtype] = json.load(open(path)) if os.path.exists(path) else {}
    return _PMC[dtype]


def pmc_stale(dtype):
    """True when the kernels changed since the PMC passes were taken (or none exists): every
    `traffic` figure on the line is then from older code."""
    meta = pmc_data(dtype).get("_meta", {})
    return meta.get("csrc_hash") != csrc_hash()


def pmc_traffic(name, dtype="f32"):
    """HBM bytes per launch of a kernel class from the committed PMC passes (rocprofv3 --pmc
    FETCH_SIZE and --pmc WRITE_SIZE in separate runs of this bench; FETCH_SIZE doubled as
    MI355X_MICROARCH.md §HBM prescribes for gfx950; tools/pmc_summary.py).  None if unknown."""
    data = pmc_data(dtype)
    if name not in PMC_MAP or not data:
        return None
    tot = n = 0
    for k, v in data.items():
        if k == "_meta":
            continue
        pats = PMC_MAP[name]
        pats = pats if isinstance(pats[0], list) else [pats]        # alternatives, each a list of substrings
        if name in ("gemm_fwd_nt", "gemm_dx_nn"):                   # (their large products run as the one weight-planes kernel)
            pats = pats + PMC_MAP["gemm_weight_planes"]
        if any(all(sub in k for sub in alt) for alt in pats):
            tot += v["hbm_bytes_per_launch"] * v["launches"]
            n += v["launches"]
    return tot / n if n else None


def mfma_busy(name, dtype="f32"):
    """Share of the matrix pipe's cycles the class's kernels kept it busy (SQ_VALU_MFMA_BUSY_CYCLES over
    GRBM_GUI_ACTIVE x SIMDs, one rocprofv3 --pmc pass of the serial step: tools/pmc_mfma_summary.py), or None."""
    path = os.path.join(ROOT, "profiles", f"{ROUND}_pmc_mfma_busy{'' if dtype == 'f32' else '_' + dtype}.json")
    if path not in _PMC:
        _PMC[path] = json.load(open(path)) if os.path.exists(path) else {}
    data = _PMC[path]
    names = [name] + (["gemm_weight_planes"] if name in ("gemm_fwd_nt", "gemm_dx_nn") else [])
    busy = cyc = 0.0
    for nm in names:
        if nm not in PMC_MAP:
            continue
        pats = PMC_MAP[nm]
        pats = pats if isinstance(pats[0], list) else [pats]
        for k, v in data.items():
            if k != "_meta" and any(all(sub in k for sub in alt) for alt in pats):
                busy += v["mfma_busy_cycles"]
                cyc += v["simd_cycles"]
    return busy / cyc if cyc else None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="avazu", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--rows", type=int, default=1 << 22, help="synthetic training rows resident in HBM")
    ap.add_argument("--uniform", action="store_true", help="uniform ids inside a field instead of Zipf(1.1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preroll", type=int, default=400,
                    help="untimed real training steps before the warm-up, so that the lazy table optimizer "
                         "carries a realistic replay debt (rows re-touched after long gaps)")
    ap.add_argument("--cpu-steps", type=int, default=14)
    ap.add_argument("--fake-world", type=int, default=0, metavar="N",
                    help="data-parallel tail rehearsal on ONE GPU: N real forward/backward passes on different "
                         "batches give N ranks' sparse-gradient messages; the merge (gather exchange and "
                         "owner-partitioned) and the optimizer over the gathered lists are timed.  No bench line.")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="arithmetic type of the dense trunk (BASELINE configs[1] is f32, configs[2] bf16: fp32 "
                         "master weights / tables / optimizer, bf16 GEMM operands and activations)")
    return ap.parse_args()


def build(args, device, rank):
    from mapx.arguments import Config, TrainingArguments
    from mapx.dataset import OurDataset, synth_table
    from mapx.models import BaseModel
    from mapx.trainer import Trainer
    wl = WORKLOADS[args.workload]
    F, V = wl["F"], wl["V"]
    ids, labels, _, _ = synth_table(args.rows, F, V, seed=42, uniform=args.uniform)
    feat_count = torch.from_numpy(np.bincount(ids.reshape(-1), minlength=V).astype(np.float32))
    pt = getattr(args, "pt", "MFP")              # tools/step_bench.py times the RFD and finetune (CTR) steps too
    cfg = Config(model_name=getattr(args, "model", "DCNv2"), data_dir=None, input_size=V, num_fields=F, embed_size=16,
                 embed_dropout_rate=0.0, embed_norm=False, hidden_size=1000, num_hidden_layers=3,
                 hidden_act="relu", hidden_dropout_rate=0.0, num_cross_layers=3, pt_neg_num=25,
                 proj_size=32, pretrain=pt != "CTR", pt_type="MFP" if pt == "CTR" else pt, RFD_replace="Unigram",
                 feat_count=feat_count, seed=42, rank=rank, compute_dtype="bf16" if args.dtype == "bf16" else "fp32",
                 # (the secondary backbones of tools/step_bench.py --model: the reference scripts' values)
                 cin_layer_units="50,50", use_lr=False, num_attn_layers=2, attn_size=40, num_attn_heads=1,
                 attn_probs_dropout_rate=0.0, res_conn=False, attn_scale=False, dnn_size=1000, num_dnn_layers=0,
                 dnn_act="relu", dnn_drop=0.0)
    torch.manual_seed(42)
    model = BaseModel.from_config(cfg)
    # the cosine schedule must outlast the run at every world size (the sharded epoch shrinks with N),
    # else the measured steps would be lr=0 steps whose table updates are cheaper than real ones
    world = int(os.environ.get("WORLD_SIZE", "1"))
    steps_per_epoch = max(1, args.rows // (args.batch * world))
    need = 2 * (args.preroll + args.warmup + args.steps + 64)
    epochs = max(3, -(-need // steps_per_epoch))
    targs = TrainingArguments(output_dir="/tmp/mapx_bench", per_gpu_train_batch_size=args.batch,
                              per_gpu_eval_batch_size=args.batch, learning_rate=1e-3, lr_sched="cosine",
                              weight_decay=5e-2, num_train_epochs=epochs, pretrain=pt != "CTR",
                              pt_type="MFP" if pt == "CTR" else pt, sampling_method="randint", mask_ratio=0.3, seed=42)
    targs._device = device
    tr = Trainer(model, cfg, targs, OurDataset(ids, labels), OurDataset(ids[:args.batch], labels[:args.batch]))
    return tr, cfg, ids, labels, feat_count


def mfp_step(tr, X, Y):
    """The Trainer's own step: captured hipGraph replay after 3 eager steps (1 GPU), eager with
    the RCCL gradient exchange otherwise."""
    return tr.run_step("mfp", X, Y)[0]


def cpu_baseline(cfg, ids, labels, feat_count, batch, steps):
    """The oracle (CPU restatement of the reference step, dense HF-AdamW over every parameter
    as the reference does) timed on this box's host cores on `steps` batches of the same
    synthetic stream."""
    from oracle import ref_model as R
    # a 1-GPU box owns a 16-core share of the host (256 logical CPUs are visible; torch's
    # intra-op pool at 256 threads is 10x slower on these memory-bound ops than at 16)
    cores = min(os.cpu_count() or 1, int(os.environ.get("MAPX_CPU_THREADS", "16")))
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    F, V, E, H, P, K = cfg.num_fields, cfg.input_size, cfg.embed_size, cfg.hidden_size, cfg.proj_size, cfg.pt_neg_num
    D = F * E
    shapes = {"embed.embedding.weight": (V, E), "feat_encoder.weight": (F * P, D + H), "feat_encoder.bias": (F * P,),
              "mfp_criterion.emb.weight": (V, P), "mfp_criterion.bias.weight": (V, 1)}
    for i in range(cfg.num_cross_layers):
        shapes[f"cross_net.cross_layers.{i}.weight"] = (D, D)
        shapes[f"cross_net.cross_layers.{i}.bias"] = (D,)
    d_in = D
    for i in range(cfg.num_hidden_layers):
        shapes[f"parallel_dnn.dnn.{3 * i}.weight"] = (H, d_in)
        shapes[f"parallel_dnn.dnn.{3 * i}.bias"] = (H,)
        d_in = H
    params = {k: (0.05 * torch.randn(*s, generator=g)).requires_grad_(True) for k, s in shapes.items()}
    m = {k: torch.zeros_like(p) for k, p in params.items()}
    v = {k: torch.zeros_like(p) for k, p in params.items()}
    logq, _, q = R.nce_buffers(feat_count)
    from mapx import ops
    prob, alias = ops.alias_build(q)                      # host C++ builder (python loop = minutes)
    L = int(F * 0.3)
    X = torch.from_numpy(ids)

    def one(step):
        xb = X[step * batch:(step + 1) * batch]
        mi = torch.randint(0, F, (batch, L), generator=g)
        masked, lab = R.dynamic_mask_mfp(xb, mi)
        kk = torch.randint(0, V, (batch, L, K), generator=g)
        noise = R.alias_draw(prob, alias, kk, torch.rand(batch, L, K, generator=g))
        fin = R.trunk(params, masked, cfg.num_cross_layers, cfg.num_hidden_layers)
        loss, _, _ = R.mfp_head(params, fin, lab, mi, noise, logq, F, P, K)
        loss.backward()
        with torch.no_grad():
            for k, p in params.items():
                R.hf_adamw_step(p, p.grad, m[k], v[k], step + 1, 1e-3, wd=5e-2 if R.decays(k) else 0.0)
                p.grad = None
    one(0)                                               # warm-up (allocations, thread pool)
    t0 = time.perf_counter()
    for s in range(1, steps + 1):
        one(s)
    dt = time.perf_counter() - t0
    return dict(value=batch * steps / dt, unit="samples/s", cores=cores, kind="port",
                sample=f"{steps} steps of batch {batch} (fwd+bwd+dense AdamW over all {sum(p.numel() for p in params.values())} "
                       f"parameters, oracle/ref_model.py, torch CPU fp32, {cores} threads), {dt:.1f} s")


XGMI_LINK_GBS = 153.0        # per link and direction; 7 links per GPU, fully connected


def fake_world(tr, args, next_batch):
    """What the tail of a data-parallel step costs at world N, measured on one GPU: the other ranks'
    messages are REAL (N forward/backward passes on N different batches with their own masks and
    negatives, packed exactly as parallel.pack_table packs them), only the wire is missing.
    Timed with HIP events, GPU parked first: (a) gather exchange: merge of N sorted lists per table
    (mapx_seg_plan_merge + segment reduction) and the optimizer over the merged rows; (b) owner-
    partitioned exchange: this rank merges only the ids it owns (1/N of the id range) out of every
    list, the optimizer then walks the concatenation of the owners' merged lists (same rows).
    Wire time is MODELLED, never measured here: bytes / (7 links x 153 GB/s), direct sends."""
    from mapx import ops, parallel
    N = args.fake_world
    opt = tr.optimizer
    tables = [t.table for t in opt.tables]
    for t in opt.tables:
        t.early_ok = False            # as with N > 1: no table update before the exchange
    per_rank = []                     # per rank: [(plan, r0, r1) per table]
    for r in range(N):
        X, Y = next_batch()
        tr._mfp_fwd_bwd(X, Y)
        ops.run_side_tasks()
        per_rank.append([tb.sparse_grad for tb in tables])
        for tb in tables:
            tb.sparse_grad = None
    torch.cuda.synchronize()
    counts = [[int(sg[0].n_uniq[0]) for sg in rank] for rank in per_rank]
    sizes = [parallel.message_size(max(c[t] for c in counts)) for t in range(len(tables))]
    msgs = []                         # per table: (k_all [N*size], r_all [N*size, Wp])
    for t, tb in enumerate(tables):
        ks, rs = [], []
        for r in range(N):
            plan, r0, r1 = per_rank[r][t]
            k, rows = ops.pack_sparse(plan, r0, r1, sizes[t], 1.0 / N, pad_id=-1)
            ks.append(k)
            rs.append(rows)
        msgs.append((torch.cat(ks), torch.cat(rs)))
    del per_rank

    def timed(fn, reps=20):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(20_000_000)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    def merge(t, lists, k_all, r_all):
        tb = tables[t]
        W0 = tb.p0.shape[1]
        mplan = ops.SegPlan(k_all, tb.num_rows + 1, sorted_lists=lists)
        if tb.p1 is not None:
            m0, m1 = ops.seg_reduce_rows_extra(mplan, r_all, W0, r_all[:, W0], 1, extra_stride=r_all.stride(0))
        else:
            m0, m1 = ops.seg_reduce_rows(mplan, r_all, W0), None
        return mplan, m0, m1

    out = {"world": N, "tables": {}}
    merged = []
    for t, tb in enumerate(tables):
        k_all, r_all = msgs[t]
        ms_merge = timed(lambda: merge(t, N, k_all, r_all))
        merged.append(merge(t, N, k_all, r_all))
        # owner-partitioned: out of every rank's list only the ids this rank owns (range 0 of N)
        chunk = -(-tb.num_rows // N)
        own = [k_all[r * sizes[t]:(r + 1) * sizes[t]] for r in range(N)]
        n_own = [int(((o >= 0) & (o < chunk)).sum()) for o in own]
        osz = parallel.message_size(max(1, max(n_own)))
        ko = torch.full((N * osz,), -1, dtype=torch.int32, device=k_all.device)
        ro = torch.zeros(N * osz, r_all.shape[1], device=k_all.device)
        for r in range(N):              # ids are ascending inside a list: the owned ones are its head
            ko[r * osz:r * osz + n_own[r]] = own[r][:n_own[r]]
            ro[r * osz:r * osz + n_own[r]] = r_all[r * sizes[t]:r * sizes[t] + n_own[r]]
        ms_owner = timed(lambda: merge(t, N, ko, ro))
        U = int(merged[-1][0].n_uniq[0])
        out["tables"][tb.name] = {
            "rows_per_rank": [c[t] for c in counts], "message_rows": sizes[t], "row_bytes": 4 * r_all.shape[1] + 4,
            "merged_rows": U, "merge_ms_gather": round(ms_merge, 4), "merge_ms_owner": round(ms_owner, 4),
            "owner_rows_per_list": n_own}

    def tail():
        for tb, m in zip(tables, merged):
            tb.sparse_grad = m
        opt.step()
    ms_tail = timed(tail, reps=10)
    for tb in tables:
        tb.sparse_grad = None
    out["optimizer_ms"] = round(ms_tail, 4)
    # wire model (direct sends over 7 links): gather = every rank's message to every other rank;
    # owner = slices to owners (1/N of a message to each peer) + the owners' merged lists to everyone
    link = XGMI_LINK_GBS * 1e9
    msg_bytes = sum(sizes[t] * out["tables"][tb.name]["row_bytes"] for t, tb in enumerate(tables))
    merged_bytes = sum(out["tables"][tb.name]["merged_rows"] * out["tables"][tb.name]["row_bytes"] for tb in tables)
    dense_bytes = sum(g["g"].numel() * 4 for g in opt.groups)
    peers = max(N - 1, 1)
    out["wire_model_ms"] = {
        "gather_allgather": round(1e3 * msg_bytes / link, 4) if N > 1 else 0.0,     # one message per link, links in parallel
        "owner_alltoall": round(1e3 * (msg_bytes / N) / link, 4) if N > 1 else 0.0,
        "owner_allgather": round(1e3 * (merged_bytes / N) / link, 4) if N > 1 else 0.0,
        "dense_allreduce": round(1e3 * 2 * (dense_bytes / N) / link, 4) if N > 1 else 0.0,
        "bytes": {"message": msg_bytes, "merged": merged_bytes, "dense": dense_bytes, "links": min(peers, 7)}}
    g_ms = sum(v["merge_ms_gather"] for v in out["tables"].values())
    o_ms = sum(v["merge_ms_owner"] for v in out["tables"].values())
    w = out["wire_model_ms"]
    out["tail_ms"] = {"gather": round(g_ms + ms_tail + w["gather_allgather"] + w["dense_allreduce"], 4),
                      "owner": round(o_ms + ms_tail + w["owner_alltoall"] + w["owner_allgather"] + w["dense_allreduce"], 4)}
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and mapx has no CPU path")
    local %= torch.cuda.device_count()          # rehearsal: several ranks may share one GPU (gloo)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1 or os.environ.get("MAPX_FORCE_DP", "0") == "1":   # FORCE_DP: one-rank RCCL rehearsal of the exchange
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("MAPX_DIST_BACKEND", "nccl")     # "nccl" = RCCL over xGMI
        if backend == "nccl":
            from mapx import parallel as _par
            _par.init_rccl(device)
        else:
            torch.distributed.init_process_group(backend=backend)
    from mapx import ops, parallel
    dp_path = parallel.exchanging()
    tr, cfg, ids, labels, feat_count = build(args, device, rank)
    train = tr._begin("bench")
    B = args.batch
    gen = tr._generator()
    batches = train.batches(B, True, gen, (rank, world), rows=True)     # as Trainer.MFP_pretrain deals them
    tr.model.train()

    def next_batch():
        nonlocal batches
        try:
            return next(batches)
        except StopIteration:
            batches = train.batches(B, True, gen, (rank, world), rows=True)
            return next(batches)

    for _ in range(args.preroll + args.warmup):
        mfp_step(tr, *next_batch())
    if args.fake_world:
        torch.cuda.synchronize()
        res = {"fake_world": [fake_world(tr, argparse.Namespace(**{**vars(args), "fake_world": n}), next_batch)
                              for n in sorted({1, 2, 4, args.fake_world})],
               "note": "merge and optimizer measured on one MI355X from N real per-rank messages; wire times are a "
                       "model (bytes / 153 GB/s per link, 7 links, direct sends) — unmeasured on xGMI"}
        print(json.dumps(res))
        return
    if os.environ.get("MAPX_GRAPH", "1") != "0" and args.preroll + args.warmup > tr.GRAPH_AFTER:
        # the headline number is the captured step: a silent fall-back to eager must not pass for it
        live = [g for g in tr._graphs.values() if not isinstance(g, int)]
        if not (tr.use_graph and live):
            raise SystemExit("bench.py: the step is not running from its hipGraph (capture failed?); "
                             "set MAPX_GRAPH=0 to measure the eager step on purpose")
    graphed = [g for g in tr._graphs.values() if getattr(g, "early", False)]
    for g in graphed:
        g.host_s = [0.0] * len(g.host_s)
        g.poll_s = 0.0
        g.spin_s = 0.0
    captures0 = sum(g.captures for g in graphed)
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        # the batch is cut from the HBM-resident split inside the timed region: its row numbers (a slice of
        # the epoch's device permutation) go to the step, whose mask kernel reads the rows through them
        loss = mfp_step(tr, *next_batch())
    parallel.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    staged = [next_batch() for _ in range(max(5, min(20, args.steps)))]      # for the per-kernel pass below
    dp_info = None
    if graphed:     # host-side phases of the data-parallel step (this rank), microseconds per step
        names = ("launch", "counts", "tail_capture", "dense_allreduce", "tail")
        dp_info = {"host_us_per_step": {n: round(1e6 * sum(g.host_s[i] for g in graphed) / args.steps, 1)
                                        for i, n in enumerate(names)},
                   # host time between the replay's launch and its segment counts: mostly asleep (wait), the
                   # last part polling the mailbox with a back-off (spin)
                   "poll_us_per_step": round(1e6 * sum(g.poll_s for g in graphed) / args.steps, 1),
                   "spin_us_per_step": round(1e6 * sum(g.spin_s for g in graphed) / args.steps, 1),
                   "tail_captures_in_timed_region": sum(g.captures for g in graphed) - captures0,
                   "message_sizes": sorted({sz for g in graphed for sz in g.tails})}
    if world > 1:                                       # the slowest rank sets the time
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    # per-kernel durations: HIP events on the launch stream around every kernel class, taken on
    # the same step run eagerly right after the timed region (a graph replay has no
    # per-kernel launch points to bracket; kernel durations do not depend on the launch mode)
    ops.serialize_streams = True          # no side streams: every kernel is timed running alone
    for t in tr.optimizer.tables:         # ... and in program order: no row update squeezed in front of the
        t.early_ok = False                # cross tower's GEMMs (it leaves them a cold L2: +14 us each)
    with ops.Timers() as timers:
        for X, Y in staged:
            # park the GPU for ~4 ms first so that the host enqueues the whole step ahead of it:
            # the event pairs then bracket back-to-back kernels, not host launch gaps
            torch.cuda._sleep(8_000_000)
            tr._mfp_step(X, Y)
        torch.cuda._sleep(8_000_000)
        timers.calibrate()
    torch.cuda.synchronize()
    ops.serialize_streams = False
    ksteps = max(5, min(20, args.steps))
    if torch.distributed.is_initialized():
        parallel.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return
    final_loss = float(loss.detach())
    ksum = timers.summary()
    kernels = {}
    # kernels whose work depends on the data (how many DISTINCT / stale rows a batch touches): the
    # a-priori byte count is only an upper bound, so their rate is priced with the HBM bytes the
    # committed PMC passes measured per launch
    data_dependent = ("table_adam_update", "table_adam_catchup", "nce_table_grad", "seg_reduce_rows")
    # the grouped feat_encoder kernels compute in fp32 in either mode; the dense GEMM classes follow --dtype
    from mapx import ops as _ops
    f32_kind = "f32h2" if _ops.H2 else "f32x3"          # (the grouped encoder kernels compute in fp32 in either mode)
    mfma_peak = lambda name: MFMA_PEAK_TFLOPS[f32_kind if ("grouped" in name or args.dtype == "f32") else args.dtype]
    gemm_flops = 0.0
    for name, s in ksum.items():
        per_launch = s["work"] / s["launches"]
        measured = pmc_traffic(name, args.dtype) if name in data_dependent else None
        if measured:
            per_launch = measured
        rate = per_launch / (s["avg_us"] * 1e-6)
        if name.startswith("gemm"):
            gemm_flops += s["work"] / ksteps
            kernels[name] = dict(bound="mfma", achieved=rate / 1e12, peak=mfma_peak(name), unit="TFLOP/s",
                                 frac=rate / 1e12 / mfma_peak(name))
            if mfma_peak(name) in (MFMA_PEAK_TFLOPS["f32x3"], MFMA_PEAK_TFLOPS["f32h2"]):
                # algorithmic fp32 flop/s; `peak` = the 16-bit dense MFMA peak / MFMAs per product; for comparison the
                # same rate against the fp32 MFMA instruction's own peak
                six = mfma_peak(name) == MFMA_PEAK_TFLOPS["f32x3"]
                kernels[name].update(peak_is="bf16 dense MFMA peak / 6 (three-piece bf16 split, six MFMAs per product)" if six else
                                     "fp16 dense MFMA peak / 3 (operands scaled by their magnitude records and cut into two "
                                     "fp16 pieces, three MFMAs per product)",
                                     frac_of_f32_mfma_peak=rate / 1e12 / MFMA_PEAK_TFLOPS["f32"])
            mb = mfma_busy(name, args.dtype)
            if mb is not None:
                kernels[name]["mfma_busy"] = mb
        else:
            kernels[name] = dict(bound="hbm", achieved=rate / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                                 frac=rate / 1e9 / HBM_PEAK_GBS)
        kernels[name].update(avg_us=s["avg_us"], launches_per_step=s["launches"] / ksteps,
                             ms_per_step=s["total_ms"] / ksteps, traffic=pmc_traffic(name, args.dtype),
                             algorithmic_per_launch=per_launch,
                             bytes_from="pmc" if measured else "model")
    dominant = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
    roofline = dict(kernel=dominant, **{k: kernels[dominant][k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic",
                                                                         "peak_is", "frac_of_f32_mfma_peak", "mfma_busy")
                                        if k in kernels[dominant]})
    hbm_name = "nce_fwd"
    out = {
        "metric": "pretrain samples/sec (DCNv2+MFP, Avazu, bs4096)", "value": world * B * args.steps / dt,
        "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"DCNv2+MFP pretrain step, {args.workload}-shaped synthetic ids "
                               f"({'uniform' if args.uniform else 'Zipf(1.1)'} per field), F={cfg.num_fields}, "
                               f"V={cfg.input_size}, E=16, H=1000x3, cross x3, P=32, K=25, mask_ratio 0.3"
                               + (", bf16 GEMM operands / activations over fp32 master weights, tables and optimizer"
                                  if args.dtype == "bf16" else
                                  (", fp32 GEMMs as three fp16 MFMAs per product (operands scaled by per-tensor magnitude records "
                                   "and cut into two fp16 pieces, fp32 accumulation; weights cut once per optimizer step)"
                                   if _ops.H2 else
                                   ", fp32 GEMMs as six bf16 MFMAs per product (operands cut into three bf16 pieces, fp32 accumulation)")),
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                   "launch": ("eager" if not tr.use_graph else
                              "hipGraph replay of mask + forward + backward; pack and merge + optimizer replayed per message size, RCCL calls eager"
                              if dp_path else "hipGraph replay of the whole step"),
                   "table_optimizer": "row-sparse AdamW with lazy replay of untouched rows, "
                                      + f"{args.preroll} untimed pre-roll steps"},
        "roofline": roofline,
        "roofline_hbm": dict(kernel=hbm_name, **{k: kernels[hbm_name][k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}),
        # all GEMM flops of a step over the whole step's time, against the dtype's dense MFMA peak:
        # what the step as a whole makes of the matrix cores (the per-kernel `roofline.frac` is a
        # kernel running alone)
        "step_mfma_frac": gemm_flops / (dt / args.steps) / 1e12 / MFMA_PEAK_TFLOPS[f32_kind if args.dtype == "f32" else args.dtype],
        "step_gemm_gflop": gemm_flops / 1e9,
        "traffic_stale": pmc_stale(args.dtype), "traffic_from": os.path.relpath(pmc_file(args.dtype), ROOT),
        "kernels": kernels, "final_loss": final_loss,
    }
    if dp_info:
        out["dp"] = dp_info
    if world == 1 and not args.no_cpu_baseline:
        del tr
        torch.cuda.empty_cache()
        out["cpu_baseline"] = cpu_baseline(cfg, ids, labels, feat_count, B, args.cpu_steps)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
