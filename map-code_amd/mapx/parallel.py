"""Data parallelism for the pretraining step: one process per GPU, RCCL over xGMI
(torch.distributed backend "nccl"), gloo for CPU rehearsal.

The reference has no working multi-GPU path (init_process_group, then
unsynchronised replicas); this is new design.  The minibatch shards by rows: every rank runs
forward/backward on its own `per_gpu_train_batch_size` rows with its own masks/negatives
(rank-offset Philox stream), then

  * dense gradients: ONE all-reduce(sum) per flat group (~15 MB fp32 at Avazu-MFP), / world;
  * table gradients: each rank's deduplicated (row id, gradient row) list is all-gathered
    (padded to the largest rank's count with zero rows on id 0), merged by the same
    deterministic reduce-by-key as the local gradient (csrc/segreduce.h), / world.  The
    largest counts of all tables travel in one MAX all-reduce: one host sync per step.

Every replica then applies the identical update, so replicas stay bit-identical without
ever broadcasting parameters.  The merge is injectable (`merge_fn`) so that the exchange
logic is testable with gloo on a CPU-only box.
"""
import os

import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def exchanging():
    """True when gradients go through the collectives: more than one rank, or a single rank with
    MAPX_FORCE_DP=1 (a one-rank RCCL group on a one-GPU box runs the very calls — all-reduce,
    MAX all-reduce of the counts, all-gathers, merge — that N ranks run; rehearsal only)."""
    if world() > 1:
        return True
    return os.environ.get("MAPX_FORCE_DP", "0") == "1" and dist.is_available() and dist.is_initialized()


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _staged(t):
    """gloo moves host memory only."""
    return t.cpu() if (dist.get_backend() == "gloo" and t.is_cuda) else t


def allreduce_mean_(flat):
    if not exchanging():
        return flat
    s = _staged(flat)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    s.div_(world())
    if s is not flat:
        flat.copy_(s)
    return flat


def max_counts(counts_dev):
    """Element-wise MAX over ranks of a small int64 device vector -> Python list (ONE host sync
    per step for all tables together)."""
    cs = _staged(counts_dev.clone())
    dist.all_reduce(cs, op=dist.ReduceOp.MAX)
    return [max(1, int(c)) for c in cs.tolist()]


def gather_sparse(uniq, rows, n_uniq_dev, maxc):
    """All-gather every rank's first n_uniq (id, row) pairs, padded to `maxc` (the largest count
    over ranks) with zero rows on id 0; the local count stays on the device (a mask, no sync).
    -> (keys int32 [world*maxc], rows f32 [world*maxc, W])."""
    w = world()
    dev = rows.device
    W = rows.shape[1]
    live = torch.arange(maxc, device=dev) < n_uniq_dev
    k_loc = torch.where(live, uniq[:maxc], torch.zeros((), dtype=uniq.dtype, device=dev))
    r_loc = torch.where(live.unsqueeze(1), rows[:maxc], torch.zeros((), dtype=rows.dtype, device=dev))
    k_all = torch.empty(w * maxc, dtype=torch.int32, device=_staged(k_loc).device)
    r_all = torch.empty(w * maxc, W, dtype=torch.float32, device=_staged(r_loc).device)
    dist.all_gather_into_tensor(k_all, _staged(k_loc))
    dist.all_gather_into_tensor(r_all, _staged(r_loc))
    return k_all.to(dev), r_all.to(dev)


def hip_merge(keys, rows, num_rows):
    """Product merge: sort + deterministic reduce-by-key on the GPU -> (plan, merged rows)."""
    from . import ops
    plan = ops.SegPlan(keys, num_rows)
    return plan, ops.seg_reduce_rows(plan, rows, rows.shape[1])


def sync_table_grad(table, maxc, merge_fn=hip_merge):
    """Replace table.sparse_grad by the mean over ranks of all ranks' sparse gradients."""
    plan, r0, r1 = table.sparse_grad
    W0 = r0.shape[1]
    if r1 is not None:                      # ride the scalar-per-row gradient in 4 extra columns
        packed = torch.zeros(r0.shape[0], W0 + 4, dtype=torch.float32, device=r0.device)
        packed[:, :W0] = r0
        packed[:, W0] = r1
    else:
        packed = r0
    if packed.shape[0] < maxc:              # capacity = this rank's key count; another rank may hold more
        packed = torch.cat([packed, packed.new_zeros(maxc - packed.shape[0], packed.shape[1])])
    uniq = plan.uniq if plan.uniq.shape[0] >= maxc else torch.cat(
        [plan.uniq, plan.uniq.new_zeros(maxc - plan.uniq.shape[0])])
    keys, rows = gather_sparse(uniq, packed, plan.n_uniq[0], maxc)
    mplan, merged = merge_fn(keys, rows, table.num_rows)
    merged = merged / world()
    if r1 is not None:
        table.sparse_grad = (mplan, merged[:, :W0].contiguous(), merged[:, W0].contiguous())
    else:
        table.sparse_grad = (mplan, merged, None)


def sync_gradients(optimizer, merge_fn=hip_merge):
    """Call between backward() and optimizer.step()."""
    if not exchanging():
        return
    from . import ops
    optimizer.collect_torch_grads()
    ops.flush_deferred()                # dense gradients must be final before the all-reduce
    for g in optimizer.groups:
        allreduce_mean_(g["g"])
    tabs = [t.table for t in optimizer.tables if t.table.sparse_grad is not None]
    if not tabs:
        return
    counts = torch.stack([tb.sparse_grad[0].n_uniq[0] for tb in tabs]).to(torch.int64)
    for tb, maxc in zip(tabs, max_counts(counts)):
        sync_table_grad(tb, maxc, merge_fn)


def barrier():
    if world() > 1:
        dist.barrier()
