"""Data parallelism for the pretraining step: one process per GPU, RCCL over xGMI
(torch.distributed backend "nccl"), gloo for CPU rehearsal.

The reference has no working multi-GPU path (init_process_group, then
unsynchronised replicas); this is new design.  The minibatch shards by rows: every rank runs
forward/backward on its own `per_gpu_train_batch_size` rows with its own masks/negatives
(rank-offset Philox stream), then

  * dense gradients: ONE all-reduce(sum) per flat group (~15 MB fp32 at Avazu-MFP), / world;
  * table gradients: each rank's deduplicated (row id, gradient row) list is all-gathered
    (padded to the largest rank's count with zero rows on id 0), merged by the same
    deterministic reduce-by-key as the local gradient (csrc/segreduce.h), / world.

Every replica then applies the identical update, so replicas stay bit-identical without
ever broadcasting parameters.  The merge is injectable (`merge_fn`) so that the exchange
logic is testable with gloo on a CPU-only box.
"""
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _staged(t):
    """gloo moves host memory only."""
    return t.cpu() if (dist.get_backend() == "gloo" and t.is_cuda) else t


def allreduce_mean_(flat):
    if world() == 1:
        return flat
    s = _staged(flat)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    s.div_(world())
    if s is not flat:
        flat.copy_(s)
    return flat


def gather_sparse(uniq, rows, count):
    """All-gather every rank's first `count` (id, row) pairs.
    -> (keys int32 [world*maxc], rows f32 [world*maxc, W]); padding = id 0 with a zero row."""
    w = world()
    dev = rows.device
    counts = torch.zeros(w, dtype=torch.int64)
    counts[rank()] = count
    cs = counts.to(dev) if dist.get_backend() != "gloo" else counts
    dist.all_reduce(cs, op=dist.ReduceOp.SUM)
    maxc = max(1, int(cs.max()))
    W = rows.shape[1]
    k_loc = torch.zeros(maxc, dtype=torch.int32, device=dev)
    r_loc = torch.zeros(maxc, W, dtype=torch.float32, device=dev)
    k_loc[:count] = uniq[:count]
    r_loc[:count] = rows[:count]
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


def sync_table_grad(table, merge_fn=hip_merge):
    """Replace table.sparse_grad by the mean over ranks of all ranks' sparse gradients."""
    if world() == 1 or table.sparse_grad is None:
        return
    plan, r0, r1 = table.sparse_grad
    count = plan.count()
    W0 = r0.shape[1]
    if r1 is not None:                      # ride the scalar-per-row gradient in 4 extra columns
        packed = torch.zeros(r0.shape[0], W0 + 4, dtype=torch.float32, device=r0.device)
        packed[:, :W0] = r0
        packed[:, W0] = r1
    else:
        packed = r0
    keys, rows = gather_sparse(plan.uniq, packed, count)
    mplan, merged = merge_fn(keys, rows, table.num_rows)
    merged = merged / world()
    if r1 is not None:
        table.sparse_grad = (mplan, merged[:, :W0].contiguous(), merged[:, W0].contiguous())
    else:
        table.sparse_grad = (mplan, merged, None)


def sync_gradients(optimizer, merge_fn=hip_merge):
    """Call between backward() and optimizer.step()."""
    if world() == 1:
        return
    from . import ops
    ops.flush_deferred()                # dense gradients must be final before the all-reduce
    for g in optimizer.groups:
        allreduce_mean_(g["g"])
    for t in optimizer.tables:
        sync_table_grad(t.table, merge_fn)


def barrier():
    if world() > 1:
        dist.barrier()
