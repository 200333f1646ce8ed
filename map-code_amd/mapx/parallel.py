"""Data parallelism for the pretraining step: one process per GPU, RCCL over xGMI
(torch.distributed backend "nccl"), gloo for CPU rehearsal.

The reference has no working multi-GPU path (init_process_group, then
unsynchronised replicas); this is new design.  The minibatch shards by rows: every rank runs
forward/backward on its own `per_gpu_train_batch_size` rows with its own masks/negatives
(rank-offset Philox stream), then

  * dense gradients: ONE all-reduce(sum) per flat group (~15 MB fp32 at Avazu-MFP), / world;
  * table gradients: each rank's deduplicated (row id, gradient row) list is all-gathered
    (padded to the largest rank's count with zero rows on id -1, which nothing applies), merged by the same
    deterministic reduce-by-key as the local gradient (csrc/segreduce.h), / world.  The
    largest counts of all tables travel in one MAX all-reduce: one host sync per step.

Every replica then applies the identical update, so replicas stay bit-identical without
ever broadcasting parameters.  The merge is injectable (`merge_fn`) so that the exchange
logic is testable with gloo on a CPU-only box.
"""
import os

import torch
import torch.distributed as dist


def init_rccl(device=None):
    """init_process_group("nccl").  MAPX_NCCL_PRIO=1 puts RCCL's stream at high priority (hardware
    queues of their own class); measured on the one-rank rehearsal it costs 0.3-0.5 ms per step —
    a high-priority queue pre-empts the step's GEMM waves — so the default is normal priority."""
    kw = {"device_id": device} if device is not None else {}
    if os.environ.get("MAPX_NCCL_PRIO", "0") == "1":
        from torch.distributed import ProcessGroupNCCL
        kw["pg_options"] = ProcessGroupNCCL.Options(is_high_priority_stream=True)
    dist.init_process_group(backend="nccl", **kw)


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
    if dist.get_backend() == "nccl":            # RCCL averages in the collective: no extra pass over the bucket
        dist.all_reduce(s, op=dist.ReduceOp.AVG)
        return flat
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


# The gather exchange of one table on device tensors, in the three pieces a captured tail needs
# (trainer.GraphedExchangeTail): pack (one kernel, 1/world folded in) | two all-gathers | merge
# (one plan, one reduction that splits rows / scalar again).
def pack_table(table, maxc):
    from . import ops
    plan, r0, r1 = table.sparse_grad
    return ops.pack_sparse(plan, r0, r1, maxc, 1.0 / world(), pad_id=-1)


def gather_buffers(keys, rows):
    """Receive buffers of one table's all-gather, on the messages' device (a captured merge reads them in place)."""
    w = world()
    return (torch.empty(w * keys.shape[0], dtype=torch.int32, device=keys.device),
            torch.empty(w * rows.shape[0], rows.shape[1], dtype=torch.float32, device=rows.device))


def all_gather_table(keys, rows, k_all, r_all):
    if dist.get_backend() == "gloo" and keys.is_cuda:        # gloo moves host memory only: staged both ways
        ks, rs = torch.empty(k_all.shape, dtype=k_all.dtype), torch.empty(r_all.shape, dtype=r_all.dtype)
        dist.all_gather_into_tensor(ks, keys.cpu())
        dist.all_gather_into_tensor(rs, rows.cpu())
        k_all.copy_(ks)
        r_all.copy_(rs)
        return
    dist.all_gather_into_tensor(k_all, keys)
    dist.all_gather_into_tensor(r_all, rows)


def all_gather_tables(msgs, gathered):
    """The (keys, rows) messages of all tables in ONE grouped RCCL launch (keys travel as raw
    32-bit words) instead of two collectives per table; plain calls on other backends."""
    if dist.get_backend() != "nccl":
        for (k, r), (k_all, r_all) in zip(msgs, gathered):
            all_gather_table(k, r, k_all, r_all)
        return
    ins = [t for k, r in msgs for t in (k.view(torch.float32), r.view(-1))]
    outs = [t for k, r in gathered for t in (k.view(torch.float32), r.view(-1))]
    work = dist.group.WORLD.allgather_into_tensor_coalesced(outs, ins)
    if work is not None:
        work.wait()                 # stream-level: the current stream waits for the RCCL stream


def merge_table(table, k_all, r_all):
    from . import ops
    _, r0, r1 = table.sparse_grad
    W0 = r0.shape[1]
    # every rank's message is sorted and ends with its -1 padding: the plan is a merge of world()
    # sorted lists (one ranking launch; as unsigned keys the padding sorts behind every real id,
    # as one run of zero rows which the table optimizer skips)
    mplan = ops.SegPlan(k_all, table.num_rows + 1, sorted_lists=world())
    if r1 is not None:
        m0, m1 = ops.seg_reduce_rows_extra(mplan, r_all, W0, r_all[:, W0], 1, extra_stride=r_all.stride(0))
    else:
        m0, m1 = ops.seg_reduce_rows(mplan, r_all, W0), None
    table.sparse_grad = (mplan, m0, m1)


def _sync_table_grad_hip(table, maxc):
    dev = table.sparse_grad[1].device
    keys, rows = pack_table(table, maxc)
    k_all, r_all = gather_buffers(keys, rows)
    all_gather_table(keys, rows, k_all, r_all)
    merge_table(table, k_all, r_all)


def sync_table_grad(table, maxc, merge_fn=hip_merge):
    """Replace table.sparse_grad by the mean over ranks of all ranks' sparse gradients."""
    plan, r0, r1 = table.sparse_grad
    if merge_fn is hip_merge and r0.is_cuda:
        return _sync_table_grad_hip(table, maxc)
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


# ------------------------------------------------------------------------- owner-partitioned exchange
# the "reduce-scatter of sparse gradients".  Row ids are owned by contiguous id
# ranges (V / world each).  A rank's deduplicated list is sorted by id, so the rows it owes each
# owner are one contiguous slice: (1) all ranks all-gather the [world] slice sizes of every table
# (ONE host sync per step), (2) all-to-all-v of (id, row) slices, (3) each owner merges what it
# received — 1/world of the keys instead of all of them — with the deterministic reduce-by-key,
# (4) all-gather of the owners' merged lists, padded with id -1 (the table optimizer skips
# negative ids) to the largest owner load, which every rank can compute from the sizes of (1).
# Against the all-gather of raw lists an owner merges world times fewer keys (measured on real per-rank
# messages, DESIGN §5: NCE table at N = 8, 0.100 vs 0.147 ms).  What it does NOT save is volume on the way back:
# step (4) must be sized by the host BEFORE the owners have merged, and the only bound the sizes of (1) give is
# the number of rows an owner RECEIVES (`cap_o` below) — with the NCE table's heavily overlapping row sets
# (8 ranks x 86 k rows merge to 101 k distinct rows) that is ~7x the merged list, so the padded owner gather
# moves what the raw gather moves.  Sizing it by the merged counts needs a second host synchronisation in the
# middle of the tail (merge -> publish -> MAX over ranks -> gather), i.e. a host round trip with nothing to
# overlap it.  And contiguous id ranges are uneven in ROWS (56.7 k of a rank's 86 k NCE rows fall to owner 0:
# every field's vocabulary starts with its frequent values).  "gather" therefore stays the default and the one
# with a captured tail (trainer.GraphedExchangeTail); this path is kept, eager, for an N-GPU measurement
# (MAPX_DP_EXCHANGE=owner).
EXCHANGE = os.environ.get("MAPX_DP_EXCHANGE", "gather")     # "gather" | "owner"


class GatheredPlan:
    """Row-id list of a gathered table gradient: `uniq` [n] int32 with -1 padding, all rows valid
    or zero.  Stands in for a SegPlan where the optimizer / clipping read a sparse gradient."""

    def __init__(self, uniq):
        self.uniq, self.n, self.n_uniq = uniq, uniq.numel(), None

    def count(self):
        return int((self.uniq >= 0).sum())


def _slice_sizes(plan, num_rows, w):
    """[w] int64 (device): how many of this rank's unique ids fall into each owner's id range."""
    dev = plan.uniq.device
    cap = plan.uniq.shape[0]
    big = torch.iinfo(torch.int32).max
    keys = torch.where(torch.arange(cap, device=dev) < plan.n_uniq[0], plan.uniq,
                       torch.full((), big, dtype=plan.uniq.dtype, device=dev))
    chunk = -(-num_rows // w)
    edges = torch.arange(w + 1, device=dev, dtype=torch.int64) * chunk
    edges[w] = big                                   # the last owner takes the remainder
    bounds = torch.searchsorted(keys.to(torch.int64), edges)
    return bounds[1:] - bounds[:-1]


def _all_to_all(out, inp, out_splits, in_splits):
    so, si = _staged(out), _staged(inp)
    dist.all_to_all_single(so, si, out_splits, in_splits)
    if so is not out:
        out.copy_(so)


def exchange_owner(tables, merge_fn=hip_merge):
    w, r = world(), rank()
    dev = tables[0].sparse_grad[1].device
    sizes = torch.stack([_slice_sizes(tb.sparse_grad[0], tb.num_rows, w) for tb in tables])      # [T, w]
    flat = _staged(sizes).contiguous().view(-1)
    all_sizes = torch.empty(w * flat.numel(), dtype=torch.int64, device=flat.device)
    dist.all_gather_into_tensor(all_sizes, flat)
    all_sizes = all_sizes.view(w, len(tables), w).tolist()      # the step's one host sync: [from][table][to]
    for t, tb in enumerate(tables):
        plan, r0, r1 = tb.sparse_grad
        W0 = r0.shape[1]
        send = [all_sizes[r][t][k] for k in range(w)]
        recv = [all_sizes[j][t][r] for j in range(w)]
        n_send, n_recv = sum(send), sum(recv)
        loads = [sum(all_sizes[j][t][k] for j in range(w)) for k in range(w)]     # rows each owner receives
        cap_o = max(1, max(loads))
        if r1 is not None:                           # the scalar-per-row gradient rides in 4 extra columns
            rows = torch.zeros(n_send, W0 + 4, dtype=torch.float32, device=dev)
            rows[:, :W0] = r0[:n_send]
            rows[:, W0] = r1[:n_send]
        else:
            rows = r0[:n_send].contiguous()
        rows = rows / w                              # mean over ranks
        keys = plan.uniq[:n_send].contiguous()
        Wp = rows.shape[1]
        k_in = torch.empty(n_recv, dtype=torch.int32, device=dev)
        r_in = torch.empty(n_recv, Wp, dtype=torch.float32, device=dev)
        _all_to_all(k_in, keys, recv, send)
        _all_to_all(r_in, rows, recv, send)
        k_own = torch.full((cap_o,), -1, dtype=torch.int32, device=dev)
        r_own = torch.zeros(cap_o, Wp, dtype=torch.float32, device=dev)
        if n_recv > 0:
            mplan, merged = merge_fn(k_in, r_in, tb.num_rows)
            live = torch.arange(n_recv, device=dev) < mplan.n_uniq[0]
            k_own[:n_recv] = torch.where(live, mplan.uniq[:n_recv], k_own[:n_recv])
            r_own[:n_recv] = torch.where(live.unsqueeze(1), merged[:n_recv], r_own[:n_recv])
        k_all = torch.empty(w * cap_o, dtype=torch.int32, device=_staged(k_own).device)
        r_all = torch.empty(w * cap_o, Wp, dtype=torch.float32, device=_staged(r_own).device)
        dist.all_gather_into_tensor(k_all, _staged(k_own))
        dist.all_gather_into_tensor(r_all, _staged(r_own))
        k_all, r_all = k_all.to(dev), r_all.to(dev)
        gplan = GatheredPlan(k_all)
        if r1 is not None:
            tb.sparse_grad = (gplan, r_all[:, :W0].contiguous(), r_all[:, W0].contiguous())
        else:
            tb.sparse_grad = (gplan, r_all, None)


def sync_dense(optimizer):
    """Mean over ranks of the dense gradients: one all-reduce per flat group."""
    from . import ops
    optimizer.collect_torch_grads()
    ops.join_pending()                  # side streams of the backward pass (table gradients)
    ops.flush_deferred()                # dense gradients must be final before the all-reduce
    if exchanging() and dist.get_backend() == "nccl" and len(optimizer.groups) > 1:
        opts = dist.AllreduceCoalescedOptions()
        opts.reduceOp = dist.ReduceOp.AVG
        work = dist.group.WORLD.allreduce_coalesced([g["g"] for g in optimizer.groups], opts)
        if work is not None:
            work.wait()
        return
    for g in optimizer.groups:
        allreduce_mean_(g["g"])


def message_size(maxc):
    """Exchange message length for a largest count of `maxc`: rounded up to 1/16 of its power of
    two (<= 6 % padding), so that a run sees a handful of distinct sizes and the tail of the step
    can be replayed from a graph captured per size (trainer.GraphedExchangeTail).  Padding
    entries are zero rows on id -1, like the entries that level a rank up to the largest count."""
    g = max(256, (1 << max(0, int(maxc).bit_length() - 1)) >> 4)
    return -(-int(maxc) // g) * g


def sync_gradients(optimizer, merge_fn=hip_merge, known_max=None):
    """Call between backward() and optimizer.step().  `known_max`: the tables' largest
    unique-row counts over the ranks when the caller already has them (trainer.GraphedBackward
    fetches them while backward still runs); otherwise one MAX all-reduce + host sync here."""
    if not exchanging():
        return
    sync_dense(optimizer)
    tabs = [t.table for t in optimizer.tables if t.table.sparse_grad is not None]
    if not tabs:
        return
    if EXCHANGE == "owner":
        exchange_owner(tabs, merge_fn)
        return
    if known_max is None:
        known_max = max_counts(torch.stack([tb.sparse_grad[0].n_uniq[0] for tb in tabs]).to(torch.int64))
    for tb, maxc in zip(tabs, known_max):
        sync_table_grad(tb, maxc, merge_fn)


def barrier():
    if world() > 1:
        dist.barrier()


def all_agree(ok):
    """True iff `ok` is true on EVERY rank (one MIN all-reduce of a host int; no collective and
    plain `ok` on a single rank).  For decisions that change which collectives a rank issues
    next — e.g. falling back from the captured step to the eager one — so that no rank takes
    them alone."""
    if not exchanging():
        return bool(ok)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    if dist.get_backend() == "nccl":
        flag = flag.cuda()
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(int(flag.item()))
