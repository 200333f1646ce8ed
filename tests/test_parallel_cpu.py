"""Data-parallel gradient exchange (mapx/parallel.py) on 2 gloo ranks, CPU only: the wire
logic (counts, padding, all-gather, mean) with an injected torch merge — the product merge
is the HIP reduce-by-key, exercised on the GPU box by tests/test_dp_gpu.py."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Plan:
    def __init__(self, uniq, cap):
        self.n = cap
        self.uniq = torch.zeros(cap, dtype=torch.int32)
        self.uniq[:uniq.numel()] = uniq.to(torch.int32)
        self.n_uniq = torch.tensor([uniq.numel()], dtype=torch.int32)

    def count(self):
        return int(self.n_uniq)


class _Table:
    def __init__(self, V, W, with_bias):
        self.num_rows, self.sparse_grad = V, None
        self.W, self.with_bias = W, with_bias


def _torch_merge(keys, rows, num_rows):
    """Test-side stand-in for parallel.hip_merge (sort + reduce-by-key)."""
    uniq, inv = torch.unique(keys.long(), return_inverse=True)
    out = torch.zeros(keys.numel(), rows.shape[1]).index_add_(0, inv, rows)
    return _Plan(uniq, keys.numel()), out


def _local_grad(rank, V, W, with_bias):
    g = torch.Generator().manual_seed(100 + rank)
    n = 5 + 3 * rank                                   # ranks hold different counts
    uniq = torch.randperm(V, generator=g)[:n].sort().values
    cap = n + 1 + 3 * rank                             # capacity > count: tail is garbage; rank 0's
    r0 = torch.randn(cap, W, generator=g)              # capacity (6) is below rank 1's count (8)
    r1 = torch.randn(cap, generator=g) if with_bias else None
    return uniq, n, r0, r1


def _worker(rank, world, port, V, W, with_bias, out):
    sys.path.insert(0, os.path.join(ROOT, "map-code_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mapx import parallel
    uniq, n, r0, r1 = _local_grad(rank, V, W, with_bias)
    table = _Table(V, W, with_bias)
    table.sparse_grad = (_Plan(uniq, r0.shape[0]), r0, r1)
    (maxc,) = parallel.max_counts(table.sparse_grad[0].n_uniq[:1].to(torch.int64))
    assert maxc == 5 + 3 * (world - 1)                 # the largest rank's count, known to everyone
    parallel.sync_table_grad(table, maxc, merge_fn=_torch_merge)
    plan, m0, m1 = table.sparse_grad
    U = plan.count()
    dense0 = torch.zeros(V, W).index_add_(0, plan.uniq[:U].long(), m0[:U])
    dense1 = torch.zeros(V).index_add_(0, plan.uniq[:U].long(), m1[:U]) if with_bias else torch.zeros(V)
    flat = torch.full((10,), float(rank + 1))
    parallel.allreduce_mean_(flat)
    torch.save((dense0, dense1, flat), f"{out}.{rank}")
    dist.destroy_process_group()


def _owner_worker(rank, world, port, V, out):
    """Two tables (one with a scalar-per-row secondary) through the owner-partitioned exchange."""
    sys.path.insert(0, os.path.join(ROOT, "map-code_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mapx import parallel
    tables = []
    for W, with_bias in ((16, False), (32, True)):
        uniq, n, r0, r1 = _local_grad(rank, V, W, with_bias)
        tb = _Table(V, W, with_bias)
        tb.sparse_grad = (_Plan(uniq, r0.shape[0]), r0, r1)
        tables.append(tb)
    parallel.exchange_owner(tables, merge_fn=_torch_merge)
    res = []
    for tb in tables:
        plan, m0, m1 = tb.sparse_grad
        keep = plan.uniq >= 0
        assert plan.n_uniq is None and plan.n == plan.uniq.numel()
        ids = plan.uniq[keep].long()
        assert ids.unique().numel() == ids.numel(), "an id reached the optimizer twice"
        assert float(m0[~keep].abs().sum()) == 0.0, "padding rows must be zero"
        d0 = torch.zeros(V, tb.W).index_add_(0, ids, m0[keep])
        d1 = torch.zeros(V).index_add_(0, ids, m1[keep]) if tb.with_bias else torch.zeros(V)
        res.append((d0, d1))
    torch.save(res, f"{out}.{rank}")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_owner_partitioned_exchange(tmp_path, world):
    """all-gather of slice sizes -> all-to-all-v of (id, row) slices -> owners merge -> all-gather of the
    merged lists: every rank ends with the mean over ranks of all sparse gradients, each id once."""
    V = 50
    out = str(tmp_path / "res")
    mp.spawn(_owner_worker, args=(world, _free_port(), V, out), nprocs=world, join=True)
    for ti, (W, with_bias) in enumerate(((16, False), (32, True))):
        exp0, exp1 = torch.zeros(V, W), torch.zeros(V)
        for r in range(world):
            uniq, n, r0, r1 = _local_grad(r, V, W, with_bias)
            exp0.index_add_(0, uniq, r0[:n])
            if with_bias:
                exp1.index_add_(0, uniq, r1[:n])
        exp0 /= world
        exp1 /= world
        for r in range(world):
            d0, d1 = torch.load(f"{out}.{r}")[ti]
            assert torch.allclose(d0, exp0, atol=1e-6), "every rank must hold the same merged mean"
            assert torch.allclose(d1, exp1, atol=1e-6)


@pytest.mark.parametrize("W,with_bias", [(16, False), (32, True)])
def test_sparse_gradient_exchange_two_ranks(tmp_path, W, with_bias):
    V, world = 50, 2
    out = str(tmp_path / "res")
    mp.spawn(_worker, args=(world, _free_port(), V, W, with_bias, out), nprocs=world, join=True)
    exp0, exp1 = torch.zeros(V, W), torch.zeros(V)
    for r in range(world):
        uniq, n, r0, r1 = _local_grad(r, V, W, with_bias)
        exp0.index_add_(0, uniq, r0[:n])
        if with_bias:
            exp1.index_add_(0, uniq, r1[:n])
    exp0 /= world
    exp1 /= world
    for r in range(world):
        d0, d1, flat = torch.load(f"{out}.{r}")
        assert torch.allclose(d0, exp0, atol=1e-6), "every rank must hold the same merged mean"
        assert torch.allclose(d1, exp1, atol=1e-6)
        assert torch.allclose(flat, torch.full((10,), 1.5))


def test_single_process_is_a_no_op():
    sys.path.insert(0, os.path.join(ROOT, "map-code_amd"))
    from mapx import parallel
    assert parallel.world() == 1 and parallel.rank() == 0
    t = torch.ones(3)
    assert parallel.allreduce_mean_(t) is t
    parallel.barrier()


def test_message_size_buckets():
    """Exchange message sizes: never below the count, at most 1/16 of a power of two above it (so
    <= 6.25 % padding), monotone, and few distinct values over a +-2 % drift of the count."""
    from mapx import parallel
    for c in list(range(1, 3000, 7)) + [17_000, 17_408, 17_409, 86_000, 90_113, 250_000, 2_000_000]:
        m = parallel.message_size(c)
        assert m >= c and m >= 256
        if c >= 4096:
            assert m - c <= (1 << (c.bit_length() - 1)) // 16
    for c in range(1, 200_000, 997):
        assert parallel.message_size(c + 1) >= parallel.message_size(c)
    assert len({parallel.message_size(c) for c in range(84_000, 88_000)}) <= 2


# ------------------------------------------------------------------ the ranks stay in lockstep
class _ToySplit:
    pass


def test_ragged_epoch_deals_full_batches_in_whole_rounds():
    """n = 9 full batches + 100 rows, B = 512, 2 ranks used to hand the ragged
    batch to the last rank only.  With world > 1 only full batches are dealt, in whole rounds."""
    from mapx.trainer import DeviceSplit
    import numpy as np

    class DS:
        X = np.arange((512 * 9 + 100) * 3, dtype=np.int64).reshape(-1, 3)
        Y = np.arange(512 * 9 + 100, dtype=np.int64)
    sp = DeviceSplit(DS, "cpu")
    seen = []
    for w in (2, 3, 4):
        per_rank = [[tuple(x.shape) for x, _ in sp.batches(512, False, None, (r, w))] for r in range(w)]
        assert all(shapes == per_rank[0] for shapes in per_rank), "every rank must see the same batch shapes"
        assert all(s == (512, 3) for s in per_rank[0])
        assert len(per_rank[0]) == sp.num_batches(512, w) == (9 // w)
        rows = torch.cat([y for r in range(w) for _, y in sp.batches(512, False, None, (r, w))])
        assert rows.unique().numel() == rows.numel() == 512 * w * (9 // w)      # no row twice
        seen.append(len(per_rank[0]))
    # one process keeps the reference's drop_last=False epoch (trainer.py:51-58)
    shapes = [x.shape[0] for x, _ in sp.batches(512, False)]
    assert shapes == [512] * 9 + [100] and sp.num_batches(512) == 10


def _lockstep_worker(rank, world, port, out):
    """Trainer.run_step's graph/eager switch with a capture that fails on rank 1 only: both ranks
    must go back to the eager step together and issue the same collectives step for step."""
    sys.path.insert(0, os.path.join(ROOT, "map-code_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mapx import parallel, trainer as T
    calls = []
    real_allreduce = dist.all_reduce

    def spy(t, op=dist.ReduceOp.SUM, **kw):
        calls.append(("all_reduce", str(op), tuple(t.shape)))
        return real_allreduce(t, op=op, **kw)
    dist.all_reduce = spy

    class Args:
        per_gpu_train_batch_size = 8
        sampling_method = "randint"

    class Opt:
        max_grad_norm = 0.0
        tables = []

    tr = T.Trainer.__new__(T.Trainer)
    tr.args, tr.optimizer, tr.use_graph, tr._graphs = Args(), Opt(), True, {}
    modes = []

    def eager_step(X, Y):                      # what an eager DP step does on the wire
        flat = torch.ones(4)
        parallel.allreduce_mean_(flat)
        modes.append("eager")
        return flat

    class FakeGraph:                           # stands in for GraphedBackward (needs a GPU)
        def __init__(self, trainer, fn, X, Y):
            if rank == 1:
                raise RuntimeError("injected capture failure")

        def __call__(self, X, Y):
            c = torch.ones(2, dtype=torch.int64)
            dist.all_reduce(c, op=dist.ReduceOp.MAX)      # a graphed step starts with the counts
            modes.append("graph")
            return c
    T.GraphedBackward = FakeGraph
    tr._ctr_step = eager_step
    tr._mfp_step = tr._rfd_step = eager_step
    tr._ctr_fwd_bwd = tr._mfp_fwd_bwd = tr._rfd_fwd_bwd = eager_step
    X, Y = torch.zeros(8, 3, dtype=torch.int64), torch.zeros(8, dtype=torch.int64)
    for _ in range(T.Trainer.GRAPH_AFTER + 3):
        tr.run_step("ctr", X, Y)
    torch.save((calls, modes, tr.use_graph), f"{out}.{rank}")
    dist.destroy_process_group()


def test_capture_failure_on_one_rank_sends_every_rank_back_to_eager(tmp_path):
    out = str(tmp_path / "res")
    mp.spawn(_lockstep_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    (c0, m0, g0), (c1, m1, g1) = (torch.load(f"{out}.{r}") for r in range(2))
    assert c0 == c1, "ranks issued different collective sequences"
    assert m0 == m1 == ["eager"] * len(m0)
    assert g0 is False and g1 is False
    # the one extra collective is the agreement itself (MIN), at the step where capture was tried
    assert sum(1 for c in c0 if "MIN" in c[1]) == 1
