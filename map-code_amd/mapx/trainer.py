"""Step loops of the hot path (host mirror of reference code/trainer.py): Trainer.train /
MFP_pretrain / RFD_pretrain / eval / test / dynamic_mask / save_model / load_model, with the
reference's step order  mask -> forward -> backward -> (clip) -> optimizer -> scheduler ->
zero_grad  and its checkpoint rules.

Differences that are deliberate:
  * the id matrix of a split is resident in HBM and a batch is a row-gather by a shuffled
    index on the device (the reference collates 4096 numpy rows per step in Python);
  * masks / replacements / negatives are drawn on the device from Philox streams;
  * step metrics stay on the device and are fetched every `logging_steps` steps (the
    reference synchronises twice per step with .item());
  * the reference's NameError at trainer.py:341 (`log` undefined) is not reproduced.
"""
import contextlib
import gc
import logging
import os
import time
import weakref

import numpy as np
import torch

from . import ops, parallel
from .optim import MapxOptimizer

logger = logging.getLogger(__name__)


class RowsRef:
    """A batch that is still inside the HBM-resident split: the rows `sel` of split.X / split.Y.  The MFP
    mask kernel reads the rows through `sel` itself (mapx_dynamic_mask_mfp_rows), so a step needs no
    gather of X and Y and no copy of them into a captured step's static buffers — only the 32 KB of
    row numbers.  Quacks like the (X, Y) tensors where the step loop touches them."""

    def __init__(self, split, sel, order=None, start=0, stride=0, cursor=None, batch=None):
        self.split, self.sel = split, sel
        # where the rows come from: sel == order[start : start + B], the same rank's next batch starts
        # `stride` later.  A captured step (GraphedStep) keeps its own copy of `order` and a device-side
        # cursor instead of receiving 32 KB of row numbers in front of every replay: then `sel` is the
        # whole permutation and the batch is sel[*cursor : *cursor + batch].
        self.order, self.start, self.stride, self.cursor, self.batch = order, start, stride, cursor, batch

    @property
    def shape(self):
        return (self.batch if self.cursor is not None else self.sel.shape[0], self.split.X.shape[1])

    @property
    def is_cuda(self):
        return self.sel.is_cuda

    @property
    def device(self):
        return self.sel.device

    @property
    def X(self):
        return self.split.X[self.sel]

    @property
    def Y(self):
        return self.split.Y[self.sel]

    def clone(self):
        return RowsRef(self.split, self.sel.clone())

    def copy_(self, other):
        self.sel.copy_(other.sel)
        return self


class DeviceSplit:
    """A dataset split resident on the GPU: X int64 [N,F], Y int64 [N]."""

    def __init__(self, dataset, device):
        self.X = torch.as_tensor(np.ascontiguousarray(dataset.X)).to(device)
        self.Y = torch.as_tensor(np.ascontiguousarray(dataset.Y)).to(device)
        self.n = self.X.shape[0]

    def batches(self, batch_size, shuffle, generator=None, shard=(0, 1), rows=False):
        """Yield (X_b, Y_b) — or, with rows=True and shuffling, (RowsRef, RowsRef): the batch as row
        numbers into the resident split (MFP pretraining: the mask kernel gathers the rows itself).
        shard = (rank, world): rank r takes batches r, r+world, ...
        With world > 1 only FULL batches are dealt, and only whole rounds of `world` of them (the
        ragged tail of an epoch is dropped for every rank): all ranks then run the same number of
        steps on the same batch shape, so they take the graph / eager decision together and issue
        the same sequence of collectives, and the mean over ranks is the mean over the global
        batch's rows."""
        order = torch.randperm(self.n, device=self.X.device, generator=generator) if shuffle else None
        r, w = shard
        starts = list(range(0, self.n, batch_size))
        if w > 1:
            starts = starts[:self.n // batch_size]
            starts = starts[:len(starts) - len(starts) % w]
        for bi in range(r, len(starts), w):
            s = starts[bi]
            if order is None:
                yield self.X[s:s + batch_size], self.Y[s:s + batch_size]
            else:
                sel = order[s:s + batch_size]
                if rows:
                    ref = RowsRef(self, sel, order=order, start=s, stride=w * batch_size)
                    yield ref, ref
                else:
                    yield self.X[sel], self.Y[sel]

    def num_batches(self, batch_size, world=1):
        if world > 1:
            return (self.n // batch_size) // world          # full batches only, whole rounds (see batches)
        return (self.n + batch_size - 1) // batch_size


def _weak(obj):
    return obj if isinstance(obj, weakref.ProxyTypes) else weakref.proxy(obj)


@contextlib.contextmanager
def _capture(graph):
    """torch.cuda.graph(graph), hardened against what other code may do while a stream captures:
    * with a process group alive the capture runs in "thread_local" error mode.  In the default
      "global" mode ANY thread's unsafe HIP call invalidates the capture, and ProcessGroupNCCL's
      watchdog thread queries events all the time: about one capture in eight died with "operation
      failed due to a previous error during capture" (and took the process with it).
      thread_local still rejects unsafe calls from the capturing thread itself;
    * Python's cyclic garbage collector is off for the duration: a collection that starts inside
      the capture may destroy graphs / events / pinned buffers of objects that died earlier
      (e.g. a previous Trainer and its captured step reference each other), and those
      destructors are not capture-safe — observed as "Fatal Python error: Aborted ...
      Garbage-collecting" in the middle of a forward pass being captured."""
    kw = {}
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        kw["capture_error_mode"] = "thread_local"
    was_enabled = gc.isenabled()
    gc.collect()
    gc.disable()
    ops.amax_capture_begin(torch.device("cuda", torch.cuda.current_device()))     # the capture's magnitude records
    try:
        with torch.cuda.graph(graph, **kw):
            yield
    finally:
        ops.amax_capture_end()
        if was_enabled:
            gc.enable()


class GraphedStep:
    """One whole training step (mask -> forward -> backward -> optimizer -> schedule) captured
    in a hipGraph and replayed on static input buffers: ~115 kernel launches per step cost one
    graph launch, and the step runs at GPU speed instead of Python launch speed.  Everything
    that varies per step lives on the device (update counter, Philox offsets, LR table)."""

    def __init__(self, trainer, step_fn, X, Y):
        self.trainer = _weak(trainer)       # no trainer <-> graph cycle: both die by reference count
        # Batches that are rows of the resident split in an epoch's permutation (RowsRef with `order`): the
        # graph owns a copy of the permutation and a device-side cursor that it advances itself, so a replay
        # is preceded by no copy at all (before: two 5-us copies + launch gaps in front of every step).
        self.walk = (isinstance(X, RowsRef) and X.order is not None and X.stride > 0 and Y is X
                     and os.environ.get("MAPX_WALK", "1") == "1")
        if self.walk:
            self.order = X.order.clone()
            self.cursor = torch.full((1,), X.start, dtype=torch.int64, device=X.device)
            self.stride, self._src, self._expect = X.stride, X.order, X.start
            self.X = self.Y = RowsRef(X.split, self.order, cursor=self.cursor, batch=X.shape[0])
        else:
            self.X = X.clone()
            self.Y = self.X if Y is X else Y.clone()
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        gs, sd = trainer.global_step, trainer.optimizer.steps_done
        try:
            if self.walk:      # the step's last launch (update counter += 1) also moves the cursor to the next batch
                trainer.optimizer.walk_cursor = (self.cursor, self.stride)
            with _capture(self.graph):
                self.out = step_fn(self.X, self.Y)
        finally:
            # the capture pass ran the Python bookkeeping but no kernel
            trainer.optimizer.walk_cursor = None
            trainer.global_step, trainer.optimizer.steps_done = gs, sd

    def __call__(self, X, Y):
        if self.walk:
            if not (isinstance(X, RowsRef) and X.order is not None and Y is X):
                raise TypeError("this captured step walks an epoch's permutation: it takes the RowsRef batches of "
                                "DeviceSplit.batches(rows=True), not tensors (run_step keys its graphs by input kind)")
            if X.start + X.shape[0] > X.order.numel():
                raise IndexError(f"batch rows [{X.start}, {X.start + X.shape[0]}) run past the epoch's "
                                 f"{X.order.numel()} rows")
            if X.order is not self._src:               # a new epoch's permutation
                self.order.copy_(X.order)
                self._src, self._expect = X.order, None
            if X.start != self._expect:                # (first batch of an epoch, or a caller that skips around)
                self.cursor.fill_(X.start)
            self._expect = X.start + self.stride
        else:
            self.X.copy_(X)
            if self.Y is not self.X:
                self.Y.copy_(Y)
        self.graph.replay()
        # the step's host-side effects (a replay runs no Python)
        opt = self.trainer.optimizer
        opt.steps_done += 1
        for t in opt.tables:
            t.stale = True
        self.trainer.global_step += 1
        return self.out


class GraphedBackward:
    """Data-parallel steps (and steps with gradient clipping): mask -> forward -> backward
    captured in a hipGraph, the gradient exchange and the optimizer after each replay.  The graph
    leaves the dense gradients in the optimizer's flat buffers and the tables' sparse gradients
    in buffers of fixed address; the Python references to them (cleared by optimizer.step()) are
    put back after every replay.

    Exchange sizes.  The exchange needs one host-side number per table: the largest unique-row
    count over the ranks.  A count exists as soon as the table's segment plan does — it depends
    on the step's ids, not on any gradient — so during capture a tiny kernel right behind each
    plan stores the count, stamped with the replay number, straight into coherent host memory
    (mapx_publish_i32; a node appended at the end of the capture would also run at the end of
    the replay).  The host picks the counts up while forward/backward still run, takes the MAX
    over ranks on a side stream, and enqueues the exchange + optimizer tail
    (GraphedExchangeTail) behind the graph before it has finished."""

    MAX_TAILS = 12

    def __init__(self, trainer, fwd_bwd_fn, X, Y):
        from . import layers
        self.trainer = _weak(trainer)       # no trainer <-> graph cycle: both die by reference count
        self.X = X.clone()
        self.Y = self.X if Y is X else Y.clone()        # (row references: one list of row numbers, one copy per step)
        # (gloo stages every message through host memory and synchronises the device doing so: its exchange stays
        # eager unless MAPX_DP_GLOO_GRAPH=1 asks for the captured tail — the 2-rank test on one GPU does)
        self.early = (parallel.exchanging() and parallel.EXCHANGE == "gather" and X.is_cuda
                      and (torch.distributed.get_backend() == "nccl"
                           or os.environ.get("MAPX_DP_GLOO_GRAPH", "0") == "1"))
        self.replays = 0
        self.sizes_floor = None
        self.tails, self.captures, self.poll_s, self.spin_s = {}, 0, 0.0, 0.0
        self._arrival = None            # seconds after the replay's launch at which the counts usually land
        self.host_s = [0.0] * 5         # launch | counts | tail capture | dense all-reduce | tail
        tables = [t.table for t in trainer.optimizer.tables]
        if self.early:                  # nothing below may be allocated while a stream is capturing
            self.stamps = torch.zeros(len(tables), dtype=torch.int32, device=X.device)
            self.mailbox = ops.HostMailbox(4 * len(tables))                 # per table [count, stamp, check, -]
            self.pinned_np = self.mailbox.np.reshape(len(tables), 4)
            self.staging = torch.zeros(len(tables), dtype=torch.int64).pin_memory()
            self.counts_dev = torch.zeros(len(tables), dtype=torch.int64, device=X.device)
            self.comm = ops.aux_stream("counts", X.device, high=True)
            self.published = []

            def publish(table, plan):
                i = next(k for k, tb in enumerate(tables) if tb is table)
                ops.publish_i32(plan.n_uniq, 1, self.stamps[i:i + 1], self.mailbox, at=4 * i)
                self.published.append(i)
            layers.plan_observers.append(publish)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        try:
            with _capture(self.graph):
                self.out = fwd_bwd_fn(self.X, self.Y)
                # task queues a backward node may have left for a later node (the encoder's dW / db, a head's dW, the
                # NCE table's gradient): run HERE, inside the graph — left to optimizer.step() they would run once,
                # eagerly, and every replay would miss that gradient
                ops.run_main_tasks()
                ops.run_side_tasks()
                ops.run_late_tasks()
                # deferred partial sums (bias gradients) belong to this graph: the list that names them
                # exists only while the capture runs, a replay would leave them unsummed for the tail
                ops.flush_deferred()
                ops.join_pending()        # side streams a backward node forked: joined inside this graph
        finally:
            if self.early:
                layers.plan_observers.remove(publish)
        self.sparse = [tb.sparse_grad for tb in tables]
        if self.early:
            with_grad = [i for i, tb in enumerate(tables) if tb.sparse_grad is not None]
            if sorted(self.published) != with_grad:
                raise RuntimeError(f"tables with a gradient {with_grad} != tables that published a count {self.published}")
            self.published = with_grad

    def _max_counts(self):
        """Wait until this replay's counts have landed, then MAX over ranks on a side stream (the main stream is
        still busy with the graph).  The counts land a fixed part of the way into the replay (behind the segment
        plans).  The host sleeps through the first 60 % of the time they took on earlier replays — an estimate
        that only replays whose landing was SEEN by a poll update, and that shrinks whenever the counts were
        already there on waking (a sleep may overshoot by more than it was asked for) — then polls: busily for
        at most 300 us, with 50-us sleeps after that.  `poll_s` is the whole wait, `spin_s` the busy part."""
        a, t0 = self.pinned_np, time.perf_counter()
        landed = lambda i: int(a[i, 1]) == self.replays and int(a[i, 0]) + int(a[i, 1]) == int(a[i, 2])
        all_landed = lambda: all(landed(i) for i in self.published)
        if self._arrival is not None and not all_landed():
            nap = self._t_launch + 0.6 * self._arrival - time.perf_counter()
            if nap > 50e-6:
                time.sleep(nap)
        t1 = time.perf_counter()
        seen_landing = not all_landed()
        if not seen_landing and self._arrival is not None:
            self._arrival *= 0.9                                  # overslept, or an idle GPU: sleep less next time
        busy = 0.0
        for i in self.published:
            while not landed(i):
                now = time.perf_counter()
                if now - t1 > 300e-6:
                    time.sleep(50e-6)
                else:
                    busy = now - t1
                if now - t0 > 60.0:
                    seen = {j: tuple(int(x) for x in a[j, :3]) for j in self.published}
                    try:
                        torch.cuda.synchronize()                 # surface the HIP error behind a dead replay
                    except RuntimeError as e:
                        raise RuntimeError(f"the step graph failed before publishing its segment counts: {e}; "
                                           f"mailbox (count, stamp, check) per table = {seen}, "
                                           f"expected stamp {self.replays}") from e
                    raise RuntimeError("the step graph did not publish its segment counts within 60 s: mailbox "
                                       f"(count, stamp, check) per table = {seen}, expected stamp {self.replays} "
                                       "(a stalled collective on another rank keeps the previous tail, and so "
                                       "this replay, from running)")
        now = time.perf_counter()
        self.spin_s += min(busy, now - t1) if seen_landing else 0.0
        if seen_landing:
            took = now - self._t_launch
            self._arrival = took if self._arrival is None else 0.8 * self._arrival + 0.2 * took
        self.poll_s += time.perf_counter() - t0
        T = len(self.published)
        for k, i in enumerate(self.published):
            self.staging[k] = int(a[i, 0])
        with torch.cuda.stream(self.comm):
            self.counts_dev[:T].copy_(self.staging[:T], non_blocking=True)
            return parallel.max_counts(self.counts_dev[:T])

    def __call__(self, X, Y):
        t = [time.perf_counter()]
        self.X.copy_(X)
        if self.Y is not self.X:
            self.Y.copy_(Y)
        self._t_launch = time.perf_counter()
        self.graph.replay()
        self.replays += 1
        for tb, sg in zip(self.trainer.optimizer.tables, self.sparse):
            tb.table.sparse_grad = sg
        if not (self.early and self.published):
            self.trainer._optimizer_step()
            return self.out
        t.append(time.perf_counter())
        # Message sizes never shrink: the counts wander by a few per cent around a bucket boundary, and
        # every new size tuple is a new capture of the tail (~0.1 s, measured: one capture inside 200
        # timed steps = +0.6 ms per step on average).  Holding the largest bucket seen so far (at most
        # one 6 % bucket of extra zero rows) makes the run settle on ONE tail graph; every rank sees
        # the same all-reduced counts, so every rank holds the same sizes.
        sizes = tuple(max(parallel.message_size(m), lo) for m, lo in
                      zip(self._max_counts(), self.sizes_floor or (0,) * len(self.published)))
        self.sizes_floor = sizes
        t.append(time.perf_counter())
        tail = self.tails.get(sizes)
        if tail is None:
            if len(self.tails) >= self.MAX_TAILS:       # sizes drifted: drop the oldest capture
                self.tails.pop(next(iter(self.tails)))
            tables = [self.trainer.optimizer.tables[i].table for i in self.published]
            tail = self.tails[sizes] = GraphedExchangeTail(self.trainer, tables, sizes)
            self.captures += 1
        t.append(time.perf_counter())
        tail(self.host_s)
        self.trainer.global_step += 1
        t.append(time.perf_counter())
        for k, j in enumerate((0, 1, 2, 4)):            # host seconds per phase (bench.py reports them)
            self.host_s[j] += t[k + 1] - t[k]
        return self.out


PACK_FIRST = os.environ.get("MAPX_DP_PACK_FIRST", "1") == "1"      # A/B switch (GraphedExchangeTail)
MERGE_SIDE = os.environ.get("MAPX_DP_MERGE_SIDE", "1") == "1"      # A/B switch: one branch per table's merge


class GraphedExchangeTail:
    """Everything between the last backward kernel and the next step, for ONE tuple of exchange
    message sizes: graph 1 packs every table's (id, row) message; the all-gathers run eagerly as
    one grouped RCCL launch (RCCL calls stay outside the captures); graph 2 merges the gathered
    lists (one branch per table) and applies the optimizer.  ~45 launches and as many tensor
    allocations cost two graph launches and two RCCL calls, which the host issues while backward
    is still running."""

    def __init__(self, trainer, tables, sizes):
        self.trainer = _weak(trainer)       # no trainer <-> graph cycle: both die by reference count
        opt = trainer.optimizer
        saved = [tb.sparse_grad for tb in tables]
        torch.cuda.synchronize()
        self.pack = torch.cuda.CUDAGraph()
        with _capture(self.pack):
            self.msgs = [parallel.pack_table(tb, m) for tb, m in zip(tables, sizes)]
        self.gathered = [parallel.gather_buffers(k, r) for k, r in self.msgs]
        self.merge = torch.cuda.CUDAGraph()
        sd = opt.steps_done
        try:
            with _capture(self.merge):
                # the tables' merges are independent chains of small kernels: one branch each
                # (largest message on the capture stream), joined before the optimizer.  (The
                # dense AdamW as a third branch was measured slower)
                main = torch.cuda.current_stream()
                order = sorted(range(len(tables)), key=lambda i: -sizes[i])
                side = [ops.aux_stream(f"merge{j}", main.device) for j in range(len(order) - 1)] if MERGE_SIDE else []
                for st in side:
                    ops.stream_wait(st, main)
                for j, i in enumerate(order):
                    with torch.cuda.stream(main if (j == 0 or not side) else side[j - 1]):
                        parallel.merge_table(tables[i], *self.gathered[i])
                for st in side:
                    ops.stream_wait(main, st)
                opt.step()
        finally:
            opt.steps_done = sd                         # the capture ran the bookkeeping but no kernel
            for tb, sg in zip(tables, saved):
                tb.sparse_grad = sg

    def __call__(self, host_s):
        opt = self.trainer.optimizer
        if PACK_FIRST:
            # the messages are packed BEFORE the dense all-reduce is issued: the two collectives then follow each other
            # on the communicator's stream without a hop to the main stream and back between them
            self.pack.replay()
        t0 = time.perf_counter()
        parallel.sync_dense(opt)
        host_s[3] += time.perf_counter() - t0
        if not PACK_FIRST:
            self.pack.replay()
        parallel.all_gather_tables(self.msgs, self.gathered)
        self.merge.replay()
        opt.steps_done += 1
        for t in opt.tables:
            if t.table.sparse_grad is not None:
                t.table.sparse_grad = None
                t.stale = True


class Trainer:
    def __init__(self, model, model_config, training_args, train_dataset, eval_dataset):
        self.model, self.model_config, self.args = model, model_config, training_args
        self.device = self.args.device
        self.train_dataset, self.eval_dataset = train_dataset, eval_dataset
        self.global_step = 0
        self.eval_metrics = []
        self.optimizer = None
        self.best_eval_auc, self.best_eval_step = 0, -1
        self.rank, self.world = parallel.rank(), parallel.world()
        self.use_graph = os.environ.get("MAPX_GRAPH", "1") != "0"
        self._graphs = {}
        self._splits = {}
        self._mask_calls = 0
        self._gen = None
        logger.info(f"setting device {self.device}")

    # ------------------------------------------------------------------ plumbing
    def _split(self, dataset):
        key = id(dataset)
        if key not in self._splits:
            self._splits[key] = DeviceSplit(dataset, self.device)
        return self._splits[key]

    def _generator(self):
        if self._gen is None:
            self._gen = torch.Generator(device=self.device)
            self._gen.manual_seed(int(self.args.seed))     # same permutation on every rank
        return self._gen

    def get_optimizer(self, num_training_steps, num_warmup_steps):
        return MapxOptimizer(self.model, self.args, num_training_steps, num_warmup_steps)

    def _begin(self, what):
        train = self._split(self.train_dataset)
        steps_per_epoch = train.num_batches(self.args.per_gpu_train_batch_size, self.world)
        t_total = int(steps_per_epoch * self.args.num_train_epochs)
        t_warmup = int(t_total * self.args.warmup_ratio)
        self.model.to(self.device)
        self._one = torch.ones((), device=self.device)       # dLoss/dLoss, allocated once (no fill launch per step)
        ops.unit_gradient[0] = self._one
        self.optimizer = self.get_optimizer(t_total, t_warmup)
        self.scheduler = self.optimizer                      # get_last_lr() lives there
        if hasattr(self.model, "mfp_criterion"):
            self.model.mfp_criterion.step_counter = self.optimizer.done
        from .layers import HipDropout
        drops = [m for m in self.model.modules() if isinstance(m, HipDropout)]
        for i, m in enumerate(drops):           # masks advance with the device-side update counter; a site's
            m.step_counter, m.seed, m.rank, m.site = self.optimizer.done, int(self.args.seed), self.rank, i + 1
        self._graphs = {}
        logger.info(f"***** running {what} *****")
        for k, v in (("dataset_name", self.args.dataset_name), ("input_size", self.model_config.input_size),
                     ("num_fields", self.model_config.num_fields), ("num_examples", len(self.train_dataset)),
                     ("num_epochs", self.args.num_train_epochs), ("batch_size", self.args.train_batch_size),
                     ("per_gpu_train_batch_size", self.args.per_gpu_train_batch_size),
                     ("total_steps", t_total), ("warmup_steps", t_warmup),
                     ("learning_rate", self.args.learning_rate), ("weight_decay", self.args.weight_decay),
                     ("lr_sched", self.args.lr_sched)):
            logger.info(f"  {k} = {v}")
        self.model.validate_model_config()
        self.global_step, self.eval_metrics = 0, []
        return train

    def _optimizer_step(self, max_counts=None):
        parallel.sync_gradients(self.optimizer, known_max=max_counts)
        self.optimizer.step()                                # + scheduler.step() + zero_grad()
        self.global_step += 1

    @contextlib.contextmanager
    def _step_window(self):
        """Brackets forward + loss.backward() of a step whose gradient exchange / optimizer.step() follows at once
        (MapxOptimizer.backward_window): tables may move their rows as soon as their gradient is final, side streams
        are left for the optimizer to join, and the NCE head leaves its loss totals to its backward's first launch."""
        self.optimizer.backward_window(True)
        try:
            yield
        finally:
            self.optimizer.backward_window(False)

    def _backward(self, loss):
        loss.backward(self._one.view(loss.shape))

    # forward + backward of one step (what a data-parallel graph captures); *_step adds the
    # gradient exchange, the optimizer and the schedule
    def _mfp_fwd_bwd(self, X, Y):
        inputs = self.dynamic_mask({"input_ids": X, "labels": Y}, self.args.sampling_method)
        with self._step_window():
            loss, count, acc = self.model(**inputs)
            self._backward(loss)
        return loss.detach(), self.model.mfp_criterion.last_acc_ratio      # = acc / count, from the loss kernel

    @staticmethod
    def _rows_of(X, Y, labels=True):
        """The tensors of a batch dealt as row references (DeviceSplit.batches(rows=True)): cut from the resident
        split by one launch inside the step — a captured step walks the epoch's permutation by itself."""
        if not isinstance(X, RowsRef):
            return X, Y
        ref = X
        ids = ops.take_rows(ref.split.X, ref.sel, ref.cursor, ref.batch if ref.cursor is not None else None)
        if isinstance(Y, RowsRef):        # (the same object, or its twin in a captured step's static inputs)
            Y = ops.take_rows(Y.split.Y, Y.sel, Y.cursor, Y.batch if Y.cursor is not None else None,
                              as_f32=True) if labels else None          # (the BCE head takes labels.float())
        return ids, Y

    def _rfd_fwd_bwd(self, X, Y):
        X, Y = self._rows_of(X, Y, labels=False)            # (RFD makes its own labels)
        inputs = self.dynamic_mask({"input_ids": X, "labels": Y}, self.args.sampling_method)
        with self._step_window():
            loss, count, acc, pos_ratio = self.model(**inputs)
            self._backward(loss)
        return loss.detach(), acc

    def _ctr_fwd_bwd(self, X, Y):
        X, Y = self._rows_of(X, Y)
        with self._step_window():
            loss, logits = self.model(input_ids=X, labels=Y)
            self._backward(loss)
        return loss.detach(), logits.detach().view(-1)

    def _mfp_step(self, X, Y):
        out = self._mfp_fwd_bwd(X, Y)
        self._optimizer_step()
        return out

    def _rfd_step(self, X, Y):
        out = self._rfd_fwd_bwd(X, Y)
        self._optimizer_step()
        return out

    def _ctr_step(self, X, Y):
        out = self._ctr_fwd_bwd(X, Y)
        self._optimizer_step()
        return out

    GRAPH_AFTER = 3      # eager steps per (kind, shape) before capture (allocator / workspace warm-up)

    def run_step(self, kind, X, Y):
        """One training step.  Full-size batches run from a captured hipGraph after a few eager
        steps: the whole step on one GPU; mask + forward + backward when a gradient exchange
        (N > 1) or gradient clipping needs host-side decisions before the optimizer.  Ragged
        batches and host-side mask sampling ("normal") stay eager."""
        fn = {"mfp": self._mfp_step, "rfd": self._rfd_step, "ctr": self._ctr_step}[kind]
        if isinstance(X, RowsRef) and kind == "mfp" and self.args.pt_type != "MFP":
            ref = X                                        # (the MFP mask kernel is what reads rows through `sel`)
            X, Y = ref.X, (ref.Y if Y is ref else Y)
        graphable = (self.use_graph and X.shape[0] == self.args.per_gpu_train_batch_size
                     and (kind == "ctr" or self.args.sampling_method == "randint"))
        if not graphable:
            return fn(X, Y)
        # (a step captured on row references walks the permutation by itself and cannot take tensor batches)
        key = (kind, tuple(X.shape), isinstance(X, RowsRef) and X.order is not None)
        g = self._graphs.get(key, 0)
        if isinstance(g, int):
            if g < self.GRAPH_AFTER:
                self._graphs[key] = g + 1
                return fn(X, Y)
            err = None
            try:
                if not parallel.exchanging() and self.optimizer.max_grad_norm <= 0:
                    g = GraphedStep(self, fn, X, Y)
                else:   # exchange sizes / the clipping norm are host decisions: graph up to the gradients
                    half = {"mfp": self._mfp_fwd_bwd, "rfd": self._rfd_fwd_bwd, "ctr": self._ctr_fwd_bwd}[kind]
                    g = GraphedBackward(self, half, X, Y)
            except RuntimeError as e:          # a runtime that cannot capture this step: stay eager
                err = e
            # The graphed and the eager step issue different collective sequences, so the ranks
            # switch together or not at all: one rank's failed capture sends every rank back to
            # eager (a capture issues no collective itself, so this all-reduce is the next one on
            # every rank whatever happened above).
            if not parallel.all_agree(err is None):
                if err is None:
                    err = "another rank could not capture it"
                logger.warning(f"hipGraph capture of the {kind} step failed ({err}); continuing eagerly")
                ops.reset_aux_streams()         # side streams forked into the dead capture are unusable
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                self.use_graph = False
                self._graphs[key] = 0
                for t in self.optimizer.tables:
                    t.table.sparse_grad = None
                return fn(X, Y)
            self._graphs[key] = g
        return g(X, Y)

    # ------------------------------------------------------------------ masking (a1, a2)
    def dynamic_mask(self, inputs, sampling_method="normal", masked_index=None, replace_feat=None):
        ids = inputs["input_ids"]
        sel = sel_cursor = batch = None
        if isinstance(ids, RowsRef):                   # rows of the resident split
            if self.args.pt_type == "MFP":
                ref = ids
                ids, sel = ref.split.X, ref.sel
                sel_cursor, batch = ref.cursor, ref.shape[0]
            else:
                ids = ids.X
        F = self.model_config.num_fields
        L = int(F * self.args.mask_ratio)
        seed = int(self.args.seed)
        if self.model.training and self.optimizer is not None:
            # training: the stream offset advances with the optimizer's device-side update counter,
            # so a captured step replays with fresh masks
            offset, offset_dev = (self.rank << 40) + (2 << 36), self.optimizer.done
        else:
            self._mask_calls += 1
            offset, offset_dev = (self.rank << 40) + (3 << 36) + self._mask_calls, None
        if masked_index is None:
            if sampling_method == "normal":        # L distinct fields per row (trainer.py:222)
                masked_index = torch.rand(ids.shape[0] if sel is None else batch, F, device=ids.device,
                                          generator=self._generator()).argsort(1)[:, :L].contiguous()
            elif sampling_method != "randint":
                raise NotImplementedError(sampling_method)
        if self.args.pt_type == "MFP":
            inputs["input_ids"], inputs["labels"], inputs["masked_index"] = ops.dynamic_mask_mfp(
                ids, L, masked_index=masked_index, seed=seed, offset=offset, offset_dev=offset_dev, sel=sel,
                sel_cursor=sel_cursor, batch=batch)
        elif self.args.pt_type == "RFD":
            x_train = self._split(self.train_dataset).X
            cfg = self.model_config
            low = getattr(cfg, "idx_low", None)
            high = getattr(cfg, "idx_high", None)
            if self.args.RFD_replace == "Uniform" and replace_feat is None:
                low, high = low.to(ids.device).contiguous(), high.to(ids.device).contiguous()
            inputs["input_ids"], inputs["labels"], _ = ops.dynamic_mask_rfd(
                ids, L, masked_index=masked_index, replace_feat=replace_feat, x_train=x_train,
                seed=seed, offset=offset, offset_dev=offset_dev, mode=self.args.RFD_replace,
                idx_low=low if self.args.RFD_replace == "Uniform" else None,
                idx_high=high if self.args.RFD_replace == "Uniform" else None, vocab=cfg.input_size)
        else:
            raise NotImplementedError(self.args.pt_type)
        return inputs

    # ------------------------------------------------------------------ MFP (north-star loop)
    def MFP_pretrain(self):
        train = self._begin("pretraining")
        logger.info(f"  mask_ratio = {self.args.mask_ratio}")
        logger.info(f"  pt_neg_num = {self.model_config.pt_neg_num}")
        logger.info(f"  pt_type = {self.model_config.pt_type}")
        B = self.args.per_gpu_train_batch_size
        win_loss = torch.zeros((), device=self.device)
        win_acc = torch.zeros((), device=self.device)
        start_time = time.time()
        for epoch in range(self.args.num_train_epochs):
            logger.info(f"-------------------- epoch-{epoch} --------------------")
            self.model.train()
            for X, Y in train.batches(B, True, self._generator(), (self.rank, self.world), rows=True):
                loss, step_acc = self.run_step("mfp", X, Y)
                win_loss += loss
                win_acc += step_acc
                if self.global_step % self.args.logging_steps == 0:
                    n = self.args.logging_steps
                    _log = {"window_loss": float(win_loss) / n, "window_acc": float(win_acc) / n,
                            "time_cost": time.time() - start_time}
                    logger.info(f"step = {self.global_step}, {_log}")
                    win_loss.zero_(); win_acc.zero_()
                    start_time = time.time()
            self.optimizer.flush()      # on EVERY rank: replicas must replay their lazy rows at the same steps
            if self.args.local_rank in [-1, 0]:
                self.MFP_pretrain_eval()
        self.optimizer.flush()
        if self.args.local_rank in [-1, 0]:
            self.save_model(self.args.output_dir)
        logger.info(str(self.eval_metrics))

    def MFP_pretrain_eval(self):
        ev = self._split(self.eval_dataset)
        logger.info("***** running eval *****")
        logger.info(f"  num examples = {ev.n}")
        self.optimizer.flush()
        self.model.eval()
        tot_loss = torch.zeros((), device=self.device)
        tot_acc = torch.zeros((), device=self.device)
        count = 0
        t0 = time.time()
        with torch.no_grad():
            for X, Y in ev.batches(self.args.per_gpu_eval_batch_size, False):
                inputs = self.dynamic_mask({"input_ids": X, "labels": Y}, self.args.sampling_method)
                loss, n, acc = self.model(**inputs)
                tot_loss += loss * n
                tot_acc += acc.float()
                count += n
        _log = {"learning_rate": self.scheduler.get_last_lr()[0], "eval_mfp_loss": float(tot_loss) / count,
                "eval_mfp_acc": float(tot_acc) / count, "eval_time_cost": time.time() - t0}
        self.eval_metrics.append([_log["eval_mfp_loss"], _log["eval_mfp_acc"]])
        logger.info(str(_log))
        return _log

    # ------------------------------------------------------------------ RFD
    def RFD_pretrain(self):
        train = self._begin("pretraining")
        logger.info(f"  pt_type = {self.model_config.pt_type}")
        logger.info(f"  mask_ratio = {self.args.mask_ratio}")
        logger.info(f"  RFD_replace = {self.args.RFD_replace}")
        B = self.args.per_gpu_train_batch_size
        win = torch.zeros(2, device=self.device)
        start_time = time.time()
        for epoch in range(self.args.num_train_epochs):
            logger.info(f"-------------------- epoch-{epoch} --------------------")
            self.model.train()
            for X, Y in train.batches(B, True, self._generator(), (self.rank, self.world), rows=True):
                loss, acc = self.run_step("rfd", X, Y)
                win += torch.stack([loss, acc])
                if self.global_step % self.args.logging_steps == 0:
                    n = self.args.logging_steps
                    w = win.tolist()
                    _log = {"window_rfd_loss": w[0] / n, "window_rfd_acc": w[1] / n,
                            "time_cost": time.time() - start_time}
                    logger.info(f"step = {self.global_step}, {_log}")
                    win.zero_()
                    start_time = time.time()
            self.optimizer.flush()      # on EVERY rank: replicas must replay their lazy rows at the same steps
            if self.args.local_rank in [-1, 0]:
                self.RFD_pretrain_eval()
        self.optimizer.flush()
        if self.args.local_rank in [-1, 0]:
            self.save_model(self.args.output_dir)
        logger.info(str(self.eval_metrics))

    def RFD_pretrain_eval(self):
        ev = self._split(self.eval_dataset)
        logger.info("***** running eval *****")
        logger.info(f"  num examples = {ev.n}")
        self.optimizer.flush()
        self.model.eval()
        tot = torch.zeros(2, device=self.device)
        count = 0
        t0 = time.time()
        with torch.no_grad():
            for X, Y in ev.batches(self.args.per_gpu_eval_batch_size, False):
                inputs = self.dynamic_mask({"input_ids": X, "labels": Y}, self.args.sampling_method)
                loss, n, acc = self.model(**inputs)[:3]
                tot += torch.stack([loss, acc]) * n
                count += n
        t = tot.tolist()
        _log = {"learning_rate": self.scheduler.get_last_lr()[0], "eval_rfd_loss": t[0] / count,
                "eval_rfd_acc": t[1] / count, "eval_time_cost": time.time() - t0}
        self.eval_metrics.append([_log["eval_rfd_loss"], _log["eval_rfd_acc"]])
        logger.info(str(_log))
        return _log

    # ------------------------------------------------------------------ CTR (scratch / finetune)
    def train(self):
        train = self._begin("training")
        self._patience, self._stop_training = 0, False
        B = self.args.per_gpu_train_batch_size
        win_loss = torch.zeros((), device=self.device)
        win_logits, win_labels = [], []
        for epoch in range(self.args.num_train_epochs):
            logger.info(f"-------------------- epoch-{epoch} --------------------")
            self.model.train()
            for X, Y in train.batches(B, True, self._generator(), (self.rank, self.world), rows=True):
                loss, logits = self.run_step("ctr", X, Y)
                win_loss += loss
                win_logits.append(logits.clone())
                win_labels.append(Y.sel if isinstance(Y, RowsRef) else Y)     # row numbers: the labels are cut below
                if self.global_step % self.args.logging_steps == 0:
                    if isinstance(Y, RowsRef):
                        win_labels = [train.Y[torch.cat(win_labels)]]
                    try:
                        auc = ops.eval_metrics(torch.cat(win_logits), torch.cat(win_labels))["auc"]
                    except ValueError:                       # a window with one class only
                        auc = float("nan")
                    _log = {"window_auc": auc, "window_loss": float(win_loss) / self.args.logging_steps}
                    logger.info(f"step = {self.global_step}, {_log}")
                    win_loss.zero_()
                    win_logits, win_labels = [], []
            self.eval()
            if self._stop_training:
                break
        logger.info(str(self.eval_metrics))

    def eval(self, eval_dataset=None, test_eval=False):
        ev = self._split(self.eval_dataset if eval_dataset is None else eval_dataset)
        logger.info("***** running TEST *****" if test_eval else "***** running eval *****")
        logger.info(f"  num examples = {ev.n}")
        if self.optimizer is not None:
            self.optimizer.flush()
        self.model.eval()
        all_logits = []
        with torch.no_grad():
            for X, Y in ev.batches(self.args.per_gpu_eval_batch_size, False):
                all_logits.append(self.model(input_ids=X, labels=Y)[1].view(-1))
        # AUC / log-loss on the device over the fp32 sigmoid, ties included (csrc/metrics.hip):
        # what sklearn's roc_auc_score / log_loss give the reference on its host copies
        met = ops.eval_metrics(torch.cat(all_logits), ev.Y)
        auc, ll = met["auc"], met["logloss"]
        self.eval_metrics.append([auc, ll])
        lr = self.scheduler.get_last_lr()[0] if self.optimizer is not None else float("nan")
        _log = {"learning_rate": lr, "eval_auc": auc, "eval_loss": ll, "avg_logits": met["avg_logits"],
                "avg_probs": met["avg_probs"]}
        logger.info(str(_log))
        if not test_eval:
            if auc > self.best_eval_auc:
                self.best_eval_auc, self.best_eval_step, self._patience = auc, self.global_step, 0
                if self.args.local_rank in [-1, 0]:
                    self.save_model(self.args.output_dir)
            else:
                self._patience += 1
            if self._patience > self.args.patience:
                self._stop_training = True
            # every rank runs eval() and takes the same branch (replicas are bit-identical); rank 0
            # may have written {step}.model above, which other ranks load in test()
            parallel.barrier()
        return _log

    # ------------------------------------------------------------------ checkpoints (a15)
    def save_model(self, model_dir):
        """{global_step}.model = torch.save(state_dict): weights + buffers only, fp32, reference
        key layout; lazy table rows are flushed first so the file holds reference-equivalent
        weights."""
        if self.optimizer is not None:
            self.optimizer.flush()
        sd = {k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()}
        path = os.path.join(model_dir, f"{self.global_step}.model")
        tmp = f"{path}.tmp.{os.getpid()}"
        torch.save(sd, tmp)
        os.replace(tmp, path)           # readers (other ranks' test()) never see a partial file

    def save_training_state(self, path):
        """Full resume state, beyond the reference's weights-only checkpoint: raw (un-flushed)
        weights, optimizer state, step counters."""
        torch.save(dict(model={k: v.detach().cpu() for k, v in self.model.state_dict().items()},
                        optimizer=self.optimizer.state_dict(), global_step=self.global_step), path)

    def load_training_state(self, path):
        st = torch.load(path, map_location="cpu")
        with torch.no_grad():
            own = self.model.state_dict()
            for k, v in st["model"].items():
                own[k].copy_(v)
        self.optimizer.load_state_dict(st["optimizer"])      # (re-derives the bf16 weight shadows)
        self.global_step = int(st["global_step"])
        self._graphs = {}

    def load_model(self, load_step, model_dir):
        sd = torch.load(os.path.join(model_dir, f"{load_step}.model"), map_location="cpu")
        with torch.no_grad():
            own = self.model.state_dict()
            missing = set(own) - set(sd)
            unexpected = set(sd) - set(own)
            if missing or unexpected:
                raise RuntimeError(f"Error(s) in loading state_dict: missing {sorted(missing)}, "
                                   f"unexpected {sorted(unexpected)}")
            for k, v in sd.items():
                own[k].copy_(v)          # in place: parameters may be views of the flat buffers
        if self.optimizer is not None:
            self.optimizer.refresh_bf16()

    def test(self, test_dataset, load_step=-1, model_dir=None):
        if load_step == -1:
            load_step = self.best_eval_step
        self.load_model(load_step, self.args.output_dir if model_dir is None else model_dir)
        return self.eval(test_dataset, test_eval=True)
