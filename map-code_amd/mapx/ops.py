"""Functional wrappers over the C ABI: tensors in, tensors out, no autograd.

Every function enqueues HIP kernels on torch's current stream and returns immediately.
Scratch memory is a per-(device, stream) byte buffer grown on demand; kernels that share it
run in stream order, so one buffer suffices.
"""
import math
import os

import torch

import ctypes

from . import native as N
from .native import check, lib, ptr, require_gpu, stream


def native_int():
    return ctypes.c_int(0)


def native_byref(x):
    return ctypes.byref(x)

_scratch = {}
_scratch_retired = []

# --------------------------------------------------------------------------- live kernel timing
# bench.py brackets kernel classes with HIP events on the launch stream (torch's current
# stream) while the timed region runs; nothing is recorded unless a Timers object is active.
_timers = None


class Timers:
    def __init__(self):
        self.events = {}          # name -> [(start, stop, work)]

    def __enter__(self):
        global _timers
        _timers = self
        return self

    def __exit__(self, *a):
        global _timers
        _timers = None

    def calibrate(self, n=32):
        """Record `n` empty brackets: their mean is the cost of the event pair itself (two
        event commands + one dependent dispatch gap), subtracted from every interval."""
        for _ in range(n):
            with _timed("__empty__"):
                pass

    def summary(self):
        """name -> dict(launches, total_ms, avg_us, work) after a device sync."""
        out = {}
        empty = self.events.pop("__empty__", None)
        over_ms = sum(a.elapsed_time(b) for a, b, _ in empty) / len(empty) if empty else 0.0
        for name, evs in self.events.items():
            ms = [max(a.elapsed_time(b) - over_ms, 0.0) for a, b, _ in evs]
            out[name] = dict(launches=len(evs), total_ms=sum(ms), avg_us=1e3 * sum(ms) / len(evs),
                             work=sum(w for _, _, w in evs), bracket_overhead_us=1e3 * over_ms)
        return out


class _timed:
    def __init__(self, name, work=0.0):
        self.name, self.work = name, work

    def __enter__(self):
        if _timers is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *e):
        if _timers is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            _timers.events.setdefault(self.name, []).append((self.a, b, self.work))


# --------------------------------------------------------------------------- auxiliary streams
# Independent pieces of a step run on named side streams so that small or tail-heavy kernels
# fill the gaps of the big GEMMs: "plan" (sort + runs of the row ids) and "tower" (DCNv2's cross
# tower next to the deep tower).
# serialize_streams = True maps them all onto the current stream (per-kernel timing passes).
_aux_streams = {}
serialize_streams = os.environ.get("MAPX_SERIAL", "0") == "1"


def aux_stream(name, device, high=False):
    """Named side stream; `high`: high priority, i.e. hardware queues of its own class (work on it
    does not wait behind packets of normal-priority streams that happen to share a queue)."""
    if serialize_streams:
        return torch.cuda.current_stream()
    key = (name, device.index)
    st = _aux_streams.get(key)
    if st is None:
        st = _aux_streams[key] = torch.cuda.Stream(device=device, priority=-1 if high else 0)
    return st


# Side tasks: HBM-bound work that is independent of the trunk's backward (e.g. the row update of a
# table whose gradient is final) queued by one backward node and run by another at the START of a
# side-stream chain that has slack (the cross tower's backward) — the same placement that works in
# forward for the NCE sampling; whatever is still queued when the optimizer starts runs there.
_side_tasks = []
pending_joins = []                # (waiting stream, side stream) left open by a backward node; the optimizer joins
HEAD_SIDE_REDUCE_ONLY = os.environ.get("MAPX_HEAD_SIDE_DP", "1") == "1"


def join_pending():
    """Join the side streams that backward nodes forked and left open (optimizer.step, or the end of a captured
    forward + backward whose tail runs elsewhere)."""
    while pending_joins:
        waiter, side = pending_joins.pop()
        stream_wait(waiter, side)
# "every dense gradient of this backward pass is final": an event recorded by the model's LAST backward node
# (the embedding gather of a tower model) before its own kernels; the optimizer's dense half may start there
dense_ready = [None, None]        # [event, stream it was recorded on]
# the Trainer's device scalar 1.0 that seeds loss.backward() (held here, so its address cannot be recycled):
# a head whose incoming gradient IS that tensor skips the multiply by it
unit_gradient = [None]
# True between MapxOptimizer.backward_window(True) and (False): optimizer.step() follows this backward pass at
# once and joins what it left open; outside the window backward() joins its side streams itself
step_window = [False]
# "1" (default): the optimizer's dense half forks from there onto the tower stream, beside the tables' half; "0":
# everything on the main stream.  (With early row updates in the bf16 mode — its default until the end of round 4 — the
# dense half queued up behind the NCE table's gradient and update on the tower stream while the main queue idled for
# 72 us, and "0" was better there: 0.5454 / 0.5425 -> 0.5397 / 0.5397 ms; without early updates "1" wins in both modes:
# optim.py.)  "auto" = "1".
TAIL_OVERLAP = os.environ.get("MAPX_TAIL_OVERLAP", "1")


def tail_overlap(bf16_trunk):
    return TAIL_OVERLAP in ("1", "auto")


def add_side_task(fn):
    _side_tasks.append(fn)


def run_side_tasks():
    while _side_tasks:
        _side_tasks.pop(0)()


# Late tasks: work of a backward node that nothing but the optimizer waits for (the grouped encoder's weight and
# bias gradients), queued instead of run on the chain that the towers' backward passes wait behind; the cross
# tower's backward node runs them at its END (its chain is the shorter one), optimizer.step() whatever is left.
_late_tasks = []


def add_late_task(fn, dense=False):
    """dense: the task forms gradients of DENSE parameters (a head's or the encoder's dW / db) — those run first, and an
    event behind them (late_dense_done) lets the optimizer's dense half start without waiting for the tables' work that
    the other late tasks put on the same stream."""
    _late_tasks.append((fn, bool(dense)))


LATE_DENSE_FIRST = os.environ.get("MAPX_LATE_DENSE_FIRST", "0") == "1"      # A/B switch (with MAPX_DENSE_BEFORE_JOIN)
late_dense_done = [None, None]        # [event, stream] behind the last dense-gradient late task of this backward pass


def run_late_tasks():
    if not _late_tasks:
        return
    tasks = list(_late_tasks)
    _late_tasks.clear()
    if not LATE_DENSE_FIRST:
        for fn, _ in tasks:
            fn()
        return
    for fn, dense in tasks:
        if dense:
            fn()
    if any(not dense for _, dense in tasks) and torch.cuda.is_available():
        late_dense_done[0], late_dense_done[1] = record_event(), torch.cuda.current_stream()
    for fn, dense in tasks:
        if not dense:
            fn()


# Main tasks: stream joins a backward node wants on the MAIN stream but not yet (the wait for a segment plan that the
# node's successors do not need); the towers' join node (layers._JoinColumns.backward) runs them, optimizer.step()
# whatever is left — always before the late tasks, which may read what they joined.
_main_tasks = []


def add_main_task(fn):
    _main_tasks.append(fn)


def run_main_tasks():
    while _main_tasks:
        _main_tasks.pop(0)()


def clear_side_tasks():
    """Drop tasks that an aborted backward pass left behind."""
    _side_tasks.clear()
    _late_tasks.clear()
    _main_tasks.clear()


_DBG_SLEEP = os.environ.get("MAPX_DBG_SLEEP", "")


def dbg_sleep(site):
    """Race hunting: MAPX_DBG_SLEEP=<site>[,<site>] parks the CURRENT stream for ~0.5 ms at the named sites (a missing
    cross-stream dependency then shows as a graph != eager or a parity failure instead of depending on timing)."""
    if _DBG_SLEEP and site in _DBG_SLEEP.split(",") and torch.cuda.is_available():
        torch.cuda._sleep(1_000_000)


def reset_aux_streams():
    """Forget the named side streams (new ones are made on demand) and everything a backward pass may have
    left queued on them: after a failed graph capture the streams may be stuck in the invalidated capture,
    and the per-backward lists (joins left open, deferred partial sums, side tasks, the dense-ready event)
    still name that capture's streams, events and partial buffers — an eager retry that found them would
    wait on a dead stream in optimizer.step() or sum stale partials into the bias gradients."""
    _aux_streams.clear()
    _scratch_retired.extend(_scratch.values())
    _scratch.clear()
    pending_joins.clear()
    step_window[0] = False
    _deferred.clear()
    _side_tasks.clear()
    _late_tasks.clear()
    _main_tasks.clear()
    dense_ready[0] = dense_ready[1] = None


def _capturing(st):
    """True if HIP stream `st` currently belongs to a stream capture (hipStreamIsCapturing)."""
    with torch.cuda.stream(st):
        return torch.cuda.is_current_stream_capturing()


class CaptureIsolationError(RuntimeError):
    pass


def _check_capture_join(waiter, waited, what):
    """A stream that belongs to a capture may only wait for work of the SAME capture.  Waiting for a
    stream that never joined it (it did not fork, directly or through other streams, from the capture's
    origin) puts un-captured work in front of captured work: HIP answers with
    hipErrorStreamCaptureIsolation at best and — seen in round 1 as a crash inside
    hipStreamEndCapture for a side-stream-to-side-stream join — invalidates the capture at worst.
    Refused here, in Python, before the runtime sees it."""
    if _capturing(waiter) and not _capturing(waited):
        raise CaptureIsolationError(
            f"{what}: the waiting stream belongs to a hipGraph capture but the stream it would wait for does not "
            "(it never forked from the capture's origin): fork it from the origin first (ops.stream_wait(side, "
            "origin)) or run that work on a stream of the capture")


def stream_wait(waiter, waited):
    """waiter.wait_stream(waited), skipped when both are the same HIP stream (a self-wait is a
    no-op when run eagerly, but inside a stream capture it hands hipStreamEndCapture a node that
    depends on itself); refused when it would join un-captured work into a capture."""
    if waiter.cuda_stream != waited.cuda_stream:
        _check_capture_join(waiter, waited, "stream_wait")
        waiter.wait_stream(waited)
        return True
    return False


def record_event():
    """An event at the current tail of the current stream: a fork point another stream can be
    made to wait on later (stream_wait_event), after more work has been enqueued here."""
    ev = torch.cuda.Event()
    ev.record()
    return ev


def stream_wait_event(waiter, event, origin):
    """waiter.wait_event(event) unless waiter is the stream the event was recorded on (`origin`);
    refused when `waiter` belongs to a capture that `origin` is not part of (see stream_wait)."""
    if waiter.cuda_stream != origin.cuda_stream:
        _check_capture_join(waiter, origin, "stream_wait_event")
        waiter.wait_event(event)
        return True
    return False


# --------------------------------------------------------------------------- deferred sums
# Split-K slabs (weight gradients) and row-chunk partials (bias gradients) of a backward pass
# are summed by ONE kernel launch right before their consumer (optimizer / gradient exchange)
# instead of one tiny launch each.  Only used when the destination is an optimizer-owned slot.
_deferred = []
COLSUM_CHUNKS = lib.mapx_colsum_chunks()


# measured on MI355X inside the full step: deferring is 2-3 % SLOWER (1.56 vs 1.52 ms) — by then the
# slabs have left L2 and one launch has few blocks — so it is off by default.  Sending each layer's
# sums to a side stream instead was far worse (1.78 vs 1.45 ms): inside the captured graph every
# extra parallel branch of tiny kernels disturbs the queue assignment of the GEMM chains.
# Letting the dense optimizer kernel sum the slabs while it reads the gradient (no extra launch
# at all) was also slower (1.49 vs 1.42 ms): the slabs are cold by then, whereas a reduce right
# behind the GEMM finds them in L2 / Infinity Cache.
DEFER = os.environ.get("MAPX_DEFER", "0") == "1"
# the bias gradients' row-chunk partials alone (7 tiny second-stage launches per step on the GEMM chains,
# 128 x N floats each: nothing to go cold)
# measured: 0.943 / 0.950 vs 0.952 / 0.954 ms per step, same box: on by default
DEFER_COLSUM = DEFER or os.environ.get("MAPX_DEFER_COLSUM", "1") == "1"
# the same for the bf16 mode's column-sum kernels (round 3: each bias gradient had a second-stage launch of its own,
# 7-10 us apiece on the backward chains of a 0.5-ms step)
DEFER_COLSUM_H = os.environ.get("MAPX_DEFER_COLSUM_H", "1") == "1"


def defer_sum(dst, src, stride, nsplit, n):
    _deferred.append((dst, src, int(stride), int(nsplit), int(n)))


def flush_deferred():
    """Launch the pending slab sums (32 tasks per launch)."""
    global _deferred
    cur = torch.cuda.current_stream() if torch.cuda.is_available() else None
    while _deferred:
        batch, _deferred = _deferred[:32], _deferred[32:]
        if cur is not None:
            for (_, src, _, _, _) in batch:       # partial buffers die with this list: keep them from being
                src.record_stream(cur)            # recycled while another stream than their own still reads them
        arr = (N.SumTask * len(batch))()
        for i, (dst, src, stride, nsplit, n) in enumerate(batch):
            arr[i].dst, arr[i].src = dst.data_ptr(), src.data_ptr()
            arr[i].stride, arr[i].n, arr[i].nsplit = stride, n, nsplit
        check(lib.mapx_sum_tasks(arr, len(batch), stream()))


def scratch(nbytes, device):
    key = (device.index, stream())
    buf = _scratch.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _scratch_retired.append(buf)    # a captured graph may have the old address baked in: never free it
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _scratch[key] = buf
    return buf


def _err_flag(device):
    return torch.zeros(1, dtype=torch.int32, device=device)


# --------------------------------------------------------------------------- embedding
BF16 = torch.bfloat16


def is_bf16(t):
    return t is not None and t.dtype == torch.bfloat16


def cast_bf16(x):
    """fp32 -> bf16 copy (round to nearest even) by the library's kernel."""
    require_gpu(x)
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=BF16, device=x.device)
    check(lib.mapx_cast_f32_bf16(ptr(x), x.numel(), ptr(out), stream()))
    return out


def cast_f32(x):
    require_gpu(x)
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(lib.mapx_cast_bf16_f32(ptr(x), x.numel(), ptr(out), stream()))
    return out


def bf16_weight(w):
    """The bf16 operand of a dense fp32 master weight: the shadow the optimizer keeps current
    (`w._mapx_bf16`, written by the AdamW kernel itself), or a fresh conversion when no optimizer
    owns the parameter (inference / tests)."""
    sh = getattr(w, "_mapx_bf16", None)
    return sh if sh is not None else cast_bf16(w.detach())


def emb_gather(ids, table, validate=False, out_dtype=torch.float32, lazy=None):
    """ids int64 [...], table [V,E] -> [..., E]  (layers.py:97-102); bf16 rows in bf16 compute mode.
    lazy (native.LazyRows): rows are read through their pending zero-gradient updates (no catch-up pass ran)."""
    require_gpu(ids, table)
    ids = ids.contiguous()
    out = torch.empty(*ids.shape, table.shape[1], dtype=out_dtype, device=table.device)
    err = _err_flag(table.device) if validate else None
    rec = amax_record(table.device) if out_dtype != BF16 else None
    with _timed("emb_gather", ids.numel() * (8 + (4 + out.element_size()) * table.shape[1])):
        if out_dtype == BF16:
            check(lib.mapx_emb_gather_fwd_bf16(ptr(ids), ids.numel(), ptr(table), table.shape[0], table.shape[1],
                                               ptr(out), ptr(err), None if lazy is None else native_byref(lazy), stream()))
        else:
            check(lib.mapx_emb_gather_fwd(ptr(ids), ids.numel(), ptr(table), table.shape[0], table.shape[1],
                                          ptr(out), ptr(err), ptr(rec), None if lazy is None else native_byref(lazy),
                                          stream()))
    tag(out, rec)
    if validate and int(err.item()):
        raise IndexError("index out of range in self")          # reference CPU behaviour
    return out


_keys_of = [None, None]      # (id matrix, its int32 keys) as the last dynamic_mask_mfp produced them


def ids_to_i32(ids, V, validate=False):
    require_gpu(ids)
    if not validate and _keys_of[0] is ids:
        return _keys_of[1]       # the mask kernel wrote them already
    ids = ids.contiguous()
    out = torch.empty(ids.numel(), dtype=torch.int32, device=ids.device)
    err = _err_flag(ids.device) if validate else None
    check(lib.mapx_ids_to_i32(ptr(ids), ids.numel(), V, ptr(out), ptr(err), stream()))
    if validate and int(err.item()):
        raise IndexError("index out of range in self")
    return out


class SegPlan:
    """Sorted-run description of n int32 keys (see include/mapx_hip.h: mapx_seg_plan)."""

    def __init__(self, keys_i32, V, sorted_lists=0, launch=True):
        """`sorted_lists` = w > 0: the keys are w concatenated lists of equal length, each ascending
        as unsigned values and free of repeats except trailing -1 padding (the gathered messages
        of mapx.parallel): ranked by one merge launch instead of the radix passes, same outputs.
        `launch=False`: only allocate the outputs (SegPlan.build_many launches several plans at once)."""
        require_gpu(keys_i32)
        n, dev = keys_i32.numel(), keys_i32.device
        self.n, self.V = n, V
        i32 = dict(dtype=torch.int32, device=dev)
        self.sorted_keys = torch.empty(max(n, 1), **i32)
        self.perm = torch.empty(max(n, 1), **i32)
        self.rank = torch.empty(max(n, 1), **i32)
        self.uniq = torch.empty(max(n, 1), **i32)
        self.seg_start = torch.empty(n + 1, **i32)
        self.n_uniq = torch.empty(2, **i32)         # [number of runs, owner counter (zeroed by the plan)]
        self._counter_fresh = n > 0
        if not launch:
            return
        nb = lib.mapx_seg_plan_workspace_bytes(n, V)
        ws = scratch(nb, dev)
        with _timed("seg_plan", n * 4.0):
            if sorted_lists > 0:
                if n % sorted_lists:
                    raise ValueError("sorted_lists must divide the number of keys")
                check(lib.mapx_seg_plan_merge(ptr(keys_i32), sorted_lists, n // sorted_lists, ptr(ws), ws.numel(),
                                              ptr(self.sorted_keys), ptr(self.perm), ptr(self.rank), ptr(self.uniq),
                                              ptr(self.seg_start), ptr(self.n_uniq), stream()))
            else:
                check(lib.mapx_seg_plan(ptr(keys_i32), n, V, ptr(ws), ws.numel(), ptr(self.sorted_keys),
                                        ptr(self.perm), ptr(self.rank), ptr(self.uniq), ptr(self.seg_start),
                                        ptr(self.n_uniq), stream()))

    @staticmethod
    def build_many(key_lists, Vs):
        """The plans of several key lists (the step's tables) from ONE chain of launches
        (mapx_seg_plan_multi: 8 launches for two 24-bit lists instead of 8 each) -> [SegPlan]."""
        import ctypes as C
        plans = [SegPlan(k, V, launch=False) for k, V in zip(key_lists, Vs)]
        cnt = len(plans)
        if cnt == 1:          # same kernels, one problem
            pass
        I64, PP = C.c_int64 * cnt, C.c_void_p * cnt
        n_arr, v_arr = I64(*[p.n for p in plans]), I64(*[int(v) for v in Vs])
        dev = key_lists[0].device
        ws = scratch(lib.mapx_seg_plan_multi_workspace_bytes(cnt, n_arr, v_arr), dev)
        arr = lambda ts: PP(*[t.data_ptr() for t in ts])
        with _timed("seg_plan", sum(p.n for p in plans) * 4.0):
            check(lib.mapx_seg_plan_multi(cnt, arr(key_lists), n_arr, v_arr, ptr(ws), ws.numel(),
                                          arr([p.sorted_keys for p in plans]), arr([p.perm for p in plans]),
                                          arr([p.rank for p in plans]), arr([p.uniq for p in plans]),
                                          arr([p.seg_start for p in plans]), arr([p.n_uniq for p in plans]),
                                          stream()))
        return plans

    def tensors(self):
        return (self.sorted_keys, self.perm, self.rank, self.uniq, self.seg_start, self.n_uniq)

    def count(self):
        """Number of unique keys (host sync)."""
        return int(self.n_uniq[0].item())

    def take_counter(self):
        """Address of the plan's zeroed owner counter for its first segment reduction, None after."""
        if not self._counter_fresh:
            return None
        self._counter_fresh = False
        return self.n_uniq.data_ptr() + 4


def seg_reduce_rows(plan, src, W, src2=None):
    """out[u,:] = sum of src rows whose key is plan.uniq[u]; out has capacity plan.n rows.
    `src2`: a second tensor like src, added to it element by element inside the kernel (the two towers'
    dL/dX0 of DCNv2: no elementwise launch in front of the reduction)."""
    require_gpu(src)
    if src2 is not None:
        require_gpu(src2)
        if src2.dtype != src.dtype or src2.shape != src.shape or not src2.is_contiguous():
            raise ValueError("seg_reduce_rows: src2 must match src (dtype, shape, contiguous)")
    out = torch.empty(max(plan.n, 1), W, dtype=torch.float32, device=src.device)
    nb = lib.mapx_seg_reduce_workspace_bytes(plan.n, W)
    ws = scratch(nb, src.device)
    fn = lib.mapx_seg_reduce_rows_bf16 if is_bf16(src) else lib.mapx_seg_reduce_rows
    nsrc = 2 if src2 is not None else 1
    with _timed("seg_reduce_rows", plan.n * (float(src.element_size()) * W * nsrc + 8)):
        check(fn(plan.n, ptr(plan.perm), ptr(plan.rank), ptr(plan.seg_start),
                 ptr(src), ptr(src2), W, ptr(out), ptr(ws), ws.numel(), plan.take_counter(), stream()))
    return out


def seg_reduce_rows_extra(plan, src, W, extra, group, extra_stride=1):
    """-> (rows [cap,W], scalars [cap]): the row reduction of seg_reduce_rows over the first W columns
    of src plus, per unique key, the sum of extra[(position // group) * extra_stride] (DeepFM's LR
    weight gradient rides with the embedding's; DP merge of rows that carry a scalar column)."""
    require_gpu(src, extra)
    out = torch.empty(max(plan.n, 1), W, dtype=torch.float32, device=src.device)
    out1 = torch.empty(max(plan.n, 1), dtype=torch.float32, device=src.device)
    ws = scratch(lib.mapx_seg_reduce_workspace_bytes(plan.n, W), src.device)
    with _timed("seg_reduce_rows", plan.n * (4.0 * W + 12)):
        check(lib.mapx_seg_reduce_rows_extra(plan.n, ptr(plan.perm), ptr(plan.rank), ptr(plan.seg_start),
                                             src.data_ptr(), W, src.stride(0), extra.data_ptr(), group,
                                             extra_stride, ptr(out), ptr(out1), ptr(ws), ws.numel(),
                                             plan.take_counter(), stream()))
    return out, out1


def pack_sparse(plan, rows0, rows1, maxc, scale, pad_id=0):
    """The first n_uniq (id, row) pairs of a sparse gradient as a message of exactly `maxc` entries
    (mapx/parallel.py): -> (keys int32 [maxc], rows f32 [maxc, W0 (+4 with rows1)])."""
    require_gpu(rows0)
    W0 = rows0.shape[1]
    Wp = W0 + 4 if rows1 is not None else W0
    keys = torch.empty(maxc, dtype=torch.int32, device=rows0.device)
    rows = torch.empty(maxc, Wp, dtype=torch.float32, device=rows0.device)
    check(lib.mapx_pack_sparse(ptr(plan.uniq), ptr(rows0), W0, ptr(rows1), ptr(plan.n_uniq),
                               min(plan.uniq.shape[0], rows0.shape[0]), maxc, float(scale), pad_id, ptr(keys),
                               ptr(rows), stream()))
    return keys, rows


class HostMailbox:
    """`n` int32 words of fine-grained pinned host memory that a kernel can store into and the host
    can poll while the stream is still running (mapx_host_alloc_coherent): `.np` is the numpy
    view, `.address(i)` the address of word i for the kernel."""

    def __init__(self, n):
        import ctypes
        import numpy as np
        p = ctypes.c_void_p()
        check(lib.mapx_host_alloc_coherent(4 * n, ctypes.byref(p)))
        self._p, self.n = p.value, n
        self.np = np.ctypeslib.as_array((ctypes.c_int32 * n).from_address(p.value))

    def address(self, i=0):
        return self._p + 4 * i

    def __del__(self):
        p, self._p = getattr(self, "_p", None), None
        if p and lib is not None:
            self.np = None
            lib.mapx_host_free(p)


def publish_i32(src, n, stamp_dev, mailbox, at=0):
    """++stamp; mailbox[at : at+n] = src[:n] (int32, device); mailbox[at+n] = stamp;
    mailbox[at+n+1] = checksum.  A kernel that stores straight into coherent host memory: usable
    from inside a captured graph, readable by the host before the graph ends (mapx_publish_i32)."""
    require_gpu(src, stamp_dev)
    if src.dtype != torch.int32 or stamp_dev.dtype != torch.int32:
        raise ValueError("publish_i32 moves int32 values")
    if at < 0 or at + n + 2 > mailbox.n:
        raise ValueError("publish_i32 needs n + 2 words of mailbox")
    check(lib.mapx_publish_i32(ptr(src), n, ptr(stamp_dev), mailbox.address(at), stream()))


# --------------------------------------------------------------------------- xDeepFM: CIN pieces
def transpose_batched(x):
    """[B,R,C] -> [B,C,R] (contiguous)."""
    require_gpu(x)
    x = x.contiguous()
    B, R, C = x.shape
    out = torch.empty(B, C, R, dtype=torch.float32, device=x.device)
    check(lib.mapx_transpose_batched(ptr(x), B, R, C, ptr(out), stream()))
    return out


def cin_outer_fwd(x0t, xi, pad_to=1):
    """had[r, h*H+m] = x0t[r,h] * xi[r,m]: x0t [R,F], xi [R,H] -> [R, F*H]  (layers.py:714-715); pad_to = 8: the row
    is padded with zero columns to a multiple of 8 floats (-> [R, ceil8(F*H)], the GEMM's vectorised operand path)."""
    require_gpu(x0t, xi)
    R, F = x0t.shape
    H = xi.shape[1]
    ld = (F * H + pad_to - 1) // pad_to * pad_to
    had = torch.empty(R, ld, dtype=torch.float32, device=x0t.device)
    check(lib.mapx_cin_outer_fwd(ptr(x0t), F, ptr(xi), H, R, ptr(had), ld, stream()))
    return had


def cin_outer_bwd(dhad, x0t, xi, dx0t, accumulate_x0):
    """-> dxi [R,H]; dx0t [R,F] is written or accumulated in place."""
    require_gpu(dhad, x0t, xi, dx0t)
    R, F = x0t.shape
    H = xi.shape[1]
    dxi = torch.empty(R, H, dtype=torch.float32, device=x0t.device)
    if dhad.stride(1) != 1 or dhad.shape[1] < F * H:
        raise ValueError("cin_outer_bwd: dhad [R, >= F*H] with unit column stride")
    check(lib.mapx_cin_outer_bwd(ptr(dhad), dhad.stride(0), ptr(x0t), F, ptr(xi), H, R, ptr(dx0t), int(accumulate_x0),
                                 ptr(dxi), stream()))
    return dxi


def cin_pool_fwd(xt, B, E, out):
    """out[b, :H] = sum_d xt[(b,d), :]: xt [B*E, H]; `out` may be a column slice (row stride ld)."""
    require_gpu(xt, out)
    H = xt.shape[1]
    check(lib.mapx_cin_pool_fwd(ptr(xt), B, E, H, out.data_ptr(), out.stride(0), stream()))
    return out


def cin_pool_bwd(g, B, E, dxt, accumulate):
    """dxt[(b,d), :] (+)= g[b, :H]; `g` may be a column slice."""
    require_gpu(g, dxt)
    H = dxt.shape[1]
    check(lib.mapx_cin_pool_bwd(g.data_ptr(), g.stride(0), B, E, H, ptr(dxt), int(accumulate), stream()))
    return dxt


# --------------------------------------------------------------------------- DeepFM terms
def lr_sum(ids, w, validate=False):
    """out[b] = sum_f w[ids[b,f]]   (reference models.py:137-140, before the bias)."""
    require_gpu(ids, w)
    ids = ids.contiguous()
    B, F = ids.shape
    out = torch.empty(B, dtype=torch.float32, device=ids.device)
    err = _err_flag(ids.device) if validate else None
    check(lib.mapx_lr_sum_fwd(ptr(ids), B, F, ptr(w), w.numel(), ptr(out), ptr(err), stream()))
    if validate and int(err):
        raise IndexError("index out of range in self")
    return out


def fm_fwd(x3):
    """x3 [B,F,E] -> (fm [B], s [B,E] = sum over fields, kept for backward)   (layers.py:123-131)."""
    require_gpu(x3)
    x3 = x3.contiguous()
    B, F, E = x3.shape
    out = torch.empty(B, dtype=torch.float32, device=x3.device)
    s = torch.empty(B, E, dtype=torch.float32, device=x3.device)
    check(lib.mapx_fm_fwd(ptr(x3), B, F, E, ptr(out), ptr(s), stream()))
    return out, s


def fm_bwd(g, s, x3):
    B, F, E = x3.shape
    dx = torch.empty_like(x3)
    check(lib.mapx_fm_bwd(ptr(g.contiguous()), ptr(s), ptr(x3), B, F, E, ptr(dx), stream()))
    return dx


# --------------------------------------------------------------------------- AutoInt attention core
def attn_fwd(q, k, v, G, F, A, scaled):
    """q, k, v: G*F*A floats each (G = B*heads groups of [F, A]) -> (o same size, p [G,F,F])."""
    require_gpu(q, k, v)
    o = torch.empty_like(q)
    p = torch.empty(G, F, F, dtype=torch.float32, device=q.device)
    check(lib.mapx_attn_fwd(ptr(q), ptr(k), ptr(v), G, F, A, int(bool(scaled)), ptr(o), ptr(p), stream()))
    return o, p


def attn_bwd(q, k, v, p, d_o, G, F, A, scaled):
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    check(lib.mapx_attn_bwd(ptr(q), ptr(k), ptr(v), ptr(p), ptr(d_o.contiguous()), G, F, A, int(bool(scaled)),
                            ptr(dq), ptr(dk), ptr(dv), stream()))
    return dq, dk, dv


# --------------------------------------------------------------------------- NCE
def alias_build(probs_cpu):
    """Host Walker table, bit-identical to the reference's (alias_multinomial.py:39-72)."""
    probs_cpu = probs_cpu.detach().to("cpu", torch.float32).contiguous()
    V = probs_cpu.numel()
    prob = torch.empty(V, dtype=torch.float32)
    alias = torch.empty(V, dtype=torch.int64)
    check(lib.mapx_alias_build_host(probs_cpu.data_ptr(), V, prob.data_ptr(), alias.data_ptr()))
    return prob, alias


def alias_pack(prob, alias):
    require_gpu(prob, alias)
    packed = torch.empty(prob.numel(), 2, dtype=torch.int32, device=prob.device)
    check(lib.mapx_alias_pack(ptr(prob), ptr(alias), prob.numel(), ptr(packed), stream()))
    return packed


def alias_draw(packed, targets, K, seed, offset, offset_dev=None):
    """-> idx int32 [T, K+1]; column 0 = targets.  offset_dev: int32 device scalar added to
    `offset` inside the kernel (hipGraph-replay safe)."""
    require_gpu(packed, targets)
    targets = targets.contiguous()
    T = targets.numel()
    idx = torch.empty(T, K + 1, dtype=torch.int32, device=targets.device)
    check(lib.mapx_alias_draw(ptr(packed), packed.shape[0], ptr(targets), T, K, seed, offset,
                              ptr(offset_dev), ptr(idx), stream()))
    return idx


def nce_pack_idx(targets, noise, V, validate=False):
    require_gpu(targets, noise)
    targets, noise = targets.contiguous(), noise.contiguous()
    T, K = targets.numel(), noise.shape[-1]
    idx = torch.empty(T, K + 1, dtype=torch.int32, device=targets.device)
    err = _err_flag(targets.device) if validate else None
    check(lib.mapx_nce_pack_idx(ptr(targets), ptr(noise), T, K, V, ptr(idx), ptr(err), stream()))
    if validate and int(err.item()):
        raise IndexError("index out of range in self")
    return idx


def lazy_rows_supported(P, K1):
    """mapx_nce_fwd reads rows through their pending updates (lazy=) for this proj_size / sample count."""
    return P == 32 and K1 <= 32


def nce_fwd(enc, masked_index, idx, emb, bias, logq, F, P, want_logits=False, hpos=None, dh_slots=None,
            totals_later=False, lazy=None):
    """-> dict(loss [2] = {mean loss, accuracy}, acc [1] i32, h [T,P], dlogit [T,K+1], dh [T,P], logits or None).
    hpos (grouped encoder): `enc` is h_slots [slots,P]; dh_slots receives dh at the slots too.
    lazy (a native.LazyRows: optim.TableAdam.lazy_rows()): emb / bias rows are read through their pending
    zero-gradient updates — no catch-up pass ran before this call.
    totals_later: loss / acc are left UNWRITTEN and out["totals"] = (partials buffer, count, loss, acc) goes to the
    nce_scatter_dh that must follow (the head's backward inside a training step)."""
    require_gpu(enc, masked_index, idx, emb, bias, logq)
    B, L = masked_index.shape
    T, K1 = idx.shape
    assert T == B * L and (hpos is not None or enc.shape == (B, F * P))
    dev = enc.device
    f32 = dict(dtype=torch.float32, device=dev)
    out = dict(loss=torch.empty(2, **f32), acc=torch.empty(1, dtype=torch.int32, device=dev),
               h=torch.empty(T, P, **f32), dlogit=torch.empty(T, K1, **f32),
               dh=torch.empty(T, P, **f32),
               logits=torch.empty(T, K1, **f32) if want_logits else None)
    nws = lib.mapx_nce_fwd_workspace_bytes()
    # (partials that outlive this call get a buffer of their own, not the stream's shared scratch)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev) if totals_later else scratch(nws, dev)
    left = native_int() if totals_later else None
    rec = amax_record(dev) if (dh_slots is not None and P == 32 and K1 <= 32) else None     # (the p32 kernel writes it)
    # algorithmic bytes: per (target, sample) one table row + bias + log q + id
    with _timed("nce_fwd", T * K1 * (4.0 * P + 8 + 4)):
        check(lib.mapx_nce_fwd(ptr(enc), B, L, F, P, ptr(masked_index.contiguous()), ptr(idx), K1 - 1,
                               ptr(emb), ptr(bias), ptr(logq), emb.shape[0], ptr(out["h"]),
                               ptr(out["dlogit"]), ptr(out["dh"]), ptr(out["logits"]), ptr(out["loss"]),
                               ptr(out["acc"]), ptr(ws), ws.numel(), ptr(hpos), ptr(dh_slots),
                               None if left is None else native_byref(left), ptr(rec),
                               None if lazy is None else native_byref(lazy), stream()))
    tag(dh_slots, rec)
    tag(out["dh"], rec)
    out["totals"] = (ws, left.value, out["loss"], out["acc"]) if (left is not None and left.value > 0) else None
    return out


class EncGroups:
    """Padded by-field slot layout of the B*L targets (csrc/gemm.hip: grouped feat_encoder)."""

    def __init__(self, masked_index, F):
        require_gpu(masked_index)
        B, L = masked_index.shape
        T, dev = B * L, masked_index.device
        self.T, self.L, self.F = T, L, F
        # >= T + 127 F (every group padded to 128); whole rounds of 8 tiles, one per XCD (enc_grouped_fwd's order)
        self.cap = (T + 127 * F + 1023) // 1024 * 1024
        masked_index = masked_index.contiguous()
        i32 = dict(dtype=torch.int32, device=dev)
        self.rowmap = torch.empty(self.cap, **i32)
        self.hpos = torch.empty(T, **i32)
        self.tile_group = torch.empty(self.cap // 128, **i32)
        self.group_start = torch.empty(F + 1, **i32)
        check(lib.mapx_enc_group_layout(ptr(masked_index), T, L, F, self.cap, ptr(self.rowmap), ptr(self.hpos),
                                        ptr(self.tile_group), ptr(self.group_start), stream()))


def _enc_groups_tensors(self):
    return (self.rowmap, self.hpos, self.tile_group, self.group_start)


EncGroups.tensors = _enc_groups_tensors


def enc_grouped_fwd(final, w, b, groups, zero_slots=None):
    """h_slots [cap, 32]: the masked fields' encoder blocks only (26 % of the dense GEMM).
    zero_slots: a [cap, 32] buffer cleared by the same launch."""
    require_gpu(final, w, b)
    h = torch.empty(groups.cap, 32, dtype=torch.float32, device=final.device)
    ra, rb = amax_of(final), amax_of(w)
    if AUTO_AMAX and H2:
        ra = ra if ra is not None else amax(final)
        rb = rb if rb is not None else amax(w)
    sc = _scale_arg(ra, rb) if (ra is not None and rb is not None) else None
    with _timed("gemm_enc_grouped_fwd", 2.0 * groups.T * 32 * final.shape[1]):
        check(lib.mapx_enc_grouped_fwd(final.data_ptr(), final.stride(0), final.shape[0], final.shape[1],
                                       ptr(w), w.stride(0), ptr(b), ptr(groups.rowmap), ptr(groups.tile_group),
                                       ptr(groups.group_start), groups.F, groups.cap, ptr(h), ptr(zero_slots),
                                       None if sc is None else native_byref(sc), stream()))
    return h


def enc_grouped_dw(dh_slots, final, groups, out=None, gscale=None):
    """dW [F*32, D+H] of feat_encoder from the slot-ordered dh (every row written), times the
    device scalar `gscale` if given."""
    require_gpu(dh_slots, final)
    Nn = final.shape[1]
    if out is None:
        out = torch.empty(groups.F * 32, Nn, dtype=torch.float32, device=final.device)
    ra, rb = amax_of(dh_slots), amax_of(final)
    if AUTO_AMAX and H2:
        ra = ra if ra is not None else amax(dh_slots)
        rb = rb if rb is not None else amax(final)
    sc = _scale_arg(ra, rb) if (ra is not None and rb is not None) else None
    with _timed("gemm_enc_grouped_dw", 2.0 * groups.T * 32 * Nn):
        check(lib.mapx_enc_grouped_dw(ptr(dh_slots), final.data_ptr(), final.stride(0), final.shape[0], Nn,
                                      ptr(groups.rowmap), ptr(groups.group_start), groups.F, ptr(gscale),
                                      ptr(out), out.stride(0), None if sc is None else native_byref(sc), stream()))
    return out


def nce_scatter_dh(dh, masked_index, F, P, gscale=None, totals=None):
    """`totals`: nce_fwd(totals_later=True)'s out["totals"] — this launch then also writes the loss / accuracy."""
    require_gpu(dh, masked_index)
    B, L = masked_index.shape
    denc = torch.empty(B, F * P, dtype=torch.float32, device=dh.device)
    ws, n, loss, acc = totals if totals is not None else (None, 0, None, None)
    rec = amax_record(dh.device)
    check(lib.mapx_nce_scatter_dh(ptr(dh), ptr(masked_index.contiguous()), ptr(gscale), B, L, F, P,
                                  ptr(denc), ptr(ws), n, ptr(loss), ptr(acc), ptr(rec), stream()))
    return tag(denc, rec)


def nce_table_grad(plan, dlogit, h, K, P, gscale=None):
    """-> (emb grad rows [cap,P], bias grad rows [cap]) for plan.uniq; dlogit is multiplied by the
    device scalar `gscale` (incoming gradient of the loss) on the fly."""
    dev = dlogit.device
    out_emb = torch.empty(max(plan.n, 1), P, dtype=torch.float32, device=dev)
    out_bias = torch.empty(max(plan.n, 1), dtype=torch.float32, device=dev)
    nb = lib.mapx_nce_table_grad_workspace_bytes(plan.n, P)
    ws = scratch(nb, dev)
    with _timed("nce_table_grad", plan.n * (4.0 * P + 4)):
        check(lib.mapx_nce_table_grad(plan.n, ptr(plan.perm), ptr(plan.rank), ptr(plan.seg_start),
                                      ptr(dlogit), ptr(h), K, P, ptr(gscale), ptr(out_emb), ptr(out_bias),
                                      ptr(ws), ws.numel(), plan.take_counter(), stream()))
    return out_emb, out_bias


def scale_(x, g):
    check(lib.mapx_scale_inplace(ptr(x), x.numel(), ptr(g), stream()))
    return x


# --------------------------------------------------------------------------- dense
_EPI_RELU_MASK_COLSUM = N.EPI_RELU_MASK_COLSUM      # (gemm_bf16's N is a size)


def gemm_bf16(a, b, a_kc, b_kc, M, N, K, out=None, out_dtype=BF16, ldc=None, epi=N.EPI_NONE, bias=None,
              aux1=None, aux2=None, out2=None, nsplit=1, lda=None, ldb=None, tile=-1):
    """The bf16-operand GEMM (include/mapx_hip.h: mapx_gemm_bf16): a, b, aux2, out2 bf16; C bf16 or
    fp32 (`out_dtype`, or the dtype of `out`); aux1 bf16 or fp32 (EPI_ADD only); fp32 accumulation."""
    require_gpu(a, b)
    if not (is_bf16(a) and is_bf16(b)):
        raise TypeError("gemm_bf16 takes bf16 operands")
    dev = a.device
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype, device=dev)
    c_f32 = out.dtype == torch.float32
    if not (c_f32 or is_bf16(out)):
        raise TypeError("gemm_bf16 writes bf16 or fp32")
    ldc = ldc if ldc is not None else out.stride(0)
    lda = lda if lda is not None else a.stride(0)
    ldb = ldb if ldb is not None else b.stride(0)
    ws, wsn = None, 0
    if nsplit > 1:
        wsn = lib.mapx_gemm_splitk_workspace_bytes(M, N, nsplit)
        ws = scratch(wsn, dev)
        wsn = ws.numel()
    if bias is not None and bias.dtype != torch.float32:
        raise TypeError("gemm_bf16: the bias stays fp32")
    aux1_f32 = aux1 is not None and aux1.dtype == torch.float32
    for t_ in (aux2, out2):
        if t_ is not None and not is_bf16(t_) and not (t_ is out2 and epi == _EPI_RELU_MASK_COLSUM):
            raise TypeError("gemm_bf16: aux2 / out2 are bf16 (out2 of EPI_RELU_MASK_COLSUM: fp32 partial rows)")
    kind = "gemm_fwd_nt" if (a_kc and b_kc) else ("gemm_dx_nn" if a_kc else "gemm_dw_tn")
    with _timed(kind, 2.0 * M * N * K):
        check(lib.mapx_gemm_bf16(int(a_kc), int(b_kc), M, N, K, a.data_ptr(), lda, b.data_ptr(), ldb,
                                 out.data_ptr(), ldc, int(c_f32), epi, ptr(bias),
                                 aux1.data_ptr() if aux1 is not None else None,
                                 aux1.stride(0) if aux1 is not None else 0, int(aux1_f32),
                                 aux2.data_ptr() if aux2 is not None else None,
                                 aux2.stride(0) if aux2 is not None else 0,
                                 out2.data_ptr() if out2 is not None else None,
                                 out2.stride(0) if out2 is not None else 0, nsplit, tile,
                                 ws.data_ptr() if ws is not None else None, wsn, stream()))
    return out


# Magnitude records (include/mapx_hip.h: mapx_gemm_scale; csrc/amax.h): every kernel that writes a tensor a GEMM will
# read leaves max |x| in the tensor's 8-byte record; a product whose two operands carry one is formed by the two-piece
# fp16 arithmetic (csrc/gemm_h2.hip), any other by the six-product bf16 one (csrc/gemm_x3.hip).  Here a record travels
# as the attribute `_amax` of the tensor object: outputs of autograd Functions, saved INPUTS and gradients handed from
# one backward node to the next keep their attributes; saved OUTPUTS and slices do not (carry / amax_pack below).
H2 = os.environ.get("MAPX_GEMM_H2", "1") == "1"          # A/B switch: 0 = no records anywhere, x3 arithmetic only
_epoch_word = {}
_cap_pool = [None, 0]          # records of the capture in progress: (int32 [2 n] tensor, next free)
_cap_pools = []                # ... of finished captures: a replayed graph has their addresses baked in
_ring, _RING = {}, 1 << 12     # eager records
REC = 128                      # int32 words of a record (MAPX_AMAX_RECORD_BYTES / 4: 64 slots of {bits, epoch})


def _ensure_epoch(device):
    """The process-wide epoch word of the records (mapx_amax_epoch_source): allocated once, never freed."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _epoch_word:
        w = torch.zeros(1, dtype=torch.int32, device=device)
        check(lib.mapx_amax_epoch_source(w.data_ptr()))
        _epoch_word[key] = w
    return _epoch_word[key]


def amax_capture_begin(device, n=1024):
    """Called right before a hipGraph capture: the records handed out during the capture come from one zeroed
    block allocated HERE (a record allocated inside the capture would be a memset node of its own; the epoch tags
    make resets unnecessary)."""
    if H2:
        _ensure_epoch(device)
        _cap_pool[0], _cap_pool[1] = torch.zeros(REC * n, dtype=torch.int32, device=device), 0
        _cap_pools.append(_cap_pool[0])


def amax_capture_end():
    _cap_pool[0] = None


def amax_record(device):
    """A zeroed magnitude record, or None where none can be had (records switched off; inside a capture that
    did not announce itself: the product then simply takes the six-product arithmetic)."""
    if not H2:
        return None
    if torch.cuda.is_current_stream_capturing():
        pool, i = _cap_pool
        if pool is None or REC * (i + 1) > pool.numel():
            return None
        _cap_pool[1] = i + 1
        return pool[REC * i:REC * (i + 1)]
    _ensure_epoch(device)
    # eager launches: a ring of records that is never freed (a record outlives the tensor object it hangs on
    # whenever a side stream still runs the kernel that raises it), each zeroed when it is handed out
    key = device.index if device.index is not None else torch.cuda.current_device()
    ring = _ring.get(key)
    if ring is None:
        ring = _ring[key] = [torch.zeros(REC * _RING, dtype=torch.int32, device=device), 0]
    i = ring[1]
    ring[1] = (i + 1) % _RING
    rec = ring[0][REC * i:REC * (i + 1)]
    rec.zero_()
    return rec


def amax_of(t):
    """The record of tensor `t`, or None.  A parameter's record is kept by the optimizer's kernel; when something else
    wrote the parameter through torch (load_state_dict, copy_: its version counter moved) it is recomputed here."""
    if not H2 or t is None:
        return None
    rec = getattr(t, "_amax", None)
    if rec is not None:
        ver = getattr(t, "_amax_ver", None)
        if ver is not None and ver != t._version:
            amax(t.detach(), rec=rec, reset=True)
            t._amax_ver = t._version
            refresh_weight_planes(t)
    return rec


def tag(t, rec):
    if rec is not None and t is not None:
        t._amax = rec
    return t


def carry(dst, src):
    """`dst` is a slice / view / alias of `src`: the maximum over src bounds the maximum over dst."""
    return tag(dst, amax_of(src))


def flat_rows(x):
    """x.flatten(start_dim=1) with x's record (the flattened embedding rows are the towers' first operand)."""
    return carry(x.flatten(start_dim=1), x)


def cols(t, c0, c1):
    """t[:, c0:c1] with t's record (and, for a weight, the note which columns of which parameter it is)."""
    v = carry(t[:, c0:c1], t)
    if getattr(t, "_planes", None) is not None:
        v._wslice = (t, c0, c1)
    return v


def amax_pack(*ts):
    return tuple(amax_of(t) for t in ts)


def amax_unpack(ts, recs):
    for t, r in zip(ts, recs):
        tag(t, r)


def out_record(out, device):
    """The record a kernel raises for its output: the one the caller put on `out` (several kernels writing
    column ranges of one buffer share the buffer's record), else a fresh one."""
    rec = amax_of(out)
    return rec if rec is not None else amax_record(device)


def amax(x, rec=None, reset=True):
    """The magnitude record of a tensor no mapx kernel produced: one pass over x [rows, cols] (or 1-D)."""
    require_gpu(x)
    if x.dtype != torch.float32:
        raise TypeError("amax: fp32 tensors")
    _ensure_epoch(x.device)
    if rec is None:
        rec = torch.empty(REC, dtype=torch.int32, device=x.device)
        reset = True
    x2 = x if x.dim() == 2 else x.reshape(1, -1)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    check(lib.mapx_amax_f32(x2.data_ptr(), x2.shape[0], x2.shape[1], x2.stride(0), rec.data_ptr(), int(reset), stream()))
    return rec


def amax_value(rec):
    """Host value of a record (tests)."""
    return float(rec[0::2].max().view(1).view(torch.float32).item())      # (bit patterns of values >= 0 order as the values)


# MAPX_AMAX_CHECK=1 (tests): every record a product is given is compared with the operand on the host — a record
# below the operand's true maximum would overflow fp16, one far above it wastes precision.  H2_USED counts the
# products that were handed both records (tests assert that the hot path really takes the new arithmetic).
AMAX_CHECK = os.environ.get("MAPX_AMAX_CHECK", "0") == "1"
# tests: a product whose operand comes without a record gets one computed on the spot (and its weight-like operand B
# its planes), so that every GEMM test also runs through the two-piece fp16 kernels
AUTO_AMAX = os.environ.get("MAPX_AUTO_AMAX", "0") == "1"
H2_USED = [0, 0]               # products with both records / without


def _check_record(x, rec, what):
    if rec is None:
        return
    xs = x.detach().float()
    xs = xs[torch.isfinite(xs)]
    true = float(xs.abs().max()) if xs.numel() else 0.0
    got = amax_value(rec)
    if not (got >= true and (true == 0.0 or got <= true * 64.0 or got < 1e-30)):
        raise AssertionError(f"magnitude record of operand {what}: {got!r} for a tensor whose max |x| is {true!r}")


# Weight planes (include/mapx_hip.h: mapx_h2_weight_planes; csrc/gemm_h2w.hip): a weight that is operand B of large
# products is cut into its two fp16 pieces once per optimizer step instead of once per row tile of every product.  The
# optimizer owns the cache: `p._planes` {(b_kc, c0, c1): planes} on a parameter, filled at first use, refreshed by
# MapxOptimizer right after the AdamW kernel (and by refresh_bf16 after anything else wrote the parameter).
H2W = os.environ.get("MAPX_GEMM_H2W", "1") == "1"


def h2_weight_planes(w, b_kc, rec, out=None):
    """The planes of w as operand B (b_kc: B(k,n) = w[n,k], forward; else B(k,n) = w[k,n], input gradient; w may
    be a column slice) at the scale its record `rec` gives now."""
    require_gpu(w)
    Nn, K = (w.shape[0], w.shape[1]) if b_kc else (w.shape[1], w.shape[0])
    if out is None:
        out = torch.empty(lib.mapx_h2_weight_planes_bytes(Nn, K), dtype=torch.uint8, device=w.device)
    check(lib.mapx_h2_weight_planes(w.data_ptr(), w.stride(0), Nn, K, int(b_kc), rec.data_ptr(), out.data_ptr(), stream()))
    return out


def planes_wanted(M, Nn, K):
    """Would gemm_h2w.hip take a product of this shape?  (>= 128 tiles of 128 x 128 — or of 128 x 64 —, K >= 64, K % 8 == 0)"""
    return math.ceil(M / 128) * math.ceil(Nn / 64) >= 128 and K >= 64 and K % 8 == 0


def weight_planes(w, b_kc, M):
    """The cached planes of weight (or weight slice, ops.cols) `w` for a product with M rows, or None: no optimizer
    owns the parameter, records are off, or the product is not one gemm_h2w.hip takes."""
    if not (H2 and H2W) or w is None or w.dim() != 2 or w.dtype != torch.float32:
        return None
    base, c0, c1 = getattr(w, "_wslice", (w, 0, None))
    reg = getattr(base, "_planes", None)
    rec = amax_of(base)
    if reg is None or rec is None:
        return None
    if _dirty_planes:
        refresh_dirty_planes()            # (nobody re-cut them at the start of this step: do it before the first reader)
    Nn, K = (w.shape[0], w.shape[1]) if b_kc else (w.shape[1], w.shape[0])
    if not planes_wanted(M, Nn, K):
        return None
    key = (bool(b_kc), c0, c1)
    pl = reg.get(key)
    if pl is None:
        if torch.cuda.is_current_stream_capturing():
            return None               # (registered by the eager steps that precede a capture)
        pl = reg[key] = h2_weight_planes(base.detach()[:, c0:c1] if (c0, c1) != (0, None) else base.detach(), b_kc, rec)
    return pl


# Where the planes are re-cut: right behind the AdamW kernel (end of the step, "0"), or at the START of the next step
# ("1"): the optimizer only notes which parameters it wrote; the model's forward re-cuts them on a stream that is idle
# while the step's head (mask, catch-up, gather) runs, and any product that asks for planes before that re-cuts on the
# spot — a stale plane is never read.
PLANES_AT_START = os.environ.get("MAPX_PLANES_AT_START", "0") == "1"
_dirty_planes = []


def planes_written(params):
    """The optimizer wrote these parameters (their records are current): re-cut their planes now or note them."""
    if PLANES_AT_START:
        _dirty_planes.append(list(params))
    else:
        refresh_weight_planes(params)


def refresh_dirty_planes():
    """Re-cut the planes of every parameter noted by planes_written, on the current stream.  -> True if any."""
    if not _dirty_planes:
        return False
    todo = [p for ps in _dirty_planes for p in ps]
    del _dirty_planes[:]
    refresh_weight_planes(todo)
    return True


def refresh_weight_planes(params):
    """Re-cut every registered plane set of the parameters (their records must be current on this stream): one
    launch per 16 sets."""
    if isinstance(params, torch.Tensor):
        params = [params]
    todo = []
    for p in params:
        for (b_kc, c0, c1), pl in (getattr(p, "_planes", None) or {}).items():
            w = p.detach()[:, c0:c1] if (c0, c1) != (0, None) else p.detach()
            Nn, K = (w.shape[0], w.shape[1]) if b_kc else (w.shape[1], w.shape[0])
            todo.append((w, Nn, K, b_kc, p._amax, pl))
    for i in range(0, len(todo), 16):
        part = todo[i:i + 16]
        arr = (N.PlaneTask * len(part))()
        for j, (w, Nn, K, b_kc, rec, pl) in enumerate(part):
            arr[j].W, arr[j].ldw, arr[j].N, arr[j].K, arr[j].b_kc = w.data_ptr(), w.stride(0), Nn, K, int(b_kc)
            arr[j].amax_record, arr[j].planes = rec.data_ptr(), pl.data_ptr()
        check(lib.mapx_h2_weight_planes_multi(arr, len(part), stream()))


def _as2d(t, ld):
    """The [rows, ld] matrix an operand pointer + leading dimension describes (AUTO_AMAX: a superset of the operand)."""
    return t if t.dim() == 2 else t.reshape(-1, ld)


def _scale_arg(amax_a, amax_b, amax_c=None, amax_c2=None, b_planes=None):
    if amax_a is None and amax_b is None and amax_c is None and amax_c2 is None:
        return None
    sc = N.GemmScale()
    sc.b_planes = b_planes.data_ptr() if (b_planes is not None and amax_a is not None) else None
    sc.amax_a = amax_a.data_ptr() if amax_a is not None else None
    sc.amax_b = amax_b.data_ptr() if amax_b is not None else None
    sc.amax_c = amax_c.data_ptr() if amax_c is not None else None
    sc.amax_c2 = amax_c2.data_ptr() if amax_c2 is not None else None
    return sc


def gemm(a, b, a_kc, b_kc, M, N, K, out=None, ldc=None, epi=N.EPI_NONE, bias=None, aux1=None,
         aux2=None, out2=None, nsplit=1, lda=None, ldb=None, tile=-1, defer=False, out_dtype=None,
         amax_a=None, amax_b=None, amax_c=None, record=True, b_planes=None):
    """C[M,N] = epi(sum_k A(m,k) B(k,n)); see include/mapx_hip.h: mapx_gemm_f32 (fp32 operands) /
    mapx_gemm_bf16 (bf16 operands; `out_dtype` picks a bf16 or fp32 result).  amax_a / amax_b: the operands'
    magnitude records (both given: the two-piece fp16 arithmetic); amax_c: record raised with max |C|."""
    require_gpu(a, b)
    if is_bf16(a):
        return gemm_bf16(a, b, a_kc, b_kc, M, N, K, out=out, out_dtype=out_dtype or BF16, ldc=ldc, epi=epi,
                         bias=bias, aux1=aux1, aux2=aux2, out2=out2, nsplit=nsplit, lda=lda, ldb=ldb, tile=tile)
    dev = a.device
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=dev)
    ldc = ldc if ldc is not None else out.stride(0)
    lda = lda if lda is not None else a.stride(0)
    ldb = ldb if ldb is not None else b.stride(0)
    ws, wsn = None, 0
    if nsplit > 1:
        wsn = lib.mapx_gemm_splitk_workspace_bytes(M, N, nsplit)
        # deferred slabs must outlive this call: their own buffer, not the shared scratch
        ws = torch.empty(wsn, dtype=torch.uint8, device=dev) if defer else scratch(wsn, dev)
        wsn = ws.numel()
    got = native_int() if (defer and nsplit > 1) else None
    ld1 = aux1.stride(0) if aux1 is not None else 0
    ld2 = aux2.stride(0) if aux2 is not None else 0
    ldo2 = out2.stride(0) if out2 is not None else 0
    kind = "gemm_fwd_nt" if (a_kc and b_kc) else ("gemm_dx_nn" if a_kc else "gemm_dw_tn")
    amax_a = amax_a if amax_a is not None else amax_of(a)
    amax_b = amax_b if amax_b is not None else amax_of(b)
    if AUTO_AMAX and H2:
        amax_a = amax_a if amax_a is not None else amax(_as2d(a, lda))
        amax_b = amax_b if amax_b is not None else amax(_as2d(b, ldb))
        if b_planes is None and H2W and a_kc and nsplit == 1 and tile < 0 and planes_wanted(M, N, K) \
                and b.dim() == 2 and b.stride(1) == 1:
            b_planes = h2_weight_planes(b[:N, :K] if b_kc else b[:K, :N], b_kc, amax_b)
    if amax_c is None and record and nsplit == 1:      # (split-K outputs are weight gradients: nobody multiplies them)
        amax_c = out_record(out, dev)
    H2_USED[0 if (amax_a is not None and amax_b is not None) else 1] += 1
    if AMAX_CHECK and not torch.cuda.is_current_stream_capturing():
        _check_record(a, amax_a, "A")
        _check_record(b, amax_b, "B")
    sc = _scale_arg(amax_a, amax_b, amax_c, b_planes=b_planes if b_planes is not None else
                    (weight_planes(b, b_kc, M) if (a_kc and nsplit == 1 and tile < 0) else None))
    with _timed(kind, 2.0 * M * N * K):
        check(lib.mapx_gemm_f32(int(a_kc), int(b_kc), M, N, K, a.data_ptr(), lda, b.data_ptr(), ldb,
                                out.data_ptr(), ldc, epi, ptr(bias),
                                aux1.data_ptr() if aux1 is not None else None, ld1,
                                aux2.data_ptr() if aux2 is not None else None, ld2,
                                out2.data_ptr() if out2 is not None else None, ldo2, nsplit, tile,
                                ws.data_ptr() if ws is not None else None, wsn,
                                None if got is None else native_byref(got),
                                None if sc is None else native_byref(sc), stream()))
    if got is not None and got.value > 1:
        defer_sum(out, ws.view(torch.float32), M * N, got.value, M * N)
    return tag(out, amax_c)


# Layers with at most 32 outputs (RFD's last predictor layer, the finetune head) as fp32 streaming kernels instead of
# MFMA GEMM tiles that are mostly padding (csrc/skinny.hip).  A/B switch: MAPX_SKINNY=0.
SKINNY = os.environ.get("MAPX_SKINNY", "1") == "1"


def _rows16(t):
    return t.dim() == 2 and t.dtype == torch.float32 and t.stride(1) == 1 and t.stride(0) % 4 == 0 \
        and t.data_ptr() % 16 == 0


def _rows8h(t):
    return t.dim() == 2 and t.dtype == torch.bfloat16 and t.stride(1) == 1 and t.stride(0) % 4 == 0 \
        and t.data_ptr() % 8 == 0


# bf16 compute mode: layers of at most 8 outputs (the finetune head) on the streaming kernels' bf16 forms instead of a
# 128-wide MFMA tile per useful column
SKINNY_BF16 = os.environ.get("MAPX_SKINNY_BF16", "1") == "1"


def _skinny_h(Nn, K, *mats):
    return SKINNY and SKINNY_BF16 and 1 <= Nn <= 8 and K >= 4 and K % 4 == 0 and all(_rows8h(m) for m in mats)


_TALL_ROWS = int(os.environ.get("MAPX_TALL_ROWS", "128"))
SKINNY_MAX = int(os.environ.get("MAPX_SKINNY_MAX", "8"))            # forward: wider layers measured faster on the GEMM
# dW / dX (RFD's 23-wide layer: 14.6 / 8.6 vs 28 / 12 us).  The kernels take up to 64 outputs (Criteo's 39-wide layer:
# dW 45 -> 26 us, its RFD step -2.3 %): opt-in.  (With 33..64 on, the captured step of the 64-wide test models stopped
# agreeing with the eager one — traced to the packed FMAs of skinny_dw_tall_kernel, see there; now a test of its own.)
SKINNY_MAX_BWD = int(os.environ.get("MAPX_SKINNY_MAX_BWD", "32"))
SKINNY_TALL = os.environ.get("MAPX_SKINNY_TALL", "0") == "1"


def _skinny(Nn, K, *mats, bwd=False):
    return SKINNY and 1 <= Nn <= (SKINNY_MAX_BWD if bwd else SKINNY_MAX) and K >= 4 and K % 4 == 0 \
        and all(_rows16(m) for m in mats)


def _sum_now(dst, src, stride, nsplit, n):
    arr = (N.SumTask * 1)()
    arr[0].dst, arr[0].src, arr[0].stride, arr[0].n, arr[0].nsplit = dst.data_ptr(), src.data_ptr(), stride, n, nsplit
    check(lib.mapx_sum_tasks(arr, 1, stream()))


def linear_fwd(x, w, b, relu=False, out=None, out_dtype=None):
    """nn.Linear (+ReLU): x [M,K], w [N,K], b [N] -> [M,N]; `out` may be a column slice.  bf16 x:
    `w` is the weight's bf16 operand (ops.bf16_weight), b stays fp32, the result is bf16 or
    (`out_dtype=torch.float32`: the heads' logits) fp32."""
    M, K = x.shape
    Nn = w.shape[0]
    if out_dtype in (None, torch.float32) and _skinny(Nn, K, x, w) and (out is None or out.dtype == torch.float32):
        require_gpu(x, w)
        y = out if out is not None else torch.empty(M, Nn, dtype=torch.float32, device=x.device)
        with _timed("skinny_linear", 4.0 * (M * K + Nn * K + M * Nn)):
            check(lib.mapx_skinny_linear_fwd(ptr(x), x.stride(0), ptr(w), w.stride(0), ptr(b), M, Nn, K, int(relu),
                                             y.data_ptr(), y.stride(0), stream()))
        return y
    if out_dtype == torch.float32 and _skinny_h(Nn, K, x, w) and (out is None or out.dtype == torch.float32):
        require_gpu(x, w)
        y = out if out is not None else torch.empty(M, Nn, dtype=torch.float32, device=x.device)
        with _timed("skinny_linear", 2.0 * (M * K + Nn * K) + 4.0 * M * Nn):
            check(lib.mapx_skinny_linear_fwd_bf16(ptr(x), x.stride(0), ptr(w), w.stride(0), ptr(b), M, Nn, K, int(relu),
                                                  y.data_ptr(), y.stride(0), stream()))
        return y
    return gemm(x, w, True, True, M, Nn, K, out=out, epi=N.EPI_BIAS_RELU if relu else N.EPI_BIAS,
                bias=b, out_dtype=out_dtype)


RELU_LINK = os.environ.get("MAPX_RELU_LINK", "1") == "1"


def fused_mask_colsum_ok(dy, relu_of):
    """Can linear_bwd_input also apply the upstream ReLU's mask and form its bias gradient?  (16-byte rows,
    deferred partial sums on.)"""
    if not (RELU_LINK and DEFER_COLSUM and relu_of.dim() == 2 and dy.dtype == relu_of.dtype):
        return False
    if is_bf16(dy):
        return relu_of.shape[1] % 8 == 0 and relu_of.stride(1) == 1 and relu_of.stride(0) % 8 == 0 \
            and relu_of.data_ptr() % 16 == 0
    return dy.dtype == torch.float32 and relu_of.shape[1] % 4 == 0 and row_sliceable(relu_of)


def part_rows(M, dtype=torch.float32):
    """Partial rows of a column-sum epilogue (EPI_RELU_MASK_COLSUM, mapx_gemm_f32_bwd_fused) over M rows: the fp32
    kernels write one per 64 rows (tiles of 128 rows: the sum and a row of zeros), the bf16 ones one per 128."""
    return (M + 63) // 64 if dtype == torch.float32 else (M + 127) // 128


def linear_bwd_input(dy, w, out=None, add=None, relu_of=None, colsum_to=None):
    """dX = dY W  (+ add)  or masked by relu_of > 0.  dy [M,N], w [N,K] -> [M,K] (dtype of dy;
    bf16: `add` may be fp32 — the cross tower's running dL/dX0).
    colsum_to (with relu_of; see fused_mask_colsum_ok): the masked result is the upstream ReLU layer's dZ,
    and its column sums — that layer's bias gradient — leave the same epilogue as partial rows (part_rows),
    summed into `colsum_to` by flush_deferred()."""
    M, Nn = dy.shape
    K = w.shape[1]
    if add is None and relu_of is None and _skinny(Nn, K, w, bwd=True) and dy.dtype == torch.float32 and dy.dim() == 2 \
            and dy.stride(1) == 1 and (out is None or _rows16(out)):
        require_gpu(dy, w)
        dx = out if out is not None else torch.empty(M, K, dtype=torch.float32, device=dy.device)
        with _timed("skinny_linear", 4.0 * (M * K + Nn * K + M * Nn)):
            check(lib.mapx_skinny_linear_dx(ptr(dy), dy.stride(0), ptr(w), w.stride(0), M, Nn, K, dx.data_ptr(),
                                            dx.stride(0), stream()))
        return dx
    if add is None and relu_of is None and dy.dim() == 2 and _skinny_h(Nn, K, w) and is_bf16(dy) and dy.stride(1) == 1 \
            and (out is None or _rows8h(out)):
        require_gpu(dy, w)
        dx = out if out is not None else torch.empty(M, K, dtype=torch.bfloat16, device=dy.device)
        with _timed("skinny_linear", 2.0 * (M * K + Nn * K + M * Nn)):
            check(lib.mapx_skinny_linear_dx_bf16(ptr(dy), dy.stride(0), ptr(w), w.stride(0), M, Nn, K, dx.data_ptr(),
                                                 dx.stride(0), stream()))
        return dx
    epi, aux, out2 = N.EPI_NONE, None, None
    if add is not None:
        epi, aux = N.EPI_ADD, add
    elif relu_of is not None and colsum_to is not None:
        epi, aux = N.EPI_RELU_MASK_COLSUM, relu_of
        out2 = torch.empty(part_rows(M, dy.dtype), K, dtype=torch.float32, device=dy.device)
    elif relu_of is not None:
        epi, aux = N.EPI_RELU_MASK, relu_of
    dx = gemm(dy, w, True, False, M, K, Nn, out=out, epi=epi, aux1=aux, out2=out2)
    if out2 is not None:
        defer_sum(colsum_to, out2, K, out2.shape[0], K)
    return dx


def _splits_wide_tiles(M, Nn, Kred):
    """split-K factor for the bf16-MFMA kernels (bf16 operands, or fp32 cut into three bf16 pieces):
    128 x 128 tiles, split over K until they cover the 256 CUs, K chunk >= 256."""
    tiles = math.ceil(M / 128) * math.ceil(Nn / 128)
    ns = 1
    while ns < 16 and tiles * ns * 2 <= 288 and Kred // (ns * 2) >= 256:
        ns *= 2
    return ns


def linear_bwd_weight(dy, x, out=None, defer=False):
    """dW = dY^T X.  dy [B,N], x [B,K] -> [N,K].  defer: leave split-K slabs for flush_deferred()."""
    Bn, Nn = dy.shape
    K = x.shape[1]
    # (also: both dimensions small over many rows — AutoInt's attention projections, dW [40, 16 | 40] over B*F rows.
    # Opt-in since round 4 (MAPX_SKINNY_TALL=1): this is the kernel whose compiler-generated packed FMAs
    # gave a timing-dependent wrong result beside MFMA kernels in round 3; the re-written loop has passed every
    # bitwise graph == eager run since, but the mechanism was never found, so the default is the GEMM.)
    tall = SKINNY and SKINNY_TALL and 32 < Nn <= 64 and 4 <= K <= 64 and K % 4 == 0 and Bn >= 8192 and _rows16(x)
    ok = _skinny(Nn, K, x, bwd=True) or tall
    rows_cap = 64 if (Nn > 32 and K > 64) else _TALL_ROWS     # (33..64 outputs x many columns: the LDS-tiled form only)
    if ok and Nn > 32 and K > 64 and Bn > 64 * 2048:
        ok = False
    if Bn >= 1 and ok and dy.dtype == torch.float32 and dy.dim() == 2 \
            and dy.stride(1) == 1 \
            and (out is None or (out.dtype == torch.float32 and out.is_contiguous())):
        require_gpu(dy, x)
        chunks = lib.mapx_skinny_chunks()
        while -(-Bn // chunks) > rows_cap and chunks < 2048:  # tall problems: a chunk is a workgroup
            chunks *= 2
        dw = out if out is not None else torch.empty(Nn, K, dtype=torch.float32, device=dy.device)
        part = torch.empty(chunks, Nn * K, dtype=torch.float32, device=dy.device)
        with _timed("skinny_linear", 4.0 * (Bn * K + Bn * Nn + chunks * Nn * K)):
            check(lib.mapx_skinny_linear_dw(ptr(dy), dy.stride(0), ptr(x), x.stride(0), Bn, Nn, K, ptr(part), chunks,
                                            stream()))
        if DEFER_COLSUM and defer and out is not None:
            defer_sum(dw, part, Nn * K, chunks, Nn * K)          # with the step's other partial sums
        else:
            _sum_now(dw, part, Nn * K, chunks, Nn * K)
        return dw
    if Bn >= 1 and is_bf16(dy) and dy.dim() == 2 and dy.stride(1) == 1 and _skinny_h(Nn, K, x) \
            and (out is None or (out.dtype == torch.float32 and out.is_contiguous())):
        require_gpu(dy, x)
        chunks = lib.mapx_skinny_chunks()
        while -(-Bn // chunks) > _TALL_ROWS and chunks < 2048:
            chunks *= 2
        dw = out if out is not None else torch.empty(Nn, K, dtype=torch.float32, device=dy.device)
        part = torch.empty(chunks, Nn * K, dtype=torch.float32, device=dy.device)
        with _timed("skinny_linear", 2.0 * (Bn * K + Bn * Nn) + 4.0 * chunks * Nn * K):
            check(lib.mapx_skinny_linear_dw_bf16(ptr(dy), dy.stride(0), ptr(x), x.stride(0), Bn, Nn, K, ptr(part), chunks,
                                                 stream()))
        if DEFER_COLSUM and defer and out is not None:
            defer_sum(dw, part, Nn * K, chunks, Nn * K)
        else:
            _sum_now(dw, part, Nn * K, chunks, Nn * K)
        return dw
    # the bf16-MFMA kernels want 128 x 128 tiles (half the L2 -> LDS bytes per flop of 64 x 64 ones)
    ns = _splits_wide_tiles(Nn, K, Bn)
    if out is not None and out.stride(0) != K:
        ns = 1
    if is_bf16(dy):                          # fp32 gradient from bf16 operands
        return gemm_bf16(dy, x, False, False, Nn, K, Bn, out=out, out_dtype=torch.float32, nsplit=ns)
    return gemm(dy, x, False, False, Nn, K, Bn, out=out, nsplit=ns, defer=DEFER and defer and out is not None,
                record=False)


def _partials_h(Nn, device, defer):
    """Workspace of the bf16 column-sum kernels -> (buffer, chunk rows); defer: a buffer of its own (it lives until
    flush_deferred sums it) instead of the stream's scratch."""
    nb = lib.mapx_colsum_bf16_workspace_bytes(Nn)
    buf = torch.empty(nb, dtype=torch.uint8, device=device) if defer else scratch(nb, device)
    return buf, nb // (4 * Nn)


def _partials(Nn, device, defer):
    nb = lib.mapx_colsum_workspace_bytes(Nn)
    return torch.empty(nb, dtype=torch.uint8, device=device) if defer else scratch(nb, device)


def colsum(x, out=None, defer=False):
    """Column sums; defer (needs out): leave the row-chunk partials for flush_deferred()."""
    require_gpu(x)
    M, Nn = x.shape
    if is_bf16(x):
        later = DEFER_COLSUM and DEFER_COLSUM_H and defer and out is not None
        if out is None:
            out = torch.empty(Nn, dtype=torch.float32, device=x.device)
        if x.stride(1) != 1:
            x = x.contiguous()
        ws, chunks = _partials_h(Nn, x.device, later)
        check(lib.mapx_colsum_bf16(x.data_ptr(), x.stride(0), M, Nn, None if later else ptr(out), ptr(ws), ws.numel(),
                                   stream()))
        if later:
            defer_sum(out, ws.view(torch.float32), Nn, chunks, Nn)
        return out
    defer = DEFER_COLSUM and defer and out is not None
    if out is None:
        out = torch.empty(Nn, dtype=torch.float32, device=x.device)
    ws = _partials(Nn, x.device, defer)
    check(lib.mapx_colsum(x.data_ptr(), x.stride(0), M, Nn, None if defer else ptr(out), ptr(ws),
                          ws.numel(), stream()))
    if defer:
        defer_sum(out, ws.view(torch.float32), Nn, COLSUM_CHUNKS, Nn)
    return out


def cross_layer_fwd(x0, xi, w, b, out=None):
    """-> (X_{i+1} = Xi + X0 * (Xi W^T + b), u = Xi W^T + b)   (layers.py:200)."""
    M, D = xi.shape
    u = torch.empty(M, D, dtype=xi.dtype, device=xi.device)
    y = gemm(xi, w, True, True, M, D, D, out=out, epi=N.EPI_BIAS_CROSS, bias=b, aux1=xi, aux2=x0,
             out2=u)            # (y carries its record: the next layer's operand; u is read elementwise only)
    return y, u


def cross_bwd_pre(g, x0, u, dx0=None):
    """t = g*x0; dx0 (+)= g*u.  Returns (t, dx0)."""
    t = torch.empty_like(g)
    acc = dx0 is not None
    if dx0 is None:
        dx0 = torch.empty_like(g)
    check(lib.mapx_cross_bwd_pre(ptr(g), ptr(x0), ptr(u), g.numel(), ptr(t), ptr(dx0), int(acc),
                                 stream()))
    return t, dx0


def alias_cols(buf, col0, ncols):
    """A tensor over columns [col0, col0+ncols) of the 2-D buffer `buf` that shares its memory but
    is NOT an autograd view of it: kernels write a layer's output straight into its slot of a
    concatenated buffer, and autograd's view+in-place bookkeeping never sees a relationship."""
    return carry(torch.empty(0, dtype=buf.dtype, device=buf.device).set_(
        buf.untyped_storage(), buf.storage_offset() + col0, (buf.shape[0], ncols), (buf.stride(0), 1)), buf)


def row_sliceable(x):
    """True if x [M,N] can be read in place by the float4 row kernels: unit column stride,
    16-byte aligned rows (a column slice of a wider buffer qualifies: no .contiguous() copy)."""
    return x.dim() == 2 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 \
        and x.stride(0) >= x.shape[1]


def relu_mask_colsum(dy, y, db=None, defer=False):
    """-> (dz = y > 0 ? dy : 0, db = colsum(dz)) in one pass.  dy may be a column slice."""
    if is_bf16(dy):
        if dy.stride(1) != 1:
            dy = dy.contiguous()
        if y.stride(1) != 1:
            y = y.contiguous()
        M, Nn = dy.shape
        dz = torch.empty(M, Nn, dtype=BF16, device=dy.device)
        later = DEFER_COLSUM and DEFER_COLSUM_H and defer and db is not None
        if db is None:
            db = torch.empty(Nn, dtype=torch.float32, device=dy.device)
        ws, chunks = _partials_h(Nn, dy.device, later)
        check(lib.mapx_relu_mask_colsum_bf16(dy.data_ptr(), dy.stride(0), y.data_ptr(), y.stride(0), M, Nn, ptr(dz),
                                             None if later else ptr(db), ptr(ws), ws.numel(), stream()))
        if later:
            defer_sum(db, ws.view(torch.float32), Nn, chunks, Nn)
        return dz, db
    if not row_sliceable(dy):
        dy = dy.contiguous()
    if not row_sliceable(y):
        y = y.contiguous()
    M, Nn = dy.shape
    dz = torch.empty(M, Nn, dtype=torch.float32, device=dy.device)
    defer = DEFER_COLSUM and defer and db is not None
    if db is None:
        db = torch.empty(Nn, dtype=torch.float32, device=dy.device)
    ws = _partials(Nn, dy.device, defer)
    rec = amax_record(dy.device)
    check(lib.mapx_relu_mask_colsum(dy.data_ptr(), dy.stride(0), y.data_ptr(), y.stride(0), M, Nn, ptr(dz),
                                    None if defer else ptr(db), ptr(ws), ws.numel(), ptr(rec), stream()))
    if defer:
        defer_sum(db, ws.view(torch.float32), Nn, COLSUM_CHUNKS, Nn)
    return tag(dz, rec), db


def cross_bwd_pre_colsum(g, x0, u, dx0=None, db=None, defer=False, plus_g=False):
    """-> (t = g*x0, dx0 (+)= g*u (+ g if plus_g), db = colsum(t)) in one pass.  g may be a column slice.
    bf16 mode: g, x0, u, t bf16; the running dx0 and db fp32."""
    if is_bf16(g):
        if g.stride(1) != 1:
            g = g.contiguous()
        M, Nn = g.shape
        t = torch.empty(M, Nn, dtype=BF16, device=g.device)
        acc = dx0 is not None
        if dx0 is None:
            dx0 = torch.empty(M, Nn, dtype=torch.float32, device=g.device)
        later = DEFER_COLSUM and DEFER_COLSUM_H and defer and db is not None
        if db is None:
            db = torch.empty(Nn, dtype=torch.float32, device=g.device)
        ws, chunks = _partials_h(Nn, g.device, later)
        check(lib.mapx_cross_bwd_pre_colsum_bf16(g.data_ptr(), g.stride(0), ptr(x0), ptr(u), M, Nn, ptr(t), ptr(dx0),
                                                 int(acc) | (2 if plus_g else 0), None if later else ptr(db), ptr(ws),
                                                 ws.numel(), stream()))
        if later:
            defer_sum(db, ws.view(torch.float32), Nn, chunks, Nn)
        return t, dx0, db
    if not row_sliceable(g):
        g = g.contiguous()
    M, Nn = g.shape
    t = torch.empty(M, Nn, dtype=torch.float32, device=g.device)
    acc = dx0 is not None
    if dx0 is None:
        dx0 = torch.empty(M, Nn, dtype=torch.float32, device=g.device)
    defer = DEFER_COLSUM and defer and db is not None
    if db is None:
        db = torch.empty(Nn, dtype=torch.float32, device=g.device)
    ws = _partials(Nn, g.device, defer)
    rec = amax_record(g.device)
    check(lib.mapx_cross_bwd_pre_colsum(g.data_ptr(), g.stride(0), ptr(x0), ptr(u), M, Nn, ptr(t), ptr(dx0),
                                        int(acc) | (2 if plus_g else 0),
                                        None if defer else ptr(db), ptr(ws), ws.numel(), ptr(rec), stream()))
    if defer:
        defer_sum(db, ws.view(torch.float32), Nn, COLSUM_CHUNKS, Nn)
    return tag(t, rec), dx0, db


JOIN_FUSE = os.environ.get("MAPX_JOIN_FUSE", "1") == "1"      # the heads' dX as one product per tower with fused epilogues
# cross dX GEMMs do the next layer's elementwise backward: off by default — inside the tuned schedule (tools/flag_sweep.py,
# round 3) the fused form is 1.1 % slower (0.8314 vs 0.8223 ms per step) although it saves three launches
CROSS_FUSE = os.environ.get("MAPX_CROSS_FUSE", "0") == "1"
DW_BATCH = os.environ.get("MAPX_DW_BATCH", "1") == "1"        # the cross layers' weight gradients from one launch


def gemm_bwd_fused(dy, w, c0, add=None, mask=None, x0=None, u=None, dx0=None, plus_v=False, out=None):
    """v = dy W (+ add) with the elementwise backward that follows it done in the GEMM's epilogue
    (include/mapx_hip.h: mapx_gemm_f32_bwd_fused): columns >= c0 masked by `mask` > 0 (the ReLU layer whose output
    `mask` is), columns < c0 the cross layer's  t = v x0,  dx0 (+)= v u (+ v).  fp32.
    -> (C [M,N], t [M,c0] | None, dx0 [M,c0] | None, part [ceil(M/64), N]: partial rows of the bias gradients)."""
    require_gpu(dy, w)
    M, K = dy.shape
    Nn = w.shape[1]
    dev = dy.device
    C = out if out is not None else torch.empty(M, Nn, dtype=torch.float32, device=dev)
    t = torch.empty(M, c0, dtype=torch.float32, device=dev) if c0 > 0 else None
    accumulate = dx0 is not None
    if c0 > 0 and dx0 is None:
        dx0 = torch.empty(M, c0, dtype=torch.float32, device=dev)
    part = torch.empty(part_rows(M), Nn, dtype=torch.float32, device=dev)
    sd = lambda x: (x.data_ptr(), x.stride(0)) if x is not None else (None, 0)
    # records: C's ReLU-masked columns (>= c0: the deep tower's dZ) and t, the operands of the products that follow
    rec_c = amax_record(dev) if c0 < Nn else None
    rec_t = amax_record(dev) if c0 > 0 else None
    ra, rb, pl = amax_of(dy), amax_of(w), weight_planes(w, False, M)
    if AUTO_AMAX and H2:
        ra = ra if ra is not None else amax(dy)
        rb = rb if rb is not None else amax(w)
        if pl is None and H2W and planes_wanted(M, Nn, K) and w.stride(1) == 1:
            pl = h2_weight_planes(w, False, rb)
    sc = _scale_arg(ra, rb, rec_c, rec_t, b_planes=pl)
    with _timed("gemm_dx_nn", 2.0 * M * Nn * K):
        check(lib.mapx_gemm_f32_bwd_fused(M, Nn, K, dy.data_ptr(), dy.stride(0), w.data_ptr(), w.stride(0),
                                          C.data_ptr(), C.stride(0), *sd(add), *sd(mask), c0, *sd(x0), *sd(u), *sd(t),
                                          *sd(dx0), int(accumulate), int(plus_v), part.data_ptr(), part.stride(0),
                                          None if sc is None else native_byref(sc), stream()))
    if c0 == 0:
        tag(C, rec_c)            # (0 < c0 < N: C mixes g and dZ — callers slice it and tag the slices)
    C._amax_dz = rec_c
    return C, tag(t, rec_t), dx0, part


def skinny_join_bwd(dz, w, final, D, x0, u, plus_v):
    """The input gradient of a head of at most 8 outputs over DCNv2's towers, both towers' first backward step included
    (csrc/skinny.hip; the wide heads' form: gemm_bwd_fused + linear_bwd_input(relu_of=, colsum_to=)).
    -> (g, t, dx0 [M, D], dzr [M, H], part_cross [tiles, D], part_deep [tiles, H])."""
    require_gpu(dz, w, final, x0, u)
    M, Nn = dz.shape
    H = final.shape[1] - D
    f32 = dict(dtype=torch.float32, device=dz.device)
    g, t, dx0 = torch.empty(M, D, **f32), torch.empty(M, D, **f32), torch.empty(M, D, **f32)
    dzr = torch.empty(M, H, **f32)
    tiles = (M + 127) // 128
    pc, pd = torch.empty(tiles, D, **f32), torch.empty(tiles, H, **f32)
    rec_t, rec_z = amax_record(dz.device), amax_record(dz.device)
    with _timed("skinny_linear", 4.0 * (3.0 * M * (D + H) + 3.0 * M * D)):
        check(lib.mapx_skinny_join_bwd(ptr(dz), dz.stride(0), ptr(w), w.stride(0), M, Nn, D, H, final.data_ptr(),
                                       final.stride(0), ptr(x0), x0.stride(0), ptr(u), u.stride(0), int(plus_v), ptr(g),
                                       D, ptr(t), D, ptr(dx0), D, ptr(dzr), H, ptr(pc), ptr(pd), ptr(rec_t), ptr(rec_z),
                                       stream()))
    return g, tag(t, rec_t), dx0, tag(dzr, rec_z), pc, pd


def defer_part_rows(dst, part, col0, ncols):
    """Queue dst[0:ncols] = column sums of part[:, col0 : col0 + ncols] (partial rows of a fused epilogue)."""
    defer_sum(dst, part.view(-1)[col0:], part.stride(0), part.shape[0], ncols)


def linear_bwd_weight_batched(dys, xs, outs):
    """dW_z = dy_z^T x_z for up to four equal-shaped layers in ONE launch (+ one slab sum): dy_z [B,N],
    x_z [B,K] -> outs[z] [N,K] (dense fp32, written in place).  The cross layers' weight gradients."""
    import ctypes as C
    cnt = len(dys)
    Bn, Nn = dys[0].shape
    K = xs[0].shape[1]
    for d, x, o in zip(dys, xs, outs):
        require_gpu(d, x, o)
        if d.shape != (Bn, Nn) or x.shape != (Bn, K) or o.shape != (Nn, K) or not (d.is_contiguous() and x.is_contiguous()
                                                                                   and o.is_contiguous()):
            raise ValueError("linear_bwd_weight_batched: equal shapes, contiguous operands")
    tiles = math.ceil(Nn / 128) * math.ceil(K / 128) * cnt
    ns = 1
    while ns < 16 and tiles * ns * 2 <= 288 and Bn // (ns * 2) >= 256:
        ns *= 2
    ws = scratch(cnt * lib.mapx_gemm_splitk_workspace_bytes(Nn, K, ns), dys[0].device) if ns > 1 else None
    PP = C.c_void_p * cnt
    scales = None
    if AUTO_AMAX and H2:
        for t_ in list(dys) + list(xs):
            if amax_of(t_) is None:
                tag(t_, amax(t_))
    if all(amax_of(d) is not None and amax_of(x) is not None for d, x in zip(dys, xs)):
        scales = (N.GemmScale * cnt)()
        for z, (d, x) in enumerate(zip(dys, xs)):
            scales[z].amax_a, scales[z].amax_b = amax_of(d).data_ptr(), amax_of(x).data_ptr()
    with _timed("gemm_dw_tn", 2.0 * cnt * Nn * K * Bn):
        check(lib.mapx_gemm_f32_batched(cnt, 0, 0, Nn, K, Bn, PP(*[d.data_ptr() for d in dys]), Nn,
                                        PP(*[x.data_ptr() for x in xs]), K, PP(*[o.data_ptr() for o in outs]), ns,
                                        ws.data_ptr() if ws is not None else None,
                                        ws.numel() if ws is not None else 0, scales, stream()))
    return outs


def relu_mask(dy, y):
    out = torch.empty_like(dy)
    fn = lib.mapx_relu_mask_bf16 if is_bf16(dy) else lib.mapx_relu_mask
    check(fn(ptr(dy), ptr(y), dy.numel(), ptr(out), stream()))
    return out


# --------------------------------------------------------------------------- dropout / LayerNorm options
ACT_KINDS = {"tanh": 1, "sigmoid": 2, "none": 3, "elu": 4, "leu": 5, "gelu": 6, "gelu_new": 7, "swish": 8, "mish": 9}


def act_fwd(kind, z, out=None):
    """y = f(z) for the reference's activations other than relu (layers.py:55-80); z [M,N] contiguous fp32,
    `out`: optional destination with unit column stride (a column slice of a wider buffer)."""
    require_gpu(z)
    if z.dtype != torch.float32 or z.dim() != 2 or not z.is_contiguous():
        raise TypeError("act_fwd: z must be a contiguous fp32 matrix")
    M, Nn = z.shape
    if out is None:
        out = torch.empty_like(z)
    if out.shape != z.shape or out.stride(1) != 1:
        raise ValueError("act_fwd: `out` must have z's shape and unit column stride")
    check(lib.mapx_act_fwd(ACT_KINDS[kind], ptr(z), M, Nn, out.data_ptr(), out.stride(0), stream()))
    return out


def act_bwd(kind, dy, z):
    """dz = dy * f'(z); dy may be a row-strided slice."""
    require_gpu(dy, z)
    if dy.stride(1) != 1:
        dy = dy.contiguous()
    M, Nn = z.shape
    dz = torch.empty_like(z)
    check(lib.mapx_act_bwd(ACT_KINDS[kind], dy.data_ptr(), dy.stride(0), ptr(z), M, Nn, ptr(dz), stream()))
    return dz


def dropout(x, p, seed, offset, offset_dev=None, out=None):
    """out = keep ? x / (1 - p) : 0, keep mask = Philox(seed, offset + *offset_dev) (never stored: the
    backward pass calls this again on the gradient with the same seed / offset).  fp32."""
    require_gpu(x)
    if x.dtype != torch.float32:
        raise NotImplementedError("dropout is built for the fp32 path")
    x = x.contiguous()
    if out is None:
        out = torch.empty_like(x)
    if not out.is_contiguous():
        raise ValueError("dropout writes a contiguous tensor")
    check(lib.mapx_dropout(ptr(x), x.numel(), float(p), int(seed), int(offset), ptr(offset_dev), ptr(out), stream()))
    return out


def layernorm_fwd(x2, w, b, eps):
    """x2 [R,E] -> (y [R,E], stats [R,2] = {mean, rstd})   (nn.LayerNorm over the last dimension)."""
    require_gpu(x2, w, b)
    x2 = x2.contiguous()
    R, E = x2.shape
    y = torch.empty_like(x2)
    stats = torch.empty(R, 2, dtype=torch.float32, device=x2.device)
    check(lib.mapx_layernorm_fwd(ptr(x2), R, E, ptr(w), ptr(b), float(eps), ptr(y), ptr(stats), stream()))
    return y, stats


def layernorm_bwd(dy2, x2, w, stats):
    """-> (dx [R,E], dy * xhat [R,E]: its column sums are dL/dw; dL/db = column sums of dy)."""
    require_gpu(dy2, x2, w, stats)
    dy2 = dy2.contiguous()
    R, E = x2.shape
    dx, dyx = torch.empty_like(x2), torch.empty_like(x2)
    check(lib.mapx_layernorm_bwd(ptr(dy2), ptr(x2), ptr(w), ptr(stats), R, E, ptr(dx), ptr(dyx), stream()))
    return dx, dyx


# --------------------------------------------------------------------------- heads / masks
def bce_with_logits(logits, labels, want_grad=True):
    """-> (out3 = [loss, acc, pos_ratio] f32 device, dlogits or None)."""
    require_gpu(logits, labels)
    logits, labels = logits.contiguous(), labels.contiguous()
    n = logits.numel()
    out3 = torch.empty(3, dtype=torch.float32, device=logits.device)
    dl = torch.empty_like(logits) if want_grad else None
    ws = scratch(lib.mapx_bce_workspace_bytes(), logits.device)
    check(lib.mapx_bce_with_logits(ptr(logits), ptr(labels), n, ptr(dl), ptr(out3), ptr(ws),
                                   ws.numel(), stream()))
    return out3, dl


def eval_metrics(logits, labels):
    """ROC-AUC / log-loss / mean logit / mean probability of an eval pass, computed on the device
    (reference trainer.py:189-199 uses sklearn on host lists).  One host read of 6 doubles.
    -> dict(auc, logloss, avg_logits, avg_probs, positives, negatives)"""
    require_gpu(logits, labels)
    logits = logits.contiguous().view(-1).float()
    labels = labels.contiguous().view(-1).float()
    n = logits.numel()
    if labels.numel() != n:
        raise ValueError(f"eval_metrics: {n} logits but {labels.numel()} labels")
    out = torch.empty(6, dtype=torch.float64, device=logits.device)
    ws = scratch(lib.mapx_eval_metrics_workspace_bytes(n), logits.device)
    check(lib.mapx_eval_metrics(ptr(logits), ptr(labels), n, ptr(out), ptr(ws), ws.numel(), stream()))
    auc, ll, ml, mp, npos, nneg = out.tolist()
    if npos == 0 or nneg == 0:                     # sklearn.metrics.roc_auc_score's behaviour
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    return dict(auc=auc, logloss=ll, avg_logits=ml, avg_probs=mp, positives=int(npos), negatives=int(nneg))


def dynamic_mask_mfp(ids, L, masked_index=None, seed=0, offset=0, offset_dev=None, sel=None, sel_cursor=None,
                     batch=None):
    """-> (masked ids [B,F], labels [B,L], masked_index [B,L])  (trainer.py:217-232).
    `sel` int64 [B]: `ids` is the whole HBM-resident split [N,F] and the batch is its rows sel.
    `sel_cursor` (device int64 scalar) + `batch`: sel is a whole epoch's permutation and the batch is
    sel[*cursor : *cursor + batch] (a captured step walks the epoch without host copies)."""
    require_gpu(ids)
    ids = ids.contiguous()
    if sel is not None:
        require_gpu(sel)
        sel = sel.contiguous()
        B, F = (sel.numel() if sel_cursor is None else int(batch)), ids.shape[1]
        out = torch.empty(B, F, dtype=torch.int64, device=ids.device)
        labels = torch.empty(B, L, dtype=torch.int64, device=ids.device)
        mi_out = torch.empty(B, L, dtype=torch.int64, device=ids.device)
        mi_in = masked_index.contiguous() if masked_index is not None else None
        keys = torch.empty(B * F, dtype=torch.int32, device=ids.device)
        check(lib.mapx_dynamic_mask_mfp_rows(ptr(ids), ids.shape[0], ptr(sel), sel.numel(), ptr(sel_cursor), B, F, L, ptr(mi_in), seed, offset,
                                             ptr(offset_dev), ptr(out), ptr(labels), ptr(mi_out), ptr(keys), stream()))
        _keys_of[0], _keys_of[1] = out, keys
        return out, labels, mi_out
    B, F = ids.shape
    out = torch.empty_like(ids)
    labels = torch.empty(B, L, dtype=torch.int64, device=ids.device)
    mi_out = torch.empty(B, L, dtype=torch.int64, device=ids.device)
    mi_in = masked_index.contiguous() if masked_index is not None else None
    keys = torch.empty(B * F, dtype=torch.int32, device=ids.device)
    check(lib.mapx_dynamic_mask_mfp(ptr(ids), B, F, L, ptr(mi_in), seed, offset, ptr(offset_dev), ptr(out),
                                    ptr(labels), ptr(mi_out), ptr(keys), stream()))
    _keys_of[0], _keys_of[1] = out, keys        # Embeddings.forward asks ids_to_i32 for exactly this matrix next
    return out, labels, mi_out


RFD_MODES = {"Unigram": 0, "Uniform": 1, "Whole-Uniform": 2, "Whole-Unigram": 3}


def dynamic_mask_rfd(ids, L, masked_index=None, replace_feat=None, x_train=None, seed=0, offset=0,
                     offset_dev=None, mode="Unigram", idx_low=None, idx_high=None, vocab=0):
    """-> (replaced ids [B,F], labels f32 [B,F], masked_index [B,L])  (trainer.py:233-240)."""
    require_gpu(ids)
    ids = ids.contiguous()
    B, F = ids.shape
    out = torch.empty_like(ids)
    labels = torch.empty(B, F, dtype=torch.float32, device=ids.device)
    mi_out = torch.empty(B, L, dtype=torch.int64, device=ids.device)
    mi_in = masked_index.contiguous() if masked_index is not None else None
    rep = replace_feat.contiguous() if replace_feat is not None else None
    nrows = x_train.shape[0] if x_train is not None else 0
    if mode not in RFD_MODES:
        raise NotImplementedError(mode)                     # trainer.py:261-262
    check(lib.mapx_dynamic_mask_rfd(ptr(ids), B, F, L, ptr(mi_in), ptr(rep), ptr(x_train), nrows,
                                    RFD_MODES[mode], ptr(idx_low), ptr(idx_high), int(vocab), seed, offset,
                                    ptr(offset_dev), ptr(out), ptr(labels), ptr(mi_out), stream()))
    return out, labels, mi_out


# --------------------------------------------------------------------------- optimizer
def make_sched(lr0, lambdas, beta1, beta2):
    """Host (float64) -> device table [T,2] f32 of {step_size_s, lr_s}, s = 1..T."""
    rows = []
    for i, lam in enumerate(lambdas):
        s = i + 1
        lr = lr0 * lam
        rows.append((lr * math.sqrt(1.0 - beta2 ** s) / (1.0 - beta1 ** s), lr))
    return torch.tensor(rows, dtype=torch.float64).to(torch.float32)


REPLAY_TERMS = 7          # kJ + 1 in csrc/optim.hip
CLOSED_REPLAY = os.environ.get("MAPX_CLOSED_REPLAY", "1") == "1"     # 0: step-by-step replay + closed-form tail


def make_replay_aux(lr0, lambdas, beta1, beta2, wd):
    """Host fp64 tables [17, T+1] of the closed-form replay of zero-gradient AdamW steps
    (csrc/optim.hip: replay_coef): prefix products prod_{i<s}(1 - lr_i*wd), beta1^n, beta2^n, and
    R_i[s] = a_s + q_i / (1 - d_s) * R_i[s+1] for q_i = beta1 * beta2^(-(i+1)/2), i = 0..6, once with
    d_s = lr_s*wd and once with d_s = 0.  a_s and lr_s are the fp32-rounded values the kernels read
    from the schedule table.  MAPX_CLOSED_REPLAY=0: the first three rows only."""
    T = len(lambdas)
    sched = make_sched(lr0, lambdas, beta1, beta2).to(torch.float64)        # fp32 values, as doubles
    a, lr = sched[:, 0].tolist(), sched[:, 1].tolist()
    lr_exact = torch.tensor([lr0 * l for l in lambdas], dtype=torch.float64)
    cum = torch.ones(T + 1, dtype=torch.float64)
    cum[1:] = torch.cumprod(1.0 - lr_exact * wd, 0)
    n = torch.arange(T + 1, dtype=torch.float64)
    rows = [cum, torch.pow(torch.tensor(beta1, dtype=torch.float64), n),
            torch.pow(torch.tensor(beta2, dtype=torch.float64), n)]
    if CLOSED_REPLAY:
        beta = math.sqrt(beta2)
        cum_l = cum.tolist()
        for decay in (True, False):
            for i in range(REPLAY_TERMS):
                q = (beta1 / beta) * beta ** (-i)
                R = [0.0] * (T + 1)
                for s in range(T - 1, -1, -1):
                    keep = (cum_l[s + 1] / cum_l[s]) if decay else 1.0     # 1 - d_s, consistent with row 0
                    R[s] = a[s] + q / keep * R[s + 1]
                rows.append(torch.tensor(R, dtype=torch.float64))
    return torch.stack(rows).contiguous()


def adamw_dense(p, g, m, v, sched, done, beta1, beta2, eps, wd, shadow=None, seg_off=None, seg_amax=None):
    """`shadow`: bf16 [n] copy of the updated parameters, written by the same kernel (bf16 mode).
    seg_off (int64 [n_params + 1]) / seg_amax (int32 [n_params, 2]): the parameters' magnitude records, raised
    with what this launch writes (fp32 mode: the weights are the next step's GEMM operands)."""
    require_gpu(p, g, m, v, sched, done)
    with _timed("adamw_dense", p.numel() * (28.0 if shadow is None else 30.0)):
        if shadow is None:
            nseg = seg_off.numel() - 1 if (seg_off is not None and seg_amax is not None) else 0
            check(lib.mapx_adamw_dense(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(sched), sched.shape[0],
                                       ptr(done), beta1, beta2, eps, wd, ptr(seg_off) if nseg else None, nseg,
                                       ptr(seg_amax) if nseg else None, stream()))
        else:
            check(lib.mapx_adamw_dense_shadow(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(sched),
                                              sched.shape[0], ptr(done), beta1, beta2, eps, wd, ptr(shadow),
                                              stream()))


def take_rows(src, sel, cursor=None, batch=None, as_f32=False):
    """src[sel[c : c + batch]] for an int64 matrix / vector resident on the device (c = the device scalar `cursor`,
    0 and batch = len(sel) without one): the batch of an RFD / finetune step, cut inside the step.
    as_f32: the values as float32 (the finetune labels: no labels.float() launch behind it)."""
    require_gpu(src, sel)
    if src.dtype != torch.int64 or sel.dtype != torch.int64 or not src.is_contiguous() or not sel.is_contiguous():
        raise TypeError("take_rows: contiguous int64 source and row numbers")
    if cursor is not None and (cursor.dtype != torch.int64 or cursor.numel() != 1 or batch is None):
        raise TypeError("take_rows: the cursor is an int64 device scalar and comes with the batch size")
    B = int(batch) if batch is not None else sel.numel()
    if cursor is None and B > sel.numel():
        raise IndexError(f"take_rows: {B} rows of a list of {sel.numel()}")
    F = 1 if src.dim() == 1 else src.shape[1]
    out = torch.empty((B,) if src.dim() == 1 else (B, F), dtype=torch.float32 if as_f32 else torch.int64,
                      device=src.device)
    if B == 0:
        return out
    check(lib.mapx_take_rows_i64(ptr(src), src.shape[0], F, ptr(sel), sel.numel(), ptr(cursor), B, None if as_f32 else ptr(out),
                                 ptr(out) if as_f32 else None, stream()))
    return out


def step_advance(done, cursor=None, stride=0):
    """*done += 1; `cursor` (int64 device scalar): += stride in the same launch (the batch cursor of a step
    that walks the epoch's permutation, trainer.GraphedStep)."""
    if cursor is not None and (cursor.dtype != torch.int64 or not cursor.is_cuda):
        raise TypeError("step_advance: the cursor is an int64 device scalar")
    check(lib.mapx_step_advance(ptr(done), ptr(cursor), int(stride), stream()))


def replay_coef_table(aux, beta1, beta2, done, coef=None):
    """The closed-form replay's coefficients for every gap ending at *done (include/mapx_hip.h:
    mapx_replay_coef_table) -> coef [2, aux_len, 12] f32 (allocated once by the caller, rewritten every step)."""
    require_gpu(aux, done)
    if coef is None:
        coef = torch.zeros(lib.mapx_replay_coef_table_bytes(aux.shape[1]) // 4, dtype=torch.float32, device=aux.device)
    with _timed("replay_coef_table", 0.0):
        check(lib.mapx_replay_coef_table(ptr(aux), aux.shape[1], aux.shape[0], beta1, beta2, ptr(done), ptr(coef),
                                         stream()))
    return coef


def table_adam(p0, m0, v0, wd0, last, sched, done, aux, beta1, beta2, eps, p1=None, m1=None, v1=None,
               wd1=0.0, rows=None, n_rows_dev=None, row_begin=0, n_rows=None, grad0=None, grad1=None,
               rows_may_repeat=False):
    require_gpu(p0, m0, v0, last, sched, done)
    W0 = p0.shape[1]
    if n_rows is None:
        n_rows = rows.numel() if rows is not None else p0.shape[0] - row_begin
    kind = "table_adam_update" if grad0 is not None else ("table_adam_catchup" if rows is not None
                                                           else "table_adam_sweep")
    # m0 / v0 (and m1 / v1) are dense arrays of their own, or the two halves of one record per row (row stride 2 W0 / 2)
    def _rows(t, width):
        if t.dim() == 1:
            return t.data_ptr(), t.stride(0)
        if t.stride(1) != 1 or t.shape[1] != width:
            raise ValueError("table_adam: moment rows must be unit-stride rows of the table's width")
        return t.data_ptr(), t.stride(0)
    (pm0, ld0), (pv0, ldv0) = _rows(m0, W0), _rows(v0, W0)
    if ld0 != ldv0:
        raise ValueError("table_adam: m0 and v0 must share their row stride")
    pm1 = pv1 = None
    ld1 = 1
    if p1 is not None:
        (pm1, ld1), (pv1, ldv1) = _rows(m1, 1), _rows(v1, 1)
        if ld1 != ldv1:
            raise ValueError("table_adam: m1 and v1 must share their stride")
    with _timed(kind, float(n_rows) * (W0 + (1 if p1 is not None else 0)) * 4.0 * 7):
        check(lib.mapx_table_adam(ptr(p0), pm0, pv0, ld0, W0, wd0, ptr(p1), pm1, pv1, ld1, wd1,
                                  ptr(last), ptr(rows), row_begin, n_rows, ptr(n_rows_dev), ptr(grad0),
                                  ptr(grad1), ptr(sched), sched.shape[0], ptr(done), ptr(aux),
                                  aux.shape[1], aux.shape[0], beta1, beta2, eps, int(rows_may_repeat), stream()))
