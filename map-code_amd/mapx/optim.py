"""Optimizer of the hot path: transformers-4.26 AdamW + cosine/const schedule, as the
reference's Trainer.get_optimizer builds it (code/trainer.py:60-85), on device.

* dense parameters live in two flat fp32 buffers (decay / no-decay group, split by the
  reference's NAME rule) and take one fused kernel per group per step;
* the [V,*] tables take row-sparse updates with exact lazy replay of the zero-gradient
  updates the reference applies to untouched rows (csrc/optim.hip);
* {step size, lr} per update come from a device table and the update counter is a device
  int, so nothing is synchronised with the host during training.
"""
import math
import os

import torch

from . import ops

NO_DECAY = ("bias", "LayerNorm.weight")          # trainer.py:61
DENSE_BEFORE_JOIN = os.environ.get("MAPX_DENSE_BEFORE_JOIN", "0") == "1"     # A/B switch: MapxOptimizer.step
PACK_MOMENTS = os.environ.get("MAPX_PACK_MOMENTS", "1") == "1"      # A/B switch: m | v of a table row in one record


def decays(name):
    return not any(nd in name for nd in NO_DECAY)


def lr_lambda(kind, step, total, warmup):
    """get_{cosine,constant}_schedule_with_warmup multiplier at scheduler step `step`."""
    kind = kind.lower()
    if kind not in ("cosine", "const"):
        raise NotImplementedError(kind)                       # trainer.py:82-83
    if step < warmup:
        return float(step) / float(max(1, warmup))
    if kind == "const":
        return 1.0
    progress = float(step - warmup) / float(max(1, total - warmup))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))


class TableAdam:
    """Lazy exact AdamW state of one RowTable."""

    def __init__(self, table, wd0, wd1, hyper, sched, done, aux, max_gap):
        self.table, self.wd0, self.wd1, self.aux = table, wd0, wd1, aux
        self.b1, self.b2, self.eps = hyper
        self.sched, self.done = sched, done
        p0, p1 = table.p0.data, (table.p1.data if table.p1 is not None else None)
        # a row's two moments side by side in ONE record [V, 2 W] (m | v): the update reads and writes two random
        # places per row (the parameter row and this record) instead of three — random rows cost per access, not
        # per byte (csrc/optim.hip TableGroup).  m0 / v0 (m1 / v1) are views of the halves.
        W = p0.shape[1]
        self.mv0 = torch.zeros(p0.shape[0], 2 * W, dtype=p0.dtype, device=p0.device) if PACK_MOMENTS else None
        if PACK_MOMENTS:
            self.m0, self.v0 = self.mv0[:, :W], self.mv0[:, W:]
        else:
            self.m0, self.v0 = torch.zeros_like(p0), torch.zeros_like(p0)
        self.m1 = self.v1 = self.mv1 = None
        if p1 is not None and PACK_MOMENTS:
            self.mv1 = torch.zeros(p1.shape[0], 2, device=p1.device)
            self.m1, self.v1 = self.mv1[:, 0], self.mv1[:, 1]
        elif p1 is not None:
            self.m1, self.v1 = torch.zeros(p1.shape[0], device=p1.device), torch.zeros(p1.shape[0], device=p1.device)
        self.last = torch.zeros(p0.shape[0], dtype=torch.int32, device=p0.device)
        self.stale = False
        self.cursor = 0
        # max_gap > 0: also bring V/max_gap rows up to date every step (round robin).  Not needed
        # for speed since replays have a closed-form tail (csrc/optim.hip); kept for testing.
        self.sweep = math.ceil(p0.shape[0] / max_gap) if max_gap > 0 else 0
        table.lazy = self

    def _call(self, **kw):
        t = self.table
        ops.table_adam(t.p0.data, self.m0, self.v0, self.wd0, self.last, self.sched, self.done,
                       self.aux, self.b1, self.b2, self.eps,
                       p1=t.p1.data.view(-1) if t.p1 is not None else None,
                       m1=self.m1, v1=self.v1, wd1=self.wd1, **kw)

    def catch_up(self, plan):
        self._call(rows=plan.uniq, n_rows=plan.n, n_rows_dev=plan.n_uniq)

    def replay_in_readers(self):
        """The closed form is tabulated (17-row aux): a forward kernel may read rows through their pending updates."""
        return self.aux.shape[0] > 3

    def refresh_coef(self):
        """This step's replay coefficients (ops.replay_coef_table) for the kernels that take lazy_rows()."""
        self.coef = ops.replay_coef_table(self.aux, self.b1, self.b2, self.done, getattr(self, "coef", None))

    def lazy_rows(self):
        """native.LazyRows over this table's state (include/mapx_hip.h: mapx_lazy_rows)."""
        from . import native as N
        if getattr(self, "coef", None) is None:
            self.refresh_coef()
        lz = N.LazyRows()
        lz.m0, lz.v0, lz.ld_mv0, lz.wd0 = self.m0.data_ptr(), self.v0.data_ptr(), self.m0.stride(0), self.wd0
        if self.m1 is not None:
            lz.m1, lz.v1, lz.ld_mv1, lz.wd1 = self.m1.data_ptr(), self.v1.data_ptr(), self.m1.stride(0), self.wd1
        else:
            lz.m1, lz.v1, lz.ld_mv1, lz.wd1 = None, None, 1, 0.0
        lz.last, lz.sched, lz.sched_len, lz.done = self.last.data_ptr(), self.sched.data_ptr(), self.sched.shape[0], self.done.data_ptr()
        lz.aux, lz.aux_len, lz.aux_rows = self.aux.data_ptr(), self.aux.shape[1], self.aux.shape[0]
        lz.beta1, lz.beta2, lz.eps = self.b1, self.b2, self.eps
        lz.coef_opt = self.coef.data_ptr()
        return lz

    def catch_up_raw(self, keys_i32):
        """Catch-up straight from a raw id list (repeats allowed; no sort on the critical path)."""
        self._call(rows=keys_i32, n_rows=keys_i32.numel(), rows_may_repeat=True)

    def update(self):
        sg = self.table.sparse_grad
        if sg is None:
            return
        plan, r0, r1 = sg
        self._call(rows=plan.uniq, n_rows=plan.n, n_rows_dev=plan.n_uniq, grad0=r0, grad1=r1)
        self.table.sparse_grad = None
        self.stale = True

    def sweep_some(self):
        """Bound the replay length: every row is brought up to date at least every max_gap steps."""
        if self.sweep <= 0:
            return
        V = self.table.num_rows
        n = min(self.sweep, V - self.cursor)
        self._call(row_begin=self.cursor, n_rows=n)
        self.cursor = (self.cursor + n) % V

    def flush(self):
        if self.stale:
            self._call(row_begin=0, n_rows=self.table.num_rows)
            self.stale = False


class MapxOptimizer:
    """optimizer + scheduler of reference Trainer.get_optimizer, fused.  `step()` = the
    reference's optimizer.step(); scheduler.step(); model.zero_grad()."""

    def __init__(self, model, args, num_training_steps, num_warmup_steps, max_gap=0):
        b1, b2 = (float(x) for x in args.adam_betas.split(","))
        self.hyper = (b1, b2, float(args.adam_epsilon))
        self.lr0, self.wd = float(args.learning_rate), float(args.weight_decay)
        self.kind, self.total, self.warmup = args.lr_sched.lower(), int(num_training_steps), int(num_warmup_steps)
        self.max_grad_norm = float(getattr(args, "max_grad_norm", 0.0))
        lambdas = [lr_lambda(self.kind, s, self.total, self.warmup) for s in range(max(1, self.total))]
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("MapxOptimizer needs the model on the GPU (call model.to(device) first)")
        self.sched = ops.make_sched(self.lr0, lambdas, b1, b2).to(dev)
        self.aux = ops.make_replay_aux(self.lr0, lambdas, b1, b2, self.wd).to(dev)
        self.done = torch.zeros(1, dtype=torch.int32, device=dev)
        self.steps_done = 0
        table_ids = model.table_parameter_ids()
        named = [(n, p) for n, p in model.named_parameters() if id(p) not in table_ids and p.requires_grad]
        self.dense_params = [p for _, p in named]
        # bf16 compute mode: every dense weight gets a bf16 shadow (`p._mapx_bf16`, the operand of the
        # bf16 GEMMs) inside one flat buffer per group, written by the AdamW kernel with the update
        from .layers import compute_dtype_of
        self.bf16 = compute_dtype_of(getattr(model, "config", None)) == torch.bfloat16
        self.groups = []
        for wd, members in ((self.wd, [(n, p) for n, p in named if decays(n)]),
                            (0.0, [(n, p) for n, p in named if not decays(n)])):
            if members:
                self.groups.append(self._flatten(members, wd, dev, self.bf16))
        self.refresh_bf16()
        names = {id(p): n for n, p in model.named_parameters()}
        self.tables = []
        for t in model.row_tables():
            wd0 = self.wd if decays(names[id(t.p0)]) else 0.0
            wd1 = (self.wd if decays(names[id(t.p1)]) else 0.0) if t.p1 is not None else 0.0
            self.tables.append(TableAdam(t, wd0, wd1, self.hyper, self.sched, self.done, self.aux, max_gap))
        # a table may apply its update as soon as its gradient is final (ops.add_side_task) unless the
        # step needs all gradients first: a global clipping norm, or the gradient exchange of N ranks
        from . import parallel
        # Round 3 (tools/flag_sweep.py, one box each): in fp32, with the table gradients on the tower stream's late
        # tasks, the early row updates LOSE (0.8058 vs 0.7975 ms per step: the updates run in step(), beside the
        # optimizer's dense half); in the bf16 mode, whose GEMM chains are half as long, they WIN (0.5931 vs 0.6318).
        # Round 4, two sweeps: with gemm_h2.hip alone early won in fp32 too (0.7075 vs 0.7175), with the weights' planes
        # (gemm_h2w.hip, the shipped default) it loses again (0.7315 vs 0.7120).  End of round 4 (the deep tower's join
        # product captured first, layers.JOIN_DEEP_FIRST): they lose in the bf16 mode as well — MFP 0.5372 / 0.5417 with
        # them (and the dense half on the main stream) vs 0.5305 / 0.5327 without (dense half beside the tables' on the
        # tower stream, as in fp32), RFD 0.448 / 0.441 vs 0.428 / 0.426, Criteo-shaped equal (profiles/r04_ab_tail.txt).
        early_default = "0"
        early = (os.environ.get("MAPX_EARLY_TABLE_UPDATE", early_default) == "1" and self.max_grad_norm <= 0
                 and not parallel.exchanging())
        for t in self.tables:
            t.early_ok, t.early_now = early, False
        # (int64 device cursor, stride) of a captured step that walks the epoch's permutation: moved to the next
        # batch by the launch that advances the update counter (trainer.GraphedStep sets it around its capture)
        self.walk_cursor = None

    def backward_window(self, open_):
        """Between backward_window(True) and (False) — the Trainer brackets loss.backward() of a step
        whose optimizer.step() follows at once — a table may apply its row update as soon as its
        gradient is final.  Outside the window backward() never touches a parameter."""
        if open_:
            ops.clear_side_tasks()      # leftovers of a backward pass that raised
        ops.step_window[0] = bool(open_)
        for t in self.tables:
            t.early_now = bool(open_) and t.early_ok

    FLAT_PAD = 8          # elements every parameter's slot of a flat buffer is rounded up to (saved with the state)

    @staticmethod
    def _flatten(members, wd, dev, bf16=False):
        pad = MapxOptimizer.FLAT_PAD
        sizes = [(p.numel() + pad - 1) // pad * pad for _, p in members]   # every view 16-B aligned (fp32 and bf16)
        total = sum(sizes)
        flat_p = torch.zeros(total, device=dev)
        flat_g = torch.zeros(total, device=dev)
        flat_h = torch.zeros(total, dtype=torch.bfloat16, device=dev) if bf16 else None
        # fp32 mode: every parameter's magnitude record (ops: "Magnitude records"), kept current by the AdamW kernel
        # with what it writes — `p._amax` is what the GEMMs that read p as an operand look at
        recs = None if (bf16 or not ops.H2) else torch.zeros(len(members), ops.REC, dtype=torch.int32, device=dev)
        offs = [0]
        for sz in sizes:
            offs.append(offs[-1] + sz)
        off = 0
        for i, ((_, p), sz) in enumerate(zip(members, sizes)):
            if recs is not None:
                p._amax = recs[i]
                if p.dim() == 2:
                    p._planes = {}           # ops.weight_planes: filled at first use, refreshed behind every update
            view = flat_p[off:off + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            p.grad = None
            p._mapx_grad = flat_g[off:off + p.numel()].view_as(p)   # backward kernels write here
            p._mapx_bf16 = flat_h[off:off + p.numel()].view_as(p) if bf16 else None
            off += sz
        return dict(p=flat_p, g=flat_g, m=torch.zeros_like(flat_p), v=torch.zeros_like(flat_p), wd=wd,
                    names=[n for n, _ in members], numels=[p.numel() for _, p in members], h=flat_h,
                    amax=recs, seg_off=torch.tensor(offs, dtype=torch.int64, device=dev) if recs is not None else None,
                    params=[p for _, p in members])

    def refresh_bf16(self):
        """Re-derive what the optimizer keeps beside the fp32 master weights — the bf16 shadows (bf16 mode), the
        magnitude records (fp32 mode) — after anything other than step() wrote the parameters (construction,
        load_model / load_state_dict, a test poking .data)."""
        for g in self.groups:
            if g.get("amax") is not None:
                for p, rec in zip(g["params"], g["amax"]):
                    ops.amax(p.data.reshape(1, -1) if p.dim() != 2 else p.data, rec=rec, reset=True)
                    p._amax_ver = p._version
                    ops.refresh_weight_planes(p)
        if not self.bf16:
            return
        for g in self.groups:
            ops.check(ops.lib.mapx_cast_f32_bf16(ops.ptr(g["p"]), g["p"].numel(), ops.ptr(g["h"]), ops.stream()))

    # ------------------------------------------------------------------
    def clip_grad_norm_(self):
        sq = sum((g["g"] ** 2).sum() for g in self.groups)
        for t in self.tables:
            if t.table.sparse_grad is not None:
                plan, r0, r1 = t.table.sparse_grad
                if plan.n_uniq is None:          # gathered list (mapx.parallel): padding rows are zero
                    live = torch.ones(r0.shape[0], device=r0.device)
                else:
                    live = (torch.arange(r0.shape[0], device=r0.device) < plan.n_uniq[0]).float()
                sq = sq + ((r0 ** 2).sum(1) * live).sum()
                if r1 is not None:
                    sq = sq + ((r1 ** 2) * live).sum()
        coef = (self.max_grad_norm / (sq.sqrt() + 1e-6)).clamp(max=1.0)
        for g in self.groups:
            g["g"].mul_(coef)
        for t in self.tables:
            if t.table.sparse_grad is not None:
                _, r0, r1 = t.table.sparse_grad
                r0.mul_(coef)
                if r1 is not None:
                    r1.mul_(coef)

    def collect_torch_grads(self):
        """Parameters that a model uses through plain torch ops (e.g. DeepFM's one-element LR bias)
        get their gradient in `.grad`, not in the flat buffer the fused kernels write: move it."""
        moved = 0
        for p in self.dense_params:
            if p.grad is not None:
                p._mapx_grad.copy_(p.grad)
                p.grad = None
                moved += 1
        return moved

    def step(self):
        ops.run_main_tasks()            # stream joins nobody picked up
        main = torch.cuda.current_stream() if torch.cuda.is_available() else None
        ev, ev_stream = ops.dense_ready
        late_ev, late_stream = ops.late_dense_done
        ops.late_dense_done[0] = ops.late_dense_done[1] = None
        from . import parallel
        # The dense half first, when it runs on this stream (no fork: the bf16 trunk, ops.tail_overlap) and the side
        # stream's dense gradients are marked by an event: it then waits for THAT event, not for the tables' gradient
        # and row update that follow on the side stream (28 us of idle main queue in front of sum_tasks on
        # profiles/r04_step_timeline_bf16.txt); everything else is joined behind it.
        dense_first = (DENSE_BEFORE_JOIN and ev is None and late_ev is not None and main is not None
                       and self.max_grad_norm <= 0 and not parallel.exchanging() and not ops._side_tasks
                       and not ops._late_tasks)
        if dense_first:
            ops.stream_wait_event(main, late_ev, late_stream)
            if self.collect_torch_grads() == 0:
                ops.flush_deferred()
                self._dense_update()
            else:
                dense_first = False        # (a torch-op gradient arrived in .grad: the plain order below)
        ops.join_pending()              # side work a backward node forked and left open
        ops.run_side_tasks()            # early table updates nobody picked up
        ops.run_late_tasks()            # optimizer-only gradients nobody picked up
        moved = self.collect_torch_grads()
        ops.dense_ready[0] = ops.dense_ready[1] = None
        # (the copies collect_torch_grads enqueued on the main stream come AFTER the dense-ready event: a side
        # stream that waits for that event alone could read the flat gradient before they land)
        if ev is not None and moved == 0 and self.max_grad_norm <= 0 and ev_stream == main \
                and not parallel.exchanging():
            # The dense half (partial sums + dense AdamW, HBM-bound, ~30 us) beside the tables' half (the
            # embedding gradient's reduction + row updates, ~30 us) instead of behind it: it forks from the
            # point where the dense gradients were final, on the tower stream, which is idle by then.
            side = ops.aux_stream("tower", self.done.device)
            forked = ops.stream_wait_event(side, ev, main)
            with torch.cuda.stream(side):
                ops.flush_deferred()
                self._dense_update()
            for t in self.tables:
                t.update()
            if forked:
                ops.stream_wait(main, side)
        elif dense_first:
            for t in self.tables:
                t.update()
        else:
            ops.flush_deferred()            # split-K slabs / colsum partials of this backward pass
            if self.max_grad_norm > 0:
                self.clip_grad_norm_()
            self._dense_update()
            for t in self.tables:
                t.update()
        ops.step_advance(self.done, *(self.walk_cursor or ()))
        self.steps_done += 1
        for t in self.tables:
            t.sweep_some()
        self.zero_grad()

    def _dense_update(self):
        b1, b2, eps = self.hyper
        for g in self.groups:
            ops.adamw_dense(g["p"], g["g"], g["m"], g["v"], self.sched, self.done, b1, b2, eps, g["wd"],
                            shadow=g["h"], seg_off=g.get("seg_off"), seg_amax=g.get("amax"))
        for g in self.groups:
            if g.get("amax") is not None:
                ops.planes_written(g["params"])              # the weights' fp16 pieces for the next step's products

    def zero_grad(self):
        """Dense gradients are overwritten by the next backward (see layers._grad_slot); only
        the sparse side channel needs clearing."""
        for t in self.tables:
            t.table.sparse_grad = None
        ops.dense_ready[0] = ops.dense_ready[1] = None

    def flush(self):
        """Materialise reference-equivalent table weights (before eval / checkpoint)."""
        for t in self.tables:
            t.flush()

    # ------------------------------------------------------------------ resume state
    def state_dict(self):
        """Everything needed to continue training bit-exactly: dense and table Adam moments, the
        per-row replay clocks, the update counter.  (The reference saves model weights only and
        cannot resume mid-run: trainer.py:517-519.)"""
        return dict(steps_done=self.steps_done, done=self.done.cpu(), flat_pad=self.FLAT_PAD,
                    groups=[dict(names=g["names"], m=g["m"].cpu(), v=g["v"].cpu()) for g in self.groups],
                    tables=[dict(name=t.table.name, m0=t.m0.contiguous().cpu(), v0=t.v0.contiguous().cpu(),
                                 m1=None if t.m1 is None else t.m1.contiguous().cpu(),
                                 v1=None if t.v1 is None else t.v1.contiguous().cpu(), last=t.last.cpu(),
                                 stale=t.stale, cursor=t.cursor) for t in self.tables])

    def load_state_dict(self, sd):
        self.steps_done = int(sd["steps_done"])
        self.done.copy_(sd["done"])
        saved_pad = sd.get("flat_pad")
        for g, s in zip(self.groups, sd["groups"]):
            if g["names"] != s["names"]:
                raise ValueError("optimizer state: the parameter list changed since the state was saved")
            sizes = g["numels"]
            if s["v"].numel() != s["m"].numel():
                raise ValueError("optimizer state: m and v of a group differ in length")
            if s["m"].numel() == g["m"].numel() and saved_pad in (None, self.FLAT_PAD):
                g["m"].copy_(s["m"])
                g["v"].copy_(s["v"])
                continue
            # another slot size: the flat moments do not line up — re-pack them parameter by parameter.  States
            # written before the `flat_pad` key existed used slots of 4, then of 8 elements: take the one that fits.
            fits = [pad for pad in ([int(saved_pad)] if saved_pad is not None else [8, 4])
                    if sum((n + pad - 1) // pad * pad for n in sizes) == s["m"].numel()]
            if not fits:
                raise ValueError(f"optimizer state: flat moments of {s['m'].numel()} elements do not match the "
                                 f"parameters at a slot size of {saved_pad if saved_pad is not None else '8 or 4'}")
            pad = fits[0]
            src_off = dst_off = 0
            for n in sizes:
                for key in ("m", "v"):
                    g[key][dst_off:dst_off + n].copy_(s[key][src_off:src_off + n])
                src_off += (n + pad - 1) // pad * pad
                dst_off += (n + self.FLAT_PAD - 1) // self.FLAT_PAD * self.FLAT_PAD
        for t, s in zip(self.tables, sd["tables"]):
            assert t.table.name == s["name"]
            t.m0.copy_(s["m0"]); t.v0.copy_(s["v0"]); t.last.copy_(s["last"])
            if t.m1 is not None:
                t.m1.copy_(s["m1"]); t.v1.copy_(s["v1"])
            t.stale, t.cursor = bool(s["stale"]), int(s["cursor"])
        self.refresh_bf16()

    def get_last_lr(self):
        return [self.lr0 * lr_lambda(self.kind, self.steps_done, self.total, self.warmup)]
