"""MFP criterion: NCE over the whole feature vocabulary (host mirror of reference
code/nce/nce_loss.py NCELoss, nce/index_linear.py IndexLinear, nce/alias_multinomial.py
AliasMultinomial — `nce` loss type, per-word noise: the only mode the reference reaches)."""
import math
import os

import torch
from torch import nn
from torch.autograd import Function

from . import ops
from .layers import RowTable, TableWeight

BACKOFF_PROB = 1e-10


class AliasMultinomial(nn.Module):
    """Walker alias sampler.  Buffers `prob` [V] f32 and `alias` [V] i64 keep the reference's
    checkpoint layout; the table is built by the C++ host builder in the reference's visiting
    order and cached in data_dir/alias_self_{prob,alias}.h5 (torch.save files, as upstream)."""

    def __init__(self, probs, config):
        super().__init__()
        data_dir = getattr(config, "data_dir", None)
        pf = os.path.join(data_dir, "alias_self_prob.h5") if data_dir else None
        af = os.path.join(data_dir, "alias_self_alias.h5") if data_dir else None
        if pf and os.path.exists(pf) and os.path.exists(af):
            prob, alias = torch.load(pf), torch.load(af)
            if prob.numel() != probs.numel():
                raise ValueError(f"{pf} holds {prob.numel()} classes, expected {probs.numel()}")
        else:
            prob, alias = ops.alias_build(probs)
            if pf and os.path.isdir(data_dir) and int(getattr(config, "rank", 0)) == 0:
                # rank 0 alone writes the cache, atomically and `prob` last (readers test for
                # both files): another rank either loads whole files or builds the table itself
                for obj, path in ((alias, af), (prob, pf)):
                    tmp = f"{path}.tmp.{os.getpid()}"
                    torch.save(obj, tmp)
                    os.replace(tmp, path)
        self.register_buffer("prob", prob)
        self.register_buffer("alias", alias)
        self._packed = None

    def _load_from_state_dict(self, *a, **k):
        self._packed = None
        return super()._load_from_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def packed(self):
        if self._packed is None:
            self._packed = ops.alias_pack(self.prob, self.alias)
        return self._packed

    def draw_index_matrix(self, targets, K, seed, offset, offset_dev=None):
        """idx int32 [T, K+1]: column 0 = targets, then K negatives."""
        return ops.alias_draw(self.packed(), targets, K, seed, offset, offset_dev)


def _head_side():
    return HEAD_SIDE


class _NceLoss(Function):
    @staticmethod
    def forward(ctx, enc, emb_w, bias_w, logq, masked_index, idx, crit, F, P, want_logits):
        # inside a training step (the Trainer's window) the head's backward follows at once: its first launch also
        # forms the loss totals, one launch less between the loss and the backward pass
        later = TOTALS_LATER and ops.step_window[0] and ctx.needs_input_grad[0]
        o = ops.nce_fwd(enc.contiguous(), masked_index, idx, emb_w, bias_w.view(-1), logq, F, P,
                        want_logits=want_logits, totals_later=later, lazy=getattr(idx, "_lazy", None))
        ctx.totals = o["totals"]
        ctx.crit, ctx.F, ctx.P, ctx.K = crit, F, P, idx.shape[1] - 1
        ctx.plan = crit.table.plan
        ctx.save_for_backward(o["dlogit"], o["dh"], o["h"], masked_index)
        logits = o["logits"] if want_logits else torch.empty(0, device=enc.device)
        ctx.mark_non_differentiable(o["acc"], logits)
        ctx.set_materialize_grads(False)    # else autograd fills zero gradients for them: a launch on the chain
        ctx.crit.last_acc_ratio = o["loss"][1]      # device float, same launch as the loss
        return o["loss"][0], o["acc"].view(()), logits

    @staticmethod
    def backward(ctx, gl, _a, _l):
        dlogit, dh, h, mi = ctx.saved_tensors
        gl = gl.contiguous().float()
        denc = ops.nce_scatter_dh(dh, mi, ctx.F, ctx.P, gscale=gl, totals=ctx.totals)
        ctx.totals = None
        if ctx.plan is None:
            raise RuntimeError("NCE backward without a segment plan")
        lazy = ctx.crit.table.lazy
        early = lazy is not None and getattr(lazy, "early_now", False)
        if (DEFER_PLAN_JOIN and getattr(ctx.crit, "towers_follow", False) and ops.step_window[0] and _head_side()
                and dh.is_cuda and (early or ops.HEAD_SIDE_REDUCE_ONLY)):
            # The dense-encoder head (bf16 mode).  Joined HERE the sort of the sampled ids held the main stream: with
            # bf16 GEMMs the sort chain outlasts the forward pass (by 40 us at Avazu's sizes, 70 at Criteo's) and the
            # whole backward pass waited for it.  The join moves behind the encoder's backward (the towers' join node
            # runs it: ~130 us of main-stream work later, when the sort is long done), and the table's gradient + row
            # update go to the END of the cross tower's chain, as in the grouped head (a tower <- plan edge of its own
            # is what segfaults in hipStreamEndCapture; the tower stream syncs with the main one at the join node).
            slot, crit, K, P = ctx.plan, ctx.crit, ctx.K, ctx.P
            holder = {}
            ops.add_main_task(lambda: holder.__setitem__("plan", slot.get()))

            def table_work():
                plan = holder.get("plan") or slot.get()
                ge, gb = ops.nce_table_grad(plan, dlogit, h, K, P, gscale=gl)
                crit.table.sparse_grad = (plan, ge, gb)
                if early:            # (with a gradient exchange or a clipping norm ahead the row update has to wait)
                    lazy.update()
                cur = torch.cuda.current_stream()
                for t in (dlogit, h, gl) + tuple(plan.tensors()):
                    t.record_stream(cur)
            ops.add_late_task(table_work)
            return denc, None, None, None, None, None, None, None, None, None
        plan = ctx.plan.get()
        if lazy is not None and getattr(lazy, "early_now", False) and _head_side() and dh.is_cuda:
            # as in _EncNceLoss.backward: the table's gradient and row update leave the main chain
            main, side = torch.cuda.current_stream(), ops.aux_stream("tower", dh.device)
            if ops.stream_wait(side, main):
                with torch.cuda.stream(side):
                    ge, gb = ops.nce_table_grad(plan, dlogit, h, ctx.K, ctx.P, gscale=gl)
                    ctx.crit.table.sparse_grad = (plan, ge, gb)
                    lazy.update()
                for t in (ge, gb, dlogit, h, gl) + tuple(plan.tensors()):
                    t.record_stream(side)
                ops.pending_joins.append((main, side))
                return denc, None, None, None, None, None, None, None, None, None
        ge, gb = ops.nce_table_grad(plan, dlogit, h, ctx.K, ctx.P, gscale=gl)
        ctx.crit.table.sparse_grad = (plan, ge, gb)
        return denc, None, None, None, None, None, None, None, None, None


HEAD_SIDE = os.environ.get("MAPX_HEAD_SIDE", "1") == "1"     # 0.871 vs 0.900 ms per step
DEFER_PLAN_JOIN = os.environ.get("MAPX_DEFER_PLAN_JOIN", "1") == "1"    # the dense-encoder head: see _NceLoss.backward
TOTALS_LATER = os.environ.get("MAPX_TOTALS_LATER", "1") == "1"     # loss totals formed by the head's first backward launch
LATE_TABLE = os.environ.get("MAPX_LATE_TABLE", "1") == "1"
# _EncNceLoss.backward: the table's gradient on the plan stream behind its sort ("1"), at the end of the cross tower's
# chain ("0"), or ("auto") by the cross tower's width: its chain is the longer one at Criteo's 624 columns, where taking
# the gradient off it wins (1.0155 / 1.0085 -> 0.9868 / 0.9865 ms), and has slack at Avazu's 368, where the plan
# stream's queue is the deep tower's (0.668 -> 0.690 / 0.694 with "1").
TABLE_ON_PLAN = os.environ.get("MAPX_TABLE_ON_PLAN", "auto")
TABLE_ON_PLAN_MIN_D = 512


def _table_on_plan(join):
    if TABLE_ON_PLAN == "auto":
        return join is not None and join.D >= TABLE_ON_PLAN_MIN_D
    return TABLE_ON_PLAN == "1"
# "1": a training step reads the sampled table rows through their pending zero-gradient updates (no catch-up pass; the
# gradient update is a row's one read-modify-write of the step).  Built, bit-identical, and
# measured slower on the step it was asked for (one box, tools/ab_env.sh): the loss kernel 19.5 -> 64 us (639 k row
# accesses per step, 14 % of them stale and nearly all distinct: the replay runs in waves that are 70 % divergent, on
# the loss's critical path), the update 2 x 12.8 -> 2 x 17.5 us, against the 31-us catch-up pass that ran beside the deep
# tower's first GEMMs: 0.746 vs 0.730 ms per step.  Hence opt-in.
LAZY_FOLD = os.environ.get("MAPX_LAZY_FOLD", "0") == "1"


class _EncNceLoss(Function):
    """feat_encoder + field gather + NCE loss in one autograd node, computing only the encoder
    blocks that targets select (reference models.py:74-76 computes all F blocks and gathers L).
    Backward: dX through the dense scattered d_enc (unchanged), dW through the grouped GEMM."""

    @staticmethod
    def forward(ctx, final, w_enc, b_enc, emb_w, bias_w, logq, masked_index, idx, crit, F, P, want_logits,
                groups, join=None):
        final = final.contiguous()
        dh_slots = torch.empty(groups.cap, P, dtype=torch.float32, device=final.device)
        h_slots = ops.enc_grouped_fwd(final, w_enc, b_enc, groups, zero_slots=dh_slots)
        ready = getattr(idx, "_ready", None)
        if ready is not None:           # sampled behind the cross tower (models.NCE_AFTER_CROSS): only this kernel waits
            ops.stream_wait_event(torch.cuda.current_stream(), ready[0], ready[1])
        later = TOTALS_LATER and ops.step_window[0] and ctx.needs_input_grad[0]      # (see _NceLoss.forward)
        o = ops.nce_fwd(h_slots, masked_index, idx, emb_w, bias_w.view(-1), logq, F, P,
                        want_logits=want_logits, hpos=groups.hpos, dh_slots=dh_slots, totals_later=later,
                        lazy=getattr(idx, "_lazy", None))
        ctx.totals = o["totals"]
        ctx.crit, ctx.F, ctx.P, ctx.K, ctx.groups, ctx.join = crit, F, P, idx.shape[1] - 1, groups, join
        ctx.plan = crit.table.plan
        ctx.slots = (getattr(w_enc, "_mapx_grad", None), getattr(b_enc, "_mapx_grad", None))
        ctx.amax = ops.amax_pack(final, w_enc, dh_slots)
        ctx.save_for_backward(final, w_enc, o["dlogit"], o["dh"], o["h"], masked_index, dh_slots)
        logits = o["logits"] if want_logits else torch.empty(0, device=final.device)
        ctx.mark_non_differentiable(o["acc"], logits)
        ctx.set_materialize_grads(False)    # else autograd fills zero gradients for them: a launch on the chain
        ctx.crit.last_acc_ratio = o["loss"][1]      # device float, same launch as the loss
        return o["loss"][0], o["acc"].view(()), logits

    @staticmethod
    def backward(ctx, gl, _a, _l):
        final, w_enc, dlogit, dh, h, mi, dh_slots = ctx.saved_tensors
        ops.amax_unpack((final, w_enc, dh_slots), ctx.amax)
        sw, sb = ctx.slots
        gl = gl.contiguous().float()
        denc = ops.nce_scatter_dh(dh, mi, ctx.F, ctx.P, gscale=gl, totals=ctx.totals)          # dense [B, F*P]
        ctx.totals = None
        dfinal, joined = None, False
        if ctx.needs_input_grad[0]:
            join = ctx.join
            if join is not None and join.usable(denc, final):
                from .layers import join_bwd_input
                dfinal = join_bwd_input(denc, w_enc, final, join)      # one product per tower, each on its stream
                joined = True
            else:
                dfinal = ops.linear_bwd_input(denc, w_enc)
        if ctx.plan is not None:
            ctx.plan.start()         # sort of the sampled ids: forks from the draw, enqueued behind the dX GEMM
        if ctx.plan is None:
            raise RuntimeError("NCE backward without a segment plan")
        lazy = ctx.crit.table.lazy
        early = lazy is not None and getattr(lazy, "early_now", False)
        # (with a gradient exchange or a clipping norm ahead the row update has to wait; the reduction need not)
        aside = HEAD_SIDE and final.is_cuda and (early or ops.HEAD_SIDE_REDUCE_ONLY)
        if aside and joined and _table_on_plan(ctx.join) and ctx.plan.value is not None and ops.step_window[0]:
            # The table's gradient on the PLAN stream, right behind the sort that it alone needs: no stream waits for the
            # plan at this point of the step (rounds 1-4 joined the plan into the main stream here — a tower <- plan join
            # crashes hipStreamEndCapture — and the deep tower's backward chain, which continues on this stream, waited
            # for the sort chain: 20-29 us of both queues on round 4's timelines).  The plan stream waits for this point
            # of the main stream (dlogit, h and the loss scale are final) and is joined by optimizer.step().
            from .layers import _side_stream
            main, pst = torch.cuda.current_stream(), _side_stream(final.device)
            plan = ctx.plan.value
            if ops.stream_wait(pst, main):
                with torch.cuda.stream(pst):
                    ge, gb = ops.nce_table_grad(plan, dlogit, h, ctx.K, ctx.P, gscale=gl)
                    ctx.crit.table.sparse_grad = (plan, ge, gb)
                    if early:
                        lazy.update()
                for t in (ge, gb, dlogit, h, gl):
                    t.record_stream(pst)
                ops.pending_joins.append((main, pst))
                early = None
        elif aside and joined and LATE_TABLE:
            # with one product per tower the cross tower's chain starts at once on the tower stream; the table's
            # gradient reduction and row update (HBM-bound, ~70 us) go BEHIND it (ops.run_late_tasks) instead of in
            # front of it, beside the deep tower's remaining MFMA-bound GEMMs
            crit, K, P, do_update = ctx.crit, ctx.K, ctx.P, early
            plan = ctx.plan.get()        # (joined here, on the main stream: a tower <- plan join inside a capture
                                         # is the one that crashed hipStreamEndCapture in round 1)

            def table_work():
                ge, gb = ops.nce_table_grad(plan, dlogit, h, K, P, gscale=gl)
                crit.table.sparse_grad = (plan, ge, gb)
                if do_update:
                    lazy.update()
                cur = torch.cuda.current_stream()
                for t in (dlogit, h, gl) + tuple(plan.tensors()):
                    t.record_stream(cur)
            ops.add_late_task(table_work)
            early = None
        elif aside:
            # The table's gradient reduction and row update need nothing of this backward pass but the loss
            # scale; the trunk's backward needs only dfinal.  They go to the tower stream (idle between the
            # towers' forward and backward), forked behind the dX GEMM; the trunk's backward no longer waits
            # for them on the main stream.  (The encoder's dW / db there as well, or on the plan stream: 0.888 /
            # 0.883 vs 0.873 ms — the cross tower's backward queues behind whatever this stream holds.)
            plan = ctx.plan.get()
            main, side = torch.cuda.current_stream(), ops.aux_stream("tower", final.device)
            if ops.stream_wait(side, main):
                with torch.cuda.stream(side):
                    ge, gb = ops.nce_table_grad(plan, dlogit, h, ctx.K, ctx.P, gscale=gl)
                    ctx.crit.table.sparse_grad = (plan, ge, gb)
                    if early:
                        lazy.update()
                for t in (ge, gb, dlogit, h, gl) + tuple(plan.tensors()):
                    t.record_stream(side)
                ops.pending_joins.append((main, side))
                early = None
        if joined and sw is not None and sb is not None:
            # nothing but the optimizer waits for the encoder's own gradients, and both towers' backward passes
            # wait for this node to return: they go to the end of the cross tower's chain (ops.run_late_tasks)
            groups = ctx.groups

            def encoder_grads():
                cur = torch.cuda.current_stream()
                for t in (dh_slots, final, denc, gl) + tuple(groups.tensors()):
                    t.record_stream(cur)
                ops.enc_grouped_dw(dh_slots, final, groups, out=sw, gscale=gl)
                ops.colsum(denc, out=sb, defer=True)
            ops.add_late_task(encoder_grads, dense=True)
            dw = db = None
        else:
            dw = ops.enc_grouped_dw(dh_slots, final, ctx.groups, out=sw, gscale=gl)
            db = ops.colsum(denc, out=sb, defer=True)      # second stage with the other deferred partial sums
        if early is not None:
            plan = ctx.plan.get()
            ge, gb = ops.nce_table_grad(plan, dlogit, h, ctx.K, ctx.P, gscale=gl)
            ctx.crit.table.sparse_grad = (plan, ge, gb)
        if early:
            # the table's row update (HBM-bound, 26 us at the end of the step) needs nothing else of
            # this backward pass: queued as a side task, it runs at the start of the cross tower's
            # backward chain, which has slack against the deep tower's (1.212 -> 1.201 ms).  The
            # encoder's bias-gradient column sum queued the same way cost what this gains.
            def update_rows():
                cur = torch.cuda.current_stream()
                for t in (ge, gb) + tuple(plan.tensors()):
                    t.record_stream(cur)
                lazy.update()
            ops.add_side_task(update_rows)
        return (dfinal, None if sw is not None else dw, None if sb is not None else db,
                None, None, None, None, None, None, None, None, None, None, None)


class IndexLinear(nn.Module):
    """Output embedding emb [V,P] + bias [V,1] scored only at the target and K sampled
    negatives, with the NCE binary loss.  Buffers/parameters keep the reference names:
    logprob_noise, alias.{prob,alias}, emb.weight, bias.weight."""

    def __init__(self, config):
        super().__init__()
        noise = torch.as_tensor(config.feat_count, dtype=torch.float32).cpu()
        probs = (noise / noise.sum()).clamp(min=BACKOFF_PROB)
        renormed = probs / probs.sum()
        self.register_buffer("logprob_noise", renormed.log())
        self.alias = AliasMultinomial(renormed, config)
        self.noise_ratio = config.pt_neg_num
        self.norm_term = math.log(noise.numel())
        self.proj_size = config.proj_size
        self.loss_type = "nce"
        self.emb = TableWeight(config.input_size, config.proj_size)
        self.bias = TableWeight(config.input_size, 1)
        self.reset_parameters()
        self.table = RowTable("mfp_criterion", self.emb.weight, self.bias.weight)
        self.seed = int(getattr(config, "seed", 42))
        self.rank = int(getattr(config, "rank", 0))
        self._draws = 0
        self.step_counter = None     # int32 device scalar (the optimizer's update counter): when set
                                     # and training, it advances the Philox stream (graph-replay safe)
        self.return_logits = False

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.proj_size)
        with torch.no_grad():
            self.emb.weight.uniform_(-stdv, stdv)
            # unigram initialisation of the bias (index_linear.py:44-48)
            self.bias.weight.copy_((self.logprob_noise + self.norm_term).unsqueeze(1))

    def get_noise_index(self, target):
        if self.training and self.step_counter is not None:
            offset, dev = (self.rank << 40) + (1 << 36), self.step_counter
        else:
            self._draws += 1
            offset, dev = (self.rank << 40) + self._draws, None
        return self.alias.draw_index_matrix(target.reshape(-1), self.noise_ratio, self.seed, offset, dev)

    def forward(self, target, input, masked_index=None, noise_samples=None, idx=None):
        """target [B,L] i64.  `input` is either the selected hidden [B,L,P] (reference call
        form, models.py:76) or, with `masked_index` [B,L], the whole encoder output [B,F*P] —
        then the field gather is fused into the loss kernel.
        -> (loss, logits [B,L,K+1] or empty, indices int32 [B,L,K+1]); `self.last_acc` holds
        the number of targets ranked first (device scalar)."""
        B, L = target.shape
        P = self.proj_size
        if masked_index is None:
            enc = input.reshape(B, L * P)
            F = L
            masked_index = torch.arange(L, device=target.device).expand(B, L).contiguous()
        else:
            enc, F = input, input.shape[1] // P
        if idx is None:                      # else: sample_ids() ran earlier (ids drawn, rows caught up)
            idx = self.sample_ids(target, noise_samples)
        loss, acc, logits = _NceLoss.apply(enc, self.emb.weight, self.bias.weight, self.logprob_noise,
                                           masked_index, idx, self, F, P, self.return_logits)
        self.last_acc = acc
        if self.return_logits:
            logits = logits.view(B, L, -1)
        return loss, logits, idx.view(B, L, -1)

    def supports_grouped_encoder(self):
        return self.proj_size == 32 and self.noise_ratio + 1 <= 32

    def sample_ids(self, target, noise_samples=None, catch_up=True):
        """The step's sampled row ids [B*L, K+1] (target in column 0) and the lazy catch-up of the
        rows they name.  Depends on the targets only, not on the trunk: a model may call it early,
        on a stream that has slack (DCNV2 does, ahead of the cross tower), and hand the result to
        forward_with_encoder."""
        B, L = target.shape
        V = self.emb.weight.shape[0]
        if noise_samples is not None:
            idx = ops.nce_pack_idx(target.reshape(-1), noise_samples.reshape(B * L, -1), V)
        else:
            idx = self.get_noise_index(target)
        need_grad = torch.is_grad_enabled() and self.emb.weight.requires_grad
        # a training step's rows are read through their pending updates by the loss kernel and written once, by the
        # gradient update; without a gradient (eval) the catch-up pass brings them up to date
        lazy = self.table.lazy
        fold = (LAZY_FOLD and need_grad and lazy is not None and lazy.replay_in_readers()
                and ops.lazy_rows_supported(self.proj_size, idx.shape[1]))
        self.table.prepare(idx.view(-1), need_grad, defer_plan=True, through_replay=fold, catch_up=catch_up)
        idx._lazy = lazy.lazy_rows() if fold else None
        return idx

    def forward_with_encoder(self, target, final, encoder, masked_index, noise_samples=None, groups=None,
                             idx=None, join=None):
        """The MFP head from the trunk output: `encoder` (feat_encoder) is applied only to the
        field blocks that `masked_index` selects.  Same returns as forward()."""
        B, L = target.shape
        P, F = self.proj_size, encoder.out_features // self.proj_size
        if idx is None:
            idx = self.sample_ids(target, noise_samples)
        if groups is None:
            groups = ops.EncGroups(masked_index, F)       # one launch: counting sort of the targets by field
        loss, acc, logits = _EncNceLoss.apply(final, encoder.weight, encoder.bias, self.emb.weight,
                                              self.bias.weight, self.logprob_noise, masked_index, idx, self,
                                              F, P, self.return_logits, groups, join)
        self.last_acc = acc
        if self.return_logits:
            logits = logits.view(B, L, -1)
        return loss, logits, idx.view(B, L, -1)
