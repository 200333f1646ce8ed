"""BaseModel + DCNV2 (host mirror of reference code/models.py:21-127, 282-322): same factory,
same forward signature and output tuples, same state_dict key layout; every arithmetic step
is a gfx950 kernel."""
import logging
import os

import torch
from torch import nn

from . import ops
from .arguments import Config
from .layers import (_JoinColumns, CIN, CrossNetV2, Embeddings, HipLinear, MLPBlock, MultiHeadSelfAttention,
                     RowTable, TableWeight, bce_with_logits, fm_product_sum)
from .nce import IndexLinear

logger = logging.getLogger(__name__)
GROUPED_ENCODER = os.environ.get("MAPX_GROUPED_ENC", "1") == "1"
NCE_EARLY = os.environ.get("MAPX_NCE_EARLY", "1") == "1"
# both tables' segment plans from one chain of launches (8 launches instead of 16, 0.112 instead of
# 0.19 ms of sorting per step): "fwd" = started in forward behind the towers, "bwd" = started by the
# head's backward node (where the sampled ids' sort alone starts otherwise), "off" = one chain per
# table; "auto" = fwd.  What decided it was WHERE the graph runtime runs the chain: captured as a branch
# of its own it was run ahead of the deep tower's GEMMs on the queue the two share, and the joint chain
# lost in fp32 (1.306 fwd / 1.305 bwd vs 1.200 off) and won only a little in bf16 (0.752 vs 0.793).  Made
# to wait for the deep tower's GEMMs (PLAN_AFTER_DNN below) it wins in both: fp32 0.908 / 0.912 vs 0.935 /
# 0.937 ms (off), bf16 0.690 vs 0.744 (fwd without the wait) vs 0.783 (off).  (Also measured in fp32: the
# embedding's sort started by the head's backward node right behind the sampled ids' sort — so that the
# head's backward no longer waits 34 us for the latter — 1.13 vs 0.96 ms.)
JOINT_PLAN = os.environ.get("MAPX_JOINT_PLAN", "auto")
# the joint plan in forward waits for the deep tower's GEMMs to be on their way (see above)
# the grouped encoder's slot layout ahead of the deep tower (main stream) or on the cross tower's stream.  Round 2: main
# won (0.869 / 0.873 vs 0.876 / 0.882 ms); round 3, after the backward pass changed (tools/flag_sweep.py): the tower
# stream wins, 0.8058 vs 0.8223 ms; round 4: with the two-piece fp16 GEMMs cutting both operands main won (0.7175 vs
# 0.7307), with the weights' planes (gemm_h2w.hip) the tower stream again (0.7120 vs 0.7175) — kept there
# "auto": on the main stream when the cross tower is at least 512 columns wide (Criteo-shaped: its stream is the longer
# one of the forward pass, 1.0026 -> 0.9924 ms with the layout off it), on the cross tower's stream otherwise (Avazu-
# shaped: equal either way, 0.7064 vs 0.7070)
LAYOUT_ON_MAIN = os.environ.get("MAPX_LAYOUT_ON_MAIN", "auto")
PLAN_AFTER_DNN = os.environ.get("MAPX_PLAN_AFTER_DNN", "1")       # 1 | tower | 0: what the joint plan goes behind
# RFD / finetune steps: what the one table's sort goes behind: auto | main | tower | 0 (A/B switch)
PLAN_AFTER_TRUNK = os.environ.get("MAPX_PLAN_AFTER_TRUNK", "auto")

X0_LINK = os.environ.get("MAPX_X0_LINK", "1") == "1"       # A/B switch of layers._X0Link
# Round 4 (the two-piece fp16 GEMMs made the towers a third shorter): the NCE head's sampling + catch-up (HBM-bound,
# 80 us in the graph) BEHIND the cross tower's GEMMs on the tower stream instead of in front of them — in front, the
# cross tower ended 50 us after the deep one and the encoder waited for it; the main stream joins the tower stream at
# the cross tower's end (an event), the loss kernel alone waits for the sampled ids (`nce_idx._ready`)
CATCHUP_AFTER_CROSS = os.environ.get("MAPX_CATCHUP_AFTER_CROSS", "0") == "1"     # A/B switch (DCNV2.forward)
NCE_AFTER_CROSS = os.environ.get("MAPX_NCE_AFTER_CROSS", "0") == "1"
EARLY_NCE_ALL = os.environ.get("MAPX_EARLY_NCE_ALL", "1") == "1"   # A/B switch: BaseModel._sample_early

_OTHER_BACKBONES = ("trans", "fignn", "fgcnn")


class _RfdPredictor(nn.ModuleDict):
    """Linear -> ReLU -> Linear with the reference nn.Sequential's keys "0" and "2"
    (models.py:119-123); the ReLU is fused into the first GEMM's epilogue."""

    def __init__(self, input_dim, hidden, out):
        super().__init__({"0": HipLinear(input_dim, hidden, relu=True), "2": HipLinear(hidden, out, out_fp32=True)})

    def forward(self, x, link_in=None):
        return self["2"](self["0"](x, link_in=link_in))


class BaseModel(nn.Module):
    used_params = []

    def __init__(self, model_name="BaseModel", config: Config = None):
        super().__init__()
        self.model_name = model_name
        self.config = config

    @classmethod
    def from_config(cls, config: Config):
        name = config.model_name.lower()
        from .layers import compute_dtype_of
        if compute_dtype_of(config) != torch.float32 and name not in ("dcnv2", "dnn"):
            raise NotImplementedError(f"compute_dtype=bf16 is built for DCNv2 and DNN, not {config.model_name}")
        if name == "dcnv2":
            return DCNV2(config)
        if name == "dnn":
            return DNN(config)
        if name == "deepfm":
            return DeepFM(config)
        if name == "autoint":
            return AutoInt(config)
        if name == "xdeepfm":
            return xDeepFM(config)
        if name in _OTHER_BACKBONES:
            raise NotImplementedError(
                f"{config.model_name}: mapx builds the DCNv2 hot path and, of the other backbones "
                "(SURVEY §8 f4), DNN, DeepFM, xDeepFM and AutoInt")
        raise NotImplementedError(config.model_name)

    def validate_model_config(self):
        logger.info(f"  model_name = {self.model_name}")
        for key in self.used_params:
            logger.info(f"  {key} = {getattr(self.config, key)}")

    # ------------------------------------------------------------------ heads
    def _sample_early(self, labels, masked_index, noise_samples):
        """MFP over a single-stream trunk (DNN, DeepFM, xDeepFM, AutoInt): the NCE head's sampling and the lazy
        catch-up of the sampled rows need only the targets — they run on the tower stream beside the trunk instead of
        between the trunk and the loss (as DCNV2.forward does by hand).  -> (ids | None, join state)."""
        if not (EARLY_NCE_ALL and self.config.pretrain and self.config.pt_type == "MFP" and masked_index is not None
                and labels is not None and labels.is_cuda):
            return None, None
        main = torch.cuda.current_stream()
        tower = ops.aux_stream("tower", labels.device)
        forked = ops.stream_wait(tower, main)
        with torch.cuda.stream(tower):
            idx = self.mfp_criterion.sample_ids(labels, noise_samples)
        return idx, (main, tower, forked, labels)

    def _plans_and_join(self, idx, state):
        """Behind the trunk: both tables' segment plans from ONE chain of launches (behind what the main stream holds),
        then the main stream takes the sampled ids over."""
        if state is None:
            self.embed.table.start_plan()
            return
        main, tower, forked, labels = state
        from .layers import PlanSlot
        if self.mfp_criterion.table.plan is not None and self.embed.table.plan is not None:
            PlanSlot.start_many([self.embed.table.plan, self.mfp_criterion.table.plan], after=main)
        else:
            self.embed.table.start_plan()
        ops.stream_wait(main, tower)
        if forked:
            labels.record_stream(tower)
            idx.record_stream(main)

    def get_outputs(self, inputs, labels=None, masked_index=None, is_pretrain=None, noise_samples=None,
                    groups=None, nce_idx=None, join=None):
        """MFP -> (loss, #signals, #targets ranked first)            (models.py:71-78)
        RFD -> (loss, #signals, accuracy, positive ratio)            (models.py:79-85)
        CTR -> (loss, logits) or (logits,)                           (models.py:88-93)
        `#targets ranked first` is a device scalar (no host sync per step); the reference
        returns a Python int after `.item()`.  `join`: the layers._JoinLink of a two-tower trunk whose output
        `inputs` is (the head's first layer then does both towers' first backward step in its dX epilogue)."""
        cfg = self.config
        if (is_pretrain is None and cfg.pretrain) or is_pretrain:
            if cfg.pt_type == "MFP":
                crit = self.mfp_criterion
                if (GROUPED_ENCODER and crit.supports_grouped_encoder() and inputs.shape[1] % 8 == 0
                        and inputs.dtype == torch.float32):
                    # only the L masked fields' blocks of feat_encoder are computed (26 %)
                    loss, _logits, _idx = crit.forward_with_encoder(labels, inputs, self.feat_encoder,
                                                                    masked_index, noise_samples=noise_samples,
                                                                    groups=groups, idx=nce_idx, join=join)
                else:
                    # (bf16 mode: the dense encoder GEMM — 16x the MFMA rate makes computing all F blocks
                    # cheaper than the grouped GEMM's gathers; its output and the whole NCE head stay fp32)
                    enc = self.feat_encoder(inputs, link_in=join)
                    loss, _logits, _idx = crit(labels, enc, masked_index=masked_index,
                                               noise_samples=noise_samples, idx=nce_idx)
                return (loss, labels.shape[0] * labels.shape[1], self.mfp_criterion.last_acc)
            if cfg.pt_type == "RFD":
                logits = self.pred_rfd(inputs, link_in=join)
                loss, stats = bce_with_logits(logits, labels)
                return (loss, labels.shape[0] * labels.shape[1], stats[1], stats[2])
            raise NotImplementedError(cfg.pt_type)
        outputs = (inputs,)
        if labels is not None:
            loss, _ = bce_with_logits(inputs.view(-1), labels.float())
            outputs = (loss,) + outputs
        return outputs

    def create_pretraining_predictor(self, input_dim):
        cfg = self.config
        if cfg.pt_type == "MFP":
            self.feat_encoder = HipLinear(input_dim, cfg.num_fields * cfg.proj_size, out_fp32=True)
            self.mfp_criterion = IndexLinear(cfg)
        elif cfg.pt_type == "RFD":
            self.pred_rfd = _RfdPredictor(input_dim, cfg.num_fields * cfg.proj_size, cfg.num_fields)
        else:
            raise NotImplementedError(cfg.pt_type)

    # ------------------------------------------------------------------ checkpoints
    def load_from_target_model(self, target_model_dict):
        """Copy every tensor whose NAME and SHAPE match; report the rest (models.py:97-107)."""
        own = self.state_dict()
        skipped = []
        for k, v in target_model_dict.items():
            if k in own and own[k].shape == v.shape:
                own[k] = v
                logger.info(f"Load tensor: {k}, {tuple(v.shape)}")
            else:
                skipped.append(k)
                logger.info(f"Unmatched tensor in the target model: {k}, {tuple(v.shape)}")
        self.load_state_dict(own)
        return skipped

    def load_for_finetune(self, model_path):
        return self.load_from_target_model(torch.load(model_path, map_location="cpu"))

    def row_tables(self):
        """The [V,*] tables with row-sparse gradients (optimised by mapx.optim.TableAdam)."""
        return [m.table for m in self.modules() if hasattr(m, "table")]

    def table_parameter_ids(self):
        ids = set()
        for t in self.row_tables():
            ids.add(id(t.p0))
            if t.p1 is not None:
                ids.add(id(t.p1))
        return ids


class DCNV2(BaseModel):
    used_params = ["embed_size", "hidden_size", "num_hidden_layers", "hidden_dropout_rate", "hidden_act",
                   "num_cross_layers"]

    def __init__(self, config: Config):
        super().__init__(model_name="DCNV2", config=config)
        self.embed = Embeddings(config)
        self.embed.defer_plan = True                 # forward() picks the fork point of the sort
        self.embed.table.mark_dense_ready = True     # the gather's backward node is this model's last one
        input_dim = config.num_fields * config.embed_size
        self.cross_net = CrossNetV2(input_dim, config.num_cross_layers)
        final_dim = input_dim
        if config.num_hidden_layers > 0:
            self.parallel_dnn = MLPBlock(input_dim=input_dim, hidden_size=config.hidden_size,
                                         num_hidden_layers=config.num_hidden_layers,
                                         hidden_dropout_rate=config.hidden_dropout_rate,
                                         hidden_act=config.hidden_act)
            final_dim += config.hidden_size
        if config.pretrain:
            self.create_pretraining_predictor(final_dim)
        else:
            self.fc_out = HipLinear(final_dim, 1, out_fp32=True)

    def _mfp_head(self, masked_index):
        return self.config.pretrain and self.config.pt_type == "MFP" and masked_index is not None

    def _grouped_head(self, masked_index):
        return (self._mfp_head(masked_index) and self.embed.compute_dtype == torch.float32
                and GROUPED_ENCODER and self.mfp_criterion.supports_grouped_encoder()
                and self.feat_encoder.in_features % 8 == 0)

    def forward(self, input_ids, labels=None, masked_index=None, noise_samples=None):
        groups, nce_idx, join = None, None, None
        planes_ev = None
        if ops._dirty_planes and input_ids.is_cuda and self.config.num_hidden_layers > 0:
            # the weights' planes of this step (ops.PLANES_AT_START): on the tower stream, beside the step's head (catch-up,
            # gather: small grids); the deep tower's first product waits for them
            tower0 = ops.aux_stream("tower", input_ids.device)
            if ops.stream_wait(tower0, torch.cuda.current_stream()):
                with torch.cuda.stream(tower0):
                    ops.refresh_dirty_planes()
                    planes_ev = ops.record_event()
        feat_embed = ops.flat_rows(self.embed(input_ids))
        if self.config.num_hidden_layers > 0:
            # Three independent chains leave the gather: the cross tower (small D x D GEMMs on a
            # second stream that fill the tails of the deep tower's big ones; autograd replays the
            # same stream assignment in backward), the deep tower (main stream) and the sort for
            # the embedding gradient's segment plan.  The sort is enqueued LAST but forks from the
            # event where its keys were final: a captured hipGraph keeps the first-captured
            # successor of a node on the node's queue, and a chain of tiny kernels there starves
            # the other queues (measured: the trunk's first GEMM started 150 us late).
            main = torch.cuda.current_stream()
            tower = ops.aux_stream("tower", feat_embed.device)
            forked = ops.stream_wait(tower, main)

            # both towers write their last layer straight into the concatenated buffer
            D, H = feat_embed.shape[1], self.config.hidden_size
            direct = self.config.num_cross_layers > 0
            if (direct and torch.is_grad_enabled() and feat_embed.dtype == torch.float32
                    and self.parallel_dnn.act == "relu"
                    and not (self.parallel_dnn.p_drop > 0 and self.training)):
                from .layers import _JoinLink
                join = _JoinLink(D)            # towers -> the head's first layer (fused backward epilogue)
            x0_link = None
            if (X0_LINK and direct and torch.is_grad_enabled() and self.embed.table.plan is not None
                    and not self.embed.embed_norm and not (self.embed.dropout.p > 0 and self.training)):
                from .layers import _X0Link
                x0_link = _X0Link(main)        # cross tower -> the gather's backward (no elementwise add, no wait)
            self.embed.table.x0_link = x0_link
            final_buf = torch.empty(feat_embed.shape[0], D + H, dtype=feat_embed.dtype, device=feat_embed.device)
            if feat_embed.dtype == torch.float32:
                # ONE magnitude record for the concatenated output: both towers' last kernels raise it (ops.out_record)
                ops.tag(final_buf, ops.amax_record(final_buf.device))
            if (LAYOUT_ON_MAIN == "1" or (LAYOUT_ON_MAIN == "auto" and D >= 512)) and self._grouped_head(masked_index):
                # the grouped encoder's slot layout (one 15-us launch) ahead of the deep tower, which by now
                # has ~45 us of slack against the cross tower's stream (round 1 had it the other way round)
                groups = ops.EncGroups(masked_index, self.config.num_fields)
            with torch.cuda.stream(tower):
                if self._grouped_head(masked_index) and groups is None:
                    # the cross tower has ~70 us of slack against the deep one: the slot layout of
                    # the grouped encoder (one single-workgroup launch) rides on its stream
                    groups = ops.EncGroups(masked_index, self.config.num_fields)
                early_nce = self._mfp_head(masked_index) and NCE_EARLY and labels is not None \
                    and (groups is not None or self.embed.compute_dtype != torch.float32)
                after_cross = early_nce and NCE_AFTER_CROSS and groups is not None and direct
                if early_nce and not after_cross:
                    # the NCE head's sampling and the lazy catch-up of the sampled rows need only
                    # the targets: HBM-bound kernels that run beside the deep tower's first GEMMs
                    # instead of alone between the towers and the loss (same branch, no new one).  (Round 2: BEHIND
                    # the cross tower's GEMMs instead, the trunk joining at the cross tower's end and only the
                    # loss kernel waiting for the sampling: 1.02 vs 0.91 ms fp32, 0.72 vs 0.66 bf16.)
                    # (CATCHUP_AFTER_CROSS: the draw stays here — the segment plans' sort waits for it — and the
                    # catch-up of the sampled rows goes BEHIND the cross tower's products on this stream, in front of
                    # the join: since the fp32 products became 1.3-1.9 x faster the cross tower's stream, not the deep
                    # tower's, is the longer one in forward)
                    nce_idx = self.mfp_criterion.sample_ids(labels, noise_samples, catch_up=not CATCHUP_AFTER_CROSS)
                cross_output = self.cross_net(feat_embed, out=ops.alias_cols(final_buf, 0, D) if direct else None,
                                              link=join, x0_link=x0_link)
                if early_nce and not after_cross and CATCHUP_AFTER_CROSS:
                    self.mfp_criterion.table.catch_up_pending()
                cross_done = None
                if after_cross:
                    cross_done = ops.record_event()
                    nce_idx = self.mfp_criterion.sample_ids(labels, noise_samples)
                    nce_idx._ready = (ops.record_event(), tower)
            if planes_ev is not None:
                ops.stream_wait_event(main, planes_ev, tower)
            dnn_output = self.parallel_dnn(feat_embed, out=ops.alias_cols(final_buf, D, H) if direct else None,
                                           link_last=join.relu if join is not None else None)
            # Both tables' segment plans from ONE chain of launches (8 instead of 8 + 8), when the
            # sampled ids exist already (drawn early on the tower stream).  (Round 1 note: the sampled
            # ids' sort as a chain of its own ahead of the embedding's cost 1.375 vs 1.21 ms.)
            mode = JOINT_PLAN
            if mode == "auto":
                mode = "fwd"
            if nce_idx is not None and mode == "fwd":
                from .layers import PlanSlot
                from .layers import IMPLIED
                # (the tower stream forked from the main one behind the gather: the sampled ids' event
                # implies that the embedding's keys are final)
                PlanSlot.start_many([self.embed.table.plan, self.mfp_criterion.table.plan],
                                    implied=[self.embed.table.plan] if IMPLIED else (),
                                    after={"1": main, "tower": tower}.get(PLAN_AFTER_DNN))
            elif nce_idx is not None and mode == "bwd" and self.mfp_criterion.table.plan is not None:
                # the head's backward node starts both (PlanSlot.start_many from IndexLinear's partner list)
                self.mfp_criterion.table.plan.partners = [self.embed.table.plan]
            else:
                # (RFD / finetune steps: one table, one chain.  Forked from the ids alone the graph runtime ran it
                # LAST, 127 us of sort + reduction + row update exposed behind the backward pass; behind the deep
                # tower's forward GEMMs ("main") it runs beside the head: RFD 0.946 -> 0.845 ms, finetune 0.765 -> 0.664.  The
                # single-stream backbones below measured neutral (RFD / CTR) or worse (DNN + MFP): left as they were)
                # Behind the CROSS tower's (the head is short then and the deep tower's backward would wait for the
                # sort on its queue): finetune 0.636 -> 0.619 ms, RFD 0.768 -> 0.786 — so by the head.
                # Round 4 (tools/mini_sweep_steps.sh, after the join's capture order changed): the finetune step with
                # the fp32 trunk at Avazu's width now also prefers "main" (0.515 / 0.513 -> 0.493 / 0.492 ms); its
                # Criteo-shaped (0.623 vs 0.641) and bf16 (0.345 vs 0.376) forms keep "tower".
                narrow_f32 = self.embed.compute_dtype == torch.float32 and D < 512
                where = PLAN_AFTER_TRUNK if PLAN_AFTER_TRUNK != "auto" else \
                    ("main" if (self.config.pretrain or narrow_f32) else "tower")
                self.embed.table.start_plan(after={"main": main, "tower": tower}.get(where))
            if cross_done is not None:
                ops.stream_wait_event(main, cross_done, tower)       # (the sampled ids: waited for by the loss kernel)
                ops.pending_joins.append((main, tower))
            else:
                ops.stream_wait(main, tower)
            if forked:
                feat_embed.record_stream(tower)
                final_buf.record_stream(tower)
                if groups is not None:
                    masked_index.record_stream(tower)
                    for t in groups.tensors():
                        t.record_stream(main)
                if nce_idx is not None:
                    labels.record_stream(tower)
                    nce_idx.record_stream(main)
            final_output = ops.carry(_JoinColumns.apply(cross_output, dnn_output, final_buf, join), final_buf) if direct \
                else torch.cat([cross_output, dnn_output], dim=-1)
            if self._mfp_head(masked_index):
                # the towers' join node and the cross tower's node run behind the head's backward: it may leave them
                # a stream join and the table's gradient (nce._NceLoss.backward)
                self.mfp_criterion.towers_follow = direct and torch.is_grad_enabled()
        else:
            self.embed.table.x0_link = None
            final_output = self.cross_net(feat_embed)
            self.embed.table.start_plan()
        if self.config.pretrain:
            return self.get_outputs(final_output, labels, masked_index, noise_samples=noise_samples, groups=groups,
                                    nce_idx=nce_idx, join=join)
        return self.get_outputs(self.fc_out(final_output, link_in=join), labels)


class DNN(BaseModel):
    """Embeddings -> MLP -> head (reference models.py:164-193).  Same heads, tables and optimizer
    path as DCNV2; only the trunk differs."""
    used_params = ["embed_size", "hidden_size", "num_hidden_layers", "hidden_dropout_rate", "hidden_act"]

    def __init__(self, config: Config):
        super().__init__(model_name="DNN", config=config)
        self.embed = Embeddings(config)
        self.embed.defer_plan = True
        self.dnn = MLPBlock(input_dim=config.embed_size * config.num_fields, hidden_size=config.hidden_size,
                            num_hidden_layers=config.num_hidden_layers,
                            hidden_dropout_rate=config.hidden_dropout_rate, hidden_act=config.hidden_act)
        if config.pretrain:
            self.create_pretraining_predictor(config.hidden_size)
        else:
            self.fc_out = HipLinear(config.hidden_size, 1, out_fp32=True)

    def forward(self, input_ids, labels=None, masked_index=None, noise_samples=None):
        feat_embed = ops.flat_rows(self.embed(input_ids))
        nce_idx, early = self._sample_early(labels, masked_index, noise_samples)
        nn_output = self.dnn(feat_embed)
        self._plans_and_join(nce_idx, early)     # the sort(s) fork from the ids, enqueued behind the trunk
        if self.config.pretrain:
            return self.get_outputs(nn_output, labels, masked_index, noise_samples=noise_samples, nce_idx=nce_idx)
        return self.get_outputs(self.fc_out(nn_output), labels)


class LR(nn.Module):
    """First-order term (reference models.py:129-143): `embed_w` [V,1] + `bias` [1].  In DeepFM the
    weight is the secondary parameter of the embedding's RowTable (same ids, one gradient plan);
    the reference initialises it like any nn.Embedding, N(0, 1)."""

    def __init__(self, config: Config):
        super().__init__()
        self.embed_w = TableWeight(config.input_size, 1)
        with torch.no_grad():
            self.embed_w.weight.normal_(0.0, 1.0)
        self.bias = nn.Parameter(torch.zeros(1))


class _InnerProductBuffers(nn.Module):
    """The reference's InnerProductLayer keeps three index tensors as frozen Parameters
    (layers.py:116-121); they appear in its state_dict, so they do here (as buffers)."""

    def __init__(self, num_fields):
        super().__init__()
        iu = torch.triu_indices(num_fields, num_fields, offset=1)
        self.register_buffer("field_p", iu[0].clone())
        self.register_buffer("field_q", iu[1].clone())
        self.register_buffer("upper_triangle_mask",
                             torch.triu(torch.ones(num_fields, num_fields), 1).bool())


class DeepFM(BaseModel):
    """LR + FM(product_sum) + MLP (reference models.py:196-233).  Pretraining feeds
    cat([dnn_vec, lr + fm]) [B, H+1] to the MFP / RFD heads; CTR sums the three logits."""
    used_params = ["embed_size", "hidden_size", "num_hidden_layers", "hidden_dropout_rate", "hidden_act"]

    def __init__(self, config: Config):
        super().__init__(model_name="DeepFM", config=config)
        self.embed = Embeddings(config)
        self.embed.defer_plan = True
        self.lr_layer = LR(config)
        # one row table for both [V,*] parameters read with input_ids
        self.embed.table = RowTable("embed.embedding", self.embed.embedding.weight, self.lr_layer.embed_w.weight)
        self.dnn = MLPBlock(input_dim=config.num_fields * config.embed_size, hidden_size=config.hidden_size,
                            num_hidden_layers=config.num_hidden_layers,
                            hidden_dropout_rate=config.hidden_dropout_rate, hidden_act=config.hidden_act)
        self.ip_layer = _InnerProductBuffers(config.num_fields)
        if config.pretrain:
            self.create_pretraining_predictor(config.hidden_size + 1)
        else:
            self.dnn_fc_out = HipLinear(config.hidden_size, 1)

    def forward(self, input_ids, labels=None, masked_index=None, noise_samples=None):
        x3, lr = self.embed.forward_with_linear(input_ids, self.lr_layer.embed_w.weight)
        nce_idx, early = self._sample_early(labels, masked_index, noise_samples)
        dnn_vec = self.dnn(ops.flat_rows(x3))
        self._plans_and_join(nce_idx, early)
        lr_fm = lr.view(-1, 1) + self.lr_layer.bias + fm_product_sum(x3)
        if self.config.pretrain:
            final_vec = torch.cat([dnn_vec, lr_fm], dim=1)
            return self.get_outputs(final_vec, labels, masked_index, noise_samples=noise_samples, nce_idx=nce_idx)
        return self.get_outputs(self.dnn_fc_out(dnn_vec) + lr_fm, labels)


class AutoInt(BaseModel):
    """Stacked multi-head self-attention over the field embeddings (reference models.py:440-488);
    the flattened [B, F*heads*attn_size] output feeds the MFP / RFD heads or `attn_out`; the finetune model adds
    the LR term (use_lr: its weight is the secondary parameter of the embedding's RowTable, as in DeepFM) and an MLP
    tower over the flattened embeddings (num_dnn_layers > 0: `dnn` + `dnn_out`), models.py:464-471, 482-486.
    Not built: attention dropout > 0."""
    used_params = ["embed_size", "num_attn_layers", "attn_size", "num_attn_heads", "attn_probs_dropout_rate",
                   "use_lr", "res_conn", "attn_scale", "dnn_size", "num_dnn_layers", "dnn_act", "dnn_drop"]

    def __init__(self, config: Config):
        super().__init__(model_name="AutoInt", config=config)
        self.embed = Embeddings(config)
        self.embed.defer_plan = True
        HA = config.num_attn_heads * config.attn_size
        self.self_attention = nn.Sequential(*[
            MultiHeadSelfAttention(config.embed_size if i == 0 else HA, attention_dim=config.attn_size,
                                   num_heads=config.num_attn_heads, dropout_rate=config.attn_probs_dropout_rate,
                                   use_residual=config.res_conn, use_scale=config.attn_scale)
            for i in range(config.num_attn_layers)])
        final_dim = config.num_fields * HA
        if config.pretrain:
            self.create_pretraining_predictor(final_dim)
        else:
            self.attn_out = HipLinear(final_dim, 1)
            # (the reference creates these for the finetune model only: models.py:463-471)
            self.lr_layer = LR(config) if config.use_lr else None
            if self.lr_layer is not None:       # one row table for both [V, *] parameters read with input_ids
                self.embed.table = RowTable("embed.embedding", self.embed.embedding.weight, self.lr_layer.embed_w.weight)
            # The reference sizes the tower's input as final_dim (fields x heads x attn_size, models.py:466) and feeds
            # it the flattened EMBEDDINGS (fields x embed_size, models.py:486): the option runs only when the two
            # agree; the same shapes and the same error otherwise.
            if config.num_dnn_layers and final_dim != config.num_fields * config.embed_size:
                raise ValueError("AutoInt with num_dnn_layers > 0 needs embed_size == num_attn_heads * attn_size "
                                 "(reference models.py:466, 486: the tower is sized for the attention output and fed "
                                 "the embeddings)")
            self.dnn = MLPBlock(input_dim=final_dim, hidden_size=config.dnn_size,
                                num_hidden_layers=config.num_dnn_layers, hidden_dropout_rate=config.dnn_drop,
                                hidden_act=config.dnn_act) if config.num_dnn_layers else None
            self.dnn_out = HipLinear(config.dnn_size, 1) if config.num_dnn_layers else None

    def forward(self, input_ids, labels=None, masked_index=None, noise_samples=None):
        lr = None
        if not self.config.pretrain and self.lr_layer is not None:
            x, lr = self.embed.forward_with_linear(input_ids, self.lr_layer.embed_w.weight)
        else:
            x = self.embed(input_ids)
        nce_idx, early = self._sample_early(labels, masked_index, noise_samples)
        attention_out = self.self_attention(x).flatten(start_dim=1)
        self._plans_and_join(nce_idx, early)
        if self.config.pretrain:
            return self.get_outputs(attention_out, labels, masked_index, noise_samples=noise_samples, nce_idx=nce_idx)
        logits = self.attn_out(attention_out)
        if lr is not None:
            logits = logits + (lr.view(-1, 1) + self.lr_layer.bias)         # models.py:483-484
        if self.dnn is not None:
            logits = logits + self.dnn_out(self.dnn(ops.flat_rows(x)))      # models.py:485-486
        return self.get_outputs(logits, labels)


class xDeepFM(BaseModel):
    """CIN + MLP (reference models.py:235-279): cat([CIN(embed), MLP(embed.flatten)]) — or the CIN alone when
    num_hidden_layers = 0 — feeds the MFP / RFD heads or `fc` (+ the LR term when use_lr).  The LR weight, when present, is the
    secondary parameter of the embedding's RowTable as in DeepFM."""
    used_params = ["embed_size", "hidden_size", "num_hidden_layers", "hidden_dropout_rate", "hidden_act",
                   "cin_layer_units", "use_lr"]

    def __init__(self, config: Config):
        super().__init__(model_name="xDeepFM", config=config)
        self.embed = Embeddings(config)
        self.embed.defer_plan = True
        units = [int(c) for c in str(config.cin_layer_units).split(",")]
        self.cin = CIN(config.num_fields, units)
        if config.num_hidden_layers > 0:
            self.dnn = MLPBlock(input_dim=config.num_fields * config.embed_size, hidden_size=config.hidden_size,
                                num_hidden_layers=config.num_hidden_layers,
                                hidden_dropout_rate=config.hidden_dropout_rate, hidden_act=config.hidden_act)
            final_dim = sum(units) + config.hidden_size
        else:                                   # models.py:253-255: the CIN alone feeds the heads
            self.dnn = None
            final_dim = sum(units)
        if config.pretrain:
            self.create_pretraining_predictor(final_dim)
        else:
            self.lr_layer = LR(config) if config.use_lr else None
            if self.lr_layer is not None:
                self.embed.table = RowTable("embed.embedding", self.embed.embedding.weight,
                                            self.lr_layer.embed_w.weight)
            self.fc = HipLinear(final_dim, 1)

    def forward(self, input_ids, labels=None, masked_index=None, noise_samples=None):
        lr = None
        if not self.config.pretrain and self.lr_layer is not None:
            x3, lr = self.embed.forward_with_linear(input_ids, self.lr_layer.embed_w.weight)
        else:
            x3 = self.embed(input_ids)
        nce_idx, early = self._sample_early(labels, masked_index, noise_samples)
        final_vec = self.cin(x3)
        if self.dnn is not None:
            final_vec = torch.cat([final_vec, self.dnn(ops.flat_rows(x3))], dim=1)
        self._plans_and_join(nce_idx, early)
        if self.config.pretrain:
            return self.get_outputs(final_vec, labels, masked_index, noise_samples=noise_samples, nce_idx=nce_idx)
        logits = self.fc(final_vec)
        if lr is not None:
            logits = logits + lr.view(-1, 1) + self.lr_layer.bias
        return self.get_outputs(logits, labels)
