"""Training path of the hot loop: tasks/viewpoint_select/pretrain.py:150-193 on HIP kernels.

``PretrainEngine`` owns what the reference's step does around ``model(**batch)``:

    model.zero_grad(); loss, ... = model(**batch); [all-reduce]; loss.backward(); optimizer.step(); scheduler.step()

* parameters are re-pointed into ONE flat fp32 slab (decay group first, then the no-decay group of
  pretrain.py:109-127), gradients into a second slab of the same layout (``p.grad`` are views), and
  a bf16 mirror of the parameter slab is what the GEMM kernels read; query/key/value weights (and
  biases) are laid out adjacently so the packed [3H,H] projection and its gradient are plain views;
* forward saves per-layer activations; backward = hand-written HIP kernels (encoder: one C call);
* the MLM / token heads are evaluated on the SUPERVISED rows only (labels != -1): the 7-tuple the
  reference returns depends on nothing else (encoder.py:377-431), so the result is identical while
  the 30522-wide decoder GEMM shrinks by ~12x;
* the optimizer is a fused AdamW kernel with the pytorch-transformers update rule;
* data parallelism: one process per GPU, bucketed all-reduce of the flat gradient slab
  (``torch.distributed``; backend "nccl" = RCCL over xGMI), see ``visitron_amd.distributed``.
"""
import ctypes
import math

import torch
import os

import torch.nn.functional as F

from . import _lib, ops
from .modeling import BertImgModelwithLocationEmbeds, PreTrainOscar, _i64, invalidate_packed_weights
from .ops import ACT_MUL, ACT_GELU, ACT_NONE, ACT_TANH, BF16, round_up

ALIGN = 64  # elements; keeps every parameter view 256-byte aligned inside the slabs


def _is_no_decay(name):
    return ("bias" in name) or ("LayerNorm.weight" in name)  # pretrain.py:109


class FlatParams(object):
    """Flat fp32 parameter / gradient / moment slabs + bf16 mirror for a PreTrainOscar."""

    def __init__(self, model, attach_grads=True):
        self.attach_grads = attach_grads
        named = list(model.named_parameters())
        dev = named[0][1].device
        # order: decay params then no-decay params; model order inside each group already puts
        # query, key, value (weights, and separately their biases) of a layer next to each other
        decay = [(n, p) for n, p in named if not _is_no_decay(n)]
        nodecay = [(n, p) for n, p in named if _is_no_decay(n)]
        self.entries, off = [], 0
        for grp, items in ((0, decay), (1, nodecay)):
            for n, p in items:
                # q|k|v must be exactly adjacent: no padding between them
                adjacent = n.endswith(("attention.self.key.weight", "attention.self.value.weight",
                                       "attention.self.key.bias", "attention.self.value.bias"))
                if not adjacent:
                    off = round_up(off, ALIGN)
                self.entries.append((n, p, off, p.numel(), grp))
                off += p.numel()
            off = round_up(off, ALIGN)
            if grp == 0:
                self.n_decay = off
        self.total = off
        self.p = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.g = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.m = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.mirror = torch.zeros(self.total, dtype=BF16, device=dev)
        self.off = {}
        for n, p, o, cnt, _ in self.entries:
            self.p[o:o + cnt].copy_(p.data.reshape(-1))
            p.data = self.p[o:o + cnt].view(p.shape)
            if attach_grads:
                p.grad = self.g[o:o + cnt].view(p.shape)
            self.off[n] = (o, cnt, tuple(p.shape))
        self.refresh_mirror()

    def owns_params(self):
        """False once something re-pointed a parameter's storage (another engine built over the same parameters -- the
        full model's and the bare trunk's -- or model.to / load with assign): this slab is then stale."""
        base = self.p.data_ptr()
        return all(p.data_ptr() == base + 4 * o for _, p, o, _, _ in self.entries)

    def refresh_mirror(self):
        self.mirror.copy_(self.p)
        self._versions = self._version_key()

    def _version_key(self):
        return tuple(p._version for _, p, _, _, _ in self.entries)

    def mirror_is_stale(self):
        return self._version_key() != self._versions

    def mark_fresh(self):
        self._versions = self._version_key()

    def view(self, slab, name, count=None, shape=None):
        o, cnt, shp = self.off[name]
        if count is not None:
            return slab[o:o + count].view(shape)
        return slab[o:o + cnt].view(shp)

    def reattach_grads(self):
        """optimizer.zero_grad(set_to_none=True) drops p.grad; point them back at the slab."""
        if not self.attach_grads:
            return
        for n, p, o, cnt, _ in self.entries:
            if p.grad is None or p.grad.data_ptr() != self.g.data_ptr() + 4 * o:
                p.grad = self.g[o:o + cnt].view(p.shape)


# VT_ATTN_KEEP_BITS=0: the attention backward re-derives the dropout mask from the hash instead of reading the words the
# forward wrote (same mask either way; A/B switch)
KEEP_BITS = os.environ.get("VT_ATTN_KEEP_BITS", "1") != "0"


class _TrainBuffers(object):
    """Per-(B,S) activation store and gradient scratch, plus the ctypes tables for the C loops."""

    def __init__(self, L, M, B, S, H, I, nh, dev):
        mk = lambda n: torch.empty((M, n), dtype=BF16, device=dev)
        # the residual stream at fp16 precision (ops.F16_STREAM): the pre-LayerNorm sums are fp16 tensors (the LayerNorm
        # backward reads them) and every LayerNorm output has a second, fp16 copy that the next residual add reads -- those
        # copies are transient: ONE pair of buffers serves all layers (include/visitron_hip.h, vt_layer_acts::ln1_h)
        mkpre = (lambda: torch.empty((M, H), dtype=ops.F16, device=dev)) if ops.F16_STREAM else (lambda: mk(H))
        # ops.LN_RESIDUAL: no fp16 copies at all -- the residual adds rebuild each LayerNorm from its fp16 input and the row
        # statistics its kernel wrote (4 x [M] fp32 per layer; vt_layer_acts::ln_residual_mode)
        self.ln_residual = bool(ops.LN_RESIDUAL)
        self.ln_h = (mkpre(), mkpre()) if (ops.F16_STREAM and not self.ln_residual) else None
        self.layers = []
        self.acts = (_lib.LayerActs * L)()
        for i in range(L):
            d = dict(qkv=mk(3 * H), ctx=mk(H), attn_pre=mkpre(), attn_out=mk(H), mid_pre=mk(I), mid=mk(I),
                     out_pre=mkpre(), out=mk(H), lse=torch.empty((B, nh, S), dtype=torch.float32, device=dev))
            if self.ln_h is not None:
                d["ln1_h"], d["ln2_h"] = self.ln_h
            if self.ln_residual:
                for k in ("ln1_mean", "ln1_rstd", "ln2_mean", "ln2_rstd"):
                    d[k] = torch.empty(M, dtype=torch.float32, device=dev)
                self.acts[i].ln_residual_mode = 1
            if KEEP_BITS:   # the attention dropout's keep decisions, forward -> backward (25 MB per layer at B = 256)
                d["keep_bits"] = torch.empty(ops.keep_words(B, nh, S), dtype=torch.int32, device=dev)
            self.layers.append(d)
            for k, v in d.items():
                setattr(self.acts[i], k, v.data_ptr())
        self._mk, self._ctx_raw = mk, {}
        self._H = H
        self.x0 = mk(H)
        self.x0c = mk(H)      # compacted copy of x0 (real rows only), the encoder's input in the compact path
        self.g = mk(H)
        self.g_pad = mk(H)    # dL/dx0 scattered back to the padded row order for the embedding / region backward
        self.g_seq32 = torch.empty((M, H), dtype=torch.float32, device=dev)
        self.ws_t = dict(g_pre=mk(H), g_pre2=mk(H), g_mid=mk(I), g_ctx=mk(H), g_qkv=mk(3 * H),
                         g_pre_d=mk(H), g_pre2_d=mk(H),   # dropout-masked copies (hidden dropout > 0)
                         delta=torch.empty((B, nh, S), dtype=torch.float32, device=dev),
                         ln_partial=torch.empty(ops.LN_BWD_WS_ROWS * 2 * H, dtype=torch.float32, device=dev))
        if S > 256:  # attention backward over several key blocks accumulates dQ in fp32
            self.ws_t["dq32"] = torch.empty((M, H), dtype=torch.float32, device=dev)
        self.ws = _lib.BwdWorkspace()
        for k, v in self.ws_t.items():
            setattr(self.ws, k, v.data_ptr())
        # second set of the buffers a layer's weight-gradient launch reads: with it the wgrad of layer l runs on a side
        # stream beside layer l-1's dgrad chain (vt_encoder_backward_overlap_bf16); the rest is shared
        self.ws_t_b = dict(self.ws_t)
        for k, n in (("g_pre", H), ("g_pre2", H), ("g_mid", I), ("g_qkv", 3 * H), ("g_pre_d", H), ("g_pre2_d", H)):
            self.ws_t_b[k] = mk(n)
        self.ws_b = _lib.BwdWorkspace()
        for k, v in self.ws_t_b.items():
            setattr(self.ws_b, k, v.data_ptr())


def _ctx_raw(self, layer):
    """Per-layer buffer for the attention kernel's unscaled context (head_mask in training), allocated on first use."""
    if layer not in self._ctx_raw:
        self._ctx_raw[layer] = self._mk(self._H)
    return self._ctx_raw[layer]


_TrainBuffers.ctx_raw = _ctx_raw


class _TrunkOnly(torch.nn.Module):
    """A bare BertImgModelwithLocationEmbeds presented to the engine under the parameter names it has inside a
    PreTrainOscar ("bert." + name): train.py:47 hands `model.bert` to the rollout agent, which trains through it."""

    def __init__(self, trunk):
        super().__init__()
        self.bert = trunk
        self.config = trunk.config


class _State(object):
    """The locals of a trunk forward that its backward reads (PretrainEngine._trunk_fwd / _trunk_bwd)."""

    def __init__(self, d):
        self.__dict__.update({k: v for k, v in d.items() if k != "self"})


class PretrainEngine(object):
    def __init__(self, model, lr=5e-5, weight_decay=0.05, eps=1e-8, betas=(0.9, 0.999), correct_bias=True,
                 schedule="linear", warmup_steps=0, t_total=20000, process_group=None, bucket_mb=64,
                 loss_scale_by_world=True, attach_grads=True, grad_comm_dtype=None):
        if isinstance(model, BertImgModelwithLocationEmbeds):
            model = _TrunkOnly(model)    # trunk-level training (the rollout's OscarEncoder owns just the trunk)
        assert isinstance(model, (PreTrainOscar, _TrunkOnly))
        self.has_heads = isinstance(model, PreTrainOscar)
        cfg = model.config
        if cfg.hidden_size != 64 * cfg.num_attention_heads:
            raise NotImplementedError("the HIP training path serves head size 64")
        for p_ in (cfg.hidden_dropout_prob, cfg.attention_probs_dropout_prob):
            if not 0.0 <= p_ < 1.0:
                raise ValueError("dropout probability must be in [0, 1)")
        # the attention sites run p in steps of 1 / 65536 (1 / 256 under VT_ATTN_DROPOUT_BITS=8; ops.attn_drop_p raises where a
        # p is not served): the value the
        # step actually uses is recorded in state_dict()["hyper"] and in bench.py's line, not only the configured one
        self.attention_dropout_effective = ops.attn_drop_p(float(cfg.attention_probs_dropout_prob))
        self.model, self.cfg = model, cfg
        self.flat = FlatParams(model, attach_grads=attach_grads)
        self.lr, self.wd, self.eps, self.betas, self.correct_bias = lr, weight_decay, eps, betas, correct_bias
        self.schedule, self.warmup_steps, self.t_total = schedule, warmup_steps, t_total
        self.step_count = 0       # optimizer steps taken (Adam's t)
        self.sched_step = 0       # scheduler.step() calls (LambdaLR's last_epoch)
        self.pg = process_group
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        self.bucket_elems = int(bucket_mb * 1024 * 1024 // 4)
        self.loss_scale_by_world = loss_scale_by_world
        # data-parallel gradient exchange: the reference's DDP moves 452 MB of fp32 buckets per step (pretrain.py:96-102);
        # here each range of the fp32 gradient slab is cast to a bf16 communication copy as soon as its backward kernels
        # are enqueued, THAT is all-reduced (226 MB: on xGMI a ring is bound per link), and the fused
        # AdamW reads the reduced bf16 gradients into its fp32 moments.  VT_GRAD_COMM=fp32 / grad_comm_dtype="fp32": the
        # fp32 slab itself is all-reduced.
        if grad_comm_dtype is None:
            grad_comm_dtype = os.environ.get("VT_GRAD_COMM", "bf16")
        if grad_comm_dtype not in ("bf16", "fp32"):
            raise ValueError("grad_comm_dtype must be 'bf16' or 'fp32'")
        self.grad_comm_dtype = grad_comm_dtype
        self.gemm_policy = "persistent GEMM on every CU (one rank, no collective beside it)"
        self.g16 = None
        if self.world > 1 and grad_comm_dtype == "bf16":
            self.g16 = torch.zeros(self.flat.total, dtype=BF16, device=self.flat.p.device)
        self._bufs = {}
        self._tables = None
        # dropout: masks come from a counter-based hash of (seed, site, element); the seed of a
        # forward/backward pair = base (torch's seed, decorrelated per rank) + number of pairs so far
        rank = torch.distributed.get_rank(process_group) if self.world > 1 else 0
        self.drop_seed_base = (int(torch.initial_seed()) + 0x9E3779B97F4A7C15 * (rank + 1)) & 0xFFFFFFFFFFFFFFFF
        self.fb_count = 0
        self.last_drop_seed = 0
        self._wt_dirty = True
        self._wt_batch = None
        # weight gradients on a side stream beside the next layer's dgrad chain: opt-in (VT_OVERLAP_WGRAD=1 or the
        # attribute).  Measured on one GPU: +1 % when the GEMMs are the one-tile-per-workgroup kernels, -1 % with the
        # persistent kernel (its workgroups queue behind the wgrad's and still do a full share each); with a second
        # process on the same GPU it lost a lot, and beside RCCL's kernels it could not be measured here.
        self.overlap_wgrad = os.environ.get("VT_OVERLAP_WGRAD", "0") != "0"
        # one rank: AdamW per parameter range on a side stream under the backward (_train_step_adamw_under_backward)
        self.overlap_adamw = os.environ.get("VT_OVERLAP_ADAMW", "0") != "0"
        self._adam_stream = None
        # the training step on the real rows only (no padding rows; vt_encoder_*_seq_bf16): same losses and gradients
        # as the padded run (tests/test_gpu_train.py); VT_COMPACT_ROWS=0 or the attribute turns it off
        self.compact_rows = os.environ.get("VT_COMPACT_ROWS", "1") != "0"
        self.last_rows = None
        self.last_layout = None
        self._tuned_rows = set()
        # (padded) token rows below which the step stays on the padded layout.  Rounds 2 - 5: 16 384 -- the layout then cost
        # two launches and a host synchronisation of its own.  Since the row counts and lists ride on the step's one
        # synchronisation it wins at every batch measured (round 6, profiles/r06/compaction_by_batch.txt: B = 4 ... 36 x 228
        # +0.2 ... +3.6 %, 8 x 767 +3.2 %), so the default is 0; VT_COMPACT_MIN_ROWS restores a threshold
        self.compact_min_rows = int(os.environ.get("VT_COMPACT_MIN_ROWS", "0"))
        self._side_stream = None
        self._fwd_serial = 0      # forwards issued: a backward must belong to the latest one (the buffers are shared)
        self._build_tables()

    # ------------------------------------------------------------------------------ tables
    def _build_tables(self):
        self._wt_batch = None
        m, f, cfg = self.model, self.flat, self.cfg
        L, H, I = cfg.num_hidden_layers, cfg.hidden_size, cfg.intermediate_size
        self.w_tab = (_lib.LayerWeights * L)()
        self.wt_tab = (_lib.LayerWeightsT * L)()
        self.g_tab = (_lib.LayerGrads * L)()
        self._keep, self.wt = [], []
        self.layer_ranges = []   # per layer: [(start, end) in the decay region, (start, end) in the no-decay region]
        for i in range(L):
            pre = "bert.encoder.layer.%d." % i
            offs = [(o, o + c) for n_, _, o, c, _ in f.entries if n_.startswith(pre)]
            dec = [r_ for r_ in offs if r_[0] < f.n_decay]
            nod = [r_ for r_ in offs if r_[0] >= f.n_decay]
            self.layer_ranges.append([(min(a_ for a_, _ in dec), max(b_ for _, b_ in dec)),
                                      (min(a_ for a_, _ in nod), max(b_ for _, b_ in nod))])
            qw, qb = pre + "attention.self.query.weight", pre + "attention.self.query.bias"
            names = dict(
                w_ao=pre + "attention.output.dense.weight", b_ao=pre + "attention.output.dense.bias",
                ln1_g=pre + "attention.output.LayerNorm.weight", ln1_b=pre + "attention.output.LayerNorm.bias",
                w_in=pre + "intermediate.dense.weight", b_in=pre + "intermediate.dense.bias",
                w_out=pre + "output.dense.weight", b_out=pre + "output.dense.bias",
                ln2_g=pre + "output.LayerNorm.weight", ln2_b=pre + "output.LayerNorm.bias")
            t = dict(w_qkv=f.view(f.mirror, qw, 3 * H * H, (3 * H, H)), b_qkv=f.view(f.p, qb, 3 * H, (3 * H,)))
            gr = dict(d_w_qkv=f.view(f.g, qw, 3 * H * H, (3 * H, H)), d_b_qkv=f.view(f.g, qb, 3 * H, (3 * H,)))
            for k, n in names.items():
                t[k] = f.view(f.mirror if k.startswith("w_") else f.p, n)
                gr["d_" + k] = f.view(f.g, n)
            wt = dict(wt_qkv=torch.empty((H, 3 * H), dtype=BF16, device=f.p.device),
                      wt_ao=torch.empty((H, H), dtype=BF16, device=f.p.device),
                      wt_in=torch.empty((H, I), dtype=BF16, device=f.p.device),
                      wt_out=torch.empty((I, H), dtype=BF16, device=f.p.device))
            self._keep.append((t, gr))
            self.wt.append((t, wt))
            for k, v in t.items():
                setattr(self.w_tab[i], k, v.data_ptr())
            for k, v in gr.items():
                setattr(self.g_tab[i], k, v.data_ptr())
            for k, v in wt.items():
                setattr(self.wt_tab[i], k, v.data_ptr())
        # non-encoder weights: bf16 mirror views and transposed copies
        dev = f.p.device
        self.head_t = dict(pool=torch.empty((H, H), dtype=BF16, device=dev))
        if self.has_heads:
            V = m.mlmhead.predictions.decoder.weight.shape[0]
            C = m.token_head[0].weight.shape[0]
            A = m.next_action.linear.weight.shape[0]
            self.Vp, self.Cp, self.Ap = round_up(V, 64), round_up(C, 64), round_up(A, 64)
            self.head_t.update(
                dec=torch.zeros((H, self.Vp), dtype=BF16, device=dev), tr=torch.empty((H, H), dtype=BF16, device=dev),
                tok=torch.zeros((H, self.Cp), dtype=BF16, device=dev), act=torch.zeros((H, self.Ap), dtype=BF16, device=dev))
        D = m.bert.img_dim
        self.kpad = round_up(D + 128, 64)
        self.w_img = torch.zeros((H, self.kpad), dtype=BF16, device=dev)
        self.b_img = torch.zeros(H, dtype=torch.float32, device=dev)
        self.dw_img = torch.zeros((H, self.kpad), dtype=torch.float32, device=dev)
        self.db_img = torch.zeros(H, dtype=torch.float32, device=dev)

    def _name_of(self, param):
        for n, p, _, _, _ in self.flat.entries:
            if p is param:
                return n
        raise KeyError("parameter not in the flat layout")

    def _mirror(self, param):
        return self.flat.view(self.flat.mirror, self._name_of(param))

    def _grad(self, param):
        return self.flat.view(self.flat.g, self._name_of(param))

    def refresh_derived_weights(self):
        """Transposed / packed bf16 copies that the dgrad GEMMs and the region projection read."""
        if self.flat.mirror_is_stale():
            self.flat.refresh_mirror()
            self._wt_dirty = True
        if not self._wt_dirty:
            return
        m, D = self.model, self.model.bert.img_dim
        if self._wt_batch is None:   # all layers' four matrices + the two square head matrices: one launch
            pairs = []
            for t, wt in self.wt:
                pairs += [(t["w_qkv"], wt["wt_qkv"]), (t["w_ao"], wt["wt_ao"]), (t["w_in"], wt["wt_in"]),
                          (t["w_out"], wt["wt_out"])]
            pairs += [(self._mirror(m.bert.pooler.dense.weight), self.head_t["pool"])]
            if self.has_heads:
                pairs += [(self._mirror(m.mlmhead.predictions.transform.dense.weight), self.head_t["tr"]),
                          (self._mirror(m.mlmhead.predictions.decoder.weight), self.head_t["dec"]),
                          (self._mirror(m.token_head[0].weight), self.head_t["tok"]),
                          (self._mirror(m.next_action.linear.weight), self.head_t["act"])]
            self._wt_batch = ops.TransposeBatch(pairs)
        self._wt_batch.run()
        self.w_img[:, :D].copy_(self._mirror(m.bert.img_embedding.weight))
        self.w_img[:, D:D + 128].copy_(self._mirror(m.bert.location_embeds.weight))
        torch.add(m.bert.img_embedding.bias.detach(), m.bert.location_embeds.bias.detach(), out=self.b_img)
        self._wt_dirty = False

    def _buffers(self, B, S):
        key = (B, S)
        b = self._bufs.get(key)
        if b is None:
            cfg = self.cfg
            if len(self._bufs) >= 2:
                self._bufs.clear()
            b = _TrainBuffers(cfg.num_hidden_layers, B * S, B, S, cfg.hidden_size, cfg.intermediate_size,
                              cfg.num_attention_heads, self.flat.p.device)
            if self.world > 1 or os.environ.get("VT_FORCE_MULTI_RANK_GEMM") == "1":
                # collectives share the CUs with the backward: ops.multi_rank_gemm_policy decides once whether the persistent
                # GEMM keeps running (on CUs - k workgroups: opt-in, VT_GEMM_RESERVE_CUS=k) or stands down for the
                # one-tile-per-workgroup kernels (default); bench.py prints the choice.  VT_FORCE_MULTI_RANK_GEMM=1 applies the
                # policy on ONE rank: the profiled single-rank twin of each multi-GPU kernel mix (profiles/r05/bench_b36_*.json)
                k, self.gemm_policy = ops.multi_rank_gemm_policy()
                if k > 0:
                    _lib.load().vt_gemm_reserve_cus(k)
                else:
                    ops.PERSISTENT_GEMM_OK = False
            ops.autotune_encoder_shapes(B * S, cfg.hidden_size, cfg.intermediate_size, training=True,
                                        device=self.flat.p.device)
            self._bufs[key] = b
        return b

    # ------------------------------------------------------------------------------ schedule
    def lr_factor(self):
        s = self.sched_step
        if self.schedule == "constant":
            return float(s) / float(max(1.0, self.warmup_steps)) if s < self.warmup_steps else 1.0
        if s < self.warmup_steps:
            return float(s) / float(max(1, self.warmup_steps))
        return max(0.0, float(self.t_total - s) / float(max(1.0, self.t_total - self.warmup_steps)))

    # ------------------------------------------------------------------------------ forward + backward
    def _trunk_fwd(self, batch, head_mask, labels, token_labels, training, allow_compact, comm=None):
        """The trunk's forward (embeddings, region projection, encoder layers, pooler) with every activation the backward
        needs kept in the (B, S) buffer set; returns the step's state (a namespace of the locals below).  labels /
        token_labels (optional): the supervised rows are located here, where the host synchronises anyway."""
        m, cfg, f = self.model, self.cfg, self.flat
        self._require_ownership()
        f.reattach_grads()
        ids = _i64(batch["input_ids"])
        dev = ids.device
        B, T = ids.shape
        img = batch.get("img_feats")
        R = 0 if img is None else img.shape[1]
        S, H, I, nh, L = T + R, cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.num_hidden_layers
        M = B * S
        am = batch.get("attention_mask")
        mask = None if am is None else am.to(torch.float32).contiguous()
        mask_additive = False
        if mask is not None and mask.dim() == 3:   # per-query mask -> additive bias [B, S, S] (encoder.py:228-229, :238-241)
            if mask.shape != (B, S, S):
                raise RuntimeError("3-D attention_mask must be [batch, text+region, text+region]")
            mask = ((1.0 - mask) * -10000.0).contiguous()
            mask_additive = True
        elif mask is not None and mask.shape != (B, S):
            raise RuntimeError("attention_mask must be [batch, text+region]")
        from .modeling import _centered_mask, _head_scale
        if mask is not None and not mask_additive:
            mask = _centered_mask(mask)   # one constant per sequence off the additive bias: same softmax, well-scaled numbers
        hs = _head_scale(head_mask, cfg.num_hidden_layers, cfg.num_attention_heads, ids.device)
        if hs is not None and comm is not None:
            raise NotImplementedError("head_mask together with the chunked data-parallel backward")
        tt, pos_ids = _i64(batch.get("token_type_ids")), _i64(batch.get("position_ids"))
        bufs = self._buffers(B, S)
        emb = m.bert.embeddings
        eps = emb.LayerNorm.variance_epsilon
        # dropout (nn.Dropout follows the module's training flag): same (p, seed) in forward and backward
        p_h = float(cfg.hidden_dropout_prob) if training else 0.0
        p_a = float(cfg.attention_probs_dropout_prob) if training else 0.0
        seed = (self.drop_seed_base + self.fb_count) & 0xFFFFFFFFFFFFFFFF
        self.fb_count += 1
        self.last_drop_seed = seed
        dp_kw = dict(p_hidden=p_h, p_attn=p_a, drop_seed=seed)
        # the text embeddings go first: their out-of-range flag is then ready when the host synchronises below
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        ops.embed_layernorm(ids, tt, pos_ids, emb.word_embeddings.weight.detach(), emb.position_embeddings.weight.detach(),
                            emb.token_type_embeddings.weight.detach(), emb.LayerNorm.weight.detach(),
                            emb.LayerNorm.bias.detach(), eps, bufs.x0, S, err_flag=err, drop=(p_h, seed, ops.SITE_EMB))
        # Everything the host must know about this batch before it can size the step -- the numbers of supervised rows
        # (the heads run on those only), the number of rows with a non-zero mask and whether the batch qualifies for the
        # compacted layout, the embedding kernel's out-of-range flag -- is reduced on the device and read back in ONE
        # synchronisation, at the start of the step (ops.batch_row_counts); the row lists and the compacted layout are then
        # written to their known sizes by a second launch (ops.batch_row_lists).
        lab = tl = idx_w = idx_t = None
        Ml = Mt = 0
        if labels is not None:
            lab, tl = _i64(labels.reshape(-1)), _i64(token_labels.reshape(-1))
        # rows nothing in the step reads: positions with attention mask 0.  As keys they weigh exactly 0, so their
        # hidden states reach no loss and their gradient is exactly 0 -- drop them from every row-wise kernel.  Needs a
        # 0/1 mask, the [CLS] position and every supervised position kept; otherwise the padded path below.
        try_compact = (allow_compact and self.compact_rows and mask is not None and mask.dim() == 2 and hs is None
                       and M >= self.compact_min_rows)
        cmask = mask.contiguous() if try_compact else None
        early = os.environ.get("VT_STEP_OVERLAP_READBACK", "1") == "0"   # A/B switch: the transposes first, as before round 6
        if early:
            self.refresh_derived_weights()
        counts = ops.batch_row_counts_begin(lab, tl, cmask, err, B, S)   # the step's one host synchronisation ...
        # ... and while its seven numbers travel to the host on a side stream, the device transposes the step's weights (the
        # copies the dgrad GEMMs read: ~95 us that used to run BEFORE the counting kernel, with the GPU idle behind it for the
        # 250 us the host needed for this read-back, three more blocking 4-byte reads and the launches that follow)
        if not early:
            self.refresh_derived_weights()
        vals, tiles = ops.batch_row_counts_end(counts)
        # out-of-range input_ids / position_ids / token_type_ids: the reference's embedding lookup raises IndexError
        # (checked before anything indexes the gradient tables with those ids)
        if vals[0] != 0:
            raise IndexError("index out of range in BertEmbeddings (input_ids / position_ids / token_type_ids)")
        # the previous step's weight gradients: a workgroup of the persistent wgrad kernel that gave up its bounded wait
        # for a dW tile added out of turn (never seen; a preempted / shared GPU could do it) -- fail loudly
        late = int(vals[5])       # (vt_step_counters: read and cleared on the device, delivered with the row counts)
        if late:
            raise RuntimeError("vt_wgrad_bf16: %d workgroup(s) ran out of their turn wait in an earlier launch; the weight "
                               "gradients of that step are unreliable" % late)
        late = int(vals[6])       # the same for the persistent GEMM's shared tiles (kernel variants 28 .. 32)
        if late:
            raise RuntimeError("vt_linear: %d finishing workgroup(s) of shared GEMM tiles ran out of their wait in an earlier "
                               "launch; that step's activations / gradients are unreliable" % late)
        Ml, Mt = (int(vals[1]), int(vals[2])) if labels is not None else (0, 0)
        compact = try_compact and int(vals[3]) < M and not vals[4]
        lay = None
        if labels is not None or compact:   # one launch: the supervised-row lists and the compacted layout
            idx_w, idx_t, lay = ops.batch_row_lists(lab, tl, cmask if compact else None, B, S, Ml, Mt, int(vals[3]), tiles)
        Mr = M if lay is None else lay.rows
        self.last_rows = Mr
        self.last_layout = lay
        if lay is not None:
            # the tile quantisation changes with the row count: tune once per bucket of 256 rows (512 below 16 384 rows), at
            # the bucket's upper edge; the library then takes the nearest tuned M.  Rounds 2 - 5 used 2048-row buckets, whose
            # upper edge is another tile count altogether: 7 091 real rows of a B = 36 batch are 28 row tiles of 256, 8 192 are
            # 32 (QKV: 252 tiles = one round against 288 = two); 50 845 rows of the B = 256 batch are 199, 51 200 are 200 (QKV:
            # 1 791 tiles = 7.00 rounds against 1 800 = 7.03, i.e. eight).  profiles/r06/tune_bucket_ab.txt: +1.9 % / +0.6 %.
            bucket = round_up(Mr, int(os.environ.get("VT_TUNE_BUCKET_LARGE", "256")) if Mr >= 16384
                              else int(os.environ.get("VT_TUNE_BUCKET_SMALL", "512")))
            if bucket not in self._tuned_rows and bucket < M:
                ops.autotune_encoder_shapes(bucket, H, I, training=True, device=dev)
                self._tuned_rows.add(bucket)
        # the MLM head's two wide GEMMs ([Ml, 768] x [768, 30528] and back) are tuned like the encoder's, once per bucket of
        # 512 supervised rows (the row count changes from batch to batch; the library takes the nearest tuned M)
        if Ml:
            hb = round_up(Ml, 512)
            if ("head", hb) not in self._tuned_rows:
                ops.autotune_linear(hb, self.Vp, H, device=dev, out_f32=True)
                ops.autotune_linear(hb, H, self.Vp, device=dev)
                self._tuned_rows.add(("head", hb))
        rows_w = rows_t = None
        if labels is not None:   # supervised rows in the layout in use
            rows_w = idx_w if lay is None else lay.inverse.index_select(0, idx_w)
            rows_t = idx_t if lay is None else lay.inverse.index_select(0, idx_t)
        cls_rows = (torch.arange(B, device=dev) * S) if lay is None else lay.start.to(torch.int64)

        # ---------------- forward ----------------
        x0 = bufs.x0
        a_img = None
        if img is not None:
            a_img = ops.pack_concat(img.reshape(B * R, -1).float().contiguous(),
                                    batch["img_location_embeddings"].reshape(B * R, -1).float().contiguous(), self.kpad)
            if getattr(m.bert, "use_img_layernorm", None):
                # encoder.py:280-284: LayerNorm on the summed image embedding, THEN dropout; the pre-LayerNorm rows are
                # kept (compact [B*R, H]) for the backward pass
                ln = m.bert.LayerNorm
                img_pre = ops.linear(a_img, self.w_img, self.b_img)
                img_ln = ops.layernorm(img_pre, ln.weight.detach(), ln.bias.detach(), ln.variance_epsilon)
                if p_h > 0.0:
                    ops.apply_dropout(img_ln, (p_h, seed, ops.SITE_IMG))
                x0.view(B, S, H)[:, T:].copy_(img_ln.view(B, R, H))
            else:
                img_pre = None
                ops.linear(a_img, self.w_img, self.b_img, out=x0[T:], ldc=H, grp_rows=R, grp_stride=S,
                           drop=(p_h, seed, ops.SITE_IMG))
        x_enc, enc_mask = x0, mask
        if lay is not None:
            x_enc, enc_mask = bufs.x0c[:Mr], None
            torch.index_select(x0, 0, lay.index, out=x_enc)
        if ops.profiling() or hs is not None:
            self._encoder_forward_unrolled(bufs, x_enc, enc_mask, B, S, p_h, p_a, seed, lay, mask_additive, hs)
        else:
            ops.encoder_forward(self.w_tab, bufs.acts, x_enc, enc_mask, mask_additive, None, B, S, H, nh, I,
                                cfg.layer_norm_eps, seq=lay, **dp_kw)
        seq = bufs.layers[-1]["out"][:Mr]
        if lay is None:
            cls_seq = None
            pooled = ops.linear(seq, self._mirror(m.bert.pooler.dense.weight), m.bert.pooler.dense.bias.detach(),
                                act=ACT_TANH, out_f32=True, M=B, lda=S * H)
        else:
            cls_seq = seq.index_select(0, cls_rows)
            pooled = ops.linear(cls_seq, self._mirror(m.bert.pooler.dense.weight), m.bert.pooler.dense.bias.detach(),
                                act=ACT_TANH, out_f32=True)
        pooled_bf = pooled.to(BF16)

        self._fwd_serial += 1
        st = _State(locals())
        st.serial = self._fwd_serial
        return st

    def forward_backward(self, batch, grad_scale=1.0, accumulate=False, comm=None, head_mask=None, backward=True):
        """One forward + backward; gradients of `grad_scale * loss` land in the flat slab (p.grad).
        Returns the reference's 7-tuple (0-d fp32 tensors).  attention_mask: [B, S] (any numeric values, the reference's
        (1 - m) * -10000 arithmetic) or the reference's 3-D form [B, S, S] (encoder.py:228-229); head_mask as the
        reference takes it (encoder.py:248-265; the layer loop then runs op by op: the head scaling sits between the
        attention kernel and the output projection in both directions).  backward=False: the forward and its 7-tuple only."""
        st = self._trunk_fwd(batch, head_mask, batch["labels"], batch["token_labels"], self.model.training, True, comm)
        m, cfg, f, dev, emb, bufs = self.model, self.cfg, self.flat, st.dev, st.emb, st.bufs
        B, S, H, Mr, lay = st.B, st.S, st.H, st.Mr, st.lay
        seq, pooled, pooled_bf, cls_seq, cls_rows = st.seq, st.pooled, st.pooled_bf, st.cls_seq, st.cls_rows
        lab, tl, idx_w, idx_t, rows_w, rows_t, Ml, Mt = st.lab, st.tl, st.idx_w, st.idx_t, st.rows_w, st.rows_t, st.Ml, st.Mt
        next_action = batch.get("next_action")

        # heads on supervised rows only
        V, C, A = cfg.vocab_size, cfg.detector_classes, cfg.action_space
        pr = m.mlmhead.predictions
        zero = torch.zeros((), dtype=torch.float32, device=dev)
        if Ml > 0:
            seq_w = seq.index_select(0, rows_w)
            y_w = lab.index_select(0, idx_w)
            h_t = torch.empty((Ml, H), dtype=BF16, device=dev)
            t1 = ops.linear(seq_w, self._mirror(pr.transform.dense.weight), pr.transform.dense.bias.detach(), act=ACT_GELU,
                            pre_act_out=h_t)
            t2 = ops.layernorm(t1, pr.transform.LayerNorm.weight.detach(), pr.transform.LayerNorm.bias.detach(),
                               pr.transform.LayerNorm.variance_epsilon)
            logits = torch.empty((Ml, self.Vp), dtype=torch.float32, device=dev)
            ops.linear(t2, self._mirror(pr.decoder.weight), pr.bias.detach(), out=logits, out_f32=True)
            # fused CE: per-row loss, argmax and the gradient (softmax - onehot) * grad_scale / Ml in one kernel
            dl = torch.empty((Ml, self.Vp), dtype=BF16, device=dev)
            loss_rows, amax_w = ops.ce_softmax_rows(logits, y_w, V, dl, float(grad_scale) / Ml)
            mask_loss = loss_rows.mean()
            words_acc = (amax_w == y_w).sum().float() / Ml
        else:
            mask_loss = zero / zero  # CrossEntropyLoss over no valid target is nan, as in the reference
            words_acc = zero / zero
        lin_tok = m.token_head[0]
        if Mt > 0:
            seq_t = seq.index_select(0, rows_t)
            y_t = tl.index_select(0, idx_t)
            lt = torch.empty((Mt, self.Cp), dtype=torch.float32, device=dev)
            ops.linear(seq_t, self._mirror(lin_tok.weight), lin_tok.bias.detach(), out=lt, out_f32=True)
            # token_head = Linear + Softmax and the criterion applies log-softmax AGAIN: loss, argmax and the gradient
            # through both softmaxes in one kernel
            dlt = torch.empty((Mt, self.Cp), dtype=BF16, device=dev)
            tok_rows, amax_t = ops.ce_double_softmax_rows(lt, y_t, C, dlt, float(grad_scale) / Mt)
            token_loss = tok_rows.mean()
            token_acc = (amax_t == y_t).sum().float() / Mt
        else:
            token_loss = zero / zero
            token_acc = zero / zero
        la = torch.empty((B, self.Ap), dtype=torch.float32, device=dev)
        ops.linear(pooled_bf, self._mirror(m.next_action.linear.weight), m.next_action.linear.bias.detach(), out=la,
                   out_f32=True)
        dla_bf = None
        if next_action is not None:
            # LogSoftmax head under a CrossEntropyLoss that applies log_softmax again (encoder.py:142-151, 387-391): loss,
            # accuracy and d(grad_scale * loss)/d(logits) in one launch
            next_loss, action_acc, dla_bf = ops.action_head(la, _i64(next_action), A, float(grad_scale), self.Ap)
        else:
            next_loss, action_acc = 0, 0
        loss = mask_loss + next_loss + token_loss
        if not backward:   # train() under torch.no_grad(): the forward with its dropout, nothing else
            return (loss, mask_loss, next_loss, token_loss, words_acc, action_acc, token_acc)

        # ---------------- backward ----------------
        gs = float(grad_scale)
        acc = bool(accumulate)
        # dL/d(sequence output): zero except at the supervised / [CLS] rows.  Assembled directly in the bf16 buffer the
        # encoder backward reads (a row normally belongs to one head; where two heads meet the sum is formed in bf16)
        g16 = bufs.g[:Mr]
        g16.zero_()
        wg = lambda dy, x, dw, db: dict(dy=dy, x=x, dw=dw, db=db, accumulate=acc)
        dec_w_is_tied = pr.decoder.weight is emb.word_embeddings.weight
        word_grad = self._grad(emb.word_embeddings.weight)
        # The word table's gradient receives the embedding backward's run sums later on (_trunk_bwd), so it must hold
        # defined values before that: with a tied decoder and supervised rows the decoder's weight gradient -- the first
        # thing written into it -- simply overwrites all of it (no 94 MB fill, no read-back by an accumulating launch)
        dec_overwrites = dec_w_is_tied and Ml > 0 and pr.decoder.weight.shape[0] == emb.word_embeddings.weight.shape[0]
        if not acc and not dec_overwrites:
            word_grad.zero_()
            if dec_w_is_tied:
                self._grad(pr.bias).zero_()  # shares the accumulate flag of the tied decoder weight below
        if Ml > 0:
            dec_grad = self._grad(pr.decoder.weight)
            ops.wgrad([dict(dy=dl[:, :V], x=t2, dw=dec_grad, db=self._grad(pr.bias),
                            accumulate=acc or (dec_w_is_tied and not dec_overwrites))], Ml)
            # d(logits)[Ml, Vp] . W_dec[Vp, H]: 51 output tiles for 477 K-steps -- the K-steps are split over the chip
            ks = ops.splitk_for(Ml, H, self.Vp)
            g_t2 = ops.linear_splitk(dl, self.head_t["dec"], ks) if ks else ops.linear(dl, self.head_t["dec"])
            g_t1 = ops.layernorm_bwd(t1, g_t2, pr.transform.LayerNorm.weight.detach(), pr.transform.LayerNorm.variance_epsilon,
                                     self._grad(pr.transform.LayerNorm.weight), self._grad(pr.transform.LayerNorm.bias),
                                     ws=bufs.ws_t["ln_partial"], accumulate=acc)
            g_ht = ops.dgelu_mul(g_t1, h_t)
            ops.wgrad([wg(g_ht, seq_w, self._grad(pr.transform.dense.weight), self._grad(pr.transform.dense.bias))], Ml)
            g16.index_add_(0, rows_w, ops.linear(g_ht, self.head_t["tr"]))
        elif not acc:
            # no supervised MLM row (the loss is NaN, as the reference's): the head's gradients must not keep the
            # previous step's values (the tied decoder weight / bias were zeroed above)
            for prm in (pr.transform.dense.weight, pr.transform.dense.bias, pr.transform.LayerNorm.weight,
                        pr.transform.LayerNorm.bias, pr.bias) + (() if dec_w_is_tied else (pr.decoder.weight,)):
                self._grad(prm).zero_()
        if Mt > 0:
            ops.wgrad([wg(dlt[:, :C], seq_t, self._grad(lin_tok.weight), self._grad(lin_tok.bias))], Mt)
            g16.index_add_(0, rows_t, ops.linear(dlt, self.head_t["tok"]))
        elif not acc:
            self._grad(lin_tok.weight).zero_()
            self._grad(lin_tok.bias).zero_()
        if next_action is not None:
            ops.wgrad([wg(dla_bf[:, :A], pooled_bf, self._grad(m.next_action.linear.weight),
                          self._grad(m.next_action.linear.bias))], B)
            g_pooled = ops.linear(dla_bf, self.head_t["act"], out_f32=True)
            g_z = (g_pooled * (1.0 - pooled * pooled)).to(BF16)
            ops.wgrad([dict(dy=g_z, x=seq.view(B, S * H)[:, :H] if lay is None else cls_seq,
                            dw=self._grad(m.bert.pooler.dense.weight), db=self._grad(m.bert.pooler.dense.bias),
                            accumulate=acc)], B)
            g16.index_add_(0, cls_rows, ops.linear(g_z, self.head_t["pool"]))
        elif not acc:
            for prm in (m.next_action.linear.weight, m.next_action.linear.bias, m.bert.pooler.dense.weight,
                        m.bert.pooler.dense.bias):
                self._grad(prm).zero_()
        if comm is not None:
            # data-parallel: the head / pooler gradients are final here, before the encoder backward has started -- their
            # all-reduce goes out first and hides under the whole backward (round 3 reduced them last, with the embeddings)
            rng = self._param_ranges([pr.transform.dense.weight, pr.transform.dense.bias, pr.transform.LayerNorm.weight,
                                      pr.transform.LayerNorm.bias, pr.bias, lin_tok.weight, lin_tok.bias,
                                      m.next_action.linear.weight, m.next_action.linear.bias, m.bert.pooler.dense.weight,
                                      m.bert.pooler.dense.bias] + ([] if dec_w_is_tied else [pr.decoder.weight]))
            comm["launch"](rng)
            comm["done"].extend(rng)
        self._trunk_bwd(st, None, acc, comm, word_grad_ready=True)
        return (loss, mask_loss, next_loss, token_loss, words_acc, action_acc, token_acc)

    def _param_ranges(self, params):
        """Slab ranges [start, end) of the given parameters, ends rounded up to the alignment granule (the padding belongs
        to no parameter), sorted and merged where they touch."""
        f, spans = self.flat, []
        for prm in params:
            o, cnt, _ = f.off[self._name_of(prm)]
            spans.append((o, min(round_up(o + cnt, ALIGN), f.total)))
        out = []
        for s_, e_ in sorted(spans):
            if out and s_ <= out[-1][1]:
                out[-1] = (out[-1][0], max(out[-1][1], e_))
            else:
                out.append((s_, e_))
        return out

    def _trunk_bwd(self, st, g32, acc, comm=None, word_grad_ready=False, d_hidden=None):
        """Back through the trunk: g32 fp32 [rows, H] = dL/d(sequence output) in the layout of the forward (st), or None
        when the caller has already put it (bf16) into bufs.g; the gradients of the encoder layers, the embeddings and the
        region projection land in the flat slab."""
        if st.serial != self._fwd_serial:
            raise RuntimeError("the activations of this forward were overwritten by a later forward of the same engine; "
                               "run backward before the next forward")
        m, cfg, f, emb, bufs = self.model, self.cfg, self.flat, st.emb, st.bufs
        B, T, R, S, H, I, nh, L, Mr, lay = st.B, st.T, st.R, st.S, st.H, st.I, st.nh, st.L, st.Mr, st.lay
        hs, x_enc, enc_mask, mask_additive, dp_kw = st.hs, st.x_enc, st.enc_mask, st.mask_additive, st.dp_kw
        p_h, p_a, seed, ids, tt, pos_ids, eps = st.p_h, st.p_a, st.seed, st.ids, st.tt, st.pos_ids, st.eps
        img, img_pre, a_img, batch = st.img, st.img_pre if st.img is not None else None, st.a_img, st.batch
        word_grad = self._grad(emb.word_embeddings.weight)
        if not acc and not word_grad_ready:
            word_grad.zero_()
        g = bufs.g[:Mr]
        if g32 is not None:
            g.copy_(g32)
        if d_hidden is not None and comm is not None:
            raise NotImplementedError("gradients of intermediate hidden states together with the chunked data-parallel backward")
        if ops.profiling() or hs is not None:
            self._encoder_backward_unrolled(bufs, x_enc, enc_mask, g, B, S, acc, p_h, p_a, seed, lay, mask_additive, hs,
                                            inject=d_hidden)
        elif d_hidden is not None:
            # a caller's gradients with respect to intermediate hidden states (output_hidden_states in trunk-level training,
            # oscar/modeling_bert.py:146-158): the C loop one layer at a time, d_hidden[l + 1] (the output of layer l; bf16
            # rows in the forward's layout) added to the running gradient in front of layer l's backward, d_hidden[0] (the
            # embedding output) behind layer 0's.  (d_hidden[L] is part of g already.)
            for lo in range(L - 1, -1, -1):
                if lo + 1 < L and d_hidden[lo + 1] is not None:
                    g.add_(d_hidden[lo + 1])
                sub = lambda arr, typ: (typ * 1).from_address(ctypes.addressof(arr) + lo * ctypes.sizeof(typ))
                x_in = x_enc if lo == 0 else bufs.layers[lo - 1]["out"]
                ops.encoder_backward(sub(self.w_tab, _lib.LayerWeights), sub(self.wt_tab, _lib.LayerWeightsT),
                                     sub(bufs.acts, _lib.LayerActs), sub(self.g_tab, _lib.LayerGrads), x_in, enc_mask, mask_additive,
                                     g, bufs.ws, B, S, H, nh, I, cfg.layer_norm_eps, accumulate=acc, layer0=lo, seq=lay,
                                     **dp_kw, **self._overlap_kw(bufs))
            if d_hidden[0] is not None:
                g.add_(d_hidden[0])
        elif comm is None:
            ops.encoder_backward(self.w_tab, self.wt_tab, bufs.acts, self.g_tab, x_enc, enc_mask, mask_additive, g, bufs.ws, B, S,
                                 H, nh, I, cfg.layer_norm_eps, accumulate=acc, seq=lay, **dp_kw, **self._overlap_kw(bufs))
        else:
            # data-parallel: backward in layer chunks (last layers first); as soon as a chunk's kernels are
            # enqueued its gradient ranges are all-reduced on the communicator's stream, under the backward
            # of the earlier layers
            step = max(1, int(comm["layers_per_chunk"]))
            hi = L
            while hi > 0:
                lo = max(0, hi - step)
                n = hi - lo
                sub = lambda arr, typ: (typ * n).from_address(ctypes.addressof(arr) + lo * ctypes.sizeof(typ))
                x_in = x_enc if lo == 0 else bufs.layers[lo - 1]["out"]
                ops.encoder_backward(sub(self.w_tab, _lib.LayerWeights), sub(self.wt_tab, _lib.LayerWeightsT),
                                     sub(bufs.acts, _lib.LayerActs), sub(self.g_tab, _lib.LayerGrads), x_in, enc_mask, mask_additive,
                                     g, bufs.ws, B, S, H, nh, I, cfg.layer_norm_eps, accumulate=acc, layer0=lo, seq=lay,
                                     **dp_kw, **self._overlap_kw(bufs))
                # (range ends rounded up to the slab's alignment granule: the padding belongs to no parameter)
                rng = [(self.layer_ranges[lo][k][0], min(round_up(self.layer_ranges[hi - 1][k][1], ALIGN), f.total))
                       for k in (0, 1)]
                comm["launch"](rng)
                comm["done"].extend(rng)
                hi = lo
        if lay is not None:   # dL/dx0 back in the padded row order (zero at the padding rows, as in the padded run)
            # gathers: padded row -> its compact row, or the zero row kept behind the compact rows (Mr < M here); the text
            # and the region positions land in two contiguous blocks (what the embedding and the region backward read)
            bufs.g[Mr].zero_()
            gi_text, gi_reg = lay.gather_index_split(Mr, T)
            g_text = torch.index_select(bufs.g[:Mr + 1], 0, gi_text, out=bufs.g_pad[:B * T])
            g_reg = torch.index_select(bufs.g[:Mr + 1], 0, gi_reg, out=bufs.g_pad[B * T:B * S]) if img is not None else None
            g_pitch = T
        else:
            g_text, g_reg, g_pitch = g, None, S
        # embeddings: text rows
        de = ops.embed_layernorm_bwd(ids, tt, pos_ids, emb.word_embeddings.weight.detach(),
                                     emb.position_embeddings.weight.detach(), emb.token_type_embeddings.weight.detach(),
                                     emb.LayerNorm.weight.detach(), eps, g_text, g_pitch, self._grad(emb.LayerNorm.weight),
                                     self._grad(emb.LayerNorm.bias), ws=bufs.ws_t["ln_partial"], accumulate=acc,
                                     drop=(p_h, seed, ops.SITE_EMB))
        pos_grad, type_grad = self._grad(emb.position_embeddings.weight), self._grad(emb.token_type_embeddings.weight)
        if not acc:
            pos_grad.zero_()
            type_grad.zero_()
        # the three table gradients without float atomics (sorted ids, one workgroup per run: ops.embed_table_grad); the
        # default position / type ids need no lookup at all: position t collects the batch's rows t, type 0 everything
        ops.embed_table_grad(ids.reshape(-1), de, word_grad, skip_id=emb.word_embeddings.padding_idx)
        pos_sum = None
        if pos_ids is None:
            pos_sum = de.view(B, T, H).sum(0)
            pos_grad[:T].add_(pos_sum)
        else:
            ops.embed_table_grad(pos_ids.reshape(-1), de, pos_grad)
        if tt is None:
            type_grad[0].add_(pos_sum.sum(0) if pos_sum is not None else de.sum(0))
        else:
            ops.embed_table_grad(tt.reshape(-1), de, type_grad)
        # region projection
        if img is not None:
            g_img = g_reg if g_reg is not None else g.view(B, S, H)[:, T:].reshape(B * R, H)
            if p_h > 0.0:
                ops.apply_dropout(g_img, (p_h, seed, ops.SITE_IMG))   # g is not read again after this point
            if img_pre is not None:   # back through the image LayerNorm
                ln = m.bert.LayerNorm
                g_img = ops.layernorm_bwd(img_pre, g_img.contiguous(), ln.weight.detach(), ln.variance_epsilon,
                                          self._grad(ln.weight), self._grad(ln.bias), ws=bufs.ws_t["ln_partial"],
                                          accumulate=acc)
            ops.wgrad([dict(dy=g_img, x=a_img, dw=self.dw_img, db=self.db_img)], B * R)
            D = m.bert.img_dim
            gi, gl = self._grad(m.bert.img_embedding.weight), self._grad(m.bert.location_embeds.weight)
            gbi, gbl = self._grad(m.bert.img_embedding.bias), self._grad(m.bert.location_embeds.bias)
            if acc:
                gi.add_(self.dw_img[:, :D]); gl.add_(self.dw_img[:, D:D + 128]); gbi.add_(self.db_img); gbl.add_(self.db_img)
            else:
                gi.copy_(self.dw_img[:, :D]); gl.copy_(self.dw_img[:, D:D + 128]); gbi.copy_(self.db_img); gbl.copy_(self.db_img)

    # ------------------------------------------------------------------------------ trunk-level training
    def trunk_forward(self, batch, head_mask=None, training=None, unmasked_only=False, want_hidden=False, want_attn=False):
        """BertImgModelwithLocationEmbeds.forward for a caller that back-propagates through it (the rollout's
        OscarEncoder, agent.py:493-518): -> (sequence_output fp32 [B,S,H], pooled_output fp32 [B,H], state).
        unmasked_only: the caller vouches that it reads sequence_output only at positions whose attention mask is not
        zero (OscarEncoder: pack_padded_sequence drops the rest); the step may then run on those rows alone (the row
        compaction of forward_backward) and the other positions of sequence_output are zero."""
        if training is None:
            training = self.model.bert.training
        st = self._trunk_fwd(batch, head_mask, None, None, bool(training), bool(unmasked_only))
        # the caller's hidden states come from the fp16 copy of the last LayerNorm's output where the layer keeps one (the
        # bf16 copy is the heads' / pooler's GEMM operand)
        n = st.seq.shape[0]
        if st.bufs.ln_h is not None:
            last = st.bufs.ln_h[1][:n].float()
        elif st.bufs.ln_residual:
            # LN_RESIDUAL (the default): no fp16 copy of a LayerNorm output exists -- the last layer's is rebuilt at fp32 from
            # its fp16 input and the row statistics its kernel wrote, as the residual adds do (one elementwise pass; the bf16
            # output stays the heads' / pooler's GEMM operand).  8 more significant bits than the bf16 rows.
            d = st.bufs.layers[-1]
            ln = self.model.bert.encoder.layer[-1].output.LayerNorm
            last = ((d["out_pre"][:n].float() - d["ln2_mean"][:n, None]) * d["ln2_rstd"][:n, None] * ln.weight.detach().float()
                    + ln.bias.detach().float())
        else:
            last = st.seq.float()
        def padded(rows32):
            if st.lay is None:
                return rows32.view(st.B, st.S, st.H)
            out = torch.zeros((st.M, st.H), dtype=torch.float32, device=st.dev)
            out.index_copy_(0, st.lay.index, rows32)
            return out.view(st.B, st.S, st.H)

        seq = padded(last)
        attn = None
        if want_attn:
            # all_attentions (oscar/modeling_bert.py:62-79, 160-167): per layer the probabilities AFTER dropout and head_mask,
            # fp32 [B, heads, S, S] -- recomputed from the layer's saved projection and log-sum-exp, the dropout decisions read
            # back from the keep words the forward wrote.  Values only: no gradient flows into them (autograd_trunk_forward
            # marks them non-differentiable).
            if st.lay is not None:
                raise NotImplementedError("output_attentions on compacted rows")
            nh_, pa = st.nh, st.p_a
            attn = []
            for l, d in enumerate(st.bufs.layers):
                pr = ops.attention_probs(d["qkv"][:n], d["lse"], st.B, st.S, nh_, mask=st.enc_mask, mask_additive=st.mask_additive,
                                         head_scale=None if st.hs is None else st.hs[l].contiguous())
                if pa > 0.0:
                    if "keep_bits" not in d:
                        raise NotImplementedError("output_attentions in training with attention dropout needs the keep words "
                                                  "(VT_ATTN_KEEP_BITS=1, the default)")
                    keep = ops.unpack_keep_bits(d["keep_bits"], st.B, nh_, st.S)
                    pr = pr * keep / (1.0 - ops.attn_drop_p(pa))
                attn.append(pr)
        if not want_hidden:
            return (seq, st.pooled.clone(), st) + ((None, attn) if want_attn else ())
        # all_hidden_states (oscar/modeling_bert.py:146-158): the embedding output, then every layer's output -- fp32, rebuilt
        # from the fp16 sums and the row statistics where the layer keeps no higher-precision copy (as `last` above)
        hidden = [padded(st.x_enc[:n].float())]
        for l, d in enumerate(st.bufs.layers[:-1]):
            if st.bufs.ln_residual:
                ln = self.model.bert.encoder.layer[l].output.LayerNorm
                hidden.append(padded((d["out_pre"][:n].float() - d["ln2_mean"][:n, None]) * d["ln2_rstd"][:n, None]
                                     * ln.weight.detach().float() + ln.bias.detach().float()))
            else:
                hidden.append(padded(d["out"][:n].float()))
        hidden.append(seq.clone())
        return (seq, st.pooled.clone(), st, hidden) + ((attn,) if want_attn else ())

    def trunk_backward(self, st, d_seq, d_pooled=None, accumulate=False, d_hidden=None):
        """Gradients of the trunk's parameters into the flat slab, given dL/d(sequence_output) [B,S,H] and / or
        dL/d(pooled_output) [B,H] (either may be None).  The pooler's gradients are zeroed when d_pooled is None, the
        region projection's when the forward had no regions."""
        m, bufs, acc = self.model, st.bufs, bool(accumulate)
        B, S, H, M, Mr, lay = st.B, st.S, st.H, st.M, st.Mr, st.lay
        g32 = bufs.g_seq32[:Mr]
        rows = lambda t: (t.detach().reshape(M, H).float() if lay is None
                          else t.detach().reshape(M, H).float().index_select(0, lay.index))
        extra = None
        if d_hidden is not None and any(t is not None for t in d_hidden):
            # d_hidden: L + 1 entries (None where the caller read nothing): the last one is dL/d(sequence_output) again
            if d_hidden[-1] is not None:
                d_seq = d_hidden[-1] if d_seq is None else d_seq + d_hidden[-1]
            if any(t is not None for t in d_hidden[:-1]):
                extra = [None if t is None else rows(t).to(BF16) for t in d_hidden[:-1]] + [None]
        if d_seq is None:
            g32.zero_()
        elif lay is None:
            g32.copy_(d_seq.detach().reshape(M, H))
        else:
            torch.index_select(d_seq.detach().reshape(M, H).float(), 0, lay.index, out=g32)
        pw, pb = m.bert.pooler.dense.weight, m.bert.pooler.dense.bias
        if d_pooled is not None:
            g_z = (d_pooled.detach().float() * (1.0 - st.pooled * st.pooled)).to(BF16)
            x_cls = st.seq.view(B, S * H)[:, :H] if lay is None else st.cls_seq
            ops.wgrad([dict(dy=g_z, x=x_cls, dw=self._grad(pw), db=self._grad(pb), accumulate=acc)], B)
            g32.index_add_(0, st.cls_rows, ops.linear(g_z, self.head_t["pool"]).float())
        elif not acc:
            self._grad(pw).zero_()
            self._grad(pb).zero_()
        self._trunk_bwd(st, g32, acc, d_hidden=extra)
        if st.img is None and not acc:
            for prm in (m.bert.img_embedding.weight, m.bert.img_embedding.bias, m.bert.location_embeds.weight,
                        m.bert.location_embeds.bias):
                self._grad(prm).zero_()
            if getattr(m.bert, "use_img_layernorm", None):
                self._grad(m.bert.LayerNorm.weight).zero_()
                self._grad(m.bert.LayerNorm.bias).zero_()

    # ---- the launch sequences of vt_encoder_forward/backward_bf16 issued op by op (bench.py's per-kernel timing)
    def _encoder_forward_unrolled(self, bufs, x0, mask, B, S, p_h=0.0, p_a=0.0, seed=0, lay=None, mask_additive=False,
                                  hs=None):
        cfg = self.cfg
        nh, eps = cfg.num_attention_heads, cfg.layer_norm_eps
        cur, cur_h, cur_ln = x0, None, None
        n = x0.shape[0]
        for l, ((t, _), a) in enumerate(zip(self._keep, bufs.layers)):
            a = {k: (v if k in ("lse", "keep_bits") else v[:n]) for k, v in a.items()}   # the rows in use (all, or the compacted ones)
            kb = a.get("keep_bits") if p_a > 0.0 else None
            ops.linear(cur, t["w_qkv"], t["b_qkv"], out=a["qkv"])
            if hs is None:
                ops.attention_fwd(a["qkv"], B, S, nh, mask=mask, mask_additive=mask_additive, out=a["ctx"], lse=a["lse"],
                                  drop=(p_a, seed, ops.site_attn(l)), seq=lay, keep_bits=kb)
            else:
                # head_mask (oscar/modeling_bert.py:65-66): the attention kernel's own context is kept for the backward
                # (ctx_raw), its per-head scaled copy is what the output projection sees
                raw = bufs.ctx_raw(l)[:n]
                ops.attention_fwd(a["qkv"], B, S, nh, mask=mask, mask_additive=mask_additive, out=raw, lse=a["lse"],
                                  drop=(p_a, seed, ops.site_attn(l)), seq=lay, keep_bits=kb)
                ops.scale_heads(raw, hs[l].contiguous(), out=a["ctx"])
            # (fp16 copies of the stream, ops.F16_STREAM: attn_pre / out_pre are fp16 tensors, the residual adds read the
            # LayerNorm outputs' fp16 copies ln1_h / ln2_h -- the library learns all of it from the dtypes)
            if bufs.ln_residual:
                # (ops.LN_RESIDUAL: no fp16 copies -- each residual add rebuilds the previous LayerNorm from its fp16 input
                # and the statistics its kernel wrote; the same arithmetic as the C loop's ln_residual_mode 1)
                ops.linear(a["ctx"], t["w_ao"], t["b_ao"], residual=cur if cur_h is None else cur_h, out=a["attn_pre"],
                           drop=(p_h, seed, ops.site_selfout(l)), residual_ln=cur_ln)
                ops.layernorm(a["attn_pre"], t["ln1_g"], t["ln1_b"], eps, out=a["attn_out"], mean=a["ln1_mean"], rstd=a["ln1_rstd"])
                ops.linear(a["attn_out"], t["w_in"], t["b_in"], act=ACT_GELU, out=a["mid"], pre_act_out=a["mid_pre"])
                ops.linear(a["mid"], t["w_out"], t["b_out"], residual=a["attn_pre"], out=a["out_pre"],
                           drop=(p_h, seed, ops.site_out(l)), residual_ln=(a["ln1_mean"], a["ln1_rstd"], t["ln1_g"], t["ln1_b"]))
                ops.layernorm(a["out_pre"], t["ln2_g"], t["ln2_b"], eps, out=a["out"], mean=a["ln2_mean"], rstd=a["ln2_rstd"])
                cur, cur_h, cur_ln = a["out"], a["out_pre"], (a["ln2_mean"], a["ln2_rstd"], t["ln2_g"], t["ln2_b"])
                continue
            ops.linear(a["ctx"], t["w_ao"], t["b_ao"], residual=cur if cur_h is None else cur_h, out=a["attn_pre"],
                       drop=(p_h, seed, ops.site_selfout(l)))
            ops.layernorm(a["attn_pre"], t["ln1_g"], t["ln1_b"], eps, out=a["attn_out"], out_h=a.get("ln1_h"))
            ops.linear(a["attn_out"], t["w_in"], t["b_in"], act=ACT_GELU, out=a["mid"], pre_act_out=a["mid_pre"])
            ops.linear(a["mid"], t["w_out"], t["b_out"], residual=a.get("ln1_h", a["attn_out"]), out=a["out_pre"],
                       drop=(p_h, seed, ops.site_out(l)))
            ops.layernorm(a["out_pre"], t["ln2_g"], t["ln2_b"], eps, out=a["out"], out_h=a.get("ln2_h"))
            cur, cur_h = a["out"], a.get("ln2_h")

    def _overlap_kw(self, bufs):
        """Second workspace set + side stream for the wgrad / dgrad overlap (off: overlap_wgrad = False)."""
        if not self.overlap_wgrad:
            return {}
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self.flat.p.device)
        return dict(ws_b=bufs.ws_b, side_stream=self._side_stream)

    def _encoder_backward_unrolled(self, bufs, x0, mask, g, B, S, acc, p_h=0.0, p_a=0.0, seed=0, lay=None,
                                   mask_additive=False, hs=None, inject=None):
        cfg = self.cfg
        nh, eps, M = cfg.num_attention_heads, cfg.layer_norm_eps, x0.shape[0]
        rowed = ("g_pre", "g_pre2", "g_mid", "g_ctx", "g_qkv", "g_pre_d", "g_pre2_d", "dq32")
        w = {k: (v[:M] if k in rowed else v) for k, v in bufs.ws_t.items()}
        for l in range(cfg.num_hidden_layers - 1, -1, -1):
            if inject is not None and l + 1 < cfg.num_hidden_layers and inject[l + 1] is not None:
                g.add_(inject[l + 1])   # dL/d(output of layer l) from a caller that read the intermediate hidden states
            (t, gr), a, (_, wt) = self._keep[l], bufs.layers[l], self.wt[l]
            a = {k: (v if k in ("lse", "keep_bits") else v[:M]) for k, v in a.items()}
            x_in = x0 if l == 0 else bufs.layers[l - 1]["out"][:M]
            hd = p_h > 0.0   # with hidden dropout the dense outputs' gradients are the masked copies
            g_pre_dn, g_pre2_dn = (w["g_pre_d"], w["g_pre2_d"]) if hd else (w["g_pre"], w["g_pre2"])
            ops.layernorm_bwd(a["out_pre"], g, t["ln2_g"], eps, gr["d_ln2_g"], gr["d_ln2_b"], dx=w["g_pre"],
                              ws=w["ln_partial"], accumulate=acc, dx_dropped=w["g_pre_d"] if hd else None,
                              drop=(p_h, seed, ops.site_out(l)))
            ops.linear(g_pre_dn, wt["wt_out"], residual=a["mid_pre"], act=ACT_MUL, out=w["g_mid"])
            ops.linear(w["g_mid"], wt["wt_in"], residual=w["g_pre"], out=g)
            ops.layernorm_bwd(a["attn_pre"], g, t["ln1_g"], eps, gr["d_ln1_g"], gr["d_ln1_b"], dx=w["g_pre2"],
                              ws=w["ln_partial"], accumulate=acc, dx_dropped=w["g_pre2_d"] if hd else None,
                              drop=(p_h, seed, ops.site_selfout(l)))
            ops.linear(g_pre2_dn, wt["wt_ao"], out=w["g_ctx"])
            g_ctx, ctx_l = w["g_ctx"], a["ctx"]
            if hs is not None:   # back through the head scaling: dL/d(raw context) = head_mask * dL/d(scaled context)
                g_ctx = ops.scale_heads(w["g_ctx"], hs[l].contiguous())
                ctx_l = bufs.ctx_raw(l)[:M]
            ops.attention_bwd(a["qkv"], g_ctx, ctx_l, a["lse"], B, S, nh, mask=mask, mask_additive=mask_additive,
                              out=w["g_qkv"], delta_ws=w["delta"], dq32_ws=w.get("dq32"), drop=(p_a, seed, ops.site_attn(l)),
                              seq=lay, keep_bits=a.get("keep_bits") if p_a > 0.0 else None)
            ops.linear(w["g_qkv"], wt["wt_qkv"], residual=w["g_pre2"], out=g)
            ops.wgrad([dict(dy=w["g_mid"], x=a["attn_out"], dw=gr["d_w_in"], db=gr["d_b_in"], accumulate=acc),
                       dict(dy=g_pre_dn, x=a["mid"], dw=gr["d_w_out"], db=gr["d_b_out"], accumulate=acc),
                       dict(dy=w["g_qkv"], x=x_in, dw=gr["d_w_qkv"], db=gr["d_b_qkv"], accumulate=acc),
                       dict(dy=g_pre2_dn, x=a["ctx"], dw=gr["d_w_ao"], db=gr["d_b_ao"], accumulate=acc)], M)
        if inject is not None and inject[0] is not None:
            g.add_(inject[0])           # ... and with respect to the embedding output

    # ------------------------------------------------------------------------------ optimizer
    def _require_ownership(self):
        """The model's parameters must still live in THIS engine's flat slab.  Building another engine over the same
        parameters (the trunk-level engine behind a training-mode `model.bert(...)` call, `model.to(...)`, a load with
        assign) re-points `p.data` elsewhere without touching `_version`: this engine would then read and update a stale
        slab and mirror while the model's real parameters never change.  Refused loudly instead."""
        if not self.flat.owns_params():
            raise RuntimeError(
                "this PretrainEngine no longer owns the model's parameters (their storage was re-pointed: another engine was "
                "built over them -- e.g. by a training-mode call through model.bert -- or the model was moved); build a new "
                "PretrainEngine(model) and carry the optimizer state over with state_dict() / load_state_dict()")

    def _adam_begin(self):
        """Advance Adam's step counter and return this step's constants (lr after the schedule, bias-corrected step size)."""
        self._require_ownership()
        self.step_count += 1
        t = self.step_count
        lr = self.lr * self.lr_factor()
        b1, b2 = self.betas
        step_size = lr
        if self.correct_bias:
            step_size = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
        return lr, step_size, b1, b2

    def _adam_ranges(self, consts, ranges, grad_scale, grads=None):
        """The fused AdamW on [start, end) ranges of the flat slabs (weight decay by region: pretrain.py:109-127)."""
        f = self.flat
        lr, step_size, b1, b2 = consts
        g = f.g if grads is None else grads   # fp32 slab, or the all-reduced bf16 communication copy
        nd = f.n_decay
        for s_, e_ in ranges:
            for lo, hi, wd in ((s_, min(e_, nd), self.wd), (max(s_, nd), e_, 0.0)):
                if hi > lo:
                    ops.adamw_flat(f.p[lo:hi], g[lo:hi], f.m[lo:hi], f.v[lo:hi], f.mirror[lo:hi], lr, step_size, b1, b2,
                                   self.eps, wd, grad_scale)

    def _adam_end(self):
        self.sched_step += 1
        self.flat.mark_fresh()
        self._wt_dirty = True
        # the fused kernel wrote the slab through raw pointers: no parameter's _version moved, so the packed bf16
        # copies the inference path caches (encoder layers, region projection, rollout modules) are told explicitly
        invalidate_packed_weights()

    def optimizer_step(self, grad_scale=1.0, grads=None):
        """AdamW.step() + scheduler.step() (pretrain.py:192-193) as two fused launches (decay / no-decay)."""
        self._adam_ranges(self._adam_begin(), [(0, self.flat.total)], grad_scale, grads)
        self._adam_end()

    def _train_step_adamw_under_backward(self, batch, scale, layers_per_chunk):
        """One rank: the fused AdamW of a parameter range runs on a side stream as soon as that range's gradients are final
        -- the heads' before the encoder backward starts, a chunk of encoder layers' while the earlier layers' backward
        runs (pretrain.py:191-193 has no gradient clipping between backward and step: a range's update needs nothing but
        its own gradients).  The update is HBM-bound (30 bytes per parameter) and finds its CUs beside the kernels that do
        not fill the chip (the grouped weight-gradient launch runs 216 of 256 workgroups; attention, LayerNorm and the
        heads are not persistent).  A layer's weights are not read again once its backward is enqueued (dgrad reads the
        transposed copies, rebuilt at the start of the next step); embeddings, region projection and everything else follow
        on the main stream, which then waits for the side stream."""
        from .distributed import complement_ranges

        if self._adam_stream is None:
            self._adam_stream = torch.cuda.Stream(device=self.flat.p.device)
        side = self._adam_stream
        consts = self._adam_begin()
        try:
            def launch(rng):
                ev = torch.cuda.Event()
                ev.record()
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    self._adam_ranges(consts, rng, 1.0)

            comm = dict(layers_per_chunk=layers_per_chunk, launch=launch, done=[])
            out = self.forward_backward(batch, grad_scale=scale, comm=comm)
        except BaseException:
            self.step_count -= 1   # no update was completed on behalf of this step's counter
            torch.cuda.current_stream().wait_stream(side)
            raise
        self._adam_ranges(consts, complement_ranges(self.flat.total, comm["done"]), 1.0)
        torch.cuda.current_stream().wait_stream(side)
        self._adam_end()
        return out

    # ------------------------------------------------------------------------------ optimizer state (resume)
    def state_dict(self):
        """What a resumed run needs beside the model's own state_dict: AdamW's moments per parameter NAME (layout-
        independent), the step / scheduler / dropout counters and the hyper-parameters.  (The reference checkpoints weights
        only, pretrain.py:247-270; this is the fused optimizer's counterpart of torch.optim's state_dict.)"""
        f = self.flat
        state = {n: dict(exp_avg=f.view(f.m, n).detach().clone(), exp_avg_sq=f.view(f.v, n).detach().clone())
                 for n, _, _, _, _ in f.entries}
        return dict(state=state, step_count=self.step_count, sched_step=self.sched_step, fb_count=self.fb_count,
                    drop_seed_base=self.drop_seed_base,
                    hyper=dict(lr=self.lr, weight_decay=self.wd, eps=self.eps, betas=tuple(self.betas),
                               correct_bias=self.correct_bias, schedule=self.schedule, warmup_steps=self.warmup_steps,
                               t_total=self.t_total,
                               # what travelled in the gradient all-reduce of the run that wrote this state (recorded, not
                               # restored: it is a property of the launch, and it changes the result at rounding level)
                               grad_comm_dtype=self.grad_comm_dtype, world_size=self.world,
                               hidden_dropout=float(self.cfg.hidden_dropout_prob),
                               attention_dropout_configured=float(self.cfg.attention_probs_dropout_prob),
                               attention_dropout_effective=self.attention_dropout_effective))

    def load_state_dict(self, sd, load_hyper=True):
        f = self.flat
        missing = [n for n, _, _, _, _ in f.entries if n not in sd["state"]]
        if missing:
            raise KeyError("optimizer state has no entry for %s" % ", ".join(missing[:5]))
        for n, _, _, _, _ in f.entries:
            f.view(f.m, n).copy_(sd["state"][n]["exp_avg"])
            f.view(f.v, n).copy_(sd["state"][n]["exp_avg_sq"])
        self.step_count, self.sched_step = int(sd["step_count"]), int(sd["sched_step"])
        self.fb_count, self.drop_seed_base = int(sd["fb_count"]), int(sd["drop_seed_base"])
        if load_hyper:
            h = sd["hyper"]
            self.lr, self.wd, self.eps, self.betas = h["lr"], h["weight_decay"], h["eps"], tuple(h["betas"])
            self.correct_bias, self.schedule = h["correct_bias"], h["schedule"]
            self.warmup_steps, self.t_total = h["warmup_steps"], h["t_total"]

    def all_reduce_grads(self):
        """Sum the flat gradient slab over the data-parallel group in fixed-size buckets (no overlap)."""
        if self.world == 1:
            return
        from .distributed import all_reduce_flat

        if self.g16 is not None:
            ops.cast_to_bf16(self.flat.g, self.g16)
            all_reduce_flat(self.g16, 2 * self.bucket_elems, self.pg)
        else:
            all_reduce_flat(self.flat.g, self.bucket_elems, self.pg)

    def train_step(self, batch, overlap=True, layers_per_chunk=3, _force_comm=None):
        """zero_grad -> forward -> backward -> gradient all-reduce -> AdamW -> schedule, as pretrain.py:150-193.
        The reference divides the loss by world_size before backward AND lets DDP average;
        loss_scale_by_world reproduces that extra 1/world factor.  With overlap, the all-reduce of the
        encoder layers' gradients runs under the backward of the earlier layers."""
        ws = self.world
        scale = (1.0 / ws) if (ws > 1 and self.loss_scale_by_world) else 1.0
        if ws == 1 and _force_comm is None and overlap and self.overlap_adamw:
            return self._train_step_adamw_under_backward(batch, scale, layers_per_chunk)
        if (ws == 1 and _force_comm is None) or not overlap:
            out = self.forward_backward(batch, grad_scale=scale)
            self.all_reduce_grads()
        else:
            from .distributed import all_reduce_ranges, complement_ranges

            handles, launched = [], []   # launched: (ranges, first handle, one past the last handle) per launch call

            def reduce_ranges(rng):
                n0 = len(handles)
                if self.g16 is None:
                    all_reduce_ranges(self.flat.g, rng, self.bucket_elems, self.pg, handles)
                else:
                    for s_, e_ in rng:   # bf16 communication copy of the range, then its all-reduce (same bytes per bucket)
                        if e_ > s_:
                            ops.cast_to_bf16(self.flat.g[s_:e_], self.g16[s_:e_])
                    all_reduce_ranges(self.g16, rng, 2 * self.bucket_elems, self.pg, handles)
                launched.append((list(rng), n0, len(handles)))

            launch = _force_comm or reduce_ranges
            comm = dict(layers_per_chunk=layers_per_chunk, launch=launch, done=[])
            out = self.forward_backward(batch, grad_scale=scale, comm=comm)
            launch(complement_ranges(self.flat.total, comm["done"]))  # embeddings, region projection, heads
            if _force_comm is None:
                # AdamW per arrived range, in launch order (last layers first, the embeddings / heads tail last): the update
                # of the encoder's 85 M parameters runs while the tail's all-reduce is still in flight.  (wait() makes this
                # stream wait for the communicator's, not the host.)
                consts = self._adam_begin()
                for rng, h0, h1 in launched:
                    for h in handles[h0:h1]:
                        h.wait()
                    self._adam_ranges(consts, rng, 1.0 / ws, self.g16)   # 1 / ws: DDP's mean over ranks
                self._adam_end()
                return out
            for h in handles:
                h.wait()
        use16 = self.g16 is not None and _force_comm is None
        self.optimizer_step(grad_scale=1.0 / ws, grads=self.g16 if use16 else None)  # DDP's mean over ranks
        return out


class _LossWithGrads(torch.autograd.Function):
    """Bridges the engine to torch autograd for the reference's own loop (`loss.backward()` then a torch
    optimizer, tasks/viewpoint_select/pretrain.py:167-193): forward has already run the HIP forward AND
    backward for d(loss)/d(params); backward hands those gradients (times the incoming scalar) to
    autograd, which accumulates them into p.grad and fires DistributedDataParallel's hooks."""

    @staticmethod
    def forward(ctx, engine, loss, *params):
        ctx.engine = engine
        ctx.names = [engine._name_of(p) for p in params]
        ctx.fb_count = engine.fb_count      # which forward / backward pair of the engine these gradients belong to
        return loss.detach().clone()

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.engine.fb_count != ctx.fb_count:
            # the gradients live in the engine's slab and the next training forward on this module has overwritten them
            raise RuntimeError("backward() of a PreTrainOscar loss after a later training forward of the same module: the HIP "
                               "step computes the gradients with the forward and keeps ONE set per module -- call "
                               "loss.backward() before the next forward (the reference's loop does: pretrain.py:169-191)")
        f = ctx.engine.flat
        grads = []
        for n in ctx.names:
            g = f.view(f.g, n)
            grads.append(g * grad_out)
        return (None, None) + tuple(grads)


class _LossLazyGrads(torch.autograd.Function):
    """eval-mode forward with grad enabled (the reference's val(), pretrain.py:291,469-481, calls model(**batch) in
    eval() with no torch.no_grad): the values come from the inference path; the engine (its slabs and saved
    activations) is built and the HIP forward + backward run only IF `.backward()` is actually called -- dropout is
    the identity in eval mode, so the recomputation sees the same function."""

    @staticmethod
    def forward(ctx, model, batch, loss, *params):
        ctx.model, ctx.batch = model, dict(batch)
        ctx.head_mask = ctx.batch.pop("__head_mask__", None)
        ctx.params = params
        return loss.detach().clone()

    @staticmethod
    def backward(ctx, grad_out):
        model = ctx.model
        eng = _bridge_engine(model)
        was = model.training
        model.eval()
        try:
            eng.forward_backward(ctx.batch, head_mask=ctx.head_mask)
        finally:
            model.train(was)
        f = eng.flat
        return (None, None, None) + tuple(f.view(f.g, eng._name_of(p)) * grad_out for p in ctx.params)


class _TrunkWithGrads(torch.autograd.Function):
    """The trunk as ONE autograd node for callers that train through it (OscarEncoder in the rollout, agent.py:493-518):
    forward = the engine's trunk forward (activations kept in its buffers), backward = its trunk backward, handing the
    parameter gradients to autograd."""

    @staticmethod
    def forward(ctx, engine, batch, head_mask, unmasked_only, want_hidden, want_attn, *params):
        out = engine.trunk_forward(batch, head_mask, unmasked_only=unmasked_only, want_hidden=want_hidden, want_attn=want_attn)
        seq, pooled, st = out[:3]
        ctx.engine, ctx.st = engine, st
        ctx.names = [engine._name_of(p) for p in params]
        ctx.n_hidden = len(out[3]) if want_hidden else 0
        ctx.set_materialize_grads(False)
        attn = tuple(out[4]) if want_attn else ()
        ctx.mark_non_differentiable(*attn)
        return (seq, pooled) + (tuple(out[3]) if want_hidden else ()) + attn

    @staticmethod
    def backward(ctx, d_seq, d_pooled, *d_rest):
        eng, st = ctx.engine, ctx.st
        d_hidden = d_rest[:ctx.n_hidden]
        eng.trunk_backward(st, d_seq, d_pooled, d_hidden=list(d_hidden) if d_hidden else None)
        f = eng.flat
        unused = set()
        if st.img is None:
            unused.update(("bert.img_embedding.", "bert.location_embeds.", "bert.LayerNorm."))
        if d_pooled is None:
            unused.add("bert.pooler.")
        grads = [None if n.startswith(tuple(unused)) else f.view(f.g, n).clone() for n in ctx.names]
        return (None, None, None, None, None, None) + tuple(grads)


class _TrunkLazyGrads(torch.autograd.Function):
    """eval() with grad enabled at trunk level (the values come from the inference path, nothing is saved): the engine's
    trunk forward + backward run only if `.backward()` is actually called -- dropout is the identity in eval mode, so the
    recomputation sees the same function (the trunk-level twin of _LossLazyGrads)."""

    @staticmethod
    def forward(ctx, trunk, batch, head_mask, seq, pooled, *params):
        ctx.trunk, ctx.batch, ctx.head_mask, ctx.params = trunk, dict(batch), head_mask, params
        ctx.set_materialize_grads(False)
        return seq.detach().clone(), pooled.detach().clone()

    @staticmethod
    def backward(ctx, d_seq, d_pooled):
        eng = _bridge_engine(ctx.trunk)
        _, _, st = eng.trunk_forward(ctx.batch, ctx.head_mask, training=False)
        eng.trunk_backward(st, d_seq, d_pooled)
        f = eng.flat
        unused = []
        if st.img is None:
            unused += ["bert.img_embedding.", "bert.location_embeds.", "bert.LayerNorm."]
        if d_pooled is None:
            unused.append("bert.pooler.")
        names = [eng._name_of(p) for p in ctx.params]
        grads = [None if n.startswith(tuple(unused)) else f.view(f.g, n).clone() for n in names]
        return (None, None, None, None, None) + tuple(grads)


def lazy_autograd_trunk(trunk, batch, head_mask, seq, pooled):
    """The inference path's (sequence_output, pooled_output) made differentiable on demand (eval mode, grad enabled)."""
    params = [p for p in trunk.parameters() if p.requires_grad]
    return _TrunkLazyGrads.apply(trunk, batch, head_mask, seq, pooled, *params)


def autograd_trunk_forward(trunk, batch, head_mask=None, unmasked_only=False, want_hidden=False, want_attn=False):
    """BertImgModelwithLocationEmbeds.forward in training mode: (sequence_output, pooled_output) that back-propagate into the
    trunk's parameters through the HIP backward; under torch.no_grad() the same forward (dropout included) without a graph.
    unmasked_only: see PretrainEngine.trunk_forward.  want_hidden: one more element, the tuple of all_hidden_states
    (embedding output + every layer's output, oscar/modeling_bert.py:146-158), each differentiable like sequence_output.
    want_attn: one more, the tuple of all_attentions (per layer, after dropout and head_mask) -- values only."""
    eng = _bridge_engine(trunk)
    if not torch.is_grad_enabled():
        out = eng.trunk_forward(batch, head_mask, unmasked_only=unmasked_only, want_hidden=want_hidden, want_attn=want_attn)
        return (out[0], out[1]) + ((tuple(out[3]),) if want_hidden else ()) + ((tuple(out[4]),) if want_attn else ())
    params = [p for p in trunk.parameters() if p.requires_grad]
    out = _TrunkWithGrads.apply(eng, batch, head_mask, bool(unmasked_only), bool(want_hidden), bool(want_attn), *params)
    L1 = trunk.config.num_hidden_layers + 1
    res = (out[0], out[1])
    k = 2
    if want_hidden:
        res = res + (tuple(out[k:k + L1]),)
        k += L1
    if want_attn:
        res = res + (tuple(out[k:]),)
    return res


def _bridge_engine(model):
    """The engine behind the autograd bridge (parameters stay the model's own: attach_grads=False).  An external torch
    optimizer owns the update there, and optimizers that write through `p.data` (the reference's pytorch-transformers
    AdamW: `p.data.addcdiv_`, `p.data.add_`) do not move `p._version` -- so nothing can tell whether the fp32 slab
    changed since the last call.  The bf16 mirror (one fused cast of the slab) and the transposed copies are therefore
    rebuilt on EVERY bridged forward, and the inference-side packed copies are invalidated."""
    eng = getattr(model, "_vt_engine", None)
    if eng is None or eng.flat.p.device != next(model.parameters()).device or not eng.flat.owns_params():
        eng = PretrainEngine(model, attach_grads=False)
        object.__setattr__(model, "_vt_engine", eng)
    else:
        eng.flat.refresh_mirror()
        eng._wt_dirty = True
    invalidate_packed_weights()
    return eng


def autograd_forward(model, batch, head_mask=None):
    """PreTrainOscar.forward in training mode with grad enabled: returns the 7-tuple whose first element
    back-propagates into the model's parameters."""
    eng = _bridge_engine(model)
    out = eng.forward_backward(batch, head_mask=head_mask)
    params = [p for p in model.parameters() if p.requires_grad]
    loss = _LossWithGrads.apply(eng, out[0], *params)
    return (loss,) + tuple(out[1:])


def lazy_autograd_loss(model, batch, loss, head_mask=None):
    """The inference path's loss made differentiable on demand (eval mode, grad enabled)."""
    params = [p for p in model.parameters() if p.requires_grad]
    if head_mask is not None:
        batch = dict(batch, __head_mask__=head_mask)
    return _LossLazyGrads.apply(model, batch, loss, *params)
