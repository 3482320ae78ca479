"""The rollout caller around the trunk, served by the HIP library -- inference, and since round 2
training: with grad enabled every block below is an autograd node whose backward is HIP kernels too
(visitron_amd/rollout_autograd.py; the trunk node is the pretrain engine's forward / backward).

Drop-ins for tasks/viewpoint_select/agent_models.py: `OscarEncoder` (:192-310), `SoftDotAttention` (:313-357) and
`AttnDecoderLSTM` (:360-428): same constructor arguments, parameter names (so the reference's state dicts load:
`lstm.weight_ih_l0`, `feat_att_layer.linear_in.weight`, ...), argument meaning and return values.  `nn.LSTM` /
`nn.LSTMCell` are kept as parameter containers only; their arithmetic runs in vt_lstm_sequence_f32 / vt_lstm_step_f32,
the dot attentions in vt_softdot_attention_f32, every dense projection in the NT GEMM.  No CPU fallback.  nn.Dropout
(the modules' `drop`) is a torch elementwise op on the device in training mode.
"""
import torch
import torch.nn as nn

from . import ops
from .ops import ACT_NONE, ACT_TANH, BF16, round_up


SKINNY_ROWS = 256   # up to this many rows the dense layers take vt_skinny_linear_f32 instead of the tiled GEMM


def _grad_mode(module, *tensors):
    """True when the call takes the autograd-node form of the module: grad enabled and a parameter or an input requires
    grad (a graph must be built), or train() with dropout on -- also under torch.no_grad() (agent.py:476-489,
    test(use_dropout=True)), where the same nodes run forward only and nn.Dropout does its work."""
    drop = getattr(module, "drop", None)
    if module.training and drop is not None and drop.p > 0.0:
        return True
    if not torch.is_grad_enabled():
        return False
    return any(p.requires_grad for p in module.parameters()) or any(t is not None and t.requires_grad for t in tensors)


class _Packed(object):
    """bf16 copies of a module's weights padded to the GEMM's K granule, rebuilt when a parameter changes."""

    def __init__(self):
        self._key, self._val = None, None

    def get(self, params, build):
        from .modeling import _param_key   # (address, version) per parameter + the process-wide weights generation

        key = _param_key(params)
        if key != self._key:
            self._val, self._key = build(), key
        return self._val


def _pad_weight(w):
    N, K = w.shape
    out = torch.zeros((N, round_up(K, 64)), dtype=BF16, device=w.device)
    out[:, :K] = w.detach().to(BF16)
    return out


def _f32c(x):
    x = x.detach()
    return x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous()


def _dense(x_parts, w_pad, bias=None, act=ACT_NONE):
    """act([x_parts...] @ W.T + bias) in fp32: x_parts = one or two fp32 [M, d] tensors K-concatenated into the bf16
    operand (vt_pack_concat_bf16), W pre-padded bf16 [N, kpad]."""
    s0 = _f32c(x_parts[0])
    s1 = _f32c(x_parts[1]) if len(x_parts) > 1 else None
    if s0.shape[0] <= SKINNY_ROWS and s0.shape[1] % 4 == 0 and (s1 is None or s1.shape[1] % 4 == 0) \
            and act in (ACT_NONE, ACT_TANH):
        return ops.skinny_linear(s0, w_pad, bias, s1, act)   # a few rows: one launch, no packing pass
    a = ops.pack_concat(s0, s1, w_pad.shape[1])
    N = w_pad.shape[0]
    buf = torch.empty((a.shape[0], round_up(N, 4)), dtype=torch.float32, device=a.device)
    ops.linear(a, w_pad, bias, act=act, out=buf, out_f32=True)
    return buf if buf.shape[1] == N else buf[:, :N]


class SoftDotAttention(nn.Module):
    """agent_models.py:313-357."""

    def __init__(self, query_dim, ctx_dim):
        super().__init__()
        self.linear_in = nn.Linear(query_dim, ctx_dim, bias=False)
        self.sm = nn.Softmax(dim=1)
        self.linear_out = nn.Linear(query_dim + ctx_dim, query_dim, bias=False)
        self.tanh = nn.Tanh()
        self._pk = _Packed()
        self._pk_t = _Packed()

    def _weights(self):
        return self._pk.get((self.linear_in.weight, self.linear_out.weight),
                            lambda: (_pad_weight(self.linear_in.weight), _pad_weight(self.linear_out.weight)))

    def attend(self, h, context, mask, want_weighted, want_attn, output_prob):
        w_in, _ = self._weights()
        target = _dense((h,), w_in).contiguous()
        return ops.softdot_attention(target, _ctx_f32(context), mask, want_weighted, want_attn, output_prob)

    def _train_packs(self):
        from .rollout_autograd import packed_linear

        return self._pk_t.get((self.linear_in.weight, self.linear_out.weight),
                              lambda: (packed_linear(self.linear_in.weight), packed_linear(self.linear_out.weight)))

    def _forward_autograd(self, h, context, mask, output_tilde, output_prob):
        """The same block as autograd nodes (training): dense -> softdot -> cat -> dense(tanh)."""
        from . import rollout_autograd as ra

        p_in, p_out = self._train_packs()
        target = ra.dense(h, self.linear_in.weight, None, ACT_NONE, p_in)
        weighted, attn = ra.softdot(target, context, mask, output_prob)
        if output_tilde:
            h_tilde = ra.dense(torch.cat((weighted, h.float()), 1), self.linear_out.weight, None, ACT_TANH, p_out)
            return h_tilde, attn
        return weighted, attn

    def forward(self, h, context, mask=None, output_tilde=True, output_prob=True):
        ops._require_hip(h, context)
        if _grad_mode(self, h, context):
            return self._forward_autograd(h, context, mask, output_tilde, output_prob)
        weighted, attn = self.attend(h, context, mask, True, True, output_prob)
        if output_tilde:
            h_tilde = _dense((weighted, h), self._weights()[1], act=ACT_TANH)
            return h_tilde, attn
        return weighted, attn


def _ctx_f32(context):
    c = context.detach()
    if c.dtype != torch.float32:
        c = c.float()
    return c if c.stride(-1) == 1 else c.contiguous()


class _DecoderStepGraph(object):
    """One inference decoder step (ten launches of 5-20 us: the step is host-launch-bound, 164 us issued directly) captured
    once per input geometry as a HIP graph and replayed: 89 us at B = 64, 71 us at B = 8 (tools/decoder_graph_bench.py).
    The caller's tensors are copied into the capture's own input buffers (replays read fixed addresses) and the four
    results come back as views of ONE cloned flat buffer (the capture's outputs are rewritten by the next replay)."""

    def __init__(self, dec, action, feature, cand_feat, prev_h1, c_0, ctx, ctx_mask):
        f32 = lambda t: torch.empty(t.shape, dtype=torch.float32, device=t.device)
        self.inp = [f32(action), f32(feature), f32(cand_feat), f32(prev_h1), f32(c_0), f32(ctx)]
        self.mask = None if ctx_mask is None else torch.empty(ctx_mask.shape, dtype=torch.bool, device=ctx_mask.device)
        self.load(action, feature, cand_feat, prev_h1, c_0, ctx, ctx_mask)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):   # warm-up off the capture: packed weights, workspace allocations, kernel attributes
            for _ in range(2):
                dec._step(*self.inp, self.mask)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            outs = dec._step(*self.inp, self.mask)
            self.sizes = [o.numel() for o in outs]
            self.shapes = [o.shape for o in outs]
            self.flat = torch.cat([o.reshape(-1) for o in outs])   # inside the capture: part of every replay

    def load(self, action, feature, cand_feat, prev_h1, c_0, ctx, ctx_mask):
        for dst, src in zip(self.inp, (action, feature, cand_feat, prev_h1, c_0, ctx)):
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src)
        if self.mask is not None:
            self.mask.copy_(ctx_mask)

    def run(self, *args):
        self.load(*args)
        self.graph.replay()
        out = self.flat.clone()
        return tuple(t.view(shp) for t, shp in zip(out.split(self.sizes), self.shapes))


class AttnDecoderLSTM(nn.Module):
    """agent_models.py:360-428: one decoder step (3 dot attentions around an LSTM cell)."""

    def __init__(self, angle_feat_size, embedding_size, hidden_size, dropout_ratio, feature_size=2048 + 4):
        super().__init__()
        self.embedding_size = embedding_size
        self.feature_size = feature_size
        self.hidden_size = hidden_size
        self.embedding = nn.Sequential(nn.Linear(angle_feat_size, self.embedding_size), nn.Tanh())
        self.drop = nn.Dropout(p=dropout_ratio)
        self.lstm = nn.LSTMCell(embedding_size + feature_size, hidden_size)   # parameter container
        self.feat_att_layer = SoftDotAttention(hidden_size, feature_size)
        self.attention_layer = SoftDotAttention(hidden_size, hidden_size)
        self.candidate_att_layer = SoftDotAttention(hidden_size, feature_size)
        self._pk = _Packed()
        self._pk_t = _Packed()
        # inference: replay the step from a captured HIP graph (one per input geometry).  Opt-in (VT_DECODER_GRAPH=1 or the
        # attribute): the reference's rollout is bound by its simulator, and a capture holds its own copies of the inputs.
        import os
        self.use_graph = os.environ.get("VT_DECODER_GRAPH", "0") == "1"
        self._graphs = {}

    def _weights(self):
        emb, cell = self.embedding[0], self.lstm

        def build():
            return dict(w_emb=_pad_weight(emb.weight), b_emb=_f32c(emb.bias), w_ih=_pad_weight(cell.weight_ih),
                        b=(cell.bias_ih.detach().float() + cell.bias_hh.detach().float()).contiguous(),
                        w_hh=cell.weight_hh.detach().to(BF16).contiguous())

        return self._pk.get((emb.weight, emb.bias, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh), build)

    def _forward_autograd(self, action, feature, cand_feat, prev_h1, c_0, ctx, ctx_mask):
        """agent_models.py:406-425 as autograd nodes (training; `drop` follows the module's training flag)."""
        from . import rollout_autograd as ra

        emb, cell = self.embedding[0], self.lstm
        packs = self._pk_t.get((emb.weight, cell.weight_ih, cell.weight_hh),
                               lambda: (ra.packed_linear(emb.weight), ra.packed_lstm(cell.weight_ih, cell.weight_hh)))
        # the step as a chain of nodes; `drop` sits where the reference applies it (:406-425): on the action embedding, on
        # the query of each of the three attentions, never on the state the cell carries forward
        emb_a = self.drop(ra.dense(action, emb.weight, emb.bias, ACT_TANH, packs[0]))
        seen, _ = self.feat_att_layer(self.drop(prev_h1), feature, output_tilde=False)
        h_1, c_1 = ra.lstm_cell(torch.cat((emb_a, seen), 1), prev_h1, c_0, cell.weight_ih, cell.weight_hh, cell.bias_ih,
                                cell.bias_hh, packs[1])
        h_att, _ = self.attention_layer(self.drop(h_1), ctx, ctx_mask)
        _, scores = self.candidate_att_layer(self.drop(h_att), cand_feat, output_prob=False)
        return h_1, c_1, scores, h_att

    def forward(self, action, feature, cand_feat, h_0, prev_h1, c_0, ctx, ctx_mask=None):
        ops._require_hip(action, feature, cand_feat, prev_h1, c_0, ctx)
        if _grad_mode(self, action, feature, cand_feat, prev_h1, c_0, ctx):
            return self._forward_autograd(action, feature, cand_feat, prev_h1, c_0, ctx, ctx_mask)
        if self.use_graph and not torch.cuda.is_current_stream_capturing():
            from .modeling import _param_key
            key = (tuple(action.shape), tuple(feature.shape), tuple(cand_feat.shape), tuple(ctx.shape), ctx_mask is None,
                   str(action.device), _param_key(self))
            g = self._graphs.get(key)
            if g is None:
                if len(self._graphs) > 8:
                    self._graphs.clear()
                g = self._graphs[key] = _DecoderStepGraph(self, action, feature, cand_feat, prev_h1, c_0, ctx, ctx_mask)
            return g.run(action, feature, cand_feat, prev_h1, c_0, ctx, ctx_mask)
        return self._step(action, feature, cand_feat, prev_h1, c_0, ctx, ctx_mask)

    def _step(self, action, feature, cand_feat, prev_h1, c_0, ctx, ctx_mask):
        """agent_models.py:406-425, inference: ten launches."""
        w = self._weights()
        action_embeds = _dense((action,), w["w_emb"], w["b_emb"], act=ACT_TANH)                 # :406
        attn_feat, _ = self.feat_att_layer.attend(prev_h1, feature, None, True, False, True)      # :411-412
        xproj = _dense((action_embeds, attn_feat), w["w_ih"], w["b"])                             # :414-417, x half
        hp = _f32c(prev_h1)
        h_1 = torch.empty_like(hp)
        c_1 = _f32c(c_0).clone()
        ops.lstm_step(xproj, hp, h_1, c_1, w["w_hh"])                                             # :417, h half + gates
        h_tilde, _ = self.attention_layer(h_1, ctx, ctx_mask)                                     # :419-420
        _, logit = self.candidate_att_layer.attend(h_tilde, cand_feat, None, False, True, False)  # :423-425
        return h_1, c_1, logit, h_tilde


class OscarEncoder(nn.Module):
    """agent_models.py:192-310: the trunk over the instruction tokens, an LSTM over its packed output, and the two
    projections that initialise the decoder."""

    def __init__(self, args, bert, hidden_size, decoder_hidden_size, dropout_ratio, bidirectional=False, num_layers=1,
                 reverse_input=False):
        super().__init__()
        self.transformer_hidden_size = 768 if bert is None else bert.config.hidden_size   # the reference hard-codes 768
        self.reverse_input = reverse_input
        self.dec_hidden_size = decoder_hidden_size
        self.args = args
        self.bert = bert
        self.hidden_size = hidden_size
        self.drop = nn.Dropout(p=dropout_ratio)
        self.num_directions = 2 if bidirectional else 1
        self.num_layers = num_layers
        self.lstm = nn.LSTM(self.transformer_hidden_size, self.hidden_size, self.num_layers, batch_first=True,
                            dropout=dropout_ratio, bidirectional=bidirectional)     # parameter container
        self.encoder_lstm2decoder_ht = nn.Linear(hidden_size * self.num_directions, decoder_hidden_size)
        self.encoder_lstm2decoder_ct = nn.Linear(hidden_size * self.num_directions, decoder_hidden_size)
        self._pk = _Packed()
        self._pk_t = _Packed()
        self.compact_rows = True    # run the trunk on the positions below `lengths` only (when the mask agrees)

    def _weights(self):
        L = self.lstm
        sfxs = ["", "_reverse"][: self.num_directions]
        ps = [getattr(L, "%s_l%d%s" % (n, l, sfx)) for l in range(self.num_layers) for sfx in sfxs
              for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        ps += [self.encoder_lstm2decoder_ht.weight, self.encoder_lstm2decoder_ct.weight]

        def build():
            layers = []
            for l in range(self.num_layers):   # nn.LSTM's stacked layers (agent_models.py:223-230): layer l > 0 reads layer l - 1's outputs
                dirs = []
                for sfx in sfxs:
                    wih, whh = getattr(L, "weight_ih_l%d%s" % (l, sfx)), getattr(L, "weight_hh_l%d%s" % (l, sfx))
                    b = (getattr(L, "bias_ih_l%d%s" % (l, sfx)).detach().float()
                         + getattr(L, "bias_hh_l%d%s" % (l, sfx)).detach().float())
                    dirs.append((_pad_weight(wih), b.contiguous(), whh.detach().to(BF16).contiguous()))
                layers.append(dirs)
            return dict(layers=layers, dirs=layers[0], w_ht=_pad_weight(self.encoder_lstm2decoder_ht.weight),
                        w_ct=_pad_weight(self.encoder_lstm2decoder_ct.weight))

        return self._pk.get(ps, build)

    @staticmethod
    def _reverse_index(att_mask):
        """reverse_input (agent_models.py:278-282): `reversed_output[att_mask] = output[:, reverse_idx][att_mask[:, reverse_idx]]`
        -- a boolean-mask assignment (a uint8 mask indexes as its non-zero pattern): the k-th marked position of a sequence
        receives the k-th marked position counted from the END, the unmarked ones stay zero.  -> (marked bool [B, S], source
        position int64 [B, S])."""
        M = att_mask != 0
        B, S = M.shape
        rank = M.long().cumsum(1) - 1                                   # rank of a marked position among its row's marked ones
        cnt = M.long().sum(1, keepdim=True)
        pos = torch.arange(S, device=M.device)[None, :].expand(B, S)
        inv = torch.zeros((B, S), dtype=torch.int64, device=M.device)   # inv[b, r] = the position of rank r
        inv.scatter_(1, torch.where(M, rank, torch.full_like(rank, S - 1)), torch.where(M, pos, inv))
        # (unmarked positions scatter their own slot's old value into rank S - 1: harmless unless the row is all marked, and
        # then nothing is unmarked)
        src = inv.gather(1, (cnt - 1 - rank).clamp_(0, S - 1))
        return M, torch.where(M, src, torch.zeros_like(src))

    def _forward_autograd(self, inputs, lens, lens_dev, T, mask, att_mask, position_ids, token_type_ids):
        """agent_models.py:256-310 as autograd nodes (training): the trunk node (the pretrain engine's forward / backward),
        one lstm_sequence node per direction, two dense nodes for the decoder's initial state."""
        from . import rollout_autograd as ra

        L, D, hs = self.lstm, self.num_directions, self.hidden_size
        B, S = inputs.shape
        output = None
        if self.compact_rows and hasattr(self.bert, "run_trunk") and self.bert.training and mask.shape == (B, S):
            # only the positions below `lengths` are read (pack_padded_sequence, :286): when those are exactly the unmasked
            # ones the trunk node may run on them alone (the engine's row compaction), forward and backward
            keep = torch.arange(S, device=inputs.device)[None, :] < lens_dev[:, None]
            if bool(((mask != 0) == ~keep).all()):
                from .training import autograd_trunk_forward

                batch = dict(input_ids=inputs, token_type_ids=token_type_ids, attention_mask=att_mask,
                             position_ids=position_ids)
                output = autograd_trunk_forward(self.bert, batch, None, unmasked_only=True)[0]
        if output is None:
            output = self.bert(inputs, token_type_ids=token_type_ids, attention_mask=att_mask,
                               position_ids=position_ids)[0].float()
        if self.reverse_input:                                         # :277-282 (differentiable: a gather and a mask)
            M, src = self._reverse_index(att_mask)
            output = torch.where(M[:, :, None], output.gather(1, src[:, :, None].expand(-1, -1, output.shape[2])),
                                 torch.zeros((), dtype=output.dtype, device=output.device))
        names = [""] + (["_reverse"] if D == 2 else [])
        NL = self.num_layers
        ps = [getattr(L, "%s_l%d%s" % (n, l, sfx)) for l in range(NL) for sfx in names for n in ("weight_ih", "weight_hh")]
        packs = self._pk_t.get(ps + [self.encoder_lstm2decoder_ht.weight, self.encoder_lstm2decoder_ct.weight], lambda: dict(
            layers=[[ra.packed_lstm(getattr(L, "weight_ih_l%d%s" % (l, sfx)), getattr(L, "weight_hh_l%d%s" % (l, sfx)))
                     for sfx in names] for l in range(NL)],
            ht=ra.packed_linear(self.encoder_lstm2decoder_ht.weight), ct=ra.packed_linear(self.encoder_lstm2decoder_ct.weight)))
        x_l = output
        for l in range(NL):
            outs = []
            for d, sfx in enumerate(names):
                outs.append(ra.lstm_sequence(x_l, getattr(L, "weight_ih_l%d%s" % (l, sfx)), getattr(L, "weight_hh_l%d%s" % (l, sfx)),
                                             getattr(L, "bias_ih_l%d%s" % (l, sfx)), getattr(L, "bias_hh_l%d%s" % (l, sfx)),
                                             lens_dev, T, d == 1, packs["layers"][l][d]))
            ctx = torch.cat((outs[0][0], outs[1][0]), 2) if D == 2 else outs[0][0]
            if l + 1 < NL:   # nn.LSTM(dropout=p): on the outputs of every layer but the last, in training mode
                x_l = torch.nn.functional.dropout(ctx, p=float(L.dropout), training=self.training)
        if D == 2:                                                     # :289-297: (reverse, forward) order for the states
            h_t = torch.cat((outs[1][1], outs[0][1]), 1)
            c_t = torch.cat((outs[1][2], outs[0][2]), 1)
        else:
            _, h_t, c_t = outs[0]
        decoder_init = ra.dense(h_t, self.encoder_lstm2decoder_ht.weight, self.encoder_lstm2decoder_ht.bias, ACT_TANH,
                                packs["ht"])                                                            # :299
        if hs * D != self.dec_hidden_size:
            c_t = ra.dense(c_t, self.encoder_lstm2decoder_ct.weight, self.encoder_lstm2decoder_ct.bias, ACT_NONE,
                           packs["ct"])                                                                 # :300-301
        return self.drop(ctx), decoder_init, c_t                                                        # :303-307

    def forward(self, inputs, lengths, mask, position_ids=None, token_type_ids=None):
        ops._require_hip(inputs)
        att_mask = ~mask                                               # :267 (uint8 masks: 254/255, the trunk keeps that)
        B, S = inputs.shape
        H = self.transformer_hidden_size
        lens = torch.as_tensor(lengths).to("cpu", torch.int64)
        if lens.numel() != B or int(lens.min()) <= 0 or int(lens.max()) > S:
            raise RuntimeError("lengths must hold one value in 1..%d per sequence" % S)
        if bool((lens[1:] > lens[:-1]).any()):   # pack_padded_sequence(enforce_sorted=True), :286
            raise RuntimeError("`lengths` array must be sorted in decreasing order")
        T = int(lens.max())
        dev = inputs.device
        lens_dev = lens.to(dev, torch.int32)
        if _grad_mode(self):
            return self._forward_autograd(inputs, lens, lens_dev, T, mask, att_mask, position_ids, token_type_ids)
        # Only the first lengths[b] positions of a sequence are read below (pack_padded_sequence, :286).  When those are
        # exactly the unmasked ones (mask = 1 on padding, agent.py:181) the trunk runs on them alone: compacted rows,
        # no masked keys -- the same values at the positions that are read.
        lay = None
        if (self.compact_rows and hasattr(self.bert, "run_trunk") and mask.shape == (B, S) and not ops.profiling()
                and not self.reverse_input):   # (reverse_input re-orders the padded positions: the padded layout)
            keep = torch.arange(S, device=dev)[None, :] < lens_dev[:, None]
            if bool(((mask != 0) == ~keep).all()):
                outs, _, _, _, _ = self.bert.run_trunk(inputs, token_type_ids, None, position_ids, keep=keep)
                lay = self.bert._last_layout
                seq = outs[-1][:lay.rows]
        if lay is not None:
            pass
        elif hasattr(self.bert, "run_trunk"):                          # the bf16 rows, without the fp32 round trip
            outs, _, _, _, _ = self.bert.run_trunk(inputs, token_type_ids, att_mask, position_ids)
            seq = outs[-1]
        else:
            seq = self.bert(inputs, token_type_ids=token_type_ids, attention_mask=att_mask,
                            position_ids=position_ids)[0].detach().reshape(B * S, H).to(BF16).contiguous()
        if self.reverse_input:                                         # :277-282, on the padded bf16 rows
            M, src = self._reverse_index(att_mask)
            rows = (torch.arange(B, device=dev)[:, None] * S + src).reshape(-1)
            seq = seq.index_select(0, rows) * M.reshape(-1, 1).to(seq.dtype)
        w = self._weights()
        hs, D = self.hidden_size, self.num_directions
        S_l = S
        for l, dirs in enumerate(w["layers"]):
            ctx = torch.empty((B, T, D * hs), dtype=torch.float32, device=dev)
            finals = []
            for d, (w_ih, b, w_hh) in enumerate(dirs):
                xproj = torch.empty((seq.shape[0], 4 * hs), dtype=torch.float32, device=dev)
                ops.linear(seq, w_ih, b, out=xproj, out_f32=True)
                h2 = (torch.zeros((B, hs), dtype=torch.float32, device=dev), torch.empty((B, hs), dtype=torch.float32, device=dev))
                c = torch.zeros((B, hs), dtype=torch.float32, device=dev)                      # init_state :238-254
                if lay is None:
                    ops.lstm_sequence(xproj.view(B, S_l, 4 * hs), h2, c, w_hh, T, lens_dev, ctx[:, :, d * hs:(d + 1) * hs],
                                      reverse=(d == 1))
                else:
                    ops.lstm_sequence_rows(xproj, lay.start, h2, c, w_hh, T, lens_dev, ctx[:, :, d * hs:(d + 1) * hs],
                                           reverse=(d == 1))
                finals.append((h2[0], c))
            if l + 1 < len(w["layers"]):   # the next stacked layer reads this one's (padded, zero beyond the lengths) outputs
                Kp = round_up(D * hs, 64)
                nxt = torch.zeros((B * T, Kp), dtype=BF16, device=dev)
                nxt[:, :D * hs] = ctx.reshape(B * T, D * hs)
                seq, lay, S_l = nxt, None, T
        if D == 2:                                                     # :289-294: (reverse, forward) order
            h_t = torch.cat((finals[1][0], finals[0][0]), 1)
            c_t = torch.cat((finals[1][1], finals[0][1]), 1)
        else:
            h_t, c_t = finals[0]
        decoder_init = _dense((h_t,), w["w_ht"], _f32c(self.encoder_lstm2decoder_ht.bias), act=ACT_TANH)   # :299
        if hs * D != self.dec_hidden_size:
            c_t = _dense((c_t,), w["w_ct"], _f32c(self.encoder_lstm2decoder_ct.bias))                       # :300-301
        return ctx, decoder_init, c_t
