"""Drop-in modules for the VISITRON encoder hot path, running on MI355X HIP kernels.

Same constructor ``(config)``, ``forward`` signatures, returned tuples and ``state_dict`` key
names as the reference classes, so the reference's pretrain / agent loops can use these in place
of their own:

  CaptionBertSelfAttention / CaptionBertAttention / CaptionBertLayer / CaptionBertEncoder
        oscar/modeling_bert.py:26-169
  NextActionPrediction, BertImgModelwithLocationEmbeds, PreTrainOscar
        tasks/viewpoint_select/encoder.py:142-158, 161-303, 306-441
  (their BERT building blocks -- BertEmbeddings, BertSelfOutput, BertIntermediate, BertOutput,
   BertPooler, BertOnlyMLMHead, BertLayerNorm, BertPreTrainedModel -- come from the reference's
   un-vendored pytorch-transformers dependency; here they are parameter containers with the
   upstream attribute names.)

The parameters stay fp32 ``nn.Parameter``s (checkpoint / optimizer / DDP compatible); the kernels
consume bf16 packed copies (query|key|value fused into one [3H,H] weight, image + location
projections K-concatenated) that are rebuilt whenever a parameter's version or storage changes.

There is no CPU path: every forward requires HIP tensors and ``libvisitron_hip.so``.
"""
import ctypes
import logging
import os

import torch
from torch import nn

from . import _lib, ops
from .config import BertConfig
from .ops import ACT_GELU, ACT_NONE, ACT_TANH, BF16, round_up

logger = logging.getLogger(__name__)

WEIGHTS_NAME = "pytorch_model.bin"


# --------------------------------------------------------------------------------------------
# parameter containers (upstream attribute names => reference state_dict keys)
# --------------------------------------------------------------------------------------------
class BertLayerNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-12):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.bias = nn.Parameter(torch.zeros(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        shp = x.shape
        y = ops.layernorm(_as_bf16_2d(x), _f32(self.weight), _f32(self.bias), self.variance_epsilon)
        return y.view(shp).to(x.dtype)


class BertEmbeddings(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def write_rows(self, out, S, input_ids, token_type_ids=None, position_ids=None):
        """Fused gather-sum-LayerNorm into rows b*S+t of ``out`` [B*S,H] bf16."""
        _no_train_dropout(self, self.dropout.p)
        err = torch.zeros(1, dtype=torch.int32, device=out.device)
        ops.embed_layernorm(
            _i64(input_ids), _i64(token_type_ids), _i64(position_ids),
            _f32(self.word_embeddings.weight), _f32(self.position_embeddings.weight),
            _f32(self.token_type_embeddings.weight), _f32(self.LayerNorm.weight), _f32(self.LayerNorm.bias),
            self.LayerNorm.variance_epsilon, out, S, err_flag=err)
        self._last_err = err
        return out

    def forward(self, input_ids, token_type_ids=None, position_ids=None):
        B, T = input_ids.shape
        out = torch.empty((B * T, self.word_embeddings.weight.shape[1]), dtype=BF16, device=input_ids.device)
        self.write_rows(out, T, input_ids, token_type_ids, position_ids)
        _check_index_error(self)
        return out.view(B, T, -1).to(self.word_embeddings.weight.dtype)


class BertSelfOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        return _dense_residual_ln(self, hidden_states, input_tensor)


class BertIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)
        if config.hidden_act != "gelu":
            raise NotImplementedError("the HIP path implements hidden_act='gelu' (erf form) only")

    def forward(self, hidden_states):
        shp = hidden_states.shape
        y = ops.linear(_as_bf16_2d(hidden_states), _bf16(self.dense.weight), _f32(self.dense.bias), act=ACT_GELU)
        return y.view(shp[:-1] + (y.shape[-1],)).to(hidden_states.dtype)


class BertOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        return _dense_residual_ln(self, hidden_states, input_tensor)


class BertPooler(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.activation = nn.Tanh()

    def pooled(self, seq_bf16, B, S):
        """tanh(dense(h[:,0])) -> fp32 [B,H]; reads token 0 of each sequence via the row stride."""
        H = self.dense.weight.shape[0]
        key = _param_key((self.dense.weight,))
        if getattr(self, "_w16_key", None) != key:   # the bf16 copy is kept until the weight changes (5 us per call otherwise)
            object.__setattr__(self, "_w16", _bf16(self.dense.weight))
            object.__setattr__(self, "_w16_key", key)
        return ops.linear(seq_bf16, self._w16, _f32(self.dense.bias), act=ACT_TANH, out_f32=True, M=B, lda=S * H)

    def forward(self, hidden_states):
        B, S, H = hidden_states.shape
        return self.pooled(_as_bf16_2d(hidden_states), B, S).to(hidden_states.dtype)


class BertPredictionHeadTransform(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)


class BertLMPredictionHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.transform = BertPredictionHeadTransform(config)
        self.decoder = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(config.vocab_size))


class BertOnlyMLMHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.predictions = BertLMPredictionHead(config)

    def scores(self, seq_bf16):
        """decoder(LN(gelu(dense(h)))) + bias -> fp32 [M, V] (view of a 16-B-row-aligned buffer)."""
        p = self.predictions
        t = ops.linear(seq_bf16, _bf16(p.transform.dense.weight), _f32(p.transform.dense.bias), act=ACT_GELU)
        t = ops.layernorm(t, _f32(p.transform.LayerNorm.weight), _f32(p.transform.LayerNorm.bias),
                          p.transform.LayerNorm.variance_epsilon, out=t)
        V = p.decoder.weight.shape[0]
        buf = torch.empty((seq_bf16.shape[0], round_up(V, 4)), dtype=torch.float32, device=seq_bf16.device)
        ops.linear(t, _bf16(p.decoder.weight), _f32(p.bias), out=buf, out_f32=True)
        return buf[:, :V]

    def forward(self, sequence_output):
        shp = sequence_output.shape
        s = self.scores(_as_bf16_2d(sequence_output))
        return s.reshape(shp[:-1] + (s.shape[-1],)).to(sequence_output.dtype)


class BertPreTrainedModel(nn.Module):
    """The base-class services the reference uses (init / tie / resize / save / load)."""

    config_class = BertConfig

    def __init__(self, config):
        super().__init__()
        self.config = config

    def init_weights(self, module):
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
        elif isinstance(module, BertLayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()

    def _tie_or_clone_weights(self, first_module, second_module):
        if getattr(self.config, "torchscript", False):
            first_module.weight = nn.Parameter(second_module.weight.clone())
        else:
            first_module.weight = second_module.weight

    def _get_resized_embeddings(self, old_embeddings, new_num_tokens=None):
        if new_num_tokens is None:
            return old_embeddings
        old_num_tokens, dim = old_embeddings.weight.size()
        if old_num_tokens == new_num_tokens:
            return old_embeddings
        new_embeddings = nn.Embedding(new_num_tokens, dim).to(old_embeddings.weight.device)
        self.init_weights(new_embeddings)
        n = min(old_num_tokens, new_num_tokens)
        new_embeddings.weight.data[:n, :] = old_embeddings.weight.data[:n, :]
        return new_embeddings

    def save_pretrained(self, save_directory):
        """config.json + pytorch_model.bin, the layout tasks/viewpoint_select/pretrain.py:263-269 writes."""
        assert os.path.isdir(save_directory), "Saving path should be a directory where the model and configuration can be saved"
        model_to_save = self.module if hasattr(self, "module") else self
        model_to_save.config.save_pretrained(save_directory)
        torch.save(model_to_save.state_dict(), os.path.join(save_directory, WEIGHTS_NAME))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, **kwargs):
        """Load ``pytorch_model.bin`` by key name (model_utils.py:89-93); keys absent from the file keep
        their fresh init, unexpected keys are ignored, as upstream does; TF checkpoints are not read."""
        config = kwargs.pop("config", None)
        if kwargs.pop("from_tf", False):
            raise NotImplementedError("TensorFlow checkpoints are not supported")
        if config is None:
            config = cls.config_class.from_pretrained(pretrained_model_name_or_path)
        model = cls(config, *model_args)
        path = pretrained_model_name_or_path
        if os.path.isdir(path):
            path = os.path.join(path, WEIGHTS_NAME)
        state = torch.load(path, map_location="cpu")
        # old-checkpoint LayerNorm naming, as upstream
        state = {k.replace("gamma", "weight").replace("beta", "bias") if k.endswith(("gamma", "beta")) else k: v
                 for k, v in state.items()}
        own = model.state_dict()
        prefix = ""
        if not any(k.startswith("bert.") for k in own) and any(k.startswith("bert.") for k in state):
            prefix = "bert."  # loading a trunk from a full-model checkpoint
        filtered = {}
        for k, v in state.items():
            kk = k[len(prefix):] if prefix and k.startswith(prefix) else k
            if kk in own and own[kk].shape == v.shape:
                filtered[kk] = v
        missing = [k for k in own if k not in filtered]
        if missing:
            logger.info("Weights of %s not initialized from pretrained model: %s", cls.__name__, missing)
        model.load_state_dict(filtered, strict=False)
        if hasattr(model, "tie_weights"):
            model.tie_weights()
        model.eval()
        return model


# --------------------------------------------------------------------------------------------
# small host helpers
# --------------------------------------------------------------------------------------------
def _f32(t):
    if t is None:
        return None
    t = t.detach()
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def _bf16(t):
    return t.detach().to(BF16).contiguous()


def _i64(t):
    if t is None:
        return None
    return t if (t.dtype == torch.int64 and t.is_contiguous()) else t.to(torch.int64).contiguous()


def _as_bf16_2d(x):
    ops._require_hip(x)
    return x.detach().reshape(-1, x.shape[-1]).to(BF16).contiguous()


def _no_train_dropout(module, p):
    if module.training and p > 0.0:
        raise NotImplementedError(
            "dropout p=%g in training mode is served by the fused training path only (PreTrainOscar.forward with "
            "grad enabled, or PretrainEngine); use model.eval() for module-level / no-grad calls" % p
        )


# Out-of-range ids: the embedding kernels skip the row and raise a device flag.  torch.nn.Embedding on a GPU reports the
# same mistake through a device-side assert, i.e. asynchronously; here the flag travels to pinned host memory behind the
# embedding launch and is looked at WITHOUT blocking at the end of the call (and at the start of the next one): a blocking
# read-back at the end of every forward drains the launch queue between two calls (measured at B = 64: 3.58 ms per forward
# without it, 3.8-4.2 ms with it).  A flag that has landed raises IndexError there; one that has not (the stream still has
# earlier work queued) raises at the next call or at check_errors().  VT_SYNC_ERRORS=1: block at the end of every forward.
_PENDING_FLAGS = []   # (event, pinned int32[1])
_FREE_FLAGS = []
_INDEX_MSG = "index out of range in BertEmbeddings (input_ids / position_ids / token_type_ids)"


def _post_index_flag(err):
    host = _FREE_FLAGS.pop() if _FREE_FLAGS else torch.empty(1, dtype=torch.int32).pin_memory()
    host.copy_(err, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    _PENDING_FLAGS.append((ev, host))


def _poll_index_flags(block=False):
    if not _PENDING_FLAGS or torch.cuda.is_current_stream_capturing():
        return
    bad, i = False, 0
    while i < len(_PENDING_FLAGS):
        ev, host = _PENDING_FLAGS[i]
        if block:
            ev.synchronize()
        if block or ev.query():
            bad = bad or int(host[0]) != 0
            _FREE_FLAGS.append(host)
            _PENDING_FLAGS.pop(i)
        else:
            i += 1
    if bad:
        raise IndexError(_INDEX_MSG)


def check_errors():
    """Wait for the error flags of the calls issued so far and raise what they report (IndexError for ids outside an
    embedding table).  The forward calls themselves only look at flags that have already arrived."""
    _poll_index_flags(block=True)


def _check_index_error(emb):
    err = getattr(emb, "_last_err", None)
    if err is None:
        return
    emb._last_err = None
    if torch.cuda.is_current_stream_capturing():
        return   # a forward being captured into a HIP graph: no host-side look at the flag (events cannot be queried there)
    _post_index_flag(err)
    _poll_index_flags(block=os.environ.get("VT_SYNC_ERRORS") == "1")


def _dense_residual_ln(mod, hidden_states, input_tensor):
    """LN(dropout(dense(h)) + residual) for BertSelfOutput / BertOutput (eval or p == 0)."""
    _no_train_dropout(mod, mod.dropout.p)
    shp = input_tensor.shape
    pre = ops.linear(_as_bf16_2d(hidden_states), _bf16(mod.dense.weight), _f32(mod.dense.bias),
                     residual=_as_bf16_2d(input_tensor))
    y = ops.layernorm(pre, _f32(mod.LayerNorm.weight), _f32(mod.LayerNorm.bias), mod.LayerNorm.variance_epsilon, out=pre)
    return y.view(shp).to(input_tensor.dtype)


def _refuse_data_parallel_replica(module):
    """torch.nn.DataParallel (the reference's `multi-gpu-dp` mode, pretrain.py:93-94) re-creates the module per forward as
    replicas whose weights are plain broadcast tensors (no Parameters): the flat slabs, the packed bf16 copies and the saved
    activations here belong to ONE device and one module object.  Refused with the ways out instead of failing somewhere
    inside: visitron_amd.parallel.DataParallel (the same one line, replicas that persist), or one process per GPU (the
    reference's `multi-gpu-ddp` mode, pretrain.py:96-102; PretrainEngine / bench.py --gpus N)."""
    if hasattr(module, "_former_parameters"):
        raise NotImplementedError("torch.nn.DataParallel replicas are not served by the HIP path; wrap the model in "
                                  "visitron_amd.parallel.DataParallel (same constructor, persistent per-device replicas) or "
                                  "run one process per GPU (DistributedDataParallel, or PretrainEngine with "
                                  "torch.distributed)")


def _centered_mask(mask_f32):   # (any dtype ops.center_mask takes; fp32 out)
    """A per-key mask [B, S] shifted so that each sequence's largest value is 1: (1 - m) * -10000 then changes by one
    constant per sequence, which a softmax over the keys does not see (encoder.py:238-241; oscar/modeling_bert.py:55-58).
    A 0/1 mask with a kept key is unchanged bit for bit.  What it is for: the rollout caller's `~mask` of a uint8 tensor
    (254 / 255, agent_models.py:267) turns into biases of +2.53e6 / +2.54e6, where fp32 resolves 0.25 -- the reference's
    own softmax is then computed on scores rounded to that grid, and a log-sum-exp of that size cannot carry the backward's
    recomputation.  Shifted, the same mask is 0 / 1 and every kernel sees well-scaled numbers.  bf16 paths only: the fp32
    parity path keeps the literal arithmetic and reproduces the reference's rounded scores (tests/test_gpu_fp32.py)."""
    return ops.center_mask(mask_f32)   # (mask - mask.amax(1, keepdim) + 1 in fp32, from the caller's dtype, one launch)


def _additive_mask_2d(attention_mask, B, S):
    """The additive extended mask CaptionBertEncoder.forward receives: [B,1,1,S] (or [B,S]) -> contiguous fp32 [B,S];
    [B,1,S,S] (the reference's 3-D attention_mask, encoder.py:226-229) -> contiguous fp32 [B,S,S]."""
    m = attention_mask
    if m.dim() == 4 and m.shape[1] == 1 and m.shape[2] == 1:
        m = m[:, 0, 0, :]
    elif m.dim() == 4 and m.shape[1] == 1 and m.shape[2] == S:
        m = m[:, 0]
        if m.shape[0] != B:
            m = m.expand(B, -1, -1)
        if m.shape[2] != S:
            raise RuntimeError("attention mask length %d does not match sequence length %d" % (m.shape[2], S))
        return m.to(torch.float32).contiguous()
    elif m.dim() != 2:
        raise NotImplementedError("attention mask of shape %s (per-head masks are not served)" % (tuple(attention_mask.shape),))
    if m.shape[0] != B:
        m = m.expand(B, -1)
    if m.shape[1] != S:
        raise RuntimeError("attention mask length %d does not match sequence length %d" % (m.shape[1], S))
    return m.to(torch.float32).contiguous()


def _attention_with_history(x_bf16, history_state, w_qkv, b_qkv, attention_mask, B, S, nh, head_scale, want_probs,
                            mask_f32=None, mask_additive=True):
    """One layer's self-attention (oscar/modeling_bert.py:34-79): x_bf16 [B*S, H]; with history_state [B, Sh, H] the
    keys and values run over cat([history, hidden], 1) (:37-41) -- the packed projection is taken over the
    concatenated rows and the fused kernel over Sh + S positions, of which the last S query rows are the layer's
    context (the history rows' own queries are computed and dropped).  attention_mask: the additive extended mask over
    the Sh + S keys (or pass mask_f32 [B, Sh+S] with its mask_additive flag).  -> (ctx bf16 [B*S, H], probs or None)."""
    H = nh * 64
    Sh = 0
    xs = x_bf16
    if history_state is not None:
        Sh = history_state.shape[1]
        xs = torch.cat([history_state.detach().to(BF16), x_bf16.view(B, S, H)], 1).reshape(B * (Sh + S), H).contiguous()
    St = Sh + S
    qkv = ops.linear(xs, w_qkv, b_qkv)
    mask = mask_f32
    if mask is None and attention_mask is not None:
        mask = _additive_mask_2d(attention_mask, B, St)
    if mask is not None and mask.dim() == 3 and Sh:
        raise NotImplementedError("per-query attention masks together with history states are not served")
    lse = torch.empty((B, nh, St), dtype=torch.float32, device=x_bf16.device) if want_probs else None
    ctx = ops.attention_fwd(qkv, B, St, nh, mask=mask, mask_additive=mask_additive, head_scale=head_scale, lse=lse)
    probs = None
    if want_probs:
        probs = ops.attention_probs(qkv, lse, B, St, nh, mask=mask, mask_additive=mask_additive, head_scale=head_scale)
        if Sh:
            probs = probs[:, :, Sh:, :].contiguous()
    if Sh:
        ctx = ctx.view(B, St, H)[:, Sh:].reshape(B * S, H).contiguous()
    return ctx, probs


def _head_scale(head_mask, L, nh, device):
    """The reference's head_mask list/tensor -> fp32 [L, nh] per-head multipliers (or None)."""
    if head_mask is None:
        return None
    if isinstance(head_mask, (list, tuple)):
        if all(h is None for h in head_mask):
            return None
        rows = []
        for h in head_mask:
            if h is None:
                rows.append(torch.ones(nh, device=device))
            else:
                if h.numel() != nh:
                    raise NotImplementedError("head_mask entries must hold one multiplier per head")
                rows.append(h.reshape(nh).to(device=device, dtype=torch.float32))
        return torch.stack(rows).contiguous()
    if head_mask.numel() == L * nh:
        return head_mask.reshape(L, nh).to(device=device, dtype=torch.float32).contiguous()
    if head_mask.numel() == nh:
        return head_mask.reshape(1, nh).expand(L, nh).to(device=device, dtype=torch.float32).contiguous()
    raise NotImplementedError("head_mask must broadcast from [heads] or [layers, heads]")


# --------------------------------------------------------------------------------------------
# oscar/modeling_bert.py:26-169
# --------------------------------------------------------------------------------------------
class CaptionBertSelfAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError(
                "The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                % (config.hidden_size, config.num_attention_heads)
            )
        self.output_attentions = config.output_attentions
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = config.hidden_size // config.num_attention_heads
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)

    def packed_qkv(self):
        w = torch.cat([self.query.weight, self.key.weight, self.value.weight], 0).detach().to(BF16).contiguous()
        b = torch.cat([self.query.bias, self.key.bias, self.value.bias], 0).detach().float().contiguous()
        return w, b

    def forward(self, hidden_states, attention_mask, head_mask=None, history_state=None):
        """oscar/modeling_bert.py:34-79 for one layer: packed QKV GEMM + fused attention kernel."""
        if self.attention_head_size != 64:
            raise NotImplementedError("the fused attention kernel serves head size 64")
        _no_train_dropout(self, self.dropout.p)
        B, S, H = hidden_states.shape
        w, b = self.packed_qkv()
        hs = _head_scale([head_mask], 1, self.num_attention_heads, hidden_states.device)
        hs0 = None if hs is None else hs[0].contiguous()
        ctx, probs = _attention_with_history(_as_bf16_2d(hidden_states), history_state, w, b, attention_mask, B, S,
                                             self.num_attention_heads, hs0, self.output_attentions)
        outputs = (ctx.view(B, S, H).to(hidden_states.dtype),)
        if self.output_attentions:   # :74-79: the probabilities after dropout (identity in eval) and head_mask
            outputs = outputs + (probs.to(hidden_states.dtype),)
        return outputs


class CaptionBertAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self = CaptionBertSelfAttention(config)
        self.output = BertSelfOutput(config)

    def forward(self, input_tensor, attention_mask, head_mask=None, history_state=None):
        self_outputs = self.self(input_tensor, attention_mask, head_mask, history_state)
        attention_output = self.output(self_outputs[0], input_tensor)
        return (attention_output,) + self_outputs[1:]


class CaptionBertLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.attention = CaptionBertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)

    def forward(self, hidden_states, attention_mask, head_mask=None, history_state=None):
        attention_outputs = self.attention(hidden_states, attention_mask, head_mask, history_state)
        attention_output = attention_outputs[0]
        intermediate_output = self.intermediate(attention_output)
        layer_output = self.output(intermediate_output, attention_output)
        return (layer_output,) + attention_outputs[1:]


class _PackedEncoder(object):
    """bf16 weight copies + the ctypes weight table for the C layer loop; rebuilt on change."""

    def __init__(self, encoder):
        keep, table = [], (_lib.LayerWeights * len(encoder.layer))()
        for i, layer in enumerate(encoder.layer):
            att, so = layer.attention.self, layer.attention.output
            w_qkv, b_qkv = att.packed_qkv()
            t = dict(
                w_qkv=w_qkv, b_qkv=b_qkv,
                w_ao=_bf16(so.dense.weight), b_ao=_f32(so.dense.bias),
                ln1_g=_f32(so.LayerNorm.weight), ln1_b=_f32(so.LayerNorm.bias),
                w_in=_bf16(layer.intermediate.dense.weight), b_in=_f32(layer.intermediate.dense.bias),
                w_out=_bf16(layer.output.dense.weight), b_out=_f32(layer.output.dense.bias),
                ln2_g=_f32(layer.output.LayerNorm.weight), ln2_b=_f32(layer.output.LayerNorm.bias),
            )
            keep.append(t)
            for k, v in t.items():
                setattr(table[i], k, v.data_ptr())
        self.tensors, self.table = keep, table


class _PackedEncoderLn(object):
    """Weights of the deferred-LayerNorm layer loop (vt_encoder_forward_ln_bf16; include/visitron_hip.h has the
    arithmetic).  The LayerNorm that closes sub-layer i is applied where its output is consumed, so its gamma / beta are
    folded into the CONSUMING projections: W' = W * gamma over the input columns (bf16), g = row sums of W' (as the kernel
    will see it, after rounding), h = W beta + b; the dense + residual GEMMs take bias + beta and gamma as vectors.
    Layer 0's input (the embedding output) is not normalised again: gamma 1, beta 0."""

    def __init__(self, encoder):
        H = encoder._hidden
        dev = next(encoder.parameters()).device
        keep, table = [], (_lib.LayerWeightsLn * len(encoder.layer))()
        gamma_in = torch.ones(H, dtype=torch.float32, device=dev)
        beta_in = torch.zeros(H, dtype=torch.float32, device=dev)

        def fold(W, b, gamma, beta):
            W = W.detach().float()
            wf = (W * gamma[None, :]).to(BF16).contiguous()
            # h = W beta + b as an elementwise product and a row sum (torch's `W @ beta` would be the one vendor-BLAS
            # kernel, rocblas_gemvt, in a trace of this library; weight prep only, cached per weights generation)
            return wf, wf.float().sum(1).contiguous(), ((W * beta[None, :]).sum(1) + b.detach().float()).contiguous()

        for i, layer in enumerate(encoder.layer):
            att, so = layer.attention.self, layer.attention.output
            Wqkv = torch.cat([att.query.weight, att.key.weight, att.value.weight], 0)
            bqkv = torch.cat([att.query.bias, att.key.bias, att.value.bias], 0)
            w_qkv, g_qkv, h_qkv = fold(Wqkv, bqkv, gamma_in, beta_in)
            g1, b1 = _f32(so.LayerNorm.weight), _f32(so.LayerNorm.bias)
            w_in, g_in, h_in = fold(layer.intermediate.dense.weight, layer.intermediate.dense.bias, g1, b1)
            t = dict(
                w_qkv=w_qkv, g_qkv=g_qkv, h_qkv=h_qkv,
                w_ao=_bf16(so.dense.weight), cb_ao=(_f32(so.dense.bias) + beta_in).contiguous(), gamma_in=gamma_in.contiguous(),
                w_in=w_in, g_in=g_in, h_in=h_in,
                w_out=_bf16(layer.output.dense.weight), cb_out=(_f32(layer.output.dense.bias) + b1).contiguous(), ln1_g=g1,
            )
            keep.append(t)
            for k, v in t.items():
                setattr(table[i], k, v.data_ptr())
            gamma_in, beta_in = _f32(layer.output.LayerNorm.weight), _f32(layer.output.LayerNorm.bias)
        self.tensors, self.table = keep, table
        self.final_gamma, self.final_beta = gamma_in, beta_in


# Packed bf16 weight copies (encoder layers, region projection, rollout modules) are cached and keyed on each
# parameter's (storage address, version) PLUS this process-wide generation.  Writers that change parameter values
# without bumping ``_version`` -- the fused AdamW kernel (raw pointers into the flat slab) and torch optimizers that
# update through ``p.data`` (the reference's pytorch-transformers AdamW does) -- are covered by the generation: the
# training engine bumps it after every optimizer step and the autograd bridge before every training forward (an
# external optimizer is expected to step after it).  Code that edits ``p.data`` by hand between two inference calls
# must call ``invalidate_packed_weights()`` itself.
_WEIGHTS_GEN = [0]


def invalidate_packed_weights():
    """Drop every cached packed-weight copy in this process (they are rebuilt on the next forward)."""
    _WEIGHTS_GEN[0] += 1


PRECISIONS = ("bf16", "fp32")


def set_precision(model, precision):
    """Select the arithmetic of `model`'s HIP forward: "bf16" (default: bf16 operands on the bf16 matrix cores, fp32
    accumulation -- the performance path, within 5e-2 of the fp32 reference) or "fp32" (every operand, activation and
    accumulation in fp32 on the fp32 matrix cores -- the parity path, within 1e-3 of the reference, which computes in
    fp32 itself: tasks/viewpoint_select/encoder.py:238-240).  "fp32" serves inference at encoder / trunk / model level;
    training stays on the bf16 kernels."""
    if precision not in PRECISIONS:
        raise ValueError("precision must be one of %s" % (PRECISIONS,))
    for m in model.modules():
        object.__setattr__(m, "_vt_precision", precision)
    return model


def _is_fp32(module):
    return getattr(module, "_vt_precision", "bf16") == "fp32"


def _param_key(module_or_params):
    ps = module_or_params.parameters() if isinstance(module_or_params, nn.Module) else module_or_params
    return (_WEIGHTS_GEN[0],) + tuple((p.data_ptr(), p._version) for p in ps)


class CaptionBertEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.output_attentions = config.output_attentions
        self.output_hidden_states = config.output_hidden_states
        self.layer = nn.ModuleList([CaptionBertLayer(config) for _ in range(config.num_hidden_layers)])
        self._hidden = config.hidden_size
        self._heads = config.num_attention_heads
        self._inter = config.intermediate_size
        self._eps = config.layer_norm_eps
        self._packed = None
        self._packed_key = None
        self._ws = {}
        self._final_f32, self._final_f32_fresh = None, False
        # inference, opt-in (VT_PRECISE_FINAL=1 or the attribute): the last layer's pre-LayerNorm sums and its output also
        # in fp32 (see run()).  It takes the returned hidden states' max-abs error against the fp32 reference from 5.9e-2 to
        # 4.9e-2 on the base config (rms 1.06e-2 -> 1.02e-2: the bf16 weights of the twelve layers set that, not the last
        # roundings) and costs 6 % of a B = 64 forward (op-by-op last layer, two fp32-output GEMM epilogues): off by default.
        self.precise_final = os.environ.get("VT_PRECISE_FINAL", "0") == "1"
        # inference, default: the layer loop with its LayerNorms deferred (run_ln): no LayerNorm pass, fp16 residual stream.
        # VT_DEFERRED_LN=0 (or the attribute) keeps the seven-launch layer with bf16 activations between all kernels.
        self.deferred_ln = os.environ.get("VT_DEFERRED_LN", "1") != "0"
        self.deferred_ln_min_rows = int(os.environ.get("VT_DEFERRED_LN_MIN_ROWS", "2800"))   # see serves_deferred_ln
        self._packed_ln = None
        self._packed_ln_key = None

    # ---- cached state -----------------------------------------------------------------
    def packed(self):
        key = _param_key(self)
        if self._packed is None or key != self._packed_key:
            self._packed = _PackedEncoder(self)
            self._packed_key = key
        return self._packed

    def packed_ln(self):
        key = _param_key(self)
        if self._packed_ln is None or key != self._packed_ln_key:
            self._packed_ln = _PackedEncoderLn(self)
            self._packed_ln_key = key
        return self._packed_ln

    def serves_deferred_ln(self, history=None, seq=None, rows=None):
        """The deferred-LayerNorm loop serves the plain eval forward (padded rows with a mask, or compacted rows: run_ln's
        seq): hidden size a multiple of 128 (<= 1024: eight statistics slices), no per-layer outputs asked for -- and, when
        the caller says how many token rows (padded) the forward has, at least deferred_ln_min_rows of them: below ~2 800
        rows the seven-launch layer is faster (its plain GEMMs may take the split-K and three-stage kernels; the LN-mode
        GEMMs exist in the 256-wide kernels only: forward B = 1 ... 10 x 228 rows 1.01 ... 1.32 ms against 1.36 ... 1.47,
        round 6, tools/r6/deferred_ab2.sh) and as accurate (fp16 residual stream since round 4: 3.6e-2 against 3.4e-2 on
        the base configuration's hidden states, bound 5e-2)."""
        if rows is not None and rows < self.deferred_ln_min_rows:
            return False
        return (self.deferred_ln and history is None and seq is None and not self.output_attentions
                and not self.output_hidden_states and self._hidden == self._heads * 64 and self._hidden % 128 == 0
                and self._hidden <= 1024 and self._inter % 128 == 0)

    def _workspace_ln(self, M, device):
        key = ("ln", M, str(device))
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        H, I = self._hidden, self._inter
        rows, np_ = round_up(M, 16), H // 128
        bf = lambda n: torch.empty((M, n), dtype=BF16, device=device)
        f16 = lambda n: torch.empty((M, n), dtype=torch.float16, device=device)
        f32 = lambda n: torch.empty((M, n), dtype=torch.float32, device=device)
        st = lambda: torch.zeros((np_, rows, 2), dtype=torch.float32, device=device)
        # a / b: the two residual streams (bf16 copy, fp16 rows, statistics); x32: the embedding output entering layer 0
        ws = dict(a=(bf(H), f16(H), st()), b=(bf(H), f16(H), st()), x32=f32(H), qkv=bf(3 * H), ctx=bf(H), mid=bf(I), out16=bf(H))
        ops.autotune_encoder_shapes_ln(M, H, I, device=device)   # once per token count
        if len(self._ws) > 4:
            self._ws.clear()
        self._ws[key] = ws
        return ws

    def run_ln(self, x32, B, S, mask_f32, mask_additive, head_scale=None, seq=None):
        """The deferred-LayerNorm layer loop: x32 fp32 [B*S, H] (the embedding output) -> (sequence output bf16 [B*S, H], the
        same in fp32).  Both live in the workspace and are rewritten by the next call.  seq (ops.SeqLayout): x32 holds the
        compacted rows [seq.rows, H] (no masked positions, no mask) and so do the first seq.rows rows of the bf16 output;
        no fp32 copy is made then."""
        for layer in self.layer:
            _no_train_dropout(layer, layer.attention.self.dropout.p)
            _no_train_dropout(layer, layer.output.dropout.p)
        pk = self.packed_ln()
        M = B * S
        ws = self._workspace_ln(M, x32.device)
        sa, sb = ws["a"], ws["b"]
        H, nh, I, eps = self._hidden, self._heads, self._inter, self._eps
        self._last_attentions = None
        if seq is not None:
            if mask_f32 is not None or x32.shape[0] != seq.rows:
                raise ValueError("compacted rows: [seq.rows, H] without a mask")
            # the row count decides the tile quantisation: tuned once per 2048-row bucket (the nearest tuned M is used)
            ops.autotune_encoder_shapes_ln(min(M, round_up(seq.rows, 2048)), H, I, device=x32.device)
            ops.ln_stream_init(x32, sa[1], sa[0], sa[2], eps)
            ops.encoder_forward_ln(pk.table, sa, sb, ws["qkv"], ws["ctx"], ws["mid"], None, False, head_scale, B, S, H, nh, I, eps,
                                   seq=seq)
            ops.ln_apply(sa[1], sa[2], pk.final_gamma, pk.final_beta, eps, out16=ws["out16"], M=seq.rows)
            self._final_f32, self._final_f32_fresh = None, False
            return ws["out16"], None
        ops.ln_stream_init(x32, sa[1], sa[0], sa[2], eps)
        if ops.profiling():   # bench.py's per-kernel timing: the same launches, issued one by one
            for i, t in enumerate(pk.tensors):
                hs_i = None if head_scale is None else head_scale[i].contiguous()
                ops.linear_ln(sa[0], t["w_qkv"], t["h_qkv"], t["g_qkv"], sa[2], eps, 1, out=ws["qkv"])
                ops.attention_fwd(ws["qkv"], B, S, nh, mask=mask_f32, mask_additive=mask_additive, head_scale=hs_i, out=ws["ctx"])
                ops.linear_ln(ws["ctx"], t["w_ao"], t["cb_ao"], t["gamma_in"], sa[2], eps, 2, out=sb[0], rs=sa[1], out_s=sb[1],
                              stats_out=sb[2])
                ops.linear_ln(sb[0], t["w_in"], t["h_in"], t["g_in"], sb[2], eps, 1, act=ACT_GELU, out=ws["mid"])
                ops.linear_ln(ws["mid"], t["w_out"], t["cb_out"], t["ln1_g"], sb[2], eps, 2, out=sa[0], rs=sb[1], out_s=sa[1],
                              stats_out=sa[2])
        else:
            ops.encoder_forward_ln(pk.table, sa, sb, ws["qkv"], ws["ctx"], ws["mid"], mask_f32, mask_additive, head_scale,
                                   B, S, H, nh, I, eps)
        # the fp32 rows are what the caller is handed: a fresh tensor per call (written once by the kernel, no copy out of
        # the workspace); the bf16 rows stay in the workspace for the pooler / heads
        out32 = torch.empty((M, H), dtype=torch.float32, device=x32.device)
        ops.ln_apply(sa[1], sa[2], pk.final_gamma, pk.final_beta, eps, out16=ws["out16"], out32=out32)
        self._final_f32, self._final_f32_fresh = out32, True
        return ws["out16"], out32

    def ln_input_buffer(self, M, device):
        """The workspace's fp32 buffer [M, H] for the embedding output (what run_ln takes as x32)."""
        return self._workspace_ln(M, device)["x32"]

    def _workspace(self, M, B, device, keep_all):
        """Activation buffers + ctypes table.  keep_all: one `out` buffer per layer
        (output_hidden_states); otherwise all layers share one scratch set and run in place."""
        key = (M, B, str(device), bool(keep_all))
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        H, I, L = self._hidden, self._inter, len(self.layer)
        mk = lambda n: torch.empty((M, n), dtype=BF16, device=device)
        shared = dict(qkv=mk(3 * H), ctx=mk(H), attn_pre=mk(H), attn_out=mk(H), mid=mk(I), out_pre=mk(H))
        if ops.F16_STREAM:   # the residual stream at fp16 precision: pre-LayerNorm sums + the LayerNorm outputs' second copies
            mkh = lambda: torch.empty((M, H), dtype=ops.F16, device=device)
            shared.update(attn_pre=mkh(), out_pre=mkh(), ln1_h=mkh(), ln2_h=mkh())
        outs = [mk(H) for _ in range(L)] if keep_all else [mk(H)] * L
        table = (_lib.LayerActs * L)()
        for i in range(L):
            for k, v in shared.items():
                setattr(table[i], k, v.data_ptr())
            table[i].out = outs[i].data_ptr()
        ws = dict(shared=shared, outs=outs, table=table)
        ops.autotune_encoder_shapes(M, H, I, training=False, device=device)  # once per token count
        if len(self._ws) > 4:
            self._ws.clear()
        self._ws[key] = ws
        return ws

    # ---- fused run --------------------------------------------------------------------
    def run(self, x_bf16, B, S, mask_f32, mask_additive, head_scale=None, history=None, seq=None):
        """x_bf16 [B*S,H] -> list of per-layer outputs (len L if output_hidden_states else 1 shared).  seq (ops.SeqLayout):
        x_bf16 holds the compacted rows [seq.rows, H] (no masked positions) and so do the outputs' first seq.rows rows."""
        if seq is not None and (history is not None or self.output_attentions or mask_f32 is not None or ops.profiling()):
            raise NotImplementedError("compacted rows are served by the fused layer loop only")
        if self._hidden != self._heads * 64:
            raise NotImplementedError("the fused encoder serves head size 64 (hidden = 64 * heads)")
        for layer in self.layer:
            _no_train_dropout(layer, layer.attention.self.dropout.p)
            _no_train_dropout(layer, layer.output.dropout.p)
        pk = self.packed()
        ws = self._workspace(B * S, B, x_bf16.device, self.output_hidden_states)
        self._last_attentions = None
        self._final_f32, self._final_f32_fresh = None, False
        if history is not None:   # encoder_history_states: layer i attends over cat([history[i], hidden], 1) (:148-155)
            probs = [] if self.output_attentions else None
            outs = self._run_unrolled(pk, ws, x_bf16, B, S, mask_f32, mask_additive, head_scale, probs, history)
            self._last_attentions = probs
            return outs
        if self.output_attentions:   # per-layer probabilities need each layer's qkv: the op-by-op launch sequence
            probs = []
            outs = self._run_unrolled(pk, ws, x_bf16, B, S, mask_f32, mask_additive, head_scale, probs)
            self._last_attentions = probs
            return outs
        if ops.profiling():  # bench.py's per-kernel timing: the same launches, issued one by one
            return self._run_unrolled(pk, ws, x_bf16, B, S, mask_f32, mask_additive, head_scale)
        L = len(self.layer)
        if seq is not None or not self.precise_final:
            ops.encoder_forward(pk.table, ws["table"], x_bf16, mask_f32, mask_additive, head_scale, B, S,
                                self._hidden, self._heads, self._inter, self._eps, seq=seq)
            if seq is None and "ln2_h" in ws["shared"]:   # the caller's hidden states from the fp16 copy of the last LayerNorm
                self._final_f32, self._final_f32_fresh = ws["shared"]["ln2_h"], False
            return ws["outs"]
        # The hidden states this call returns are handed to the caller in fp32.  Layers 0 .. L-2 run in the C loop; the
        # last layer is issued here with its two pre-LayerNorm sums kept in fp32 and its LayerNorm written twice from
        # them -- bf16 for the heads / pooler GEMMs, fp32 for the caller -- so the returned tensor is not rounded to
        # bf16 once before the last LayerNorm and once after it.
        if L > 1:
            n = L - 1
            sub = lambda arr, typ: (typ * n).from_address(ctypes.addressof(arr))
            hs = None if head_scale is None else head_scale[:n].contiguous()
            ops.encoder_forward(sub(pk.table, _lib.LayerWeights), sub(ws["table"], _lib.LayerActs), x_bf16, mask_f32,
                                mask_additive, hs, B, S, self._hidden, self._heads, self._inter, self._eps)
        cur = ws["outs"][L - 2] if L > 1 else x_bf16
        t, sh, eps = pk.tensors[L - 1], ws["shared"], self._eps
        M = B * S
        if "pre32" not in ws:
            ws["pre32"] = torch.empty((M, self._hidden), dtype=torch.float32, device=x_bf16.device)
            ws["final32"] = torch.empty((M, self._hidden), dtype=torch.float32, device=x_bf16.device)
        hs_l = None if head_scale is None else head_scale[L - 1].contiguous()
        ops.linear(cur, t["w_qkv"], t["b_qkv"], out=sh["qkv"])
        ops.attention_fwd(sh["qkv"], B, S, self._heads, mask=mask_f32, mask_additive=mask_additive, head_scale=hs_l,
                          out=sh["ctx"])
        res = sh["ln2_h"] if (L > 1 and "ln2_h" in sh) else cur   # the previous layer's output at fp16 precision
        ops.linear(sh["ctx"], t["w_ao"], t["b_ao"], residual=res, out=ws["pre32"], out_f32=True)
        ops.layernorm_rows(ws["pre32"], t["ln1_g"], t["ln1_b"], eps, out=sh["attn_out"])
        ops.linear(sh["attn_out"], t["w_in"], t["b_in"], act=ACT_GELU, out=sh["mid"])
        ops.linear(sh["mid"], t["w_out"], t["b_out"], residual=sh["attn_out"], out=ws["pre32"], out_f32=True)
        ops.layernorm_rows(ws["pre32"], t["ln2_g"], t["ln2_b"], eps, out=ws["outs"][L - 1])
        ops.layernorm_rows(ws["pre32"], t["ln2_g"], t["ln2_b"], eps, out=ws["final32"])
        self._final_f32, self._final_f32_fresh = ws["final32"], False
        return ws["outs"]

    def _run_unrolled(self, pk, ws, x, B, S, mask, mask_additive, head_scale, probs=None, history=None):
        """The launch sequence of vt_encoder_forward_bf16 issued op by op (same kernels, same buffers); with `probs` (a
        list) also each layer's attention probabilities (output_attentions)."""
        sh, nh, eps = ws["shared"], self._heads, self._eps
        cur, cur_h = x, None
        lse = torch.empty((B, nh, S), dtype=torch.float32, device=x.device) if probs is not None else None
        for i, t in enumerate(pk.tensors):
            out = ws["outs"][i]
            hs_i = None if head_scale is None else head_scale[i].contiguous()
            if history is not None:
                ctx_i, p_i = _attention_with_history(cur, history[i], t["w_qkv"], t["b_qkv"], None, B, S, nh, hs_i,
                                                     probs is not None, mask_f32=mask, mask_additive=mask_additive)
                sh["ctx"].copy_(ctx_i)
                if probs is not None:
                    probs.append(p_i)
            else:
                ops.linear(cur, t["w_qkv"], t["b_qkv"], out=sh["qkv"])
                ops.attention_fwd(sh["qkv"], B, S, nh, mask=mask, mask_additive=mask_additive, head_scale=hs_i,
                                  out=sh["ctx"], lse=lse)
                if probs is not None:
                    probs.append(ops.attention_probs(sh["qkv"], lse, B, S, nh, mask=mask, mask_additive=mask_additive,
                                                     head_scale=hs_i))
            # (with the fp16 copies of the stream -- ops.F16_STREAM -- attn_pre / out_pre are fp16 tensors and every
            # LayerNorm also writes ln1_h / ln2_h, which the residual adds read: the library learns it from the dtypes)
            ops.linear(sh["ctx"], t["w_ao"], t["b_ao"], residual=cur if cur_h is None else cur_h, out=sh["attn_pre"])
            ops.layernorm(sh["attn_pre"], t["ln1_g"], t["ln1_b"], eps, out=sh["attn_out"], out_h=sh.get("ln1_h"))
            ops.linear(sh["attn_out"], t["w_in"], t["b_in"], act=ACT_GELU, out=sh["mid"])
            ops.linear(sh["mid"], t["w_out"], t["b_out"], residual=sh.get("ln1_h", sh["attn_out"]), out=sh["out_pre"])
            ops.layernorm(sh["out_pre"], t["ln2_g"], t["ln2_b"], eps, out=out, out_h=sh.get("ln2_h"))
            cur, cur_h = out, sh.get("ln2_h")
        if "ln2_h" in sh:
            self._final_f32, self._final_f32_fresh = sh["ln2_h"], False
        return ws["outs"]

    def run_f32(self, x, B, S, mask_f32, mask_additive, head_scale=None, history=None):
        """The layer loop in fp32 (set_precision(model, "fp32")): x fp32 [B*S, H] -> list of the L layer outputs (fp32);
        per-layer attention probabilities land in self._last_attentions when output_attentions is set."""
        if self._hidden != self._heads * 64:
            raise NotImplementedError("the fp32 path serves head size 64 (hidden = 64 * heads)")
        nh, H, eps = self._heads, self._hidden, self._eps
        probs_all = [] if self.output_attentions else None
        outs, cur = [], x
        f = _f32
        for i, layer in enumerate(self.layer):
            att, so = layer.attention.self, layer.attention.output
            w_qkv = torch.cat([att.query.weight, att.key.weight, att.value.weight], 0).detach().float().contiguous()
            b_qkv = torch.cat([att.query.bias, att.key.bias, att.value.bias], 0).detach().float().contiguous()
            hs_i = None if head_scale is None else head_scale[i].contiguous()
            Sh, xs = 0, cur
            if history is not None:   # :37-41, :148-155: keys / values over cat([history_i, hidden], 1)
                Sh = history[i].shape[1]
                xs = torch.cat([history[i].detach().float(), cur.view(B, S, H)], 1).reshape(B * (Sh + S), H).contiguous()
            St = Sh + S
            qkv = ops.linear_f32(xs, w_qkv, b_qkv)
            ctx, p_i = ops.attention_f32(qkv, B, St, nh, mask=mask_f32, mask_additive=mask_additive, head_scale=hs_i,
                                         want_probs=probs_all is not None)
            if Sh:
                ctx = ctx.view(B, St, H)[:, Sh:].reshape(B * S, H).contiguous()
                if p_i is not None:
                    p_i = p_i[:, :, Sh:, :].contiguous()
            if probs_all is not None:
                probs_all.append(p_i)
            pre = ops.linear_f32(ctx, f(so.dense.weight), f(so.dense.bias), residual=cur)
            a_out = ops.layernorm_rows(pre, f(so.LayerNorm.weight), f(so.LayerNorm.bias), eps)
            mid = ops.linear_f32(a_out, f(layer.intermediate.dense.weight), f(layer.intermediate.dense.bias), act=ACT_GELU)
            pre2 = ops.linear_f32(mid, f(layer.output.dense.weight), f(layer.output.dense.bias), residual=a_out)
            cur = ops.layernorm_rows(pre2, f(layer.output.LayerNorm.weight), f(layer.output.LayerNorm.bias), eps)
            outs.append(cur)
        self._last_attentions = probs_all
        return outs

    def forward(self, hidden_states, attention_mask, head_mask=None, encoder_history_states=None):
        """oscar/modeling_bert.py:140-169.  attention_mask is the ADDITIVE extended mask [B,1,1,S]."""
        B, S, H = hidden_states.shape
        Sh = 0 if encoder_history_states is None else encoder_history_states[0].shape[1]
        mask = _additive_mask_2d(attention_mask, B, Sh + S) if attention_mask is not None else None
        hs = _head_scale(head_mask, len(self.layer), self._heads, hidden_states.device)
        if _is_fp32(self):
            ops._require_hip(hidden_states)
            outs = self.run_f32(hidden_states.detach().reshape(B * S, H).float().contiguous(), B, S, mask, True, hs,
                                history=encoder_history_states)
            if not self.output_hidden_states:
                outs = outs[-1:]
        elif self.serves_deferred_ln(history=encoder_history_states, rows=B * S):
            ops._require_hip(hidden_states)
            out16, _ = self.run_ln(hidden_states.detach().reshape(B * S, H).float().contiguous(), B, S, mask, True, hs)
            outs = [out16]
        else:
            outs = self.run(_as_bf16_2d(hidden_states), B, S, mask, True, hs, history=encoder_history_states)
        dt = hidden_states.dtype
        if not _is_fp32(self) and self._final_f32 is not None:   # (a copy when the fp32 rows live in the workspace)
            last = self._final_f32.view(B, S, H).to(dt, copy=not self._final_f32_fresh)
        else:
            last = outs[-1].view(B, S, H).to(dt)
        outputs = (last,)
        if self.output_hidden_states:
            outputs = outputs + ((hidden_states,) + tuple(o.view(B, S, H).to(dt) for o in outs[:-1]) + (last,),)
        if self.output_attentions:
            outputs = outputs + (tuple(p.to(dt) for p in self._last_attentions),)
        return outputs


# --------------------------------------------------------------------------------------------
# tasks/viewpoint_select/encoder.py:142-441
# --------------------------------------------------------------------------------------------
class NextActionPrediction(nn.Module):
    def __init__(self, hidden, actionspace):
        super().__init__()
        self.linear = nn.Linear(hidden, actionspace)
        self.softmax = nn.LogSoftmax(dim=-1)

    def forward(self, x):
        ops._require_hip(x)
        a = x.detach().reshape(-1, x.shape[-1]).to(BF16)
        K = a.shape[1]
        buf = torch.empty((a.shape[0], round_up(self.linear.weight.shape[0], 4)), dtype=torch.float32, device=x.device)
        ops.linear(a.contiguous(), _bf16(self.linear.weight), _f32(self.linear.bias), out=buf, out_f32=True)
        return self.softmax(buf[:, : self.linear.weight.shape[0]])


class BertImgModelwithLocationEmbeds(BertPreTrainedModel):
    """Expand from BertModel to handle image region features as input (encoder.py:161-303)."""

    def __init__(self, config):
        super().__init__(config)
        self.embeddings = BertEmbeddings(config)
        self.encoder = CaptionBertEncoder(config)
        self.pooler = BertPooler(config)

        self.img_dim = config.img_feature_dim
        logger.info("BertImgModel Image Dimension: {}".format(self.img_dim))
        self.img_feature_type = config.img_feature_type
        self.use_img_layernorm = config.use_img_layernorm if hasattr(config, "use_img_layernorm") else None

        self.img_embedding = nn.Linear(self.img_dim, self.config.hidden_size, bias=True)
        self.location_embeds = nn.Linear(128, self.config.hidden_size, bias=True)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        if self.use_img_layernorm:
            self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.img_layer_norm_eps)
        self.apply(self.init_weights)
        self._img_pack = None
        self._img_pack_key = None

    def resize_specific_embeddings(self, embedding_type, new_num_tokens):
        old_embeddings = getattr(self.embeddings, embedding_type)
        new_embeddings = self._get_resized_embeddings(old_embeddings, new_num_tokens)
        setattr(self.embeddings, embedding_type, new_embeddings)
        return getattr(self.embeddings, embedding_type)

    def _packed_img(self):
        """[H, Kpad] bf16 = [img_embedding.weight | location_embeds.weight | 0], bias = b_img + b_loc."""
        ps = (self.img_embedding.weight, self.img_embedding.bias, self.location_embeds.weight, self.location_embeds.bias)
        key = _param_key(ps)
        if self._img_pack is None or key != self._img_pack_key:
            D = self.img_dim
            kpad = round_up(D + 128, 64)
            H = self.config.hidden_size
            w = torch.zeros((H, kpad), dtype=BF16, device=ps[0].device)
            w[:, :D] = ps[0].detach().to(BF16)
            w[:, D : D + 128] = ps[2].detach().to(BF16)
            b = (ps[1].detach().float() + ps[3].detach().float()).contiguous()
            self._img_pack, self._img_pack_key = (w, b, kpad), key
        return self._img_pack

    def run_trunk(self, input_ids, token_type_ids=None, attention_mask=None, position_ids=None, head_mask=None,
                  img_feats=None, img_location_embeddings=None, encoder_history_states=None, keep=None):
        """Internal: returns (per-layer bf16 outputs list, pooled fp32 [B,H], embedding bf16, B, S).
        keep (bool [B,S], optional): the caller only reads the positions marked True and vouches that the others are the
        masked ones (OscarEncoder: the padding behind each instruction) -- the encoder then runs on those rows alone
        (compacted, ops.SeqLayout in self._last_layout; position 0 of every sequence must be kept) and the outputs'
        first layout.rows rows are the kept positions in order; attention_mask is not consulted."""
        ops._require_hip(input_ids)
        dev = input_ids.device
        B, T = input_ids.shape
        Sh = 0
        if encoder_history_states:
            assert img_feats is None, "Cannot take image features while using encoder history states"
            Sh = encoder_history_states[0].shape[1]
        else:
            encoder_history_states = None
        R = 0 if img_feats is None else img_feats.shape[1]
        S = T + R
        H = self.config.hidden_size

        # mask: encoder.py:215-241.  The -10000 arithmetic runs in the attention kernel on the raw fp32 mask.
        mask_f32 = None
        mask_is_additive = False
        if attention_mask is not None:
            if attention_mask.dim() == 2:
                if attention_mask.shape != (B, Sh + S):
                    raise RuntimeError(
                        "attention_mask shape %s does not match [batch, history+text+region] = [%d, %d]"
                        % (tuple(attention_mask.shape), B, Sh + S))
                if _is_fp32(self):   # (the fp32 path adds the literal bias in fp32 like the reference, rounding and all)
                    mask_f32 = attention_mask.to(device=dev, dtype=torch.float32).contiguous()
                else:
                    mask_f32 = _centered_mask(attention_mask.to(device=dev))
            elif attention_mask.dim() == 3:   # encoder.py:228-229 + :238-241: per-query mask -> additive bias [B,S,S]
                if attention_mask.shape != (B, S, S):
                    raise RuntimeError("3-D attention_mask shape %s does not match [%d, %d, %d]"
                                       % (tuple(attention_mask.shape), B, S, S))
                mask_f32 = ((1.0 - attention_mask.to(device=dev, dtype=torch.float32)) * -10000.0).contiguous()
                mask_is_additive = True
            else:
                raise NotImplementedError
        hs = _head_scale(head_mask, self.config.num_hidden_layers, self.config.num_attention_heads, dev)

        if _is_fp32(self):
            if keep is not None:
                raise NotImplementedError("compacted rows are served by the bf16 path")
            return self._run_trunk_f32(input_ids, token_type_ids, position_ids, img_feats, img_location_embeddings,
                                       encoder_history_states, mask_f32, mask_is_additive, hs, B, T, R, S, H)
        # compacted rows (the rollout's eval forward): the layout first -- its row count, not the padded one, picks the layer loop
        lay = None
        if keep is not None:
            if encoder_history_states is not None or self.encoder.output_attentions or self.encoder.output_hidden_states:
                raise NotImplementedError("compacted rows: plain forward only")
            lay = ops.SeqLayout(keep.to(device=dev, dtype=torch.bool))
        rows_eff = B * S if lay is None else lay.rows
        if (keep is None or not ops.profiling()) and self.encoder.serves_deferred_ln(history=encoder_history_states, rows=rows_eff):
            # default inference path: the embedding output in fp32, the encoder's residual stream in fp16 (not bf16)
            emb = self.embeddings
            _no_train_dropout(emb, emb.dropout.p)
            x32 = self.encoder.ln_input_buffer(B * S, dev)
            err = torch.zeros(1, dtype=torch.int32, device=dev)
            ops.embed_layernorm_f32(_i64(input_ids), _i64(token_type_ids), _i64(position_ids), _f32(emb.word_embeddings.weight),
                                    _f32(emb.position_embeddings.weight), _f32(emb.token_type_embeddings.weight),
                                    _f32(emb.LayerNorm.weight), _f32(emb.LayerNorm.bias), emb.LayerNorm.variance_epsilon, x32, S,
                                    err_flag=err)
            emb._last_err = err
            if img_feats is not None:
                _no_train_dropout(self, self.dropout.p)
                w, b, kpad = self._packed_img()
                a = ops.pack_concat(
                    img_feats.reshape(B * R, -1).float().contiguous(),
                    img_location_embeddings.reshape(B * R, -1).float().contiguous(), kpad)
                ops.linear(a, w, b, out=x32[T:], ldc=H, grp_rows=R, grp_stride=S, out_f32=True)
                if self.use_img_layernorm:
                    ops.layernorm_rows(x32[T:], _f32(self.LayerNorm.weight), _f32(self.LayerNorm.bias),
                                       self.LayerNorm.variance_epsilon, out=x32[T:], M=B * R, grp_rows=R, grp_stride=S)
            self._last_layout = None
            if keep is not None:   # compacted rows (the rollout's eval forward): the same loop on the rows that exist
                self._last_layout = lay
                out16, _ = self.encoder.run_ln(x32.index_select(0, lay.index), B, S, None, False, hs, seq=lay)
                cls = out16[:lay.rows].index_select(0, lay.start.to(torch.int64))
                pooled = self.pooler.pooled(cls, B, 1)
                _check_index_error(emb)
                return [out16], pooled, x32, B, S
            out16, _ = self.encoder.run_ln(x32, B, S, mask_f32, mask_is_additive, hs)
            pooled = self.pooler.pooled(out16, B, S)
            _check_index_error(emb)
            return [out16], pooled, x32, B, S
        x = torch.empty((B * S, H), dtype=BF16, device=dev)
        self.embeddings.write_rows(x, S, input_ids, token_type_ids, position_ids)  # rows b*S + [0,T)
        if img_feats is not None:
            _no_train_dropout(self, self.dropout.p)
            w, b, kpad = self._packed_img()
            a = ops.pack_concat(
                img_feats.reshape(B * R, -1).float().contiguous(),
                img_location_embeddings.reshape(B * R, -1).float().contiguous(), kpad)
            # rows b*S + T + r : the GEMM epilogue remaps its row m = b*R + r (replaces torch.cat, :287)
            ops.linear(a, w, b, out=x[T:], ldc=H, grp_rows=R, grp_stride=S)
            if self.use_img_layernorm:
                ops.layernorm(x[T:], _f32(self.LayerNorm.weight), _f32(self.LayerNorm.bias),
                              self.LayerNorm.variance_epsilon, out=x[T:], M=B * R, grp_rows=R, grp_stride=S)
        self._last_layout = None
        if keep is not None:
            self._last_layout = lay
            xc = x.index_select(0, lay.index)
            # the row count decides the tile quantisation: tune once per 256-row bucket (512 below 16 384 rows; nearest tuned
            # M is used) -- the engine's rule, training.py
            ops.autotune_encoder_shapes(round_up(lay.rows, 256 if lay.rows >= 16384 else 512), H, self.config.intermediate_size,
                                        training=False, device=dev)
            outs = self.encoder.run(xc, B, S, None, False, hs, seq=lay)
            cls = outs[-1][:lay.rows].index_select(0, lay.start.to(torch.int64))
            pooled = self.pooler.pooled(cls, B, 1)
            _check_index_error(self.embeddings)
            return outs, pooled, x, B, S
        outs = self.encoder.run(x, B, S, mask_f32, mask_is_additive, hs, history=encoder_history_states)
        pooled = self.pooler.pooled(outs[-1], B, S)
        _check_index_error(self.embeddings)
        return outs, pooled, x, B, S

    def _run_trunk_f32(self, input_ids, token_type_ids, position_ids, img_feats, img_location_embeddings, history,
                       mask_f32, mask_is_additive, hs, B, T, R, S, H):
        """encoder.py:204-303 with every tensor in fp32 (set_precision(model, "fp32")); same return as run_trunk."""
        dev = input_ids.device
        emb = self.embeddings
        x = torch.empty((B * S, H), dtype=torch.float32, device=dev)
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        ops.embed_layernorm_f32(_i64(input_ids), _i64(token_type_ids), _i64(position_ids), _f32(emb.word_embeddings.weight),
                                _f32(emb.position_embeddings.weight), _f32(emb.token_type_embeddings.weight),
                                _f32(emb.LayerNorm.weight), _f32(emb.LayerNorm.bias), emb.LayerNorm.variance_epsilon, x, S,
                                err_flag=err)
        emb._last_err = err
        if img_feats is not None:
            # img_embedding(img_feats) + location_embeds(loc) (:277-279) as ONE product over the K-concatenated operands,
            # written straight into rows b*S + T + r (the torch.cat of :287)
            a = torch.cat([img_feats.reshape(B * R, -1).float(), img_location_embeddings.reshape(B * R, -1).float()], 1)
            w = torch.cat([_f32(self.img_embedding.weight), _f32(self.location_embeds.weight)], 1).contiguous()
            b = _f32(self.img_embedding.bias) + _f32(self.location_embeds.bias)
            ops.linear_f32(a.contiguous(), w, b, out=x[T:], ldc=H, grp_rows=R, grp_stride=S)
            if self.use_img_layernorm:
                ops.layernorm_rows(x[T:], _f32(self.LayerNorm.weight), _f32(self.LayerNorm.bias),
                                   self.LayerNorm.variance_epsilon, out=x[T:], M=B * R, grp_rows=R, grp_stride=S)
        self._last_layout = None
        outs = self.encoder.run_f32(x, B, S, mask_f32, mask_is_additive, hs, history=history)
        pooled = ops.linear_f32(outs[-1], _f32(self.pooler.dense.weight), _f32(self.pooler.dense.bias), act=ACT_TANH,
                                M=B, lda=S * H)
        _check_index_error(emb)
        if not self.encoder.output_hidden_states:
            outs = outs[-1:]
        return outs, pooled, x, B, S

    def forward(self, input_ids, token_type_ids=None, attention_mask=None, position_ids=None, head_mask=None,
                img_feats=None, img_location_embeddings=None, encoder_history_states=None):
        _refuse_data_parallel_replica(self)
        _poll_index_flags()   # an earlier call's out-of-range flag that has arrived since
        if self.training and not _is_fp32(self) and (
                (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()))
                or self.config.hidden_dropout_prob > 0.0 or self.config.attention_probs_dropout_prob > 0.0):
            # a caller that trains THROUGH the trunk (the rollout's OscarEncoder, agent.py:493-518): one autograd node
            # backed by the engine's forward / backward kernels.  Also train() under torch.no_grad() with dropout on
            # (agent.py:476-489, test(use_dropout=True)): the same forward with its dropout, no graph.
            if encoder_history_states:
                raise NotImplementedError("trunk-level training serves the forward without encoder_history_states "
                                          "(output_hidden_states and output_attentions are served)")
            ops._require_hip(input_ids)
            from .training import autograd_trunk_forward

            batch = dict(input_ids=input_ids, token_type_ids=token_type_ids, attention_mask=attention_mask,
                         position_ids=position_ids)
            if attention_mask is not None and attention_mask.dim() not in (2, 3):
                raise NotImplementedError
            if img_feats is not None:
                batch.update(img_feats=img_feats, img_location_embeddings=img_location_embeddings)
            dt = next(self.parameters()).dtype
            out = autograd_trunk_forward(self, batch, head_mask, want_hidden=self.encoder.output_hidden_states,
                                         want_attn=self.encoder.output_attentions)
            res = (out[0].to(dt), out[1].to(dt))
            for extra in out[2:]:   # (sequence_output, pooled_output) + encoder_outputs[1:] = hidden states, then attentions (encoder.py:300-303)
                res = res + (tuple(h.to(dt) for h in extra),)
            return res
        outs, pooled, x, B, S = self.run_trunk(input_ids, token_type_ids, attention_mask, position_ids, head_mask,
                                               img_feats, img_location_embeddings, encoder_history_states)
        dt = next(self.parameters()).dtype
        H = self.config.hidden_size
        f32 = None if _is_fp32(self) else self.encoder._final_f32
        # (fp32 rows that live in the encoder's workspace are rewritten by the next call: hand out a copy of those)
        sequence_output = (outs[-1].view(B, S, H).to(dt) if f32 is None
                           else f32.view(B, S, H).to(dt, copy=not self.encoder._final_f32_fresh))
        pooled = pooled.to(dt)
        if torch.is_grad_enabled() and not self.training and not _is_fp32(self) and not encoder_history_states \
                and any(p.requires_grad for p in self.parameters()):
            # eval() with grad enabled: the values above stand; a backward through them, if it ever comes, recomputes the
            # forward in the engine (no dropout in eval mode: the same function) and runs its trunk backward
            from .training import lazy_autograd_trunk

            batch = dict(input_ids=input_ids, token_type_ids=token_type_ids, attention_mask=attention_mask,
                         position_ids=position_ids)
            if img_feats is not None:
                batch.update(img_feats=img_feats, img_location_embeddings=img_location_embeddings)
            sequence_output, pooled = lazy_autograd_trunk(self, batch, head_mask, sequence_output, pooled)
        outputs = (sequence_output, pooled)
        if self.encoder.output_hidden_states:
            hidden = (x.view(B, S, H).to(dt),) + tuple(o.view(B, S, H).to(dt) for o in outs[:-1]) + (sequence_output,)
            outputs = outputs + (hidden,)
        if self.encoder.output_attentions:
            outputs = outputs + (tuple(p.to(dt) for p in self.encoder._last_attentions),)
        return outputs


class PreTrainOscar(BertPreTrainedModel):
    """Trunk + MLM / masked-region-token / 1-in-36 action heads (encoder.py:306-441)."""

    def __init__(self, config):
        super().__init__(config)
        self.config = config
        self.bert = BertImgModelwithLocationEmbeds(config)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.next_action = NextActionPrediction(self.config.hidden_size, self.config.action_space)
        self.criterion = nn.CrossEntropyLoss(ignore_index=-1)
        self.mlmhead = BertOnlyMLMHead(self.config)
        self.token_head = nn.Sequential(
            nn.Linear(self.config.hidden_size, self.config.detector_classes),
            nn.Softmax(dim=-1),
        )
        self.apply(self.init_weights)
        self.tie_weights()

    def tie_weights(self):
        self._tie_or_clone_weights(self.mlmhead.predictions.decoder, self.bert.embeddings.word_embeddings)

    def resize_embeddings(self, embedding_size_dict):
        for embedding_type, new_embedding_size in embedding_size_dict.items():
            assert embedding_type in ["word_embeddings", "position_embeddings", "token_type_embeddings"]
            self.bert.resize_specific_embeddings(embedding_type, new_embedding_size)
            logger.info(f"Resized {embedding_type} to {new_embedding_size}")

    def head_outputs(self, seq_bf16, pooled_f32):
        """(prediction_scores [M,V] fp32, token probabilities [M,C] fp32, action log-probs [B,A] fp32)."""
        if _is_fp32(self):
            return self._head_outputs_f32(seq_bf16, pooled_f32)
        scores = self.mlmhead.scores(seq_bf16)
        lin = self.token_head[0]
        C = lin.weight.shape[0]
        buf = torch.empty((seq_bf16.shape[0], round_up(C, 4)), dtype=torch.float32, device=seq_bf16.device)
        ops.linear(seq_bf16, _bf16(lin.weight), _f32(lin.bias), out=buf, out_f32=True)
        token_prob = ops.softmax_rows_f32(buf[:, :C])   # token_head's nn.Softmax (encoder.py:323-326), in place
        action = self.next_action(pooled_f32)
        return scores, token_prob, action

    LOSS_ROWS_PER_CHUNK = 8192   # labelled rows per decoder GEMM + fused CE launch (1 GB of fp32 logits at a time)

    def _labelled_rows(self, labels, token_labels):
        """The rows that carry an MLM label / a region-token label, located with ONE host synchronisation for both heads
        (the counts and the out-of-range checks travel together; the row indices come from a stable argsort cut at the
        count, which needs no size from the device): -> ((y_all, idx, n) for the MLM head, the same for the token head)."""
        V, C = self.mlmhead.predictions.decoder.weight.shape[0], self.token_head[0].weight.shape[0]
        yw, yt = labels.reshape(-1), token_labels.reshape(-1)
        have_w, have_t = yw != -1, yt != -1
        info = torch.stack([have_w.sum(), have_t.sum(), ((yw >= V) | (yw < -1)).any().to(torch.int64),
                            ((yt >= C) | (yt < -1)).any().to(torch.int64)]).tolist()
        for bad, ncls in ((info[2], V), (info[3], C)):
            if bad:   # CrossEntropyLoss raises on a target outside [0, classes) other than ignore_index
                raise IndexError("Target out of bounds for a criterion with %d classes (ignore_index = -1)" % ncls)
        rows = lambda have, n: torch.argsort((~have).to(torch.int8), stable=True)[:n]
        return (yw, rows(have_w, info[0]), int(info[0])), (yt, rows(have_t, info[1]), int(info[1]))

    def _mlm_loss_on_labelled_rows(self, seq_bf16, located):
        """(mean CE over the positions with a label, their argmax accuracy) of the MLM head -- encoder.py:377, 387-389,
        402-413 -- from the labelled rows only, chunked: transform GEMM (+GELU), LayerNorm, decoder GEMM, fused
        log-sum-exp / argmax (vt_ce_softmax_rows without its gradient output).  No labelled row: NaN, as the criterion."""
        y_all, idx, n = located
        dev = seq_bf16.device
        if n == 0:
            nan = torch.full((), float("nan"), dtype=torch.float32, device=dev)
            return nan, nan.clone()
        p = self.mlmhead.predictions
        V = p.decoder.weight.shape[0]
        w_tr, w_dec = _bf16(p.transform.dense.weight), _bf16(p.decoder.weight)
        loss_sum = torch.zeros((), dtype=torch.float32, device=dev)
        hits = torch.zeros((), dtype=torch.int64, device=dev)
        for s0 in range(0, n, self.LOSS_ROWS_PER_CHUNK):
            rows = idx[s0:s0 + self.LOSS_ROWS_PER_CHUNK]
            y = y_all.index_select(0, rows)
            t = ops.linear(seq_bf16.index_select(0, rows), w_tr, _f32(p.transform.dense.bias), act=ACT_GELU)
            t = ops.layernorm(t, _f32(p.transform.LayerNorm.weight), _f32(p.transform.LayerNorm.bias),
                              p.transform.LayerNorm.variance_epsilon, out=t)
            z = torch.empty((rows.numel(), round_up(V, 4)), dtype=torch.float32, device=dev)
            ops.linear(t, w_dec, _f32(p.bias), out=z, out_f32=True)
            loss_rows, amax = ops.ce_softmax_rows(z, y, V, None, 1.0)
            loss_sum += loss_rows.sum()
            hits += (amax == y).sum()
        return loss_sum / n, hits.float() / n

    def _token_loss_on_labelled_rows(self, seq_bf16, located):
        """The masked-region-token head (encoder.py:323-326, 380-385, 423-431): Linear + Softmax, then CrossEntropy on
        the probabilities -- loss and argmax of the labelled rows from vt_ce_double_softmax_rows."""
        y_all, idx, n = located
        dev = seq_bf16.device
        if n == 0:
            nan = torch.full((), float("nan"), dtype=torch.float32, device=dev)
            return nan, nan.clone()
        lin = self.token_head[0]
        C = lin.weight.shape[0]
        y = y_all.index_select(0, idx)
        z = torch.empty((n, round_up(C, 4)), dtype=torch.float32, device=dev)
        ops.linear(seq_bf16.index_select(0, idx), _bf16(lin.weight), _f32(lin.bias), out=z, out_f32=True)
        loss_rows, amax = ops.ce_double_softmax_rows(z, y, C, None, 1.0)
        return loss_rows.mean(), (amax == y).sum().float() / n

    def _head_outputs_f32(self, seq, pooled):
        """The three heads in fp32 (encoder.py:377-391): seq fp32 [M, H], pooled fp32 [B, H]."""
        p = self.mlmhead.predictions
        t = ops.linear_f32(seq, _f32(p.transform.dense.weight), _f32(p.transform.dense.bias), act=ACT_GELU)
        t = ops.layernorm_rows(t, _f32(p.transform.LayerNorm.weight), _f32(p.transform.LayerNorm.bias),
                               p.transform.LayerNorm.variance_epsilon)
        scores = ops.linear_f32(t, _f32(p.decoder.weight), _f32(p.bias))
        lin = self.token_head[0]
        token_prob = ops.softmax_rows_f32(ops.linear_f32(seq, _f32(lin.weight), _f32(lin.bias)))
        act = ops.linear_f32(pooled, _f32(self.next_action.linear.weight), _f32(self.next_action.linear.bias))
        return scores, token_prob, torch.log_softmax(act, dim=-1)

    def forward(self, input_ids, token_type_ids=None, attention_mask=None, labels=None, token_labels=None,
                position_ids=None, head_mask=None, img_feats=None, img_location_embeddings=None, next_action=None,
                text_only=False):
        _refuse_data_parallel_replica(self)
        if text_only:
            return self.bert(input_ids, position_ids=position_ids, token_type_ids=token_type_ids,
                             attention_mask=attention_mask, head_mask=head_mask, img_feats=img_feats,
                             img_location_embeddings=img_location_embeddings)
        batch = dict(input_ids=input_ids, token_type_ids=token_type_ids, attention_mask=attention_mask, labels=labels,
                     token_labels=token_labels, position_ids=position_ids, img_feats=img_feats,
                     img_location_embeddings=img_location_embeddings, next_action=next_action)
        batch = {k: v for k, v in batch.items() if v is not None}
        wants_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if wants_grad and _is_fp32(self):
            raise NotImplementedError('precision "fp32" serves inference (torch.no_grad()); training runs on the bf16 kernels')
        if wants_grad and self.training:
            # training: HIP forward + backward, bridged to autograd so that `loss.backward()` and any torch
            # optimizer / DistributedDataParallel wrapper work as in the reference's loop
            if token_labels is None:
                raise NameError("token_prediction")  # the reference leaves it unbound (encoder.py:400)
            from .training import autograd_forward

            return autograd_forward(self, batch, head_mask=head_mask)
        if self.training and not _is_fp32(self) and (self.config.hidden_dropout_prob > 0.0
                                                     or self.config.attention_probs_dropout_prob > 0.0):
            # train() under torch.no_grad() (or with frozen parameters): the engine's forward with its dropout, no backward
            if token_labels is None:
                raise NameError("token_prediction")
            from .training import _bridge_engine

            return _bridge_engine(self).forward_backward(batch, head_mask=head_mask, backward=False)
        outs, pooled, _, B, S = self.bert.run_trunk(
            input_ids, token_type_ids, attention_mask, position_ids, head_mask, img_feats, img_location_embeddings)
        if token_labels is None:
            raise NameError("token_prediction")  # the reference leaves it unbound (encoder.py:400)
        if _is_fp32(self):
            # fp32 parity path: every row, exactly the reference's sequence of ops (encoder.py:377-431)
            prediction_scores, token_prob, action_scores = self.head_outputs(outs[-1], pooled)
            token_loss = self.criterion(token_prob, token_labels.reshape(-1))
            mask_loss = self.criterion(prediction_scores, labels.reshape(-1))
            words_accuracy = _supervised_accuracy(prediction_scores, labels)
            token_accuracy = _supervised_accuracy(token_prob, token_labels)
        else:
            # The 7-tuple reads the 30522-wide MLM logits and the 1601-wide token probabilities only at the positions
            # that carry a label: every criterion ignores -1 (encoder.py:321) and the accuracies drop those positions
            # (:402-431).  So the two wide heads run on the labelled rows alone and their logits go straight into the
            # fused loss + argmax kernels -- [B, S, 30522] fp32 (7 GB at B = 256) is never written.
            loc_w, loc_t = self._labelled_rows(labels, token_labels)      # the call's one host synchronisation
            mask_loss, words_accuracy = self._mlm_loss_on_labelled_rows(outs[-1], loc_w)
            token_loss, token_accuracy = self._token_loss_on_labelled_rows(outs[-1], loc_t)
            action_scores = self.next_action(pooled)
        next_loss = self.criterion(action_scores, next_action) if next_action is not None else 0
        loss = mask_loss + next_loss + token_loss
        action_accuracy = 0
        if next_action is not None:   # divides by the whole batch, ignored (-1) actions included (encoder.py:418-421)
            action_accuracy = (action_scores.argmax(1) == next_action).sum().float() / action_scores.shape[0]
        if wants_grad:
            # eval mode with grad enabled (the reference's val() never wraps its loop in no_grad): the values above stand;
            # the backward pass is prepared only if someone calls it
            from .training import lazy_autograd_loss

            loss = lazy_autograd_loss(self, batch, loss, head_mask=head_mask)
        return (loss, mask_loss, next_loss, token_loss, words_accuracy, action_accuracy, token_accuracy)


def _supervised_accuracy(scores, labels):
    """Fraction of the positions with a label (!= -1) whose argmax over the class axis equals it; 0-d fp32 tensor.

    The reference reaches the same number by overwriting the prediction with -1 wherever the label is -1, counting ALL
    equal positions and subtracting the ignored count again (encoder.py:402-431); with no labelled position both forms
    are 0/0 = NaN."""
    y = labels.reshape(-1)
    have = y != -1
    hits = ((scores.reshape(y.numel(), -1).argmax(1) == y) & have).sum()
    return hits.float() / have.sum().float()


# tasks/viewpoint_select/model_utils.py:15-26 -- the registry the reference's loader indexes by name.
# (ImageBertForSequenceClassificationwithAction is dead code in the reference: it names undefined classes.)
MODEL_CLASS = {
    "PreTrainOscar": (BertConfig, PreTrainOscar, None),
    "BertImgModelwithLocationEmbeds": (BertConfig, BertImgModelwithLocationEmbeds, None),
}
