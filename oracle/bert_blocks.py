"""BERT building blocks -- CPU oracle (TEST INFRASTRUCTURE, see oracle/__init__.py).

UPSTREAM, NOT IN MOUNT.  The reference imports these classes from
``transformers.pytorch_transformers.modeling_bert`` (un-vendored submodule,
``/root/reference/.gitmodules:1-3``; huggingface/transformers in its
pytorch-transformers 1.x layout, commit not recoverable).  What follows is a
restatement of that package's published BERT semantics, written against the
reference's call sites:

  oscar/modeling_bert.py:31-32,47-49   BertSelfAttention.__init__/transpose_for_scores
  oscar/modeling_bert.py:90,94         BertSelfOutput
  oscar/modeling_bert.py:109,119       BertIntermediate
  oscar/modeling_bert.py:110,120       BertOutput
  tasks/viewpoint_select/encoder.py:166,267   BertEmbeddings
  tasks/viewpoint_select/encoder.py:168,296   BertPooler
  tasks/viewpoint_select/encoder.py:183       BertLayerNorm
  tasks/viewpoint_select/encoder.py:322,377   BertOnlyMLMHead
  tasks/viewpoint_select/encoder.py:187,192,333  init_weights/_get_resized_embeddings/_tie_or_clone_weights

Checked block by block against the independent implementation installed in
this image by ``oracle/crosscheck_hf.py``.
"""
import math

import torch
from torch import nn


def gelu(x):
    """erf-GELU (hidden_act == "gelu"): 0.5 x (1 + erf(x / sqrt 2)).  Not the tanh form."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


ACT2FN = {"gelu": gelu, "relu": torch.relu}


class BertLayerNorm(nn.Module):
    """(x - mean) / sqrt(biased_var + eps) * weight + bias, eps inside the sqrt."""

    def __init__(self, hidden_size, eps=1e-12):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.bias = nn.Parameter(torch.zeros(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        u = x.mean(-1, keepdim=True)
        s = (x - u).pow(2).mean(-1, keepdim=True)
        x = (x - u) / torch.sqrt(s + self.variance_epsilon)
        return self.weight * x + self.bias


class BertEmbeddings(nn.Module):
    """LN(word + position + token_type) then dropout.

    ``forward(input_ids, token_type_ids=None, position_ids=None)`` -- note the
    positional order; the reference calls it with keywords (encoder.py:267-269).
    ``word_embeddings`` is built with ``padding_idx=0`` upstream: the lookup's
    backward leaves row 0 without gradient (the row itself is re-drawn by
    ``init_weights``, it is not zero).
    """

    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, input_ids, token_type_ids=None, position_ids=None):
        seq_length = input_ids.size(1)
        if position_ids is None:
            position_ids = torch.arange(seq_length, dtype=torch.long, device=input_ids.device)
            position_ids = position_ids.unsqueeze(0).expand_as(input_ids)
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        e = (
            self.word_embeddings(input_ids)
            + self.position_embeddings(position_ids)
            + self.token_type_embeddings(token_type_ids)
        )
        return self.dropout(self.LayerNorm(e))


class BertSelfAttention(nn.Module):
    """Constructor half only; the reference overrides forward (oscar/modeling_bert.py:34-79)."""

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

    def transpose_for_scores(self, x):
        # [B, S, nh*dh] -> [B, nh, S, dh]
        shape = x.size()[:-1] + (self.num_attention_heads, self.attention_head_size)
        return x.view(*shape).permute(0, 2, 1, 3)


class BertSelfOutput(nn.Module):
    """LN(dropout(dense(h)) + residual)."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        h = self.dropout(self.dense(hidden_states))
        return self.LayerNorm(h + input_tensor)


class BertIntermediate(nn.Module):
    """act(dense(h)), hidden -> intermediate."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)
        act = config.hidden_act
        self.intermediate_act_fn = ACT2FN[act] if isinstance(act, str) else act

    def forward(self, hidden_states):
        return self.intermediate_act_fn(self.dense(hidden_states))


class BertOutput(nn.Module):
    """LN(dropout(dense(h)) + residual), intermediate -> hidden."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        h = self.dropout(self.dense(hidden_states))
        return self.LayerNorm(h + input_tensor)


class BertPooler(nn.Module):
    """tanh(dense(h[:, 0]))."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.activation = nn.Tanh()

    def forward(self, hidden_states):
        return self.activation(self.dense(hidden_states[:, 0]))


class BertPredictionHeadTransform(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        act = config.hidden_act
        self.transform_act_fn = ACT2FN[act] if isinstance(act, str) else act
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)

    def forward(self, hidden_states):
        return self.LayerNorm(self.transform_act_fn(self.dense(hidden_states)))


class BertLMPredictionHead(nn.Module):
    """decoder(transform(h)) + bias; decoder has no bias of its own and is tied
    to the word embeddings by the owner's ``tie_weights``."""

    def __init__(self, config):
        super().__init__()
        self.transform = BertPredictionHeadTransform(config)
        self.decoder = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(config.vocab_size))

    def forward(self, hidden_states):
        return self.decoder(self.transform(hidden_states)) + self.bias


class BertOnlyMLMHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.predictions = BertLMPredictionHead(config)

    def forward(self, sequence_output):
        return self.predictions(sequence_output)


class BertPreTrainedModel(nn.Module):
    """The four base-class services the reference uses."""

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
