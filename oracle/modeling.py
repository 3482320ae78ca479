"""Reference hot path restated on CPU -- oracle (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows, op for op and in the same order:
  oscar/modeling_bert.py:26-79     CaptionBertSelfAttention
  oscar/modeling_bert.py:82-98     CaptionBertAttention
  oscar/modeling_bert.py:101-124   CaptionBertLayer
  oscar/modeling_bert.py:127-169   CaptionBertEncoder
  tasks/viewpoint_select/encoder.py:142-158   NextActionPrediction
  tasks/viewpoint_select/encoder.py:161-303   BertImgModelwithLocationEmbeds
  tasks/viewpoint_select/encoder.py:306-441   PreTrainOscar

Module/attribute names equal the reference's so ``state_dict()`` keys are the
checkpoint keys.  Works in any float dtype; tests use
fp32 (the reference's precision) and fp64 (gradcheck / tighter pin).
"""
import math

import torch
from torch import nn

from .bert_blocks import (
    BertEmbeddings,
    BertIntermediate,
    BertLayerNorm,
    BertOnlyMLMHead,
    BertOutput,
    BertPooler,
    BertPreTrainedModel,
    BertSelfAttention,
    BertSelfOutput,
)


class CaptionBertSelfAttention(BertSelfAttention):
    # oscar/modeling_bert.py:34-79
    def forward(self, hidden_states, attention_mask, head_mask=None, history_state=None):
        kv_source = hidden_states
        if history_state is not None:  # :37-41  K,V see [history ; current]
            kv_source = torch.cat([history_state, hidden_states], dim=1)
        q = self.transpose_for_scores(self.query(hidden_states))  # :43,47
        k = self.transpose_for_scores(self.key(kv_source))  # :44,48
        v = self.transpose_for_scores(self.value(kv_source))  # :45,49

        scores = torch.matmul(q, k.transpose(-1, -2))  # :52
        scores = scores / math.sqrt(self.attention_head_size)  # :53  scale AFTER the product
        scores = scores + attention_mask  # :55  additive mask
        probs = nn.Softmax(dim=-1)(scores)  # :58
        probs = self.dropout(probs)  # :62
        if head_mask is not None:  # :65-66
            probs = probs * head_mask
        ctx = torch.matmul(probs, v)  # :68
        ctx = ctx.permute(0, 2, 1, 3).contiguous()  # :70
        ctx = ctx.view(*(ctx.size()[:-2] + (self.all_head_size,)))  # :71-72
        return (ctx, probs) if self.output_attentions else (ctx,)  # :74-79


class CaptionBertAttention(nn.Module):
    # oscar/modeling_bert.py:82-98
    def __init__(self, config):
        super().__init__()
        self.self = CaptionBertSelfAttention(config)
        self.output = BertSelfOutput(config)

    def forward(self, input_tensor, attention_mask, head_mask=None, history_state=None):
        self_out = self.self(input_tensor, attention_mask, head_mask, history_state)
        attention_output = self.output(self_out[0], input_tensor)
        return (attention_output,) + self_out[1:]


class CaptionBertLayer(nn.Module):
    # oscar/modeling_bert.py:101-124
    def __init__(self, config):
        super().__init__()
        self.attention = CaptionBertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)

    def forward(self, hidden_states, attention_mask, head_mask=None, history_state=None):
        att = self.attention(hidden_states, attention_mask, head_mask, history_state)
        attention_output = att[0]
        layer_output = self.output(self.intermediate(attention_output), attention_output)
        return (layer_output,) + att[1:]


class CaptionBertEncoder(nn.Module):
    # oscar/modeling_bert.py:127-169
    def __init__(self, config):
        super().__init__()
        self.output_attentions = config.output_attentions
        self.output_hidden_states = config.output_hidden_states
        self.layer = nn.ModuleList(CaptionBertLayer(config) for _ in range(config.num_hidden_layers))

    def forward(self, hidden_states, attention_mask, head_mask=None, encoder_history_states=None):
        all_hidden, all_att = (), ()
        for i, layer in enumerate(self.layer):
            if self.output_hidden_states:
                all_hidden = all_hidden + (hidden_states,)
            hist = None if encoder_history_states is None else encoder_history_states[i]
            # :153 indexes head_mask[i] unconditionally -> callers pass a list of length L
            out = layer(hidden_states, attention_mask, head_mask[i], hist)
            hidden_states = out[0]
            if self.output_attentions:
                all_att = all_att + (out[1],)
        if self.output_hidden_states:
            all_hidden = all_hidden + (hidden_states,)
        outputs = (hidden_states,)
        if self.output_hidden_states:
            outputs = outputs + (all_hidden,)
        if self.output_attentions:
            outputs = outputs + (all_att,)
        return outputs


class NextActionPrediction(nn.Module):
    # tasks/viewpoint_select/encoder.py:142-158  Linear(hidden -> action_space) + LogSoftmax
    def __init__(self, hidden, actionspace):
        super().__init__()
        self.linear = nn.Linear(hidden, actionspace)
        self.softmax = nn.LogSoftmax(dim=-1)

    def forward(self, x):
        return self.softmax(self.linear(x))


class BertImgModelwithLocationEmbeds(BertPreTrainedModel):
    # tasks/viewpoint_select/encoder.py:161-303
    def __init__(self, config):
        super().__init__(config)
        self.embeddings = BertEmbeddings(config)
        self.encoder = CaptionBertEncoder(config)
        self.pooler = BertPooler(config)
        self.img_dim = config.img_feature_dim
        self.img_feature_type = config.img_feature_type  # read unconditionally (:172)
        self.use_img_layernorm = getattr(config, "use_img_layernorm", None)  # :173-176
        self.img_embedding = nn.Linear(self.img_dim, config.hidden_size, bias=True)
        self.location_embeds = nn.Linear(128, config.hidden_size, bias=True)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        if self.use_img_layernorm:
            self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.img_layer_norm_eps)
        self.apply(self.init_weights)

    def resize_specific_embeddings(self, embedding_type, new_num_tokens):  # :189-194
        old = getattr(self.embeddings, embedding_type)
        setattr(self.embeddings, embedding_type, self._get_resized_embeddings(old, new_num_tokens))
        return getattr(self.embeddings, embedding_type)

    def forward(
        self,
        input_ids,
        token_type_ids=None,
        attention_mask=None,
        position_ids=None,
        head_mask=None,
        img_feats=None,
        img_location_embeddings=None,
        encoder_history_states=None,
    ):
        if attention_mask is None:  # :215-216
            attention_mask = torch.ones_like(input_ids)
        if token_type_ids is None:  # :218-219
            token_type_ids = torch.zeros_like(input_ids)

        if attention_mask.dim() == 2:  # :226-231
            ext = attention_mask.unsqueeze(1).unsqueeze(2)
        elif attention_mask.dim() == 3:
            ext = attention_mask.unsqueeze(1)
        else:
            raise NotImplementedError
        pdtype = next(self.parameters()).dtype
        ext = ext.to(dtype=pdtype)  # :238-240
        ext = (1.0 - ext) * -10000.0  # :241  literal arithmetic, no 0/1 assumption

        L = self.config.num_hidden_layers
        if head_mask is not None:  # :248-263
            if head_mask.dim() == 1:
                head_mask = head_mask.unsqueeze(0).unsqueeze(0).unsqueeze(-1).unsqueeze(-1)
                head_mask = head_mask.expand(L, -1, -1, -1, -1)
            elif head_mask.dim() == 2:
                head_mask = head_mask.unsqueeze(1).unsqueeze(-1).unsqueeze(-1)
            head_mask = head_mask.to(dtype=pdtype)
        else:
            head_mask = [None] * L  # :265

        emb = self.embeddings(input_ids, position_ids=position_ids, token_type_ids=token_type_ids)

        if encoder_history_states:  # :271-274
            assert img_feats is None, "Cannot take image features while using encoder history states"

        if img_feats is not None:  # :276-287
            img = self.img_embedding(img_feats) + self.location_embeds(img_location_embeddings)
            if self.use_img_layernorm:
                img = self.LayerNorm(img)
            img = self.dropout(img)
            emb = torch.cat((emb, img), 1)

        enc = self.encoder(emb, ext, head_mask=head_mask, encoder_history_states=encoder_history_states)
        sequence_output = enc[0]
        pooled_output = self.pooler(sequence_output)  # :296
        return (sequence_output, pooled_output) + enc[1:]  # :299-303


class PreTrainOscar(BertPreTrainedModel):
    # tasks/viewpoint_select/encoder.py:306-441
    def __init__(self, config):
        super().__init__(config)
        self.bert = BertImgModelwithLocationEmbeds(config)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.next_action = NextActionPrediction(config.hidden_size, config.action_space)
        self.criterion = nn.CrossEntropyLoss(ignore_index=-1)
        self.mlmhead = BertOnlyMLMHead(config)
        self.token_head = nn.Sequential(
            nn.Linear(config.hidden_size, config.detector_classes),
            nn.Softmax(dim=-1),
        )
        self.apply(self.init_weights)
        self.tie_weights()

    def tie_weights(self):  # :332-335
        self._tie_or_clone_weights(self.mlmhead.predictions.decoder, self.bert.embeddings.word_embeddings)

    def resize_embeddings(self, embedding_size_dict):  # :337-345 (does NOT re-tie the decoder)
        for kind, size in embedding_size_dict.items():
            assert kind in ("word_embeddings", "position_embeddings", "token_type_embeddings")
            self.bert.resize_specific_embeddings(kind, size)

    def heads(self, sequence_output, pooled_output):
        """Not a reference method: exposes the three head outputs the 7-tuple is built
        from (prediction_scores :377, token probabilities :381, action log-probs :391)
        so parity tests can compare logits, as BASELINE.json asks."""
        return (
            self.mlmhead(sequence_output),
            self.token_head(sequence_output),
            self.next_action(pooled_output),
        )

    def forward(
        self,
        input_ids,
        token_type_ids=None,
        attention_mask=None,
        labels=None,
        token_labels=None,
        position_ids=None,
        head_mask=None,
        img_feats=None,
        img_location_embeddings=None,
        next_action=None,
        text_only=False,
    ):
        outputs = self.bert(
            input_ids,
            position_ids=position_ids,
            token_type_ids=token_type_ids,
            attention_mask=attention_mask,
            head_mask=head_mask,
            img_feats=img_feats,
            img_location_embeddings=img_location_embeddings,
        )
        if text_only:  # :371-372
            return outputs
        cls_part, lang_part = outputs[1], outputs[0]
        C = self.config

        prediction_scores = self.mlmhead(lang_part)  # :377  all S positions, regions included

        token_loss = 0
        if token_labels is not None:  # :379-385  CE over already-softmaxed probabilities
            token_prediction = self.token_head(lang_part)
            token_loss = self.criterion(token_prediction.view(-1, C.detector_classes), token_labels.view(-1))
        # token_labels=None leaves token_prediction unbound -> NameError below, as in the reference (:400)

        mask_loss = self.criterion(prediction_scores.view(-1, C.vocab_size), labels.view(-1))  # :387-389
        action_scores = self.next_action(cls_part)  # :391
        next_loss = 0
        if next_action is not None:  # :393-395
            next_loss = self.criterion(action_scores, next_action)
        loss = mask_loss + next_loss + token_loss  # :396

        predicted_action = torch.argmax(action_scores, dim=1)  # :398-400
        predicted_words = torch.argmax(prediction_scores, dim=2)
        token_prediction = torch.argmax(token_prediction, dim=2)

        predicted_words[labels == -1] = -1  # :402
        ignored_words_no = torch.sum(labels == -1)
        words_left = ((labels.shape[0] * labels.shape[1]) - ignored_words_no).type(torch.float)
        words_accuracy = (torch.sum(predicted_words == labels) - ignored_words_no) / words_left  # :408-410

        if next_action is not None:  # :412-418
            action_accuracy = torch.sum(predicted_action == next_action).type(torch.float) / predicted_action.shape[0]
        else:
            action_accuracy = 0

        token_prediction[token_labels == -1] = -1  # :420
        ignored_tokens_no = torch.sum(token_labels == -1)
        tokens_left = ((token_prediction.shape[0] * token_prediction.shape[1]) - ignored_tokens_no).type(torch.float)
        token_accuracy = (
            torch.sum(token_prediction == token_labels).type(torch.float) - ignored_tokens_no
        ) / tokens_left  # :429-431

        return (loss, mask_loss, next_loss, token_loss, words_accuracy, action_accuracy, token_accuracy)
