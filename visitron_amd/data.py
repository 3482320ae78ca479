"""Pretrain input preparation on the device: what PretrainDataset does per item in Python
before the hot path (tasks/viewpoint_select/data_loader_pretrain.py:25-49, 549-613, 615-712), batched as tensor ops so
it keeps up with the encoder.  Storage (LMDB features, JSON dialogs, tokenizer) stays out of scope: the callers hand
over token ids, per-region view ids and features; this module does the masking, the location-embedding lookup, the
region padding / truncation and the label / attention-mask assembly.

Random draws are taken from a torch.Generator (or passed in, which is how the tests pin this against
oracle/data.py): the reference's stream of torch.bernoulli / torch.randint calls per item is not reproduced, the
distribution is.

HIP tensors go through the library's two kernels (csrc/datapipe.hip: vt_mask_tokens, vt_assemble_regions -- the whole
batch in two launches, outputs bit-exact against oracle/data.py); host tensors (dataset workers, the CPU tests) take
the same steps as torch tensor ops."""
import ctypes
import math

import torch

_ANGLE_INC = math.pi / 6.0
_tables = {}


def loc_embedding_table(device="cpu"):
    """[36 agent headings, 36 absolute views, 128]: _static_loc_embeddings (data_loader_pretrain.py:25-49)."""
    key = str(device)
    if key not in _tables:
        head = torch.arange(36)[:, None]
        a = torch.arange(36)[None, :]
        rel = (a - head) % 12 + torch.div(a, 12, rounding_mode="floor") * 12
        heading = (rel % 12).to(torch.float64) * _ANGLE_INC
        elevation = (torch.div(rel, 12, rounding_mode="floor") - 1).to(torch.float64) * _ANGLE_INC
        t = torch.empty(36, 36, 128, dtype=torch.float64)
        t[..., 0:32] = torch.sin(heading)[..., None]
        t[..., 32:64] = torch.cos(heading)[..., None]
        t[..., 64:96] = torch.sin(elevation)[..., None]
        t[..., 96:128] = torch.cos(elevation)[..., None]
        _tables[key] = t.to(torch.float32).to(device)
    return _tables[key]


def region_location_embeddings(current_view_index, region_view_ids):
    """current_view_index int64 [B], region_view_ids int64 [B, R] (absolute view 0..35 of every region row) ->
    fp32 [B, R, 128]: the lookup of _extract_img_features (data_loader_pretrain.py:627-633)."""
    table = loc_embedding_table(region_view_ids.device)
    return table[current_view_index[:, None], region_view_ids]


def mask_tokens(inputs, special_mask, pad_id, mask_id, vocab_size, mlm_probability=0.15, token_classes=None,
                generator=None, draws=None):
    """PretrainDataset._mask_tokens (data_loader_pretrain.py:549-613) for a batch.
    inputs int64 [B, T] token ids; special_mask bool [B, T] (ids in tokenizer.all_special_ids); token_classes int64
    [B, T] (-1 = none) when masked_token_prediction.  draws = (u_mask, u_replace, u_random, random_words): uniform
    [0,1) tensors [B, T] and int64 [B, T] random ids; drawn from `generator` when omitted.
    -> (inputs', labels, attention_mask bool [B, T])."""
    dev = inputs.device
    if draws is None:
        u_mask, u_replace, u_random = (torch.rand(inputs.shape, generator=generator, device=dev) for _ in range(3))
        random_words = torch.randint(vocab_size, inputs.shape, generator=generator, device=dev, dtype=torch.long)
    else:
        u_mask, u_replace, u_random, random_words = draws
    if inputs.is_cuda:
        return _mask_tokens_hip(inputs, special_mask, pad_id, mask_id, mlm_probability, token_classes, u_mask, u_replace,
                                u_random, random_words)
    labels = inputs.clone()
    prob = torch.full(inputs.shape, float(mlm_probability), device=dev).masked_fill(special_mask, 0.0)
    masked = u_mask < prob                                    # torch.bernoulli(probability_matrix)
    tcm = None
    if token_classes is not None:
        tcm = token_classes != -1
        masked = masked | tcm
    attention_mask = inputs != pad_id
    labels = torch.where(masked, labels, torch.full_like(labels, -1))
    if tcm is not None:
        labels = torch.where(tcm, torch.full_like(labels, -1), labels)
    replaced = (u_replace < 0.8) & masked
    out = torch.where(replaced, torch.full_like(inputs, mask_id), inputs)
    if tcm is not None:
        replaced = replaced | tcm
        out = torch.where(tcm, torch.full_like(inputs, mask_id), out)
    rnd = (u_random < 0.5) & masked & ~replaced
    out = torch.where(rnd, random_words, out)
    return out, labels, attention_mask


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _i64c(t):
    return t if (t.dtype == torch.int64 and t.is_contiguous()) else t.to(torch.int64).contiguous()


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.to(torch.float32).contiguous()


def _mask_tokens_hip(inputs, special_mask, pad_id, mask_id, mlm_probability, token_classes, u_mask, u_replace, u_random,
                     random_words):
    from . import _lib

    ids = _i64c(inputs)
    sp = special_mask.to(torch.bool).contiguous().view(torch.uint8)
    tc = None if token_classes is None else _i64c(token_classes)
    out, labels, att = torch.empty_like(ids), torch.empty_like(ids), torch.empty_like(ids)
    rc = _lib.load().vt_mask_tokens(_p(ids), _p(sp), _p(tc), _p(_f32c(u_mask)), _p(_f32c(u_replace)), _p(_f32c(u_random)),
                                    _p(_i64c(random_words)), _p(out), _p(labels), _p(att), ids.numel(), int(pad_id),
                                    int(mask_id), float(mlm_probability),
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "vt_mask_tokens")
    return out, labels, att.to(torch.bool)


def _assemble_batch_hip(input_ids, labels, text_attention_mask, img_feats, region_counts, region_view_ids,
                        current_view_index, next_action, R, token_classes, no_action_grounding):
    from . import _lib

    B, R_in, D = img_feats.shape
    T = input_ids.shape[1]
    dev = img_feats.device
    feats = torch.empty((B, R, D), dtype=torch.float32, device=dev)
    loc = torch.empty((B, R, 128), dtype=torch.float32, device=dev)
    lab = torch.empty((B, T + R), dtype=torch.int64, device=dev)
    att = torch.empty((B, T + R), dtype=torch.int64, device=dev)
    tok = None if token_classes is None else torch.empty((B, T + R), dtype=torch.int64, device=dev)
    tc = None if token_classes is None else _i64c(token_classes)
    rc = _lib.load().vt_assemble_regions(
        _p(_f32c(img_feats)), _p(_i64c(region_counts)), _p(_i64c(region_view_ids)), _p(_i64c(current_view_index)),
        _p(loc_embedding_table(dev)), _p(_i64c(labels)), _p(_i64c(text_attention_mask)), _p(tc), _p(feats), _p(loc), _p(lab),
        _p(att), _p(tok), B, T, R, R_in, D, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "vt_assemble_regions")
    return dict(input_ids=input_ids, labels=lab, attention_mask=att, img_feats=feats, img_location_embeddings=loc,
                next_action=torch.full_like(next_action, -1) if no_action_grounding else next_action, token_labels=tok)


def assemble_batch(input_ids, labels, text_attention_mask, img_feats, region_counts, region_view_ids, current_view_index,
                   next_action, max_img_seq_length, token_classes=None, no_action_grounding=False):
    """The tail of PretrainDataset._preprocess_item (data_loader_pretrain.py:654-712) for a batch whose region rows are
    already padded to a common R_in: img_feats fp32 [B, R_in, D], region_counts int64 [B] valid rows per item (the
    reference concatenates 5 rows per view), region_view_ids int64 [B, R_in].  Items with more than max_img_seq_length
    rows keep their LAST max_img_seq_length rows (:660-665), shorter ones are zero-padded with mask 0 (:669-689); labels
    and token labels get -1 on every region position (:691-700).  Returns the PreTrainOscar.forward kwargs."""
    B, R_in, D = img_feats.shape
    R = int(max_img_seq_length)
    dev = img_feats.device
    if img_feats.is_cuda:
        return _assemble_batch_hip(input_ids, labels, text_attention_mask, img_feats, region_counts, region_view_ids,
                                   current_view_index, next_action, R, token_classes, no_action_grounding)
    loc = region_location_embeddings(current_view_index, region_view_ids)
    n = region_counts.clamp(max=R_in)
    start = (n - R).clamp(min=0)                                # first kept row of every item
    idx = start[:, None] + torch.arange(R, device=dev)[None, :]
    valid = idx < n[:, None]
    idx_c = idx.clamp(max=max(R_in - 1, 0))
    feats = torch.gather(img_feats, 1, idx_c[..., None].expand(B, R, D)) * valid[..., None]
    loc = torch.gather(loc, 1, idx_c[..., None].expand(B, R, 128)) * valid[..., None]
    att = torch.cat([text_attention_mask.to(torch.bool), valid], 1) if R > 0 else text_attention_mask.to(torch.bool)
    pad = torch.full((B, R), -1, dtype=torch.long, device=dev)
    out = dict(input_ids=input_ids, labels=torch.cat([labels, pad], 1), attention_mask=att.to(torch.long), img_feats=feats,
               img_location_embeddings=loc,
               next_action=torch.full_like(next_action, -1) if no_action_grounding else next_action)
    out["token_labels"] = None if token_classes is None else torch.cat([token_classes, pad], 1)
    return out
