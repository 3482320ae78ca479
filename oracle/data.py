"""TEST INFRASTRUCTURE (see oracle/__init__.py): per-item CPU restatement of the pretrain input preparation,
tasks/viewpoint_select/data_loader_pretrain.py:25-49 (location embeddings), :549-613 (_mask_tokens), :615-712
(_extract_img_features / _preprocess_item tail), written item by item with Python lists and numpy like the reference.
The random draws are arguments (the reference calls torch.bernoulli / torch.randint in place); PARITY UNPINNED by the
reference: it ships no fixtures for this path."""
import numpy as np
import torch

ANGLE_INC = np.pi / 6.0


def build_viewpoint_loc_embedding(view_index):   # :23-43
    emb = np.zeros((36, 128), np.float32)
    for a in range(36):
        rel = (a - view_index) % 12 + (a // 12) * 12
        h = (rel % 12) * ANGLE_INC
        e = (rel // 12 - 1) * ANGLE_INC
        emb[a, 0:32] = np.sin(h)
        emb[a, 32:64] = np.cos(h)
        emb[a, 64:96] = np.sin(e)
        emb[a, 96:] = np.cos(e)
    return emb


STATIC = [build_viewpoint_loc_embedding(v) for v in range(36)]   # :46-49


def mask_tokens_item(inputs, special_ids, pad_id, mask_id, mlm_probability, token_classes, u_mask, u_replace, u_random,
                     random_words):   # :549-613 for ONE sequence
    inputs = inputs.clone()
    labels = inputs.clone()
    prob = torch.full(labels.shape, mlm_probability)
    special = torch.tensor([v in special_ids for v in labels.tolist()], dtype=torch.bool)
    att_pad = torch.tensor([v == pad_id for v in labels.tolist()], dtype=torch.bool)
    prob.masked_fill_(special, 0.0)
    masked = u_mask < prob
    if token_classes is not None:
        tcm = torch.tensor([v != -1 for v in token_classes.tolist()], dtype=torch.bool)
        masked.masked_fill_(tcm, True)
    attention_mask = torch.full(labels.shape, True).masked_fill_(att_pad, False)
    labels[~masked] = -1
    if token_classes is not None:
        labels[tcm] = -1
    replaced = (u_replace < 0.8) & masked
    inputs[replaced] = mask_id
    if token_classes is not None:
        replaced = replaced.masked_fill(tcm, True)
        inputs[tcm] = mask_id
    rnd = (u_random < 0.5) & masked & ~replaced
    inputs[rnd] = random_words[rnd]
    return inputs, labels, attention_mask


def preprocess_item_tail(inputs, labels, attention_mask, img_features, view_ids, view_index, target_view_index,
                         max_img_seq_length, token_classes=None, no_action_grounding=False):   # :627-633, :654-712
    loc = np.concatenate([STATIC[view_index][np.newaxis, i] for i in view_ids], axis=0) if len(view_ids) else np.zeros((0, 128), np.float32)
    img = torch.as_tensor(img_features)
    loc = torch.from_numpy(loc)
    att = attention_mask.tolist()
    if img.shape[0] > max_img_seq_length:
        img = img[-max_img_seq_length:]
        loc = loc[-max_img_seq_length:]
        if max_img_seq_length > 0:
            att = att + [1] * img.shape[0]
    else:
        if max_img_seq_length > 0:
            att = att + [1] * img.shape[0]
        pad = max_img_seq_length - img.shape[0]
        img = torch.cat((img, torch.zeros((pad, img.shape[1]), dtype=img.dtype)), 0)
        loc = torch.cat((loc, torch.zeros((pad, loc.shape[1]), dtype=loc.dtype)), 0)
        if max_img_seq_length > 0:
            att = att + [0] * pad
    labels = torch.LongTensor(labels.tolist() + [-1] * img.shape[0])
    if no_action_grounding:
        target_view_index = -1
    token_labels = None
    if token_classes is not None:
        token_labels = torch.LongTensor(token_classes.tolist() + [-1] * img.shape[0])
    return dict(input_ids=inputs, labels=labels, token_labels=token_labels, attention_mask=torch.tensor(att),
                img_feats=img, img_location_embeddings=loc, next_action=target_view_index)
