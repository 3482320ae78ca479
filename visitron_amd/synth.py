"""Synthetic pretrain batches shaped like the reference's dataset output.

The reference's ``PretrainDataset.__getitem__`` emits a dict whose keys are the
keyword arguments of ``PreTrainOscar.forward``
(tasks/viewpoint_select/data_loader_pretrain.py:703-711).  No dataset is
available offline, so benches and parity tests draw the same keys from the
distributions SURVEY.md section 8(d) fixes:

  input_ids   int64 [B,T]  U{1000..vocab-1}, [:,0]=101, padded tail (id 0) of length U{0..T/4}
  attention_mask  [B,T+R]  1 on real text / first r_b~U{3R/4..R} regions, else 0
                           (region mask bits appended: data_loader_pretrain.py:666-690)
  img_feats   fp32 [B,R,img_dim]  |N(0,1)|*0.5 on the first img_dim-6 dims,
                           U[0,1] box geometry on the last 6
                           (scripts/add_orientation_to_features.py:111-128), zero rows where masked
  img_location_embeddings fp32 [B,R,128]  rows of build_viewpoint_loc_embedding
                           (data_loader_pretrain.py:25-43) for a random viewIndex
  labels      int64 [B,T+R]  15% of real text positions keep their id, rest -1,
                           region positions always -1 (data_loader_pretrain.py:576,692)
  token_labels int64 [B,T+R] -1 except ~10% of text positions ~ U{0..detector_classes-1}
  next_action int64 [B]    U{0..action_space-1}

Everything is generated on the CPU with a private ``torch.Generator`` so that the
same seed gives the same batch on every machine, then moved to ``device``.
"""
import math

import torch

_ANGLE_INC = math.pi / 6.0


def viewpoint_loc_embedding(view_index):
    """36x128 heading/elevation sinusoid table for one agent heading.

    Same values as ``build_viewpoint_loc_embedding``
    (tasks/viewpoint_select/data_loader_pretrain.py:25-43): for each of the 36
    absolute views, 32x sin(rel heading), 32x cos(rel heading), 32x sin(rel
    elevation), 32x cos(rel elevation).
    """
    a = torch.arange(36)
    rel = (a - view_index) % 12 + (a // 12) * 12
    heading = (rel % 12).to(torch.float64) * _ANGLE_INC
    elevation = (torch.div(rel, 12, rounding_mode="floor") - 1).to(torch.float64) * _ANGLE_INC
    out = torch.empty(36, 128, dtype=torch.float64)
    out[:, 0:32] = torch.sin(heading)[:, None]
    out[:, 32:64] = torch.cos(heading)[:, None]
    out[:, 64:96] = torch.sin(elevation)[:, None]
    out[:, 96:128] = torch.cos(elevation)[:, None]
    return out.to(torch.float32)


def make_batch(config, batch, text_len=128, region_len=100, seed=1234, device="cpu", with_labels=True):
    """One synthetic pretrain batch; keys == PreTrainOscar.forward kwargs."""
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    B, T, R = int(batch), int(text_len), int(region_len)
    V = int(config.vocab_size)
    lo = min(1000, max(1, V // 2))

    input_ids = torch.randint(lo, V, (B, T), generator=g, dtype=torch.int64)
    input_ids[:, 0] = 101 if V > 101 else 1
    pad = torch.randint(0, T // 4 + 1, (B,), generator=g)
    pos = torch.arange(T)[None, :]
    text_real = pos < (T - pad)[:, None]
    input_ids = torch.where(text_real, input_ids, torch.zeros_like(input_ids))

    out = {"input_ids": input_ids}
    mask = text_real.to(torch.int64)
    if R > 0:
        nreg = torch.randint((3 * R) // 4, R + 1, (B,), generator=g)
        reg_real = torch.arange(R)[None, :] < nreg[:, None]
        mask = torch.cat([mask, reg_real.to(torch.int64)], 1)
        D = int(config.img_feature_dim)
        feats = torch.empty(B, R, D)
        ngeo = min(6, D)
        feats[..., : D - ngeo] = torch.randn(B, R, D - ngeo, generator=g).abs() * 0.5
        feats[..., D - ngeo :] = torch.rand(B, R, ngeo, generator=g)
        feats = feats * reg_real[..., None]
        view = torch.randint(0, 36, (B,), generator=g)
        tables = torch.stack([viewpoint_loc_embedding(int(v)) for v in view])  # [B,36,128]
        idx = torch.arange(R) % 36
        loc = tables[:, idx, :] * reg_real[..., None]
        out["img_feats"] = feats
        out["img_location_embeddings"] = loc.contiguous()
    out["attention_mask"] = mask

    if with_labels:
        S = T + R
        labels = torch.full((B, S), -1, dtype=torch.int64)
        pick = (torch.rand(B, T, generator=g) < 0.15) & text_real
        pick[:, 0] = False
        labels[:, :T] = torch.where(pick, input_ids, labels[:, :T])
        # keep at least one supervised word per batch so the mean is defined
        if not bool(pick.any()):
            labels[0, 1] = input_ids[0, 1]
        tl = torch.full((B, S), -1, dtype=torch.int64)
        tpick = (torch.rand(B, T, generator=g) < 0.10) & text_real
        tvals = torch.randint(0, int(config.detector_classes), (B, T), generator=g)
        tl[:, :T] = torch.where(tpick, tvals, tl[:, :T])
        if not bool(tpick.any()):
            tl[0, 1] = 0
        out["labels"] = labels
        out["token_labels"] = tl
        out["next_action"] = torch.randint(0, int(config.action_space), (B,), generator=g)

    return {k: v.to(device) for k, v in out.items()}


# ---------------------------------------------------------------------------
# Deterministic weights (no RNG library involved): reproducible on every machine,
# so golden fixtures only need to hold outputs.
# ---------------------------------------------------------------------------
def _hash_uniform(n, salt):
    """n values in [-1, 1): a 64-bit integer hash of the element index (numpy uint64 wraparound)."""
    import numpy as np

    i = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = (i + np.uint64(salt)) * np.uint64(0x9E3779B97F4A7C15)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    u = (x >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return u * 2.0 - 1.0


def _name_salt(name, seed):
    h = 1469598103934665603
    for ch in (name + "#%d" % seed).encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def deterministic_state_dict(model, seed=0, weight_std=0.05):
    """A full state_dict for ``model`` drawn from the index hash above.

    Matrices/embeddings ~ U(-a, a) with std ``weight_std``; LayerNorm weights
    1 + 0.1 u; every bias 0.05 u (non-zero on purpose: zero biases would hide a
    dropped bias add).  Tied tensors are filled once.
    """
    import numpy as np

    a = weight_std * math.sqrt(3.0)
    out, seen = {}, {}
    for name, t in model.state_dict().items():
        key = (t.data_ptr(), tuple(t.shape))
        if key in seen:
            out[name] = out[seen[key]]
            continue
        seen[key] = name
        u = _hash_uniform(t.numel(), _name_salt(name, seed)).reshape(tuple(t.shape))
        if name.endswith("LayerNorm.weight"):
            v = 1.0 + 0.1 * u
        elif name.endswith("bias"):
            v = 0.05 * u
        else:
            v = a * u
        out[name] = torch.from_numpy(np.ascontiguousarray(v)).to(t.dtype)
    return out
