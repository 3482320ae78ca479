"""Generate tests/golden/ref_*.npz by EXECUTING the reference's own Python -- build container only:

    python tests/golden/make_golden_from_reference.py [case ...]

What runs is the reference's source where it lies (/root/reference is put on sys.path, nothing is copied):

    oscar/modeling_bert.py                      CaptionBertSelfAttention / Attention / Layer / Encoder   (:26-169)
    tasks/viewpoint_select/encoder.py           NextActionPrediction, BertImgModelwithLocationEmbeds, PreTrainOscar (:142-441)
    tasks/viewpoint_select/agent_models.py      OscarEncoder, SoftDotAttention, AttnDecoderLSTM         (:192-428)
    tasks/viewpoint_select/data_loader_pretrain.py   build_viewpoint_loc_embedding, PretrainDataset._mask_tokens /
                                                _extract_img_features / _preprocess_item               (:25-49, :549-712)

STAND-INS (flagged: these are NOT the reference).  The reference imports its BERT building blocks from
``transformers.pytorch_transformers.modeling_bert``, an un-vendored git submodule (/root/reference/.gitmodules:1-3,
directory empty, commit not recoverable).  ``install_standins`` registers a module of that name whose blocks are this
repo's restatement ``oracle.bert_blocks`` (cross-checked against the independent transformers 5.x by
oracle/crosscheck_hf.py) plus three constructor-only bases ``BertAttention`` / ``BertLayer`` / ``BertEncoder`` -- the
reference's subclasses overwrite every sub-module those bases would build (oscar/modeling_bert.py:87-90,106-110,
132-138) and override ``forward``.  ``utils_data`` (imported by data_loader_pretrain.py:11-17 for file loading only)
needs ``lmdb``, which the image lacks: an empty module of that name is registered; no function exercised here touches it.
So the fixtures pin, by execution, every line of the four files above that the oracle restates; what stays UNPINNED by
the reference is the arithmetic inside the pytorch-transformers blocks themselves (LayerNorm, erf-GELU, embeddings,
pooler, MLM head, init, AdamW / schedules), which rests on oracle/crosscheck_hf.py and tests/test_oracle_kat.py.

Weights come from visitron_amd.synth.deterministic_state_dict (an integer hash: no RNG), inputs from
visitron_amd.synth.make_batch (private seeded CPU generator), so the fixtures hold outputs (whole tensors at mini size,
strided slices + norms at base size) and input checksums only.  Each case also runs oracle.* on the same inputs in this
process and records max |oracle - reference| per output in tests/golden/ref_pin_report.json (0.0 = bitwise equal).

The reference's Python never travels to the GPU box: only the .npz / .json written here do.
"""
import json
import os
import sys
import types
import warnings

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from visitron_amd.config import BertConfig, mini_config  # noqa: E402
from visitron_amd.synth import deterministic_state_dict, make_batch  # noqa: E402

TRUNK_KEYS = ("input_ids", "attention_mask", "img_feats", "img_location_embeddings")
GRAD_SLICE = 2048          # elements kept per parameter at base size (strided over the flattened gradient)
REPORT = {}


# ----------------------------------------------------------------------------------------------------------------------
# stand-ins for what the image lacks (see the module docstring) and the import of the reference itself
# ----------------------------------------------------------------------------------------------------------------------
def install_standins():
    from oracle import bert_blocks as bb

    class _ConstructorOnlyBase(nn.Module):
        """Upstream BertAttention / BertLayer / BertEncoder build sub-modules the reference's subclasses replace."""

        def __init__(self, config):
            super().__init__()

    pkg = types.ModuleType("transformers")
    pkg.__path__ = []
    pt = types.ModuleType("transformers.pytorch_transformers")
    pt.__path__ = []
    mb = types.ModuleType("transformers.pytorch_transformers.modeling_bert")
    mb.__doc__ = "STAND-IN for the un-vendored pytorch-transformers 1.x package: oracle.bert_blocks, not the reference"
    for n in ("BertEmbeddings", "BertLayerNorm", "BertOnlyMLMHead", "BertPooler", "BertPreTrainedModel",
              "BertIntermediate", "BertOutput", "BertSelfAttention", "BertSelfOutput"):
        setattr(mb, n, getattr(bb, n))
    for n in ("BertAttention", "BertLayer", "BertEncoder"):
        setattr(mb, n, type(n, (_ConstructorOnlyBase,), {}))
    pkg.pytorch_transformers, pt.modeling_bert = pt, mb
    sys.modules.update({"transformers": pkg, "transformers.pytorch_transformers": pt,
                        "transformers.pytorch_transformers.modeling_bert": mb})
    if "lmdb" not in sys.modules:
        try:
            import lmdb  # noqa: F401
        except ImportError:
            sys.modules["lmdb"] = types.ModuleType("lmdb")   # file reader only; never called here


def load_reference():
    if not os.path.isdir(REF):
        raise SystemExit("make_golden_from_reference.py runs in the build container only: %s is absent" % REF)
    install_standins()
    sys.path.insert(0, os.path.join(REF, "tasks", "viewpoint_select"))
    sys.path.insert(0, REF)
    import agent_models
    import data_loader_pretrain
    import encoder
    import oscar.modeling_bert as caption

    for mod in (agent_models, data_loader_pretrain, encoder, caption):
        assert mod.__file__.startswith(REF), mod.__file__
    return types.SimpleNamespace(caption=caption, encoder=encoder, agent_models=agent_models, data=data_loader_pretrain)


# ----------------------------------------------------------------------------------------------------------------------
def _np(t):
    return t.detach().cpu().numpy()


def _same_weights(ref_model, oracle_model, seed, std):
    sd = deterministic_state_dict(ref_model, seed=seed, weight_std=std)
    missing = ref_model.load_state_dict(sd)
    assert not missing.missing_keys and not missing.unexpected_keys, missing
    missing = oracle_model.load_state_dict(sd)                       # same names: the checkpoint-key contract
    assert not missing.missing_keys and not missing.unexpected_keys, missing


def _diff(case, name, ref_t, oracle_t):
    a, b = torch.as_tensor(ref_t).double(), torch.as_tensor(oracle_t).double()
    assert a.shape == b.shape, (case, name, a.shape, b.shape)
    nan = torch.isnan(a)
    assert torch.equal(nan, torch.isnan(b)), (case, name, "NaN pattern differs")
    d = float((a - b)[~nan].abs().max()) if (~nan).any() else 0.0
    REPORT.setdefault(case, {})[name] = d
    return d


def _pretrain_run(R, O, cfg, batch, seed, std, case):
    """Reference and oracle PreTrainOscar on the same weights and batch: trunk, heads, 7-tuple, backward."""
    ref = R.encoder.PreTrainOscar(cfg).eval()
    ora = O.PreTrainOscar(cfg).eval()
    _same_weights(ref, ora, seed, std)
    with torch.no_grad():
        tk = {k: batch[k] for k in TRUNK_KEYS if k in batch}
        seq, pooled = ref.bert(**tk)[:2]
        scores = ref.mlmhead(seq)                                   # encoder.py:377
        tokp = ref.token_head(seq)                                  # encoder.py:381 (Linear + Softmax)
        act = ref.next_action(pooled)                               # encoder.py:391
        oseq, opooled = ora.bert(**tk)[:2]
        oscores, otokp, oact = ora.heads(oseq, opooled)
    out7 = ref(**batch)
    out7[0].backward()
    o7 = ora(**batch)
    o7[0].backward()
    grads = {n: p.grad.detach() for n, p in ref.named_parameters()}
    ograds = {n: p.grad.detach() for n, p in ora.named_parameters()}
    assert sorted(grads) == sorted(ograds)
    for n, a, b in (("sequence_output", seq, oseq), ("pooled_output", pooled, opooled), ("prediction_scores", scores, oscores),
                    ("token_probs", tokp, otokp), ("action_scores", act, oact)):
        _diff(case, n, a, b)
    _diff(case, "tuple7", torch.stack([x.detach().double() for x in out7]), torch.stack([x.detach().double() for x in o7]))
    REPORT[case]["grad_max_rel"] = max(float((grads[n] - ograds[n]).norm() / (grads[n].norm() + 1e-30)) for n in grads)
    return ref, dict(seq=seq, pooled=pooled, scores=scores, tokp=tokp, act=act,
                     tuple7=np.array([float(x) for x in out7], dtype=np.float64), grads=grads)


def _grad_slices(grads):
    names = sorted(grads)
    out = dict(grad_names=np.array(names), grad_norms=np.array([float(grads[n].double().norm()) for n in names]))
    sl = []
    for n in names:
        flat = grads[n].reshape(-1)
        step = max(1, flat.numel() // GRAD_SLICE)
        sl.append(_np(flat[::step][:GRAD_SLICE]).astype(np.float32))
    out["grad_slice_offsets"] = np.cumsum([0] + [len(s) for s in sl])
    out["grad_slices"] = np.concatenate(sl)
    return out


def _checksums(batch):
    out = {}
    for k, v in batch.items():
        if v.dtype.is_floating_point:
            out["in_%s_checksum" % k] = np.array([float(v.double().sum()), float(v.double().abs().max())])
        else:
            out["in_" + k] = _np(v)
    return out


# ----------------------------------------------------------------------------------------------------------------------
# cases
# ----------------------------------------------------------------------------------------------------------------------
def case_mini(R, O):
    """Mini config (L=2, H=128, 2 heads of 64), B=3, 20 text + 17 regions: whole tensors, every gradient, and the
    trunk's edge cases (head_mask, 3-D mask, history states, hidden states / attentions, text_only, image LayerNorm,
    the ignore / NaN corners of the losses, non-0/1 masks)."""
    cfg = mini_config()
    b = make_batch(cfg, 3, text_len=20, region_len=17, seed=11)
    ref, r = _pretrain_run(R, O, cfg, b, 3, 0.05, "mini")
    names = sorted(r["grads"])
    out = {"in_" + k: _np(v) for k, v in b.items()}
    out.update(sequence_output=_np(r["seq"]), pooled_output=_np(r["pooled"]), prediction_scores=_np(r["scores"]),
               token_probs=_np(r["tokp"]), action_scores=_np(r["act"]), tuple7=r["tuple7"], grad_names=np.array(names))
    for i, n in enumerate(names):
        out["grad_%03d" % i] = _np(r["grads"][n])

    ora = O.PreTrainOscar(cfg).eval()
    ora.load_state_dict(ref.state_dict())
    B, T, Rg = 3, 20, 17
    S = T + Rg
    g = torch.Generator().manual_seed(5)
    tk = {k: b[k] for k in TRUNK_KEYS}

    def both(tag, fn, n_out=2):
        with torch.no_grad():
            a, o = fn(ref), fn(ora)
        for i in range(n_out):
            _diff("mini", "%s[%d]" % (tag, i), a[i], o[i])
            out["%s_%d" % (tag, i)] = _np(a[i])
        return a

    # head_mask: 1-D [heads] and 2-D [layers, heads] (encoder.py:248-265, oscar/modeling_bert.py:65-66,153)
    hm1 = torch.tensor([1.0, 0.0])
    hm2 = torch.tensor([[1.0, 0.5], [0.0, 1.0]])
    out["in_head_mask_1d"], out["in_head_mask_2d"] = _np(hm1), _np(hm2)
    both("headmask1d", lambda m: m.bert(head_mask=hm1, **tk))
    both("headmask2d", lambda m: m.bert(head_mask=hm2, **tk))
    # 3-D mask [B, S, S] (encoder.py:228-229)
    m3 = (torch.rand(B, S, S, generator=g) < 0.8).long()
    m3[:, :, 0] = 1
    out["in_mask3d"] = _np(m3)
    both("mask3d", lambda m: m.bert(b["input_ids"], attention_mask=m3, img_feats=b["img_feats"],
                                    img_location_embeddings=b["img_location_embeddings"]))
    # non-0/1 masks: float weights and the uint8 ~mask of agent_models.py:267 (254 / 255)
    mf = b["attention_mask"].float() * 0.5 + 0.25
    both("maskfloat", lambda m: m.bert(b["input_ids"], attention_mask=mf, img_feats=b["img_feats"],
                                       img_location_embeddings=b["img_location_embeddings"]))
    mu8 = ~(b["attention_mask"][:, :T] == 0).byte()
    out["in_mask_u8"] = _np(mu8)
    both("masku8", lambda m: m.bert(b["input_ids"], attention_mask=mu8))
    # encoder_history_states: text only (encoder.py:271-274), one [B, P, H] prefix per layer (oscar/modeling_bert.py:37-41,149-151)
    P = 6
    hist = [torch.randn(B, P, cfg.hidden_size, generator=g) * 0.5 for _ in range(cfg.num_hidden_layers)]
    mh = torch.cat([torch.ones(B, P, dtype=torch.long), b["attention_mask"][:, :T]], 1)
    for i, h in enumerate(hist):
        out["in_history_%d" % i] = _np(h)
    both("history", lambda m: m.bert(b["input_ids"], attention_mask=mh, encoder_history_states=hist))
    # token types, explicit position ids
    tt = (torch.arange(T)[None, :] >= 9).long().expand(B, T).contiguous()
    pid = torch.arange(T - 1, -1, -1)[None, :].expand(B, T).contiguous()
    both("types_positions", lambda m: m.bert(b["input_ids"], token_type_ids=tt, position_ids=pid,
                                             attention_mask=b["attention_mask"][:, :T]))
    # text_only returns the trunk outputs (encoder.py:371-372)
    both("text_only", lambda m: m(b["input_ids"], attention_mask=b["attention_mask"], img_feats=b["img_feats"],
                                  img_location_embeddings=b["img_location_embeddings"], text_only=True))
    # loss corners: no supervised word -> NaN mask loss; next_action -1 everywhere; one ignored action
    def seven(tag, **over):
        bb = dict(b)
        bb.update(over)
        with torch.no_grad():
            a = torch.stack([x.double() for x in ref(**bb)])
            o = torch.stack([x.double() for x in ora(**bb)])
        _diff("mini", tag, a, o)
        out[tag] = _np(a)

    seven("tuple7_no_labels", labels=torch.full_like(b["labels"], -1))
    seven("tuple7_no_token_labels", token_labels=torch.full_like(b["token_labels"], -1))
    na = b["next_action"].clone()
    na[1] = -1
    seven("tuple7_one_action_ignored", next_action=na)
    seven("tuple7_all_actions_ignored", next_action=torch.full_like(na, -1))

    # output_hidden_states / output_attentions (oscar/modeling_bert.py:74-79,146-168) and the image LayerNorm branch
    cfg2 = mini_config(output_hidden_states=True, output_attentions=True, use_img_layernorm=True, img_layer_norm_eps=1e-5)
    ref2, ora2 = R.encoder.BertImgModelwithLocationEmbeds(cfg2).eval(), O.BertImgModelwithLocationEmbeds(cfg2).eval()
    _same_weights(ref2, ora2, 4, 0.05)
    with torch.no_grad():
        a, o = ref2(**tk), ora2(**tk)
    assert len(a) == 4 and len(a[2]) == cfg.num_hidden_layers + 1 and len(a[3]) == cfg.num_hidden_layers
    _diff("mini", "imgln_seq", a[0], o[0])
    out["imgln_seq"], out["imgln_pooled"] = _np(a[0]), _np(a[1])
    for i in range(len(a[2])):
        _diff("mini", "hidden_states[%d]" % i, a[2][i], o[2][i])
        out["hidden_states_%d" % i] = _np(a[2][i])
    for i in range(len(a[3])):
        _diff("mini", "attentions[%d]" % i, a[3][i], o[3][i])
        out["attentions_%d" % i] = _np(a[3][i])
    np.savez_compressed(os.path.join(HERE, "ref_mini.npz"), **out)
    print("ref_mini 7-tuple", list(r["tuple7"]))


def _base_case(R, O, case, fname, batch_size, text_len, region_len, seed, seq_stride):
    cfg = BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    b = make_batch(cfg, batch_size, text_len=text_len, region_len=region_len, seed=seed)
    ref, r = _pretrain_run(R, O, cfg, b, 0, 0.03, case)
    out = _checksums(b)
    out.update(
        sequence_output_slice=_np(r["seq"][:, ::seq_stride, ::31]), sequence_output_absmax=np.array([float(r["seq"].abs().max())]),
        pooled_output=_np(r["pooled"]), prediction_scores_slice=_np(r["scores"][:, ::seq_stride, ::1009]),
        prediction_scores_absmax=np.array([float(r["scores"].abs().max())]),
        token_probs_slice=_np(r["tokp"][:, ::seq_stride, ::97]), action_scores=_np(r["act"]), tuple7=r["tuple7"],
        seq_stride=np.array([seq_stride]),
    )
    out.update(_grad_slices(r["grads"]))
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, "7-tuple", list(r["tuple7"]))


def case_base_cfg0(R, O):
    """BASELINE configs[0]: base config, B = 2, 128 text + 100 region tokens."""
    _base_case(R, O, "base_cfg0", "ref_base_cfg0.npz", 2, 128, 100, 1234, 19)


def case_base_long(R, O):
    """BASELINE configs[4]'s shape: 512 text + 144 region tokens (S = 656)."""
    _base_case(R, O, "base_long", "ref_base_long.npz", 2, 512, 144, 77, 41)


def case_shipped_pretrain(R, O):
    """The reference's shipped pretrain shape: 511 text + 256 regions, B = 2 (data_loader_pretrain.py:618-626,
    run_scripts/pretrain/pretrain_ndh_r2r.sh:35-37)."""
    _base_case(R, O, "shipped_pretrain", "ref_shipped_s767.npz", 2, 511, 256, 767, 59)


class _Args:
    device = torch.device("cpu")


def case_rollout(R, O):
    """agent_models.py: SoftDotAttention in its four output modes, one AttnDecoderLSTM step, and OscarEncoder over a
    768-wide 2-layer trunk with the uint8 ~mask quirk and ragged lengths (the class hard-codes 768: agent_models.py:214)."""
    from oracle import rollout as orollout

    out = {}
    g = torch.Generator().manual_seed(21)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # nn.Softmax() without dim (agent_models.py:324)
        # SoftDotAttention
        Q, D, L, B = 128, 132, 9, 4      # sizes the HIP rollout kernels serve (hidden % 128 == 0)
        ra, oa = R.agent_models.SoftDotAttention(Q, D).eval(), orollout.SoftDotAttention(Q, D).eval()
        _same_weights(ra, oa, 7, 0.08)
        h, ctx = torch.randn(B, Q, generator=g), torch.randn(B, L, D, generator=g)
        mask = torch.zeros(B, L, dtype=torch.bool)
        mask[1, 5:] = True
        mask[3, :2] = True
        out.update(sda_h=_np(h), sda_ctx=_np(ctx), sda_mask=_np(mask))
        with torch.no_grad():
            for mi, m in enumerate((None, mask)):
                for tilde in (True, False):
                    for prob in (True, False):
                        a = ra(h, ctx, None if m is None else m.clone(), output_tilde=tilde, output_prob=prob)
                        o = oa(h, ctx, None if m is None else m.clone(), output_tilde=tilde, output_prob=prob)
                        tag = "sda_m%d_t%d_p%d" % (mi, tilde, prob)
                        for i in range(2):
                            fin = torch.isfinite(a[i])
                            assert torch.equal(fin, torch.isfinite(o[i]))
                            _diff("rollout", "%s[%d]" % (tag, i), torch.where(fin, a[i], torch.zeros_like(a[i])),
                                  torch.where(fin, o[i], torch.zeros_like(o[i])))
                            out["%s_%d" % (tag, i)] = _np(a[i])
        # AttnDecoderLSTM step
        angle, emb, hs, F = 4, 64, 128, 132
        rd = R.agent_models.AttnDecoderLSTM(angle, emb, hs, 0.5, feature_size=F).eval()
        od = orollout.AttnDecoderLSTM(angle, emb, hs, 0.5, feature_size=F).eval()
        _same_weights(rd, od, 8, 0.06)
        ins = dict(action=torch.randn(B, angle, generator=g), feature=torch.randn(B, 36, F, generator=g),
                   cand_feat=torch.randn(B, 6, F, generator=g), h_0=torch.randn(B, hs, generator=g),
                   prev_h1=torch.randn(B, hs, generator=g), c_0=torch.randn(B, hs, generator=g),
                   ctx=torch.randn(B, 7, hs, generator=g))
        cmask = torch.zeros(B, 7, dtype=torch.bool)
        cmask[0, 4:] = True
        with torch.no_grad():
            a = rd(ctx_mask=cmask.clone(), **ins)
            o = od(ctx_mask=cmask.clone(), **ins)
        for k, v in ins.items():
            out["dec_in_" + k] = _np(v)
        out["dec_in_ctx_mask"] = _np(cmask)
        for i, n in enumerate(("h_1", "c_1", "logit", "h_tilde")):
            _diff("rollout", "decoder." + n, a[i], o[i])
            out["dec_" + n] = _np(a[i])
        # OscarEncoder over a 768-wide trunk
        cfg = BertConfig(num_hidden_layers=2, vocab_size=600, max_position_embeddings=64, hidden_dropout_prob=0.0,
                         attention_probs_dropout_prob=0.0, detector_classes=40)
        T, Bn = 24, 4
        rb, ob = R.encoder.BertImgModelwithLocationEmbeds(cfg).eval(), O.BertImgModelwithLocationEmbeds(cfg).eval()
        _same_weights(rb, ob, 9, 0.03)
        re_ = R.agent_models.OscarEncoder(_Args(), rb, 128, 96, 0.5).eval()
        oe = orollout.OscarEncoder(_Args(), ob, 128, 96, 0.5).eval()
        sd = deterministic_state_dict(re_, seed=10, weight_std=0.03)
        sd.update({k: v for k, v in re_.state_dict().items() if k.startswith("bert.")})
        re_.load_state_dict(sd)
        oe.load_state_dict(sd)
        ids = torch.randint(5, 600, (Bn, T), generator=g)
        lengths = [24, 19, 19, 7]
        pad = torch.zeros(Bn, T, dtype=torch.bool)
        for i, n in enumerate(lengths):
            pad[i, n:] = True
            ids[i, n:] = 0
        for tag, m in (("bool", pad), ("u8", pad.byte())):     # agent.py:181 hands a .byte() mask
            a = re_(ids, lengths, m)
            o = oe(ids, lengths, m)
            for i, n in enumerate(("ctx", "decoder_init", "c_t")):
                _diff("rollout", "oscar_encoder_%s.%s" % (tag, n), a[i], o[i])
                out["enc_%s_%s" % (tag, n)] = _np(a[i])
        loss = a[0].sum() + a[1].sum() + a[2].sum()
        loss.backward()
        (o[0].sum() + o[1].sum() + o[2].sum()).backward()
        gr = {n: p.grad for n, p in re_.named_parameters() if p.grad is not None}
        go = {n: p.grad for n, p in oe.named_parameters() if p.grad is not None}
        assert sorted(gr) == sorted(go)
        REPORT["rollout"]["oscar_encoder_grad_max_rel"] = max(float((gr[n] - go[n]).norm() / (gr[n].norm() + 1e-30)) for n in gr)
        out["enc_in_ids"], out["enc_in_lengths"] = _np(ids), np.array(lengths)
        out.update({"enc_" + k: v for k, v in _grad_slices(gr).items()})
        # round 6: the constructor arguments no reference caller sets (agent.py:110-117 leaves the defaults) -- reverse_input
        # (agent_models.py:277-282; with a uint8 mask the byte index marks EVERY position) and stacked / bidirectional
        # encoder LSTMs (num_layers = 2).  A separate file: ref_rollout.npz stays what round 5 wrote.
        out2 = dict(enc_in_ids=_np(ids), enc_in_lengths=np.array(lengths))
        for name, kw in (("rev", dict(reverse_input=True)), ("l2", dict(num_layers=2)),
                         ("l2bi_rev", dict(num_layers=2, bidirectional=True, reverse_input=True))):
            rx = R.agent_models.OscarEncoder(_Args(), rb, 128, 96, 0.5, **kw).eval()
            ox = orollout.OscarEncoder(_Args(), ob, 128, 96, 0.5, **kw).eval()
            sdx = deterministic_state_dict(rx, seed=12, weight_std=0.03)
            sdx.update({k: v for k, v in rx.state_dict().items() if k.startswith("bert.")})
            rx.load_state_dict(sdx)
            ox.load_state_dict(sdx)
            for tag, m in (("bool", pad), ("u8", pad.byte())):
                with torch.no_grad():
                    a = rx(ids, lengths, m)
                    o = ox(ids, lengths, m)
                for i, n in enumerate(("ctx", "decoder_init", "c_t")):
                    _diff("rollout", "oscar_encoder_%s_%s.%s" % (name, tag, n), a[i], o[i])
                    out2["enc_%s_%s_%s" % (name, tag, n)] = _np(a[i])
            # gradients through the reversed / stacked encoder (bool mask; eval mode: no dropout draws)
            a = rx(ids, lengths, pad)
            o = ox(ids, lengths, pad)
            rx.zero_grad(); ox.zero_grad()
            (a[0].sum() + a[1].sum() + a[2].sum()).backward()
            (o[0].sum() + o[1].sum() + o[2].sum()).backward()
            grx = {n: p.grad for n, p in rx.named_parameters() if p.grad is not None}
            gox = {n: p.grad for n, p in ox.named_parameters() if p.grad is not None}
            assert sorted(grx) == sorted(gox)
            REPORT["rollout"]["oscar_encoder_%s_grad_max_rel" % name] = max(
                float((grx[n] - gox[n]).norm() / (grx[n].norm() + 1e-30)) for n in grx)
            out2.update({"enc_%s_%s" % (name, k): v for k, v in _grad_slices(grx).items()})
    np.savez_compressed(os.path.join(HERE, "ref_rollout.npz"), **out)
    np.savez_compressed(os.path.join(HERE, "ref_rollout2.npz"), **out2)
    print("ref_rollout, ref_rollout2 written")


def case_text511(R, O):
    """The shipped rollout shape: text-only T = 511, B = 8, base config (agent_models.py:270-275; F6)."""
    cfg = BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    b = make_batch(cfg, 8, text_len=511, region_len=0, seed=511, with_labels=False)
    ref, ora = R.encoder.PreTrainOscar(cfg).eval(), O.PreTrainOscar(cfg).eval()
    _same_weights(ref, ora, 0, 0.03)
    pad_u8 = (b["attention_mask"] == 0).byte()
    with torch.no_grad():
        a = ref.bert(b["input_ids"], attention_mask=~pad_u8)          # what OscarEncoder.forward passes (:267-275)
        o = ora.bert(b["input_ids"], attention_mask=~pad_u8)
        a01 = ref.bert(b["input_ids"], attention_mask=b["attention_mask"])
    _diff("text511", "sequence_output", a[0], o[0])
    _diff("text511", "pooled_output", a[1], o[1])
    out = _checksums(b)
    out.update(sequence_output_slice=_np(a[0][:, ::37, ::31]), pooled_output=_np(a[1]),
               sequence_output_absmax=np.array([float(a[0].abs().max())]),
               sequence_output_01mask_slice=_np(a01[0][:, ::37, ::31]), pooled_output_01mask=_np(a01[1]))
    np.savez_compressed(os.path.join(HERE, "ref_text511.npz"), **out)
    print("ref_text511 written")


class _Tokenizer:
    """The four tokenizer attributes PretrainDataset._mask_tokens reads (data_loader_pretrain.py:559-600)."""
    all_special_ids = [0, 100, 101, 102, 103]
    pad_token_id = 0
    mask_token = "[MASK]"

    def convert_tokens_to_ids(self, tok):
        assert tok == "[MASK]"
        return 103

    def __len__(self):
        return 997


def case_data(R, O):
    """data_loader_pretrain.py: the 36 location tables and PretrainDataset._mask_tokens / _extract_img_features /
    _preprocess_item called unbound on a bare object holding `args` and `tokenizer`.  The methods draw in place with
    torch.bernoulli / torch.randint / torch.rand; the draws are taken from recorded uniforms by patching those three torch
    functions FOR THE CALL (bernoulli(p) := u < p), so that the fixture can hold them and oracle/data.py -- which takes
    the draws as arguments -- can be compared on the same ones."""
    from oracle import data as odata

    out = dict(loc_tables=np.stack([R.data.build_viewpoint_loc_embedding(v) for v in range(36)]),
               static_tables=np.stack(R.data._static_loc_embeddings))
    _diff("data", "loc_tables", torch.from_numpy(out["loc_tables"]), torch.from_numpy(np.stack(odata.STATIC)))
    DS = R.data.PretrainDataset
    tok = _Tokenizer()
    g = torch.Generator().manual_seed(33)
    T = 40
    n_items = 6
    real_bernoulli, real_randint, real_rand = torch.bernoulli, torch.randint, torch.rand
    rec = {}
    for mtp in (False, True):
        for it in range(n_items):
            tag = "mtp%d_item%d" % (mtp, it)
            max_img = (180, 200, 64, 0, 180, 256)[it]
            args = types.SimpleNamespace(mlm_probability=0.15, masked_token_prediction=mtp, debug=True,
                                         max_img_seq_length=max_img, no_action_grounding=(it == 4))
            self = types.SimpleNamespace(args=args, tokenizer=tok)
            self._mask_tokens = lambda i, tc, s=self: DS._mask_tokens(s, i, tc)
            self._extract_img_features = lambda a, b_, c, s=self: DS._extract_img_features(s, a, b_, c)
            ids = real_randint(200, 997, (T,), generator=g)
            ids[0] = 101
            n_real = T - 3 * it
            ids[n_real - 1] = 102
            ids[n_real:] = 0
            tc = torch.full((T,), -1, dtype=torch.long)
            tc[3 + it] = 7 + it
            tc[10] = 2
            u = [real_rand(T, generator=g) for _ in range(3)]
            words = real_randint(997, (T,), generator=g)
            feats = [real_rand(5, 2054, generator=g) for _ in range(36)]
            draws, fq = list(u), list(feats)
            torch.bernoulli = lambda p: (draws.pop(0) < p).to(p.dtype)
            torch.randint = lambda high, shape, dtype=torch.long: words.clone()
            torch.rand = lambda *shape: fq.pop(0)
            try:
                item = dict(target_dialog_tokens_id=ids.clone(), token_classes=tc.clone(), scan="s", viewpoint="v",
                            current_view_index=(5 * it) % 36, target_rel_view_index=(7 * it + 1) % 36)
                got = DS._preprocess_item(self, item)
            finally:
                torch.bernoulli, torch.randint, torch.rand = real_bernoulli, real_randint, real_rand
            assert not draws and not fq
            # the same through oracle/data.py
            w_in, w_lab, w_att = odata.mask_tokens_item(ids.clone(), set(tok.all_special_ids), 0, 103, 0.15, tc if mtp else None,
                                                        u[0], u[1], u[2], words)
            view_ids = [v for v in range(36) for _ in range(5)]
            want = odata.preprocess_item_tail(w_in, w_lab, w_att, np.concatenate([f.numpy() for f in feats], 0), view_ids,
                                              item["current_view_index"], item["target_rel_view_index"], max_img,
                                              token_classes=tc if mtp else None, no_action_grounding=(it == 4))
            for k in ("input_ids", "labels", "token_labels", "attention_mask", "img_feats", "img_location_embeddings", "next_action"):
                a, o = got[k], want[k]
                if a is None:
                    assert o is None
                    continue
                _diff("data", "%s.%s" % (tag, k), torch.as_tensor(a), torch.as_tensor(o))
                out["%s_%s" % (tag, k)] = _np(torch.as_tensor(a)) if k != "img_feats" else \
                    np.array([float(torch.as_tensor(a).double().sum()), a.shape[0]])
            rec[tag] = dict(max_img=max_img, current_view_index=item["current_view_index"],
                            target_rel_view_index=item["target_rel_view_index"], no_action_grounding=(it == 4))
            out["%s_in_ids" % tag], out["%s_in_token_classes" % tag] = _np(ids), _np(tc)
            out["%s_in_u" % tag], out["%s_in_words" % tag] = np.stack([_np(x) for x in u]), _np(words)
            out["%s_in_feats_seed" % tag] = np.array([33])
            out["%s_in_feats_first" % tag] = _np(feats[0][:, :8])
    out["items_json"] = np.array(json.dumps(rec))
    np.savez_compressed(os.path.join(HERE, "ref_data.npz"), **out)
    print("ref_data written")


CASES = dict(mini=case_mini, base_cfg0=case_base_cfg0, base_long=case_base_long, shipped_pretrain=case_shipped_pretrain,
             rollout=case_rollout, text511=case_text511, data=case_data)


def main(argv):
    torch.set_num_threads(8)
    R = load_reference()
    from oracle import modeling as O

    path = os.path.join(HERE, "ref_pin_report.json")
    if os.path.exists(path):
        REPORT.update(json.load(open(path)).get("max_abs_oracle_minus_reference", {}))
    for name in argv or list(CASES):
        REPORT.pop(name, None)
        CASES[name](R, O)
        print(name, "max |oracle - reference| =", max(REPORT[name].values()))
    with open(path, "w") as f:
        json.dump({"generator": "tests/golden/make_golden_from_reference.py",
                   "torch": torch.__version__, "threads": torch.get_num_threads(),
                   "standins": ["transformers.pytorch_transformers.modeling_bert := oracle.bert_blocks + constructor-only "
                                "BertAttention/BertLayer/BertEncoder", "lmdb := empty module"],
                   "max_abs_oracle_minus_reference": REPORT}, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1:])
