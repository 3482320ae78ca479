"""GPU: training THROUGH the rollout caller and the trunk (tasks/viewpoint_select/agent.py:493-518
back-propagates the rollout loss through OscarEncoder and AttnDecoderLSTM, then steps Adam on both).  Every autograd node of
visitron_amd/rollout_autograd.py and the trunk node (training._TrunkWithGrads) against torch autograd on the CPU oracle:
op level first (soft-dot block, LSTM cell, LSTM over a packed sequence), then the modules, then an encoder + two decoder
steps with a cross-entropy loss.  bf16 operands in every product: gradients are compared by relative L2."""
import pytest
import torch
import torch.nn as nn

from helpers import check_close, model_pair

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()   # (a float64 oracle's gradients included)
    return float((a - b).norm() / (b.norm() + 1e-6 * (b.numel() ** 0.5)))


def _grads_close(tag, prod, ref, tol):
    """Every parameter's gradient against the reference's by relative L2.  A tensor whose reference gradient is far below the
    typical one (the query / key projections under near-uniform attention: dS = P (dP - delta) is then a difference of almost
    equal bf16-rounded numbers, 200 x below the other tensors) is measured against 3 % of the median tensor's rms instead of
    its own vanishing norm."""
    wg = dict(ref.named_parameters())
    rms = sorted(float(w.grad.float().pow(2).mean().sqrt()) for w in wg.values() if w.grad is not None)
    floor = 0.03 * rms[len(rms) // 2]
    errs = {}
    for n, p in prod.named_parameters():
        w = wg[n].grad
        if w is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, (tag, n, "reference has no gradient here")
            continue
        assert p.grad is not None, (tag, n, "no gradient")
        w = w.float()
        d = (p.grad.detach().float().cpu() - w).norm()
        scale = w.norm()
        if n.endswith("attention.self.key.bias"):
            # a key bias shifts every logit of a query alike: its true gradient is zero and both sides hold rounding noise.
            # Measured against the size of the query bias's gradient instead of its own.
            scale = wg[n.replace(".key.", ".query.")].grad.float().norm()
        errs[n] = float(d / (scale + floor * w.numel() ** 0.5))
    worst = max(errs, key=errs.get)
    if errs[worst] > tol:
        print("worst gradients:", sorted(errs.items(), key=lambda kv: -kv[1])[:8])
    check_close("%s grads worst rel-L2 (%d tensors)" % (tag, len(errs)), errs[worst], 0.0, tol)
    return errs


@pytest.mark.parametrize("D,L,prob", [(512, 80, True), (2052, 36, False), (128, 7, True), (132, 300, False), (512, 511, True)])
def test_softdot_backward_matches_autograd(dev, D, L, prob):
    from visitron_amd import ops

    g = torch.Generator().manual_seed(D + L)
    B = 5
    t = torch.randn(B, D, generator=g) * 0.2
    c = torch.randn(B, L, D, generator=g) * 0.5
    mask = torch.zeros(B, L, dtype=torch.bool)
    mask[1, L // 2:] = True
    mask[3, :2] = True
    dw = torch.randn(B, D, generator=g)
    da = torch.randn(B, L, generator=g)
    for m in (None, mask):
        tr, cr = t.clone().requires_grad_(True), c.clone().requires_grad_(True)
        logit = torch.bmm(cr, tr.unsqueeze(2)).squeeze(2)
        if m is not None:
            logit = logit.masked_fill(m, -float("inf"))
        p = torch.softmax(logit, 1)
        weighted = torch.bmm(p.unsqueeze(1), cr).squeeze(1)
        out_a = p if prob else logit
        fin = torch.isfinite(out_a)
        ((weighted * dw).sum() + (torch.where(fin, out_a, torch.zeros_like(out_a)) * da).sum()).backward()
        d_t, d_c = ops.softdot_attention_bwd(t.to(dev), c.to(dev), None if m is None else m.to(dev), dw.to(dev), da.to(dev),
                                             prob, True)
        tag = "softdot bwd D%d L%d %s%s" % (D, L, "prob" if prob else "logit", "" if m is None else " masked")
        check_close(tag + " d_target rel-L2", _rel(d_t, tr.grad), 0.0, 1e-4)
        check_close(tag + " d_context rel-L2", _rel(d_c, cr.grad), 0.0, 1e-4)
        # one of the two incoming gradients absent, d_context not wanted
        d_t2, none = ops.softdot_attention_bwd(t.to(dev), c.to(dev), None if m is None else m.to(dev), dw.to(dev), None, prob,
                                               False)
        assert none is None and torch.isfinite(d_t2).all()


def test_lstm_cell_node_matches_autograd(dev):
    """nn.LSTMCell forward + backward (inputs, both states, all four parameters) at the decoder's sizes."""
    from visitron_amd import rollout_autograd as ra

    torch.manual_seed(3)
    B, ind, hs = 9, 64 + 2052, 512
    cell = nn.LSTMCell(ind, hs)
    x = (torch.randn(B, ind) * 0.3).requires_grad_(True)
    h = (torch.randn(B, hs) * 0.3).requires_grad_(True)
    c = (torch.randn(B, hs) * 0.3).requires_grad_(True)
    gh, gc = torch.randn(B, hs), torch.randn(B, hs)
    h1, c1 = cell(x, (h, c))
    ((h1 * gh).sum() + (c1 * gc).sum()).backward()
    pc = nn.LSTMCell(ind, hs).to(dev)
    pc.load_state_dict(cell.state_dict())
    xd, hd, cd = (v.detach().to(dev).requires_grad_(True) for v in (x, h, c))
    packs = ra.packed_lstm(pc.weight_ih, pc.weight_hh)
    g1, g2 = ra.lstm_cell(xd, hd, cd, pc.weight_ih, pc.weight_hh, pc.bias_ih, pc.bias_hh, packs)
    check_close("lstm cell node h", g1, h1, 5e-2)
    check_close("lstm cell node c", g2, c1, 5e-2)
    ((g1 * gh.to(dev)).sum() + (g2 * gc.to(dev)).sum()).backward()
    for name, got, want in (("dx", xd.grad, x.grad), ("dh", hd.grad, h.grad), ("dc", cd.grad, c.grad)):
        check_close("lstm cell node %s rel-L2" % name, _rel(got, want), 0.0, 2e-2)
    _grads_close("lstm cell node", pc, cell, 2e-2)


@pytest.mark.parametrize("reverse", [False, True])
def test_lstm_sequence_node_matches_autograd(dev, reverse):
    """One nn.LSTM direction over a packed batch: outputs, final states, and the gradients of the input, W_ih, W_hh and both
    biases, with incoming gradients on the padded output AND the final states (ragged lengths, batch not a multiple of 16)."""
    from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
    from visitron_amd import rollout_autograd as ra

    torch.manual_seed(4)
    B, S, H, hs = 21, 11, 128, 128
    lens = torch.tensor([9, 9, 8, 8, 7, 7, 6, 5, 5, 5, 4, 4, 3, 3, 3, 2, 2, 1, 1, 1, 1])
    T = int(lens.max())
    lstm = nn.LSTM(H, hs, 1, batch_first=True, bidirectional=True)
    sfx = "_reverse" if reverse else ""
    x = (torch.randn(B, S, H) * 0.5).requires_grad_(True)
    out, (hn, cn) = lstm(pack_padded_sequence(x, lens, batch_first=True))
    out, _ = pad_packed_sequence(out, batch_first=True)
    d = 1 if reverse else 0
    o_d, h_d, c_d = out[:, :, d * hs:(d + 1) * hs], hn[d], cn[d]
    go, gh, gc = torch.randn(B, T, hs), torch.randn(B, hs), torch.randn(B, hs)
    ((o_d * go).sum() + (h_d * gh).sum() + (c_d * gc).sum()).backward()
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    ps = [getattr(lstm, n + sfx).detach().to(dev).requires_grad_(True) for n in names]
    xd = x.detach().to(dev).requires_grad_(True)
    packs = ra.packed_lstm(ps[0], ps[1])
    so, hT, cT = ra.lstm_sequence(xd, ps[0], ps[1], ps[2], ps[3], lens.to(dev, torch.int32), T, reverse, packs)
    tag = "lstm sequence node%s" % (" reverse" if reverse else "")
    check_close(tag + " output", so, o_d, 5e-2)
    check_close(tag + " h_T", hT, h_d, 5e-2)
    check_close(tag + " c_T", cT, c_d, 5e-2)
    ((so * go.to(dev)).sum() + (hT * gh.to(dev)).sum() + (cT * gc.to(dev)).sum()).backward()
    check_close(tag + " dx rel-L2", _rel(xd.grad, x.grad), 0.0, 3e-2)
    assert float(xd.grad[0, T:].abs().max()) == 0.0 and float(xd.grad[-1, 1:].abs().max()) == 0.0   # nothing past a length
    for n, p in zip(names, ps):
        check_close("%s d_%s rel-L2" % (tag, n), _rel(p.grad, getattr(lstm, n + sfx).grad), 0.0, 3e-2)


def test_trunk_level_training_matches_oracle(dev):
    """BertImgModelwithLocationEmbeds in train mode with grad enabled is one autograd node: arbitrary downstream losses on
    sequence_output and pooled_output reach every trunk parameter (text + regions, then the rollout's text-only call with an
    inverted uint8 mask where the pooled output is unused: the pooler then gets no gradient, like in the reference)."""
    from oracle.modeling import PreTrainOscar as OModel
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import make_batch

    cfg = mini_config(num_hidden_layers=3)
    ref, prod = model_pair(OModel, PreTrainOscar, cfg, seed=41, device=dev)
    rt, pt = ref.bert, prod.bert
    rt.train()
    pt.train()
    B, T, R = 3, 20, 7
    b = make_batch(cfg, B, text_len=T, region_len=R, seed=8, with_labels=False)
    g = torch.Generator().manual_seed(2)
    ws, wp = torch.randn(B, T + R, cfg.hidden_size, generator=g), torch.randn(B, cfg.hidden_size, generator=g)
    want = rt(**b)
    ((want[0] * ws).sum() + (want[1] * wp).sum()).backward()
    got = pt(**{k: v.to(dev) for k, v in b.items()})
    assert got[0].requires_grad and got[1].requires_grad
    check_close("trunk-level train sequence_output", got[0], want[0], 5e-2)
    check_close("trunk-level train pooled_output", got[1], want[1], 5e-2)
    ((got[0] * ws.to(dev)).sum() + (got[1] * wp.to(dev)).sum()).backward()
    _grads_close("trunk-level train (text + regions)", pt, rt, 2e-2)
    # a second forward before the first backward would overwrite the saved activations: refused, not silently wrong
    o1 = pt(**{k: v.to(dev) for k, v in b.items()})
    pt(**{k: v.to(dev) for k, v in b.items()})
    with pytest.raises(RuntimeError):
        o1[0].sum().backward()
    # the rollout's call: text only, ~mask of a uint8 padding mask (agent_models.py:267: values 254 / 255, i.e. additive
    # biases of +2.53e6 / +2.54e6), pooled output unused.  In fp32 the reference adds its scores to numbers that resolve
    # 0.25, so its own probabilities are those of scores rounded to that grid; the comparison is therefore against the
    # oracle evaluated in float64, where the same arithmetic is exact (the product centres the mask: modeling._centered_mask).
    import copy

    rt64 = copy.deepcopy(rt).double()
    rt64.zero_grad()
    pt.zero_grad()
    ids = b["input_ids"]
    pad = torch.zeros(B, T, dtype=torch.uint8)
    pad[1, 15:] = 1
    pad[2, 9:] = 1
    w2 = torch.randn(B, T, cfg.hidden_size, generator=g)
    want2 = rt64(ids, attention_mask=~pad)[0]
    (want2 * w2.double()).sum().backward()
    got2 = pt(ids.to(dev), attention_mask=~pad.to(dev))[0]
    check_close("trunk-level train (inverted uint8 mask) sequence_output", got2, want2.float(), 5e-2)
    (got2 * w2.to(dev)).sum().backward()
    errs = _grads_close("trunk-level train (text only, inverted uint8 mask)", pt, rt64, 2e-2)
    assert not any(n.startswith(("pooler.", "img_embedding.", "location_embeds.")) for n in errs)
    # eval() with grad enabled: the inference path's values, differentiable on demand (the engine recomputes on backward)
    rt.eval()
    pt.eval()
    rt.zero_grad()
    pt.zero_grad()
    want = rt(**b)
    ((want[0] * ws).sum() + (want[1] * wp).sum()).backward()
    got = pt(**{k: v.to(dev) for k, v in b.items()})
    assert got[0].requires_grad
    check_close("trunk-level eval+grad sequence_output", got[0], want[0], 5e-2)
    ((got[0] * ws.to(dev)).sum() + (got[1] * wp.to(dev)).sum()).backward()
    _grads_close("trunk-level eval mode with grad enabled", pt, rt, 2e-2)


def _rollout_pair(dev, bidirectional, dropout=0.0):
    from oracle.modeling import PreTrainOscar as OModel
    from oracle.rollout import AttnDecoderLSTM as ODec, OscarEncoder as OEnc
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.rollout import AttnDecoderLSTM, OscarEncoder

    cfg = mini_config(num_hidden_layers=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    ref_m, prod_m = model_pair(OModel, PreTrainOscar, cfg, seed=43, device=dev)
    hs = 256 if bidirectional else 128            # the recurrent kernels serve hidden sizes that are multiples of 128
    enc_h = hs // 2 if bidirectional else hs
    torch.manual_seed(6)
    r_enc = OEnc(None, ref_m.bert, enc_h, hs, dropout, bidirectional=bidirectional)
    p_enc = OscarEncoder(None, prod_m.bert, enc_h, hs, dropout, bidirectional=bidirectional)
    sd = {k: v for k, v in r_enc.state_dict().items() if not k.startswith("bert.")}
    p_enc.load_state_dict(sd, strict=False)
    r_dec = ODec(4, 64, hs, dropout, feature_size=132)
    p_dec = AttnDecoderLSTM(4, 64, hs, dropout, feature_size=132)
    p_dec.load_state_dict(r_dec.state_dict())
    return cfg, r_enc, p_enc.to(dev), r_dec, p_dec.to(dev)


@pytest.mark.parametrize("bidirectional,compact", [(False, False), (True, False), (False, True)])
def test_rollout_training_matches_oracle(dev, bidirectional, compact):
    """Encoder + two teacher-forced decoder steps + cross-entropy over the candidate logits, as agent.py:373-412 builds the
    rollout loss; `loss.backward()` must give every parameter of the trunk, the encoder LSTM, the projections and the decoder
    the oracle's gradient."""
    from visitron_amd.training import _bridge_engine

    cfg, r_enc, p_enc, r_dec, p_dec = _rollout_pair(dev, bidirectional)
    for m in (r_enc, p_enc, r_dec, p_dec):
        m.train()
    eng = _bridge_engine(p_enc.bert)
    eng.compact_min_rows = 0 if compact else 1 << 30     # the trunk node on the rows below the lengths only / on all rows
    B, S, C = 5, 24, 6
    g = torch.Generator().manual_seed(9)
    lens = [24, 20, 13, 13, 6]
    ids = torch.randint(1, cfg.vocab_size, (B, S), generator=g)
    pad = torch.zeros(B, S, dtype=torch.bool)
    for i, n in enumerate(lens):
        pad[i, n:] = True
        ids[i, n:] = 0
    action = torch.randn(B, 4, generator=g)
    feature = torch.randn(B, 36, 132, generator=g).abs() * 0.3
    cand = torch.randn(B, C, 132, generator=g).abs() * 0.3
    target = torch.randint(0, C, (2, B), generator=g)

    def run(enc, dec, to):
        ctx, h_t, c_t = enc(to(ids), lens, to(pad))
        ctx_mask = to(pad)[:, : ctx.shape[1]]
        loss, h1 = 0.0, h_t
        for step in range(2):
            h_t, c_t, logit, h1 = dec(to(action), to(feature), to(cand), h1, h_t, c_t, ctx, ctx_mask)
            loss = loss + nn.functional.cross_entropy(logit, to(target[step]))
        return loss, ctx, logit

    wl, wctx, wlogit = run(r_enc, r_dec, lambda t: t)
    wl.backward()
    gl, gctx, glogit = run(p_enc, p_dec, lambda t: t.to(dev))
    assert (eng.last_layout is not None) == compact and (not compact or eng.last_rows == sum(lens))
    tag = "rollout train%s%s" % (" bidirectional" if bidirectional else "", " compacted" if compact else "")
    check_close(tag + " ctx", gctx, wctx, 5e-2)
    check_close(tag + " logit", glogit, wlogit, 5e-2)
    check_close(tag + " loss", float(gl), float(wl), 5e-2)
    gl.backward()
    _grads_close(tag + " encoder", p_enc, r_enc, 4e-2)
    _grads_close(tag + " decoder", p_dec, r_dec, 4e-2)


def test_rollout_training_runs_with_dropout_and_adam(dev):
    """The reference's loop shape with dropout on (agent.py:497-518): zero_grad, rollout loss, backward, clip, Adam on
    encoder and decoder -- the loss of a fixed batch goes down and the inference path sees the updated weights."""
    cfg, _, p_enc, _, p_dec = _rollout_pair(dev, False, dropout=0.3)
    p_enc.train()
    p_dec.train()
    opt_e = torch.optim.Adam(p_enc.parameters(), lr=1e-3)
    opt_d = torch.optim.Adam(p_dec.parameters(), lr=1e-3)
    B, S, C = 4, 16, 5
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(1, cfg.vocab_size, (B, S), generator=g).to(dev)
    lens = [16, 12, 12, 7]
    pad = torch.zeros(B, S, dtype=torch.bool)
    for i, n in enumerate(lens):
        pad[i, n:] = True
    pad = pad.to(dev)
    action = torch.randn(B, 4, generator=g).to(dev)
    feature = (torch.randn(B, 36, 132, generator=g).abs() * 0.3).to(dev)
    cand = (torch.randn(B, C, 132, generator=g).abs() * 0.3).to(dev)
    target = torch.randint(0, C, (B,), generator=g).to(dev)

    def loss_of():
        ctx, h_t, c_t = p_enc(ids, lens, pad)
        _, _, logit, _ = p_dec(action, feature, cand, h_t, h_t, c_t, ctx, pad[:, : ctx.shape[1]])
        return nn.functional.cross_entropy(logit, target)

    losses = []
    for _ in range(8):
        opt_e.zero_grad()
        opt_d.zero_grad()
        loss = loss_of()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(p_enc.parameters(), 40.0)
        torch.nn.utils.clip_grad_norm_(p_dec.parameters(), 40.0)
        opt_e.step()
        opt_d.step()
        losses.append(float(loss))
    assert all(l == l for l in losses) and min(losses[-3:]) < losses[0], losses
    p_enc.eval()
    p_dec.eval()
    with torch.no_grad():
        e1 = float(loss_of())
    p_enc.train()
    p_dec.train()
    for _ in range(3):
        opt_e.zero_grad()
        opt_d.zero_grad()
        loss_of().backward()
        opt_e.step()
        opt_d.step()
    p_enc.eval()
    p_dec.eval()
    with torch.no_grad():
        e2 = float(loss_of())
    assert e2 != e1                     # the eval path's packed weights follow the optimizer


def test_rollout_training_at_the_reference_sizes(dev):
    """The same comparison at the sizes agent.py:110-125 builds -- base trunk (12 layers, 768 wide), encoder LSTM 768 -> 512,
    decoder hidden 512, action embedding 4 -> 64, view features 2048 + 4 (so the cell's input is 2116 wide: not a multiple
    of 8, padded for the weight-gradient launch), 36 views -- on a short instruction so that the CPU oracle stays in seconds."""
    from oracle.modeling import BertImgModelwithLocationEmbeds as OTrunk
    from oracle.rollout import AttnDecoderLSTM as ODec, OscarEncoder as OEnc
    from visitron_amd.config import BertConfig
    from visitron_amd.modeling import BertImgModelwithLocationEmbeds
    from visitron_amd.rollout import AttnDecoderLSTM, OscarEncoder

    cfg = BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    r_bert, p_bert = model_pair(OTrunk, BertImgModelwithLocationEmbeds, cfg, seed=61, device=dev, weight_std=0.03)
    torch.manual_seed(8)
    r_enc = OEnc(None, r_bert, 512, 512, 0.0)
    p_enc = OscarEncoder(None, p_bert, 512, 512, 0.0)
    p_enc.load_state_dict({k: v for k, v in r_enc.state_dict().items() if not k.startswith("bert.")}, strict=False)
    r_dec = ODec(4, 64, 512, 0.0, feature_size=2052)
    p_dec = AttnDecoderLSTM(4, 64, 512, 0.0, feature_size=2052)
    p_dec.load_state_dict(r_dec.state_dict())
    p_enc, p_dec = p_enc.to(dev), p_dec.to(dev)
    for m in (r_enc, p_enc, r_dec, p_dec):
        m.train()
    B, S, C = 3, 40, 7
    g = torch.Generator().manual_seed(13)
    lens = [40, 33, 18]
    ids = torch.randint(1000, cfg.vocab_size, (B, S), generator=g)
    pad = torch.zeros(B, S, dtype=torch.bool)
    for i, n in enumerate(lens):
        pad[i, n:] = True
        ids[i, n:] = 0
    action = torch.randn(B, 4, generator=g)
    feature = torch.randn(B, 36, 2052, generator=g).abs() * 0.3
    cand = torch.randn(B, C, 2052, generator=g).abs() * 0.3
    target = torch.randint(0, C, (B,), generator=g)

    def run(enc, dec, to):
        ctx, h_t, c_t = enc(to(ids), lens, to(pad))
        _, _, logit, _ = dec(to(action), to(feature), to(cand), h_t, h_t, c_t, ctx, to(pad)[:, : ctx.shape[1]])
        return nn.functional.cross_entropy(logit, to(target)), ctx, logit

    wl, wctx, wlogit = run(r_enc, r_dec, lambda t: t)
    wl.backward()
    gl, gctx, glogit = run(p_enc, p_dec, lambda t: t.to(dev))
    check_close("rollout train base sizes ctx", gctx, wctx, 5e-2)
    check_close("rollout train base sizes logit", glogit, wlogit, 5e-2)
    check_close("rollout train base sizes loss", float(gl), float(wl), 5e-2)
    gl.backward()
    _grads_close("rollout train base sizes encoder", p_enc, r_enc, 5e-2)
    _grads_close("rollout train base sizes decoder", p_dec, r_dec, 5e-2)


@pytest.mark.parametrize("with_head_mask", [False, True])
def test_trunk_level_training_with_output_hidden_states(dev, with_head_mask):
    """output_hidden_states in trunk-level TRAINING (round 6; oscar/modeling_bert.py:146-158, encoder.py:300-303): the model
    returns (sequence_output, pooled_output, all_hidden_states) and a loss on EVERY hidden state -- the embedding output and
    each layer's output -- reaches every parameter: the engine adds the caller's gradient of layer l's output to the running
    gradient in front of layer l's backward (the C loop one layer at a time; with a head_mask the op-by-op sequence)."""
    from oracle.modeling import PreTrainOscar as OModel
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import make_batch

    cfg = mini_config(num_hidden_layers=3, output_hidden_states=True)
    ref, prod = model_pair(OModel, PreTrainOscar, cfg, seed=47, device=dev)
    rt, pt = ref.bert, prod.bert
    rt.train()
    pt.train()
    B, T, R = 3, 20, 7
    b = make_batch(cfg, B, text_len=T, region_len=R, seed=9, with_labels=False)
    g = torch.Generator().manual_seed(3)
    L = cfg.num_hidden_layers
    ws = [torch.randn(B, T + R, cfg.hidden_size, generator=g) for _ in range(L + 2)]
    wp = torch.randn(B, cfg.hidden_size, generator=g)
    hm = None
    if with_head_mask:
        hm = torch.tensor([[1.0, 0.5], [0.0, 1.0], [0.7, 1.3]])

    def loss_of(out, to):
        seq, pooled, hidden = out
        assert len(hidden) == L + 1
        total = (seq * to(ws[L + 1])).sum() + (pooled * to(wp)).sum()
        for l, h in enumerate(hidden):
            total = total + (h * to(ws[l])).sum()
        return total

    want = rt(head_mask=hm, **b)
    loss_of(want, lambda t: t).backward()
    got = pt(head_mask=None if hm is None else hm.to(dev), **{k: v.to(dev) for k, v in b.items()})
    assert len(got) == 3 and all(h.requires_grad for h in got[2])
    tag = "trunk-level train hidden states%s" % (" head_mask" if with_head_mask else "")
    check_close(tag + " sequence_output", got[0], want[0], 5e-2)
    for l in range(L + 1):
        check_close(tag + " hidden[%d]" % l, got[2][l], want[2][l], 5e-2)
    loss_of(got, lambda t: t.to(dev)).backward()
    _grads_close(tag, pt, rt, 2e-2)
    # under torch.no_grad() (train mode, dropout off here): the same tuple without a graph
    with torch.no_grad():
        ng = pt(**{k: v.to(dev) for k, v in b.items()})
    assert len(ng) == 3 and len(ng[2]) == L + 1 and not ng[2][0].requires_grad


def test_trunk_level_training_with_output_attentions(dev):
    """output_attentions in trunk-level TRAINING (round 6; oscar/modeling_bert.py:62-79, 160-167): the fourth output is the
    tuple of per-layer probabilities after dropout and head_mask.  Without dropout against the oracle's train-mode forward
    (with a head_mask); with attention dropout the first layer's values are either zero or the eval-mode probability over
    1 - p_eff, dropped at the rate p_eff (the keep words the forward wrote); a loss on the other outputs still trains."""
    from oracle.modeling import PreTrainOscar as OModel
    from visitron_amd import ops
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import make_batch

    cfg = mini_config(num_hidden_layers=2, output_hidden_states=True, output_attentions=True)
    ref, prod = model_pair(OModel, PreTrainOscar, cfg, seed=49, device=dev)
    rt, pt = ref.bert, prod.bert
    rt.train()
    pt.train()
    B, T, R = 3, 20, 7
    S = T + R
    b = make_batch(cfg, B, text_len=T, region_len=R, seed=10, with_labels=False)
    bd = {k: v.to(dev) for k, v in b.items()}
    hm = torch.tensor([[1.0, 0.5], [0.0, 1.3]])
    want = rt(head_mask=hm, **b)
    got = pt(head_mask=hm.to(dev), **bd)
    assert len(got) == 4 and len(got[3]) == cfg.num_hidden_layers and not got[3][0].requires_grad
    for l in range(cfg.num_hidden_layers):
        assert tuple(got[3][l].shape) == (B, cfg.num_attention_heads, S, S)
        check_close("trunk-level train attentions[%d]" % l, got[3][l], want[3][l], 5e-3)
    (got[0].sum() + got[1].sum()).backward()          # the differentiable outputs still reach the parameters
    assert pt.encoder.layer[0].attention.self.query.weight.grad is not None
    # attention dropout: layer 0 of a model whose hidden dropout is off
    cfg2 = mini_config(num_hidden_layers=2, output_attentions=True, attention_probs_dropout_prob=0.25)
    _, prod2 = model_pair(OModel, PreTrainOscar, cfg2, seed=49, device=dev)
    p2 = prod2.bert
    p2.eval()
    with torch.no_grad():
        base = p2(**bd)[2][0]
    p2.train()
    with torch.no_grad():
        dropped = p2(**bd)[2][0]
    pe = ops.attn_drop_p(0.25)
    real = bd["attention_mask"].bool()[:, None, None, :].expand_as(base) & (base > 1e-6)
    kept = dropped != 0
    frac = 1.0 - float(kept[real].float().mean())
    assert abs(frac - pe) < 0.03, (frac, pe)
    err = (dropped - base / (1.0 - pe))[kept & real].abs().max()
    assert float(err) <= 5e-3, float(err)
