"""GPU, round 5: the persistent, software-pipelined attention backward (attention_bwd_d64_w16p: one workgroup per compute
unit walks its (batch, head) pairs, the next slots / keys / V fragments land under the current arithmetic) against the
one-pair-per-workgroup 16-wave kernel it restates -- same tiles, same order of the sums over keys and queries; delta (an fp32
MFMA sum here) and the exponent's bias (one fma here) are rounded differently, so the packed bf16 dQ | dK | dV agree to a last
bit of bf16 on a few elements, not bitwise -- on padded and compacted batches, with and without dropout keep words, fewer and more pairs than compute
units, one-iteration and four-iteration sequences next to each other (oscar/modeling_bert.py:52-68 under loss.backward())."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _same(a, b):
    """bf16 outputs of the two kernels: equal up to one bf16 rounding step on isolated elements."""
    a, b = a.float(), b.float()
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    scale = float(a.abs().max()) + 1e-30
    assert float((a - b).abs().max()) <= 1.0 / 64 * scale, float((a - b).abs().max()) / scale     # two bf16 steps at full scale
    assert float((a - b).norm() / (a.norm() + 1e-30)) <= 2e-3, float((a - b).norm() / (a.norm() + 1e-30))


def _run(ops, waves, fn):
    ops.set_attn_bwd_waves(waves)
    try:
        out = fn()
        torch.cuda.synchronize()
    finally:
        ops.set_attn_bwd_waves(0)
    return out


@pytest.mark.parametrize("B,S,nh,p", [(3, 228, 2, 0.1), (40, 228, 12, 0.1), (2, 256, 3, 0.0), (7, 64, 2, 0.3), (5, 65, 1, 0.1),
                                      (9, 1, 2, 0.1), (4, 129, 3, 0.0), (64, 100, 12, 0.1)])
def test_persistent_attention_backward_equals_the_16_wave_kernel_padded(dev, B, S, nh, p):
    from visitron_amd import ops

    g = torch.Generator().manual_seed(B * 131 + S)
    H = nh * 64
    qkv = (torch.randn(B * S, 3 * H, generator=g) * 0.8).to(dev, BF16)
    dctx = torch.randn(B * S, H, generator=g).to(dev, BF16)
    mask = (torch.rand(B, S, generator=g) > 0.2).float()
    mask[:, 0] = 1.0
    mask = mask.to(dev)
    drop = (p, 77, 3) if p > 0 else ops.NO_DROP
    words = torch.zeros(ops.keep_words(B, nh, S), dtype=torch.int32, device=dev) if p > 0 else None
    lse = torch.zeros(B, nh, S, device=dev)
    ctx = ops.attention_fwd(qkv, B, S, nh, mask=mask, lse=lse, drop=drop, keep_bits=words)
    a = _run(ops, 16, lambda: ops.attention_bwd(qkv, dctx, ctx, lse, B, S, nh, mask=mask, drop=drop, keep_bits=words))
    b = _run(ops, 17, lambda: ops.attention_bwd(qkv, dctx, ctx, lse, B, S, nh, mask=mask, drop=drop, keep_bits=words))
    _same(a, b)
    c = _run(ops, 17, lambda: ops.attention_bwd(qkv, dctx, ctx, lse, B, S, nh, mask=mask, drop=drop, keep_bits=words))
    assert torch.equal(b, c)                      # and reproducible from launch to launch


@pytest.mark.parametrize("B,S,nh,p", [(6, 228, 2, 0.1), (48, 228, 12, 0.1), (5, 256, 2, 0.0), (33, 90, 12, 0.2)])
def test_persistent_attention_backward_equals_the_16_wave_kernel_compacted(dev, B, S, nh, p):
    """Per-sequence lengths from 1 to S: a workgroup's consecutive pairs have one to four iterations, odd lengths, single keys."""
    from visitron_amd import ops

    g = torch.Generator().manual_seed(B * 17 + S)
    H = nh * 64
    lens = torch.randint(1, S + 1, (B,), generator=g)
    lens[0], lens[1], lens[-1] = S, 1, 63
    keep = torch.arange(S)[None, :] < lens[:, None]
    seq = ops.SeqLayout(keep.to(dev))
    qkv = (torch.randn(seq.rows, 3 * H, generator=g) * 0.8).to(dev, BF16)
    dctx = torch.randn(seq.rows, H, generator=g).to(dev, BF16)
    drop = (p, 99, 5) if p > 0 else ops.NO_DROP
    words = torch.zeros(ops.keep_words(B, nh, S), dtype=torch.int32, device=dev) if p > 0 else None
    lse = torch.zeros(B, nh, S, device=dev)
    ctx = ops.attention_fwd(qkv, B, S, nh, lse=lse, drop=drop, seq=seq, keep_bits=words)
    a = _run(ops, 16, lambda: ops.attention_bwd(qkv, dctx, ctx, lse, B, S, nh, drop=drop, seq=seq, keep_bits=words))
    b = _run(ops, 17, lambda: ops.attention_bwd(qkv, dctx, ctx, lse, B, S, nh, drop=drop, seq=seq, keep_bits=words))
    _same(a, b)


@pytest.mark.parametrize("bits", [16, 8])
def test_attention_dropout_probabilities_the_quantisation_cannot_serve(dev, bits):
    """A p that rounds to 1 at the resolution in force (steps of 2^-bits: 16 by default since round 6, 8 before) is refused by
    the library (VT_ERR_UNSUPPORTED) instead of returning 0 * inf; a p below half a step drops one key per 2^bits instead of
    none (oscar/modeling_bert.py:62 nn.Dropout(attention_probs_dropout_prob))."""
    from visitron_amd import ops

    before = ops.attn_dropout_bits()
    ops.set_attn_dropout_bits(bits)
    try:
        steps = float(1 << bits)
        B, S, nh = 1, 64, 1
        qkv = torch.randn(B * S, 3 * 64, device=dev).to(BF16)
        lse = torch.zeros(B, nh, S, device=dev)
        with pytest.raises(RuntimeError):
            ops.attention_fwd(qkv, B, S, nh, lse=lse, drop=(1.0 - 0.25 / steps, 1, 0))
        ops.attention_fwd(qkv, B, S, nh, lse=lse, drop=(1.0 - 1.0 / steps, 1, 0))           # the last step below 1 is served
        torch.cuda.synchronize()
        m = torch.stack([ops.attn_dropout_mask(256, (0.1 / steps, 7, 0), h, device=dev) for h in range(64)]).float()
        kept = float(m.mean())
        assert 1.0 - 3.0 / steps < kept < 1.0 - 0.3 / steps, kept          # ~ one key in 2^bits dropped: the dropout is on
    finally:
        ops.set_attn_dropout_bits(before)


def test_persistent_attention_backward_on_twenty_random_geometries(dev):
    """Random (batch, heads, padded length, per-sequence lengths, dropout on / off, mask density): workgroups with one pair
    and with dozens, sequences of 1 .. 256 rows side by side, every iteration count from 1 to 4."""
    from visitron_amd import ops

    g = torch.Generator().manual_seed(2025)
    for case in range(20):
        B = int(torch.randint(1, 70, (1,), generator=g))
        nh = int(torch.randint(1, 13, (1,), generator=g))
        S = int(torch.randint(1, 257, (1,), generator=g))
        p = 0.15 if case % 2 else 0.0
        compact = case % 3 == 0
        H = nh * 64
        drop = (p, 1000 + case, case) if p > 0 else ops.NO_DROP
        words = torch.zeros(ops.keep_words(B, nh, S), dtype=torch.int32, device=dev) if p > 0 else None
        lse = torch.zeros(B, nh, S, device=dev)
        if compact:
            lens = torch.randint(1, S + 1, (B,), generator=g)
            keep = torch.arange(S)[None, :] < lens[:, None]
            seq = ops.SeqLayout(keep.to(dev))
            rows, kw = seq.rows, dict(seq=seq)
        else:
            mask = (torch.rand(B, S, generator=g) > float(torch.rand(1, generator=g)) * 0.6).float()
            mask[:, 0] = 1.0
            rows, kw = B * S, dict(mask=mask.to(dev))
        qkv = (torch.randn(rows, 3 * H, generator=g) * 0.8).to(dev, BF16)
        dctx = torch.randn(rows, H, generator=g).to(dev, BF16)
        ctx = ops.attention_fwd(qkv, B, S, nh, lse=lse, drop=drop, keep_bits=words, **kw)
        a = _run(ops, 16, lambda: ops.attention_bwd(qkv, dctx, ctx, lse, B, S, nh, drop=drop, keep_bits=words, **kw))
        b = _run(ops, 17, lambda: ops.attention_bwd(qkv, dctx, ctx, lse, B, S, nh, drop=drop, keep_bits=words, **kw))
        try:
            _same(a, b)
        except AssertionError as e:
            raise AssertionError("case %d: B=%d nh=%d S=%d p=%.2f compact=%s: %s" % (case, B, nh, S, p, compact, e))


def test_persistent_attention_backward_stands_down_for_more_batches_than_its_metadata_table(dev):
    """B > 1 024 sequences (the per-batch metadata the kernel keeps in LDS): the dispatcher takes the one-pair-per-workgroup
    kernel; same answer either way."""
    from visitron_amd import ops

    g = torch.Generator().manual_seed(5)
    B, S, nh = 1030, 9, 1
    qkv = (torch.randn(B * S, 3 * 64, generator=g) * 0.8).to(dev, BF16)
    dctx = torch.randn(B * S, 64, generator=g).to(dev, BF16)
    lse = torch.zeros(B, nh, S, device=dev)
    ctx = ops.attention_fwd(qkv, B, S, nh, lse=lse)
    a = _run(ops, 16, lambda: ops.attention_bwd(qkv, dctx, ctx, lse, B, S, nh))
    b = _run(ops, 17, lambda: ops.attention_bwd(qkv, dctx, ctx, lse, B, S, nh))
    assert torch.equal(a, b)          # the same kernel ran


@pytest.mark.parametrize("M,H", [(4099, 768), (9, 256), (130, 512), (8201, 1024), (1, 768)])
@pytest.mark.parametrize("xdt", [torch.float16, torch.bfloat16])
def test_full_width_layernorm_forward_and_backward_match_fp32(dev, M, H, xdt):
    """layernorm_rows_full / layernorm_bwd_rows_full (H = 768, 512, 1 024, 256: every lane owns 8 C8 + 4 C4 columns, rows walked
    over a fixed grid with the next row prefetched, reductions on the vector ALU): forward with both outputs and the row
    statistics, backward with the dropout-masked second copy, against fp32 torch on the same rounded inputs; M covers one
    row, a ragged tail and several trips of the grid-stride loop."""
    from visitron_amd import ops

    g = torch.Generator().manual_seed(M + H)
    x = (torch.randn(M, H, generator=g) * 2.0 + 0.3).to(xdt)
    dy = torch.randn(M, H, generator=g).to(BF16)
    gamma = 1 + 0.1 * torch.randn(H, generator=g)
    beta = 0.1 * torch.randn(H, generator=g)
    xf = x.float().requires_grad_(True)
    gf, bf = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    want = torch.nn.functional.layer_norm(xf, (H,), gf, bf, 1e-12)
    want.backward(dy.float())
    mean, rstd = torch.zeros(M, device=dev), torch.zeros(M, device=dev)
    if xdt == torch.float16:
        yh = torch.empty(M, H, dtype=torch.float16, device=dev)
        y = ops.layernorm(x.to(dev), gamma.to(dev), beta.to(dev), 1e-12, mean=mean, rstd=rstd, out_h=yh)
        assert float((yh.float().cpu() - want.detach()).abs().max()) <= 2.0 ** -9 * (1 + float(want.abs().max()))
        assert torch.equal(yh.float().to(BF16), y) or float((yh.float() - y.float()).abs().max()) <= 2.0 ** -7 * (1 + float(want.abs().max()))
    else:
        y = ops.layernorm(x.to(dev), gamma.to(dev), beta.to(dev), 1e-12, mean=mean, rstd=rstd)
    assert float((y.float().cpu() - want.detach()).abs().max()) <= 2.0 ** -7 * (1 + float(want.abs().max()))
    u = x.float().mean(-1)
    var = (x.float() - u[:, None]).pow(2).mean(-1)
    assert float((mean.cpu() - u).abs().max()) < 1e-4
    assert float((rstd.cpu() - 1 / torch.sqrt(var + 1e-12)).abs().max()) < 1e-3
    drop = (0.1, 77, 5)
    keep = ops.dropout_mask(M * H, drop, device=dev).view(M, H).cpu().float()
    dgam, dbet = torch.full((H,), 3.0, device=dev), torch.full((H,), 3.0, device=dev)
    dxd = torch.empty(M, H, dtype=BF16, device=dev)
    dx = ops.layernorm_bwd(x.to(dev), dy.to(dev), gamma.to(dev), 1e-12, dgam, dbet, dx_dropped=dxd, drop=drop)
    torch.cuda.synchronize()
    scale = 1 + float(xf.grad.abs().max())
    assert float((dx.float().cpu() - xf.grad).abs().max()) <= 2.0 ** -7 * scale
    # the masked copy: exactly dx (before its bf16 rounding) * keep / (1 - p), so within one bf16 rounding of dx * keep / 0.9
    assert float((dxd.float().cpu() - xf.grad * keep / 0.9).abs().max()) <= 2.0 ** -7 * scale / 0.9
    assert bool(((dxd.float().cpu() == 0) | (keep > 0)).all())
    assert float((dgam.cpu() - gf.grad).abs().max()) <= 1e-3 * (1 + float(gf.grad.abs().max())) * max(1.0, M / 1000)
    assert float((dbet.cpu() - bf.grad).abs().max()) <= 1e-3 * (1 + float(bf.grad.abs().max())) * max(1.0, M / 1000)
    ops.layernorm_bwd(x.to(dev), dy.to(dev), gamma.to(dev), 1e-12, dgam, dbet, accumulate=True)
    assert float((dgam.cpu() - 2 * gf.grad).abs().max()) <= 2e-3 * (1 + float(gf.grad.abs().max())) * max(1.0, M / 1000)


@pytest.mark.parametrize("compact", [False, True])
def test_adamw_under_the_backward_gives_the_same_weights_bit_for_bit(dev, monkeypatch, compact):
    """VT_OVERLAP_ADAMW=1 (one rank): the fused AdamW of a parameter range runs on a side stream as soon as that range's
    gradients are final (heads before the encoder backward, encoder layers in chunks of three under the earlier layers'
    backward, embeddings / region projection last).  Same kernels on the same numbers in another order of launches: after
    three steps every parameter, both Adam moments and the bf16 mirror are identical to the plain step's, and so are the
    returned losses."""
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import deterministic_state_dict, make_batch
    from visitron_amd.training import PretrainEngine

    cfg = mini_config()
    cfg.num_hidden_layers = 5          # chunks of 3 + 2
    b = make_batch(cfg, 6, text_len=24, region_len=10, seed=4)
    bd = {k: v.to(dev) for k, v in b.items()}
    runs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("VT_OVERLAP_ADAMW", flag)
        m = PreTrainOscar(cfg)
        m.load_state_dict(deterministic_state_dict(m, seed=3, weight_std=0.05))
        m.tie_weights()
        m = m.to(dev).train()
        eng = PretrainEngine(m, lr=2e-3, weight_decay=0.05, schedule="constant", warmup_steps=0)
        assert eng.overlap_adamw == (flag == "1")
        if compact:
            eng.compact_min_rows = 1          # the real-rows-only step of large batches, on this small one
        losses = [[float(x) for x in eng.train_step(bd)[:4]] for _ in range(3)]
        torch.cuda.synchronize()
        f = eng.flat
        runs.append((losses, f.p.clone(), f.m.clone(), f.v.clone(), f.mirror.clone(), eng.step_count, eng.sched_step))
    a, c = runs
    assert a[0] == c[0], (a[0], c[0])
    for i in (1, 2, 3, 4):
        assert torch.equal(a[i], c[i]), i
    assert a[5:] == c[5:] == (3, 3)
    assert not torch.equal(a[1], torch.zeros_like(a[1])) and float(a[3].abs().max()) > 0   # the steps did update


@pytest.mark.parametrize("variant", [-1, 1, 14, 11, 15, 16, 18, 19, 20, 21, 22, 23, 9])
@pytest.mark.parametrize("M,N,K,p", [(1000, 768, 768, 0.1), (777, 768, 3072, 0.0), (300, 208, 128, 0.1), (4100, 768, 768, 0.0),
                                     (600, 832, 128, 0.1)])
def test_linear_with_a_residual_that_is_a_layernorm_never_written(dev, M, N, K, p, variant):
    """C = dropout(A W^T + b) + LayerNorm(v), LayerNorm(v) rebuilt in the epilogue from the fp16 sum v and the saved row
    statistics (vt_linear_lnres_bf16, GemmArgs::r_mean): every kernel variant's epilogue (persistent and one-tile fast
    epilogues with the vectors parked in LDS, the register epilogues, the scalar tail at N = 208; variant 9's bf16-only
    grouped epilogue is substituted by the library), fp16 and bf16 outputs, against fp32 on the same operands."""
    from visitron_amd import ops

    F16 = torch.float16
    g = torch.Generator().manual_seed(M + N + K + variant + 7)
    a = (torch.randn(M, K, generator=g)).to(BF16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF16)
    b = torch.randn(N, generator=g) * 0.1
    v = (torch.randn(M, N, generator=g) * 2.0 + 0.5).to(F16)
    gamma, beta = 1 + 0.2 * torch.randn(N, generator=g), 0.3 * torch.randn(N, generator=g)
    mean = v.float().mean(-1)
    rstd = 1.0 / torch.sqrt((v.float() - mean[:, None]).pow(2).mean(-1) + 1e-12)
    ln = (v.float() - mean[:, None]) * rstd[:, None] * gamma + beta
    drop = (p, 91, 3) if p > 0 else ops.NO_DROP
    dense = a.float() @ w.float().t() + b
    if p > 0:
        keep = ops.dropout_mask(M * N, drop, device=dev).view(M, N).cpu().float()
        dense = dense * keep / (1.0 - p)
    want = dense + ln
    ops.set_gemm_variant(variant)
    try:
        kw = dict(residual=v.to(dev), residual_ln=(mean.to(dev), rstd.to(dev), gamma.to(dev), beta.to(dev)), drop=drop)
        out = torch.empty((M, N), dtype=F16, device=dev)
        ops.linear(a.to(dev), w.to(dev), b.to(dev), out=out, **kw)
        out2 = ops.linear(a.to(dev), w.to(dev), b.to(dev), **kw)
        torch.cuda.synchronize()
    finally:
        ops.set_gemm_variant(-1)
    scale = float(want.abs().max())
    assert float((out.float().cpu() - want).abs().max()) <= scale * 2.0 ** -10
    assert out2.dtype == BF16 and float((out2.float().cpu() - want).abs().max()) <= scale * 2.0 ** -7


def test_layernorm_residual_mode_is_the_default_training_layer_and_matches_the_two_output_layer(dev, monkeypatch):
    """ops.LN_RESIDUAL (vt_layer_acts::ln_residual_mode = 1): the engine's layers write no fp16 copy of a LayerNorm output.
    Against the round-4 layer (VT_LN_RESIDUAL=0: two-output LayerNorm, fp16 copy read by the residual add) on the same
    weights and batch: the hidden states agree to the fp16 rounding the old layer applies to the residual branch, the losses
    to 2e-3, and the C loop equals the op-by-op sequence (ops.profiling) bit for bit in the new mode."""
    import importlib

    from visitron_amd import ops
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import deterministic_state_dict, make_batch
    from visitron_amd.training import PretrainEngine

    assert ops.LN_RESIDUAL and ops.F16_STREAM
    cfg = mini_config()
    cfg.hidden_dropout_prob, cfg.attention_probs_dropout_prob = 0.1, 0.1
    b = make_batch(cfg, 5, text_len=20, region_len=9, seed=11)
    bd = {k: v.to(dev) for k, v in b.items()}

    def run(ln_residual, unrolled=False):
        monkeypatch.setattr(ops, "LN_RESIDUAL", ln_residual)
        m = PreTrainOscar(cfg)
        m.load_state_dict(deterministic_state_dict(m, seed=5, weight_std=0.05))
        m.tie_weights()
        m = m.to(dev).train()
        eng = PretrainEngine(m)
        if unrolled:
            monkeypatch.setattr(ops, "profiling", lambda: True)
        out = eng.forward_backward(bd)
        if unrolled:
            monkeypatch.undo()
            monkeypatch.setattr(ops, "LN_RESIDUAL", ln_residual)
        torch.cuda.synchronize()
        bufs = eng._buffers(5, 29)
        assert bufs.ln_residual == ln_residual and (bufs.ln_h is None) == ln_residual
        # (the rows the step computed: since round 6 small batches run on the real rows only, and the buffer's tail is not written)
        last = bufs.layers[-1]["out"][:eng.last_rows].float().clone()
        return [float(x) for x in out[:4]], last, eng.flat.g.clone()

    l_new, h_new, g_new = run(True)
    l_old, h_old, g_old = run(False)
    l_unr, h_unr, g_unr = run(True, unrolled=True)
    assert torch.equal(h_new, h_unr) and l_new == l_unr and torch.equal(g_new, g_unr)
    assert float((h_new - h_old).abs().max()) <= 2.0 ** -6 * (1 + float(h_old.abs().max()))
    for x, y in zip(l_new, l_old):
        assert abs(x - y) <= 2e-3 * (1 + abs(y)), (l_new, l_old)
    rel = float((g_new - g_old).norm() / g_old.norm())
    assert rel <= 2e-2, rel
