"""Parity of each HIP kernel (through the C ABI) against plain fp32 torch math on the same
bf16-rounded inputs.  Tolerances: bf16 outputs carry 2^-8 relative rounding."""
import math

import pytest
import torch

from helpers import bf16_round, maxabs

pytestmark = pytest.mark.gpu

BF16 = torch.bfloat16


def _rand(shape, g, scale=1.0):
    return (torch.randn(shape, generator=g) * scale)


def _gelu(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


@pytest.mark.parametrize(
    "M,N,K,act,res,f32out",
    [
        (256, 256, 128, 0, False, False),
        (456, 768, 768, 0, True, False),      # cfg1 rows (B=2 x 228): M tail
        (456, 2304, 768, 0, False, False),    # packed qkv
        (300, 3072, 768, 1, False, False),    # FFN up + GELU
        (300, 768, 3072, 0, True, False),     # FFN down + residual
        (130, 36, 128, 0, False, True),       # action head: N tail, fp32 out
        (200, 1601, 768, 0, False, True),     # token head: N tail not multiple of 16
        (64, 768, 768, 2, False, True),       # pooler tanh
        (128, 30522, 768, 0, False, True),    # MLM decoder width
    ],
)
def test_linear_matches_fp32(dev, M, N, K, act, res, f32out):
    from visitron_amd import ops

    g = torch.Generator().manual_seed(M * 7 + N)
    a = bf16_round(_rand((M, K), g))
    w = bf16_round(_rand((N, K), g, 0.05))
    b = _rand((N,), g, 0.1)
    r = bf16_round(_rand((M, N), g)) if res else None
    want = a @ w.t() + b
    if act == 1:
        want = _gelu(want)
    elif act == 2:
        want = torch.tanh(want)
    if res:
        want = want + r
    ldc = (N + 7) // 8 * 8
    out = torch.full((M, ldc), 7.0, dtype=torch.float32 if f32out else BF16, device=dev)
    ops.linear(a.to(dev, BF16), w.to(dev, BF16), b.to(dev), residual=None if r is None else r.to(dev, BF16),
               act=act, out=out, out_f32=f32out)
    torch.cuda.synchronize()
    got = out[:, :N].float().cpu()
    tol = 2e-3 if f32out else 1.5e-2
    err = (got - want).abs() / (1.0 + want.abs())
    assert float(err.max()) < tol, float(err.max())
    if ldc > N:  # padding columns untouched
        assert bool((out[:, N:].float() == 7.0).all())


@pytest.mark.parametrize(
    "M,N,K,act,res,f32out",
    [
        (512, 512, 128, 0, False, False),     # 2x2 tiles, two K-steps
        (256, 256, 64, 0, False, True),       # a single K-step (prologue-only pipeline)
        (700, 768, 192, 0, True, False),      # M tail, odd number of K-steps
        (456, 2304, 768, 0, False, False),
        (300, 3072, 768, 1, False, False),    # GELU epilogue
        (1000, 768, 3072, 0, True, False),    # long K
        (130, 36, 128, 0, False, True),       # N tail inside the first 64-column block
        (520, 1601, 768, 0, False, True),     # N tail not a multiple of 16
        (64, 768, 768, 2, False, True),       # tanh
        (9000, 1024, 256, 0, True, False),    # 144 tiles: several tiles per workgroup on some XCDs only
        (70000, 768, 64, 0, False, False),    # 822 tiles of one K-step each (every step crosses a tile boundary)
    ],
)
@pytest.mark.parametrize("variant", [15, 16, 18, 19, 20, 21, 22, 23])
def test_linear_variant_256x256_agpr(dev, M, N, K, act, res, f32out, variant):
    """The 256x256-tile kernels with 128x128 wave tiles (15: one tile per workgroup, 16: persistent with the K
    pipeline running across tiles, 18 / 19: the persistent kernel on 224- / 192-row tiles) on their own: tails, odd
    K-step counts, epilogues."""
    from visitron_amd import ops

    ops.set_gemm_variant(variant)
    try:
        test_linear_matches_fp32(dev, M, N, K, act, res, f32out)
        # row remap + dropout + saved pre-activation through the same kernel
        g = torch.Generator().manual_seed(M + K)
        a, w = bf16_round(_rand((M, K), g)), bf16_round(_rand((N, K), g, 0.05))
        if not f32out and act == 0:
            drop = (0.2, 11, 3)
            keep = ops.dropout_mask(M * N, drop, device=dev).view(M, N).float().cpu()
            got = ops.linear(a.to(dev, BF16), w.to(dev, BF16), drop=drop)
            torch.cuda.synchronize()
            want = (a @ w.t()) * keep / 0.8
            assert maxabs(got, want) < 2e-2 * (1 + float(want.abs().max()))
    finally:
        ops.set_gemm_variant(-1)


@pytest.mark.parametrize("N,K,act,res,pre", [(768, 768, 0, True, False), (3072, 768, 1, False, True), (3072, 768, 3, True, False),
                                             (768, 3072, 0, True, False)])
@pytest.mark.parametrize("V", [16, 18, 19, 20, 21])
def test_linear_persistent_kernel_full_size_against_plain_kernel(dev, N, K, act, res, pre, V):
    """BASELINE-size rows (B = 256 x 228 tokens): the persistent kernel's hand-counted vmcnt waits (residual ring,
    bias through LDS-DMA, stores in flight) under a full chip's memory traffic.  The 128x128-tile kernel, whose waits
    the compiler places, accumulates in the same order: outputs must agree to a bf16 ulp of the activation."""
    from visitron_amd import ops

    M = 58368
    g = torch.Generator().manual_seed(N + K)
    a = _rand((M, K), g).to(dev, BF16)
    w = _rand((N, K), g, 0.05).to(dev, BF16)
    b = _rand((N,), g, 0.1).to(dev)
    r = _rand((M, N), g).to(dev, BF16) if res else None
    outs = []
    for variant in (1, V):
        ops.set_gemm_variant(variant)
        try:
            p_out = torch.empty((M, N), dtype=BF16, device=dev) if pre else None
            y = ops.linear(a, w, b, residual=r, act=act, pre_act_out=p_out)
            torch.cuda.synchronize()
            outs.append((y.float(), None if p_out is None else p_out.float()))
        finally:
            ops.set_gemm_variant(-1)
    (y1, p1), (y2, p2) = outs
    scale = float(y1.abs().max())
    assert float((y1 - y2).abs().max()) <= scale * 2 ** -7
    # rounding ties only
    assert float(((y1 - y2).abs() > 0).float().mean()) < 0.02
    if pre:
        assert float((p1 - p2).abs().max()) <= float(p1.abs().max()) * 2 ** -7


def test_linear_asymmetric_identity(dev):
    """A = I against an ASYMMETRIC W catches a transposed or permuted accumulator write-out."""
    from visitron_amd import ops

    K = 128
    a = torch.eye(K)
    w = (torch.arange(256 * K, dtype=torch.float32).reshape(256, K) % 251) / 64.0 - 2.0
    w = bf16_round(w)
    got = ops.linear(a.to(dev, BF16), w.to(dev, BF16), out_f32=True).cpu()
    assert torch.equal(got, w.t().contiguous())


def test_linear_row_remap(dev):
    """grp_rows/grp_stride: GEMM row b*R+r lands on buffer row b*S+T+r (replaces torch.cat)."""
    from visitron_amd import ops

    B, T, R, H, K = 3, 5, 7, 128, 64
    S = T + R
    g = torch.Generator().manual_seed(5)
    a = bf16_round(_rand((B * R, K), g))
    w = bf16_round(_rand((H, K), g, 0.1))
    x = torch.zeros((B * S, H), dtype=BF16, device=dev)
    ops.linear(a.to(dev, BF16), w.to(dev, BF16), out=x[T:], ldc=H, grp_rows=R, grp_stride=S)
    got = x.float().cpu().view(B, S, H)
    want = (a @ w.t()).view(B, R, H)
    assert float(got[:, :T].abs().max()) == 0.0
    assert maxabs(got[:, T:], bf16_round(want)) < 2e-2


def test_linear_rejects_bad_arguments(dev):
    from visitron_amd import ops

    a = torch.zeros((8, 100), dtype=BF16, device=dev)  # K not a multiple of 64
    w = torch.zeros((16, 100), dtype=BF16, device=dev)
    with pytest.raises(RuntimeError):
        ops.linear(a, w)
    with pytest.raises(RuntimeError):
        ops.linear(torch.zeros((8, 64), dtype=BF16), torch.zeros((16, 64), dtype=BF16))  # CPU tensors


def _attention_ref(q, k, v, add_mask):
    # oscar/modeling_bert.py:52-68 in fp32
    s = q @ k.transpose(-1, -2) / math.sqrt(q.shape[-1]) + add_mask[:, None, None, :]
    return torch.softmax(s, dim=-1) @ v


@pytest.mark.parametrize("B,S,nh", [(2, 228, 12), (3, 37, 2), (1, 300, 4), (2, 656, 2), (1, 128, 1), (2, 1, 2)])
def test_attention_matches_fp32(dev, B, S, nh):
    from visitron_amd import ops

    g = torch.Generator().manual_seed(S)
    H = nh * 64
    qkv = bf16_round(_rand((B * S, 3 * H), g, 1.5))
    mask = (torch.rand(B, S, generator=g) > 0.25).float()
    mask[:, 0] = 1.0
    if B > 1:
        mask[1] = 0.0  # a fully masked sequence: uniform attention, as the -10000 arithmetic gives
    t = qkv.view(B, S, 3, nh, 64).permute(2, 0, 3, 1, 4)
    want = _attention_ref(t[0], t[1], t[2], (1.0 - mask) * -10000.0).permute(0, 2, 1, 3).reshape(B * S, H)
    lse = torch.zeros((B, nh, S), dtype=torch.float32, device=dev)
    got = ops.attention_fwd(qkv.to(dev, BF16), B, S, nh, mask=mask.to(dev), lse=lse)
    torch.cuda.synchronize()
    assert maxabs(got, want) < 3e-2
    # additive-mask entry (what CaptionBertEncoder.forward receives) gives the same numbers
    got2 = ops.attention_fwd(qkv.to(dev, BF16), B, S, nh, mask=((1.0 - mask) * -10000.0).to(dev), mask_additive=True)
    assert torch.equal(got, got2)
    # log-sum-exp saved for backward
    s = (t[0] @ t[1].transpose(-1, -2)) / 8.0 + ((1.0 - mask) * -10000.0)[:, None, None, :]
    want_lse = torch.logsumexp(s, dim=-1)
    assert maxabs(lse, want_lse) < 2e-2 * (1 + float(want_lse.abs().max()) * 1e-3)


def test_attention_known_answers(dev):
    """Q = 0 -> context = masked mean of V; a masked key contributes exactly 0."""
    from visitron_amd import ops

    B, S, nh = 1, 70, 1
    g = torch.Generator().manual_seed(1)
    qkv = torch.zeros((S, 192))
    qkv[:, 64:128] = bf16_round(_rand((S, 64), g))
    qkv[:, 128:] = bf16_round(_rand((S, 64), g))
    mask = torch.ones(1, S)
    mask[0, 40:] = 0
    qkv[50, 128:] = 1000.0  # masked key with a huge value must not leak
    got = ops.attention_fwd(qkv.to(dev, BF16), B, S, nh, mask=mask.to(dev)).float().cpu()
    want = qkv[:40, 128:].mean(0, keepdim=True).expand(S, 64)
    assert maxabs(got, want) < 1e-2


def test_attention_head_scale(dev):
    from visitron_amd import ops

    B, S, nh = 2, 50, 3
    g = torch.Generator().manual_seed(2)
    qkv = bf16_round(_rand((B * S, 3 * nh * 64), g)).to(dev, BF16)
    base = ops.attention_fwd(qkv, B, S, nh).float()
    hs = torch.tensor([1.0, 0.0, 0.5], device=dev)
    got = ops.attention_fwd(qkv, B, S, nh, head_scale=hs).float()
    want = (base.view(B * S, nh, 64) * hs[None, :, None]).reshape(B * S, -1)
    assert maxabs(got, want) < 1e-2


@pytest.mark.parametrize("M,H", [(5, 128), (456, 768), (33, 1024), (7, 64)])
def test_layernorm_matches_fp32(dev, M, H):
    from visitron_amd import ops

    g = torch.Generator().manual_seed(H)
    x = bf16_round(_rand((M, H), g, 3.0) + 0.7)
    gamma, beta = 1 + 0.1 * _rand((H,), g), 0.1 * _rand((H,), g)
    u = x.mean(-1, keepdim=True)
    s = (x - u).pow(2).mean(-1, keepdim=True)
    want = (x - u) / torch.sqrt(s + 1e-12) * gamma + beta
    mean = torch.zeros(M, device=dev)
    rstd = torch.zeros(M, device=dev)
    got = ops.layernorm(x.to(dev, BF16), gamma.to(dev), beta.to(dev), 1e-12, mean=mean, rstd=rstd)
    assert maxabs(got, want) < 3e-2
    assert maxabs(mean, u[:, 0]) < 1e-4
    assert maxabs(rstd, 1 / torch.sqrt(s[:, 0] + 1e-12)) < 1e-3


def test_embed_layernorm_matches_fp32(dev):
    from visitron_amd import ops

    B, T, R, H, V = 3, 9, 4, 128, 50
    S = T + R
    g = torch.Generator().manual_seed(3)
    word, pos, typ = _rand((V, H), g), _rand((16, H), g), _rand((2, H), g)
    gamma, beta = 1 + 0.1 * _rand((H,), g), 0.1 * _rand((H,), g)
    ids = torch.randint(0, V, (B, T), generator=g)
    tt = torch.randint(0, 2, (B, T), generator=g)
    e = word[ids] + pos[torch.arange(T)][None] + typ[tt]
    want = torch.nn.functional.layer_norm(e, (H,), gamma, beta, 1e-12)
    out = torch.zeros((B * S, H), dtype=BF16, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.embed_layernorm(ids.to(dev), tt.to(dev), None, word.to(dev), pos.to(dev), typ.to(dev), gamma.to(dev),
                        beta.to(dev), 1e-12, out, S, err_flag=err)
    got = out.float().cpu().view(B, S, H)
    assert maxabs(got[:, :T], want) < 3e-2
    assert float(got[:, T:].abs().max()) == 0.0
    assert int(err.item()) == 0
    bad = ids.clone()
    bad[0, 0] = V + 3
    ops.embed_layernorm(bad.to(dev), tt.to(dev), None, word.to(dev), pos.to(dev), typ.to(dev), gamma.to(dev),
                        beta.to(dev), 1e-12, out, S, err_flag=err)
    assert int(err.item()) == 1


@pytest.mark.parametrize("d0,d1,kpad,shift", [(2054, 128, 2240, 0), (2054, 128, 2240, 1), (2053, 128, 2240, 0), (2054, 127, 2240, 0),
                                              (40, 0, 40, 0), (6, 2, 8, 0)])
def test_pack_concat(dev, d0, d1, kpad, shift):
    """[src0 | src1 | zeros] -> bf16 (the region features next to their location embedding, encoder.py:277-279): even widths
    on 8-byte bases take the float2 reads, odd widths and a base that is only 4-byte aligned (shift) the element-wise path."""
    from visitron_amd import ops

    g = torch.Generator().manual_seed(4)
    rows = 11
    a, b = _rand((rows, d0), g), _rand((rows, max(d1, 1)), g)[:, :d1]
    a_dev = torch.empty(rows * d0 + shift, device=dev)[shift:].view(rows, d0).copy_(a)
    got = ops.pack_concat(a_dev, b.contiguous().to(dev), kpad).float().cpu()
    assert torch.equal(got[:, :d0], bf16_round(a))
    assert torch.equal(got[:, d0:d0 + d1], bf16_round(b))
    assert got.shape[1] == kpad and (kpad == d0 + d1 or float(got[:, d0 + d1:].abs().max()) == 0.0)


@pytest.mark.parametrize("M", [456, 64, 1000, 14592])
def test_wgrad_grouped_matches_fp32(dev, M):
    """dW = dY^T X and db = colsum(dY) for several problems in ONE launch; M tail zero-filled."""
    from visitron_amd import ops

    g = torch.Generator().manual_seed(M)
    specs = [(256, 128, True), (768, 768, True), (36, 128, True), (130, 2240, False), (1601, 64, True)]
    if M > 2000:
        specs = [(2304, 768, True), (768, 768, True), (3072, 768, True), (768, 3072, True)]
    probs, wants = [], []
    for N, K, bias in specs:
        ldy = (N + 7) // 8 * 8
        dy = torch.zeros(M, ldy)
        dy[:, :N] = bf16_round(_rand((M, N), g, 0.5))
        x = bf16_round(_rand((M, K), g))
        dw = torch.full((N, K), 3.0, device=dev)
        db = torch.full((N,), 3.0, device=dev) if bias else None
        probs.append(dict(dy=dy.to(dev, BF16)[:, :N], x=x.to(dev, BF16), dw=dw, db=db))
        wants.append((dy[:, :N].double().t() @ x.double(), dy[:, :N].double().sum(0)))
    ops.wgrad(probs, M)
    torch.cuda.synchronize()
    for p, (w_dw, w_db) in zip(probs, wants):
        scale = float(w_dw.abs().max())
        assert float((p["dw"].cpu().double() - w_dw).abs().max()) < 2e-4 * scale + 1e-3
        if p["db"] is not None:
            assert float((p["db"].cpu().double() - w_db).abs().max()) < 2e-4 * float(w_db.abs().max()) + 1e-3
    # accumulate mode adds on top
    for p in probs:
        p["accumulate"] = True
    ops.wgrad(probs, M)
    torch.cuda.synchronize()
    for p, (w_dw, w_db) in zip(probs, wants):
        assert float((p["dw"].cpu().double() - 2 * w_dw).abs().max()) < 4e-4 * float(w_dw.abs().max()) + 2e-3


@pytest.mark.parametrize("M,specs", [
    (1000, [(256, 256, True), (512, 768, True)]),                        # M tail inside the last row block; 8 tiles on 8 workgroups
    (14592, [(2304, 768, True), (768, 768, True), (3072, 768, True), (768, 3072, True)]),   # an encoder layer's group (B = 64)
    (4100, [(768, 768, False), (256, 1024, True)]),
])
def test_wgrad_persistent_streamk_matches_plain_kernel_and_fp32(dev, M, specs):
    """The persistent stream-K weight-gradient kernel (fp32 atomics of partial tiles) against fp64 and against the
    one-tile-per-workgroup kernel on the same operands; overwrite and accumulate modes."""
    from visitron_amd import ops

    g = torch.Generator().manual_seed(M + 1)
    data = []
    for N, K, bias in specs:
        dy, x = bf16_round(_rand((M, N), g, 0.5)), bf16_round(_rand((M, K), g))
        data.append((dy, x, bias, dy.double().t() @ x.double(), dy.double().sum(0)))

    def run(mode, accumulate):
        ops.set_wgrad_kernel(mode)
        try:
            probs = [dict(dy=dy.to(dev, BF16), x=x.to(dev, BF16), dw=torch.full((dy.shape[1], x.shape[1]), 3.0, device=dev),
                          db=torch.full((dy.shape[1],), 3.0, device=dev) if bias else None, accumulate=accumulate)
                     for dy, x, bias, _, _ in data]
            ops.wgrad(probs, M)
            torch.cuda.synchronize()
            return probs
        finally:
            ops.set_wgrad_kernel(0)

    for accumulate in (False, True):
        new, old = run(8, accumulate), run(-8, accumulate)    # (8: the persistent kernel also below its 12 288-row threshold)
        for pn, po, (_, _, bias, w_dw, w_db) in zip(new, old, data):
            off = 3.0 if accumulate else 0.0
            scale = float(w_dw.abs().max())
            assert float((pn["dw"].cpu().double() - off - w_dw).abs().max()) < 2e-4 * scale + 1e-3
            assert float((pn["dw"] - po["dw"]).abs().max()) < 1e-4 * scale + 1e-4      # fp32 summation order only
            if bias:
                assert float((pn["db"].cpu().double() - off - w_db).abs().max()) < 2e-4 * float(w_db.abs().max()) + 1e-3


def test_wgrad_asymmetric_identity(dev):
    """dY = I (M = N) against an asymmetric X: dW must equal X exactly (catches transposed writes)."""
    from visitron_amd import ops

    M = N = 128
    K = 256
    x = bf16_round((torch.arange(M * K, dtype=torch.float32).reshape(M, K) % 253) / 32.0 - 3.0)
    dw = torch.zeros((N, K), device=dev)
    ops.wgrad([dict(dy=torch.eye(M).to(dev, BF16), x=x.to(dev, BF16), dw=dw)], M)
    assert torch.equal(dw.cpu(), x)


@pytest.mark.parametrize("B,S,nh", [(2, 228, 3), (1, 37, 2), (2, 256, 1), (3, 64, 2), (1, 5, 1), (2, 300, 2), (1, 656, 2), (1, 767, 1)])
def test_attention_bwd_matches_autograd(dev, B, S, nh):
    from visitron_amd import ops

    g = torch.Generator().manual_seed(S + nh)
    H = nh * 64
    qkv = bf16_round(_rand((B * S, 3 * H), g, 1.2)).requires_grad_(True)
    dctx = bf16_round(_rand((B * S, H), g, 0.7))
    mask = (torch.rand(B, S, generator=g) > 0.25).float()
    mask[:, 0] = 1.0
    t = qkv.view(B, S, 3, nh, 64).permute(2, 0, 3, 1, 4)
    ctx_ref = _attention_ref(t[0], t[1], t[2], (1.0 - mask) * -10000.0).permute(0, 2, 1, 3).reshape(B * S, H)
    ctx_ref.backward(dctx)
    want = qkv.grad
    qd = qkv.detach().to(dev, BF16)
    lse = torch.zeros((B, nh, S), dtype=torch.float32, device=dev)
    ctx = ops.attention_fwd(qd, B, S, nh, mask=mask.to(dev), lse=lse)
    got = ops.attention_bwd(qd, dctx.to(dev, BF16), ctx, lse, B, S, nh, mask=mask.to(dev))
    torch.cuda.synchronize()
    got = got.float().cpu()
    for name, sl in (("dq", slice(0, H)), ("dk", slice(H, 2 * H)), ("dv", slice(2 * H, 3 * H))):
        w = want[:, sl]
        err = float((got[:, sl] - w).abs().max())
        assert err < 2.5e-2 * (1.0 + float(w.abs().max())), (name, err, float(w.abs().max()))


@pytest.mark.parametrize("M,H", [(456, 768), (37, 128), (5000, 768), (3, 1024)])
def test_layernorm_bwd_matches_autograd(dev, M, H):
    from visitron_amd import ops

    g = torch.Generator().manual_seed(M)
    x = bf16_round(_rand((M, H), g, 2.0) + 0.3).requires_grad_(True)
    dy = bf16_round(_rand((M, H), g))
    gamma = (1 + 0.1 * _rand((H,), g)).requires_grad_(True)
    beta = torch.zeros(H, requires_grad=True)
    torch.nn.functional.layer_norm(x, (H,), gamma, beta, 1e-12).backward(dy)
    dgamma = torch.full((H,), 5.0, device=dev)
    dbeta = torch.full((H,), 5.0, device=dev)
    dx = ops.layernorm_bwd(x.detach().to(dev, BF16), dy.to(dev, BF16), gamma.detach().to(dev), 1e-12, dgamma, dbeta)
    torch.cuda.synchronize()
    assert maxabs(dx, x.grad) < 2e-2 * (1 + float(x.grad.abs().max()))
    assert maxabs(dgamma, gamma.grad) < 1e-3 * (1 + float(gamma.grad.abs().max()))
    assert maxabs(dbeta, beta.grad) < 1e-3 * (1 + float(beta.grad.abs().max()))
    ops.layernorm_bwd(x.detach().to(dev, BF16), dy.to(dev, BF16), gamma.detach().to(dev), 1e-12, dgamma, dbeta, accumulate=True)
    assert maxabs(dgamma, 2 * gamma.grad) < 2e-3 * (1 + float(gamma.grad.abs().max()))


def test_linear_pre_activation_output_and_dgelu(dev):
    from visitron_amd import ops

    g = torch.Generator().manual_seed(9)
    M, N, K = 200, 256, 128
    a = bf16_round(_rand((M, K), g))
    w = bf16_round(_rand((N, K), g, 0.1))
    b = _rand((N,), g, 0.1)
    pre = torch.zeros((M, N), dtype=BF16, device=dev)
    out = ops.linear(a.to(dev, BF16), w.to(dev, BF16), b.to(dev), act=1, pre_act_out=pre)
    h = (a @ w.t() + b).requires_grad_(True)
    _gelu(h).sum().backward()
    assert maxabs(pre, h.grad) < 1e-2          # the saved tensor is gelu'(pre-activation)
    assert maxabs(out, _gelu(h.detach())) < 2e-2
    h = h.detach()
    # dgrad through the GELU: (dy @ W2) * gelu'(h)
    hh = bf16_round(h).requires_grad_(True)
    dyv = bf16_round(_rand((M, K), g))
    w2t = bf16_round(_rand((N, K), g, 0.1))  # plays W2^T: [I, H]
    upstream = dyv @ w2t.t()
    _gelu(hh).backward(upstream)
    hg = hh.detach().clone().requires_grad_(True)
    _gelu(hg).sum().backward()
    dsaved = bf16_round(hg.grad).to(dev, BF16)   # what the forward saves
    got = ops.linear(dyv.to(dev, BF16), w2t.to(dev, BF16), residual=dsaved, act=3)
    assert maxabs(got, hh.grad) < 3e-2 * (1 + float(hh.grad.abs().max()))
    got2 = ops.dgelu_mul(bf16_round(upstream).to(dev, BF16), dsaved)
    assert maxabs(got2, hh.grad) < 3e-2 * (1 + float(hh.grad.abs().max()))


@pytest.mark.parametrize("R,C", [(768, 2304), (64, 64), (3072, 768), (136, 200)])
def test_transpose(dev, R, C):
    from visitron_amd import ops

    x = torch.arange(R * C, dtype=torch.float32).reshape(R, C) % 509
    xd = x.to(dev, BF16)
    out = torch.zeros((C, R), dtype=BF16, device=dev)
    ops.transpose(xd, out)
    assert torch.equal(out, xd.t().contiguous())


@pytest.mark.parametrize("rows,V", [(37, 30522), (5, 97), (3, 8), (64, 1601)])
def test_fused_ce_rows(dev, rows, V):
    from visitron_amd import ops

    g = torch.Generator().manual_seed(V)
    Vp = (V + 63) // 64 * 64
    z = torch.zeros(rows, Vp)
    z[:, :V] = _rand((rows, V), g, 3.0)
    z[0, min(5, V - 1)] = z[0, :V].max() + 1  # a clear argmax
    y = torch.randint(0, V, (rows,), generator=g)
    zr = z[:, :V].clone().requires_grad_(True)
    loss = torch.nn.functional.cross_entropy(zr, y, reduction="none")
    (loss.sum() * 0.25).backward()
    dz = torch.full((rows, Vp), 7.0, dtype=BF16, device=dev)
    lr, am = ops.ce_softmax_rows(z.to(dev), y.to(dev), V, dz, 0.25)
    torch.cuda.synchronize()
    assert maxabs(lr, loss) < 1e-4 * (1 + float(loss.abs().max()))
    assert torch.equal(am.cpu(), z[:, :V].argmax(1))
    assert maxabs(dz[:, :V], zr.grad) < 4e-3 * float(zr.grad.abs().max()) + 1e-6
    if Vp > V:
        assert float(dz[:, V:].float().abs().max()) == 0.0


# ---- dropout (counter-based hash masks; the backward kernels recompute them) ------------------------------
def test_dropout_mask_is_bernoulli_and_site_keyed(dev):
    from visitron_amd import ops

    n = 1 << 20
    for p in (0.1, 0.5):
        k = ops.dropout_mask(n, (p, 1234, 3), device=dev).float()
        assert abs(float(k.mean()) - (1.0 - p)) < 4.0 * math.sqrt(p * (1 - p) / n) + 1e-4
        # neighbouring elements are uncorrelated
        c = float(((k[1:] - k.mean()) * (k[:-1] - k.mean())).mean() / k.var())
        assert abs(c) < 5e-3
    a = ops.dropout_mask(n, (0.5, 1234, 3), device=dev)
    assert torch.equal(a, ops.dropout_mask(n, (0.5, 1234, 3), device=dev))         # deterministic
    for other in ((0.5, 1235, 3), (0.5, 1234, 4)):                                  # seed / site change the stream
        b = ops.dropout_mask(n, other, device=dev)
        assert abs(float((a == b).float().mean()) - 0.5) < 5e-3
    h0 = ops.dropout_mask(n, (0.5, 1234, 3), head_index=0, device=dev)
    h1 = ops.dropout_mask(n, (0.5, 1234, 3), head_index=1, device=dev)
    assert abs(float((h0 == h1).float().mean()) - 0.5) < 5e-3
    assert int(ops.dropout_mask(4096, (0.0, 7, 0), device=dev).sum()) == 4096       # p = 0 keeps everything


@pytest.mark.parametrize("M,N,K,res", [(456, 768, 768, True), (300, 768, 3072, True), (130, 200, 128, False)])
def test_linear_dropout_before_residual(dev, M, N, K, res):
    """BertSelfOutput / BertOutput: LN(dropout(dense(h)) + residual) (oscar/modeling_bert.py:94,120)."""
    from visitron_amd import ops

    g = torch.Generator().manual_seed(M + N)
    a, w, b = bf16_round(_rand((M, K), g)), bf16_round(_rand((N, K), g, 0.05)), _rand((N,), g, 0.1)
    r = bf16_round(_rand((M, N), g)) if res else None
    drop = (0.2, 99, 5)
    keep = ops.dropout_mask(M * N, drop, device=dev).view(M, N).float().cpu()
    want = (a @ w.t() + b) * keep / 0.8
    if res:
        want = want + r
    got = ops.linear(a.to(dev, BF16), w.to(dev, BF16), b.to(dev), residual=None if r is None else r.to(dev, BF16), drop=drop)
    torch.cuda.synchronize()
    assert maxabs(got, want) < 2e-2 * (1 + float(want.abs().max()))
    # dropped elements are exactly the residual (or zero)
    base = r if res else torch.zeros(M, N)
    assert torch.equal(got.float().cpu()[keep == 0], bf16_round(base)[keep == 0])


def test_apply_dropout_and_layernorm_bwd_dropped_copy(dev):
    from visitron_amd import ops

    g = torch.Generator().manual_seed(8)
    M, H = 300, 768
    x, dy, gamma = bf16_round(_rand((M, H), g)), bf16_round(_rand((M, H), g)), _rand((H,), g).abs() + 0.5
    drop = (0.3, 5, 17)
    keep = ops.dropout_mask(M * H, drop, device=dev).view(M, H)
    dg, db = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    dxd = torch.empty((M, H), dtype=BF16, device=dev)
    dx = ops.layernorm_bwd(x.to(dev, BF16), dy.to(dev, BF16), gamma.to(dev), 1e-12, dg, db, dx_dropped=dxd, drop=drop)
    dg2, db2 = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    dx_plain = ops.layernorm_bwd(x.to(dev, BF16), dy.to(dev, BF16), gamma.to(dev), 1e-12, dg2, db2)
    torch.cuda.synchronize()
    assert torch.equal(dx, dx_plain) and torch.equal(dg, dg2)       # the unmasked gradient is unchanged
    assert maxabs(dxd, dx.float() * keep / 0.7) < 2e-2 * (1 + float(dx.float().abs().max()))
    assert int((dxd[keep == 0] != 0).sum()) == 0
    y = dx.clone()
    ops.apply_dropout(y, drop)
    assert maxabs(y, dxd) < 2e-2 * (1 + float(dx.float().abs().max()))


def _attn_keep(ops, B, nh, S, drop, dev):
    return torch.stack([ops.attn_dropout_mask(S, drop, i, device=dev)
                        for i in range(B * nh)]).view(B, nh, S, S).float().cpu()


@pytest.mark.parametrize("B,S,nh,waves", [(2, 228, 3, 8), (1, 37, 2, 8), (2, 300, 2, 8), (1, 656, 1, 8), (2, 228, 2, 4),
                                          (1, 300, 1, 4), (2, 228, 3, 10), (1, 37, 2, 10), (2, 300, 2, 10),
                                          (2, 228, 3, 16), (1, 37, 2, 16), (3, 193, 2, 16), (2, 256, 1, 16), (2, 64, 2, 16),
                                          (2, 228, 3, 17), (1, 37, 2, 17), (3, 193, 2, 17), (2, 256, 1, 17), (2, 64, 2, 17), (5, 1, 2, 17)])
def test_attention_dropout_fwd_bwd_match_autograd(dev, B, S, nh, waves):
    """attention_probs dropout (oscar/modeling_bert.py:62) with the kernel's own keep-mask fed to torch."""
    from visitron_amd import ops

    g = torch.Generator().manual_seed(S * 3 + nh)
    H, p = nh * 64, 0.2
    drop = (p, 4242, 16)
    qkv = bf16_round(_rand((B * S, 3 * H), g, 1.2)).requires_grad_(True)
    dctx = bf16_round(_rand((B * S, H), g, 0.7))
    mask = (torch.rand(B, S, generator=g) > 0.25).float()
    mask[:, 0] = 1.0
    keep = _attn_keep(ops, B, nh, S, drop, dev)
    t = qkv.view(B, S, 3, nh, 64).permute(2, 0, 3, 1, 4)
    s = t[0] @ t[1].transpose(-1, -2) / 8.0 + ((1.0 - mask) * -10000.0)[:, None, None, :]
    ctx_ref = ((torch.softmax(s, dim=-1) * keep / (1 - ops.attn_drop_p(p))) @ t[2]).permute(0, 2, 1, 3).reshape(B * S, H)
    ctx_ref.backward(dctx)
    want = qkv.grad
    qd = qkv.detach().to(dev, BF16)
    lse = torch.zeros((B, nh, S), dtype=torch.float32, device=dev)
    # waves == 16 / 17: the 16-wave kernel, one pair per workgroup / persistent and pipelined (the default where it serves:
    # S <= 256, dropout through the forward's keep words)
    words = torch.zeros(ops.keep_words(B, nh, S), dtype=torch.int32, device=dev) if waves >= 16 else None
    ctx = ops.attention_fwd(qd, B, S, nh, mask=mask.to(dev), lse=lse, drop=drop, keep_bits=words)
    ops.set_attn_bwd_waves(waves)
    try:
        got = ops.attention_bwd(qd, dctx.to(dev, BF16), ctx, lse, B, S, nh, mask=mask.to(dev), drop=drop, keep_bits=words)
        torch.cuda.synchronize()
    finally:
        ops.set_attn_bwd_waves(0)       # back to the default (the persistent 16-wave kernel where it serves)
    assert maxabs(ctx, ctx_ref) < 4e-2
    got = got.float().cpu()
    for name, sl in (("dq", slice(0, H)), ("dk", slice(H, 2 * H)), ("dv", slice(2 * H, 3 * H))):
        w = want[:, sl]
        err = float((got[:, sl] - w).abs().max())
        assert err < 2.5e-2 * (1.0 + float(w.abs().max())), (name, err, float(w.abs().max()))


def test_embed_layernorm_dropout(dev):
    from visitron_amd import ops

    g = torch.Generator().manual_seed(2)
    B, T, S, H, V = 2, 9, 12, 128, 50
    ids = torch.randint(0, V, (B, T), generator=g)
    word, pos, typ = _rand((V, H), g), _rand((64, H), g), _rand((2, H), g)
    gamma, beta = _rand((H,), g).abs() + 0.5, _rand((H,), g)
    p = 0.25
    out = torch.zeros((B * S, H), dtype=BF16, device=dev)
    ops.embed_layernorm(ids.to(dev), None, None, word.to(dev), pos.to(dev), typ.to(dev), gamma.to(dev), beta.to(dev),
                        1e-12, out, S, drop=(p, 77, 0))
    plain = torch.zeros((B * S, H), dtype=BF16, device=dev)
    ops.embed_layernorm(ids.to(dev), None, None, word.to(dev), pos.to(dev), typ.to(dev), gamma.to(dev), beta.to(dev),
                        1e-12, plain, S)
    keep = ops.dropout_mask(B * T * H, (p, 77, ops.SITE_EMB), device=dev).view(B, T, H).float()
    torch.cuda.synchronize()
    o = out.view(B, S, H)[:, :T].float()
    pl = plain.view(B, S, H)[:, :T].float()
    assert maxabs(o, pl * keep / (1 - p)) < 2e-2 * (1 + float(pl.abs().max()))


@pytest.mark.parametrize("rows,V", [(37, 1601), (5, 11), (300, 40)])
def test_fused_double_softmax_ce_rows(dev, rows, V):
    """token_head = Linear + Softmax and CrossEntropyLoss on top (encoder.py:323-326, 380-385): loss, argmax and the
    gradient through both softmaxes against autograd."""
    from visitron_amd import ops

    g = torch.Generator().manual_seed(rows + V)
    Vp = (V + 63) // 64 * 64
    z = torch.zeros(rows, Vp)
    z[:, :V] = _rand((rows, V), g, 2.0)
    y = torch.randint(0, V, (rows,), generator=g)
    zr = z[:, :V].clone().requires_grad_(True)
    p = torch.softmax(zr, dim=-1)
    loss = torch.nn.functional.cross_entropy(p, y, reduction="none")
    (loss.sum() * 0.37).backward()
    dz = torch.full((rows, Vp), 9.0, dtype=BF16, device=dev)
    lr, am = ops.ce_double_softmax_rows(z.to(dev), y.to(dev), V, dz, 0.37)
    torch.cuda.synchronize()
    assert maxabs(lr, loss.detach()) < 1e-4 * (1 + float(loss.abs().max()))
    assert torch.equal(am.cpu(), p.argmax(1))
    assert maxabs(dz[:, :V], zr.grad) < 1e-2 * float(zr.grad.abs().max()) + 1e-6
    assert float(dz[:, V:].float().abs().max() if Vp > V else 0.0) == 0.0


def test_transpose_batch(dev):
    """vt_transpose_batch_bf16: several matrices of different shapes (row strides included) in one launch, against the
    single-matrix kernel's definition out[c, r] = in[r, c]."""
    from visitron_amd import ops

    g = torch.Generator().manual_seed(3)
    shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072), (64, 8), (72, 200), (1601, 768), (36, 64)]
    pairs, wants = [], []
    for i, (R, C) in enumerate(shapes):
        base = torch.randn(R, C + 8 * (i % 2), generator=g).to(dev, torch.bfloat16)
        src = base[:, :C]                                  # every other one with a row stride
        Rp = (R + 63) // 64 * 64 if R % 8 else R           # a row count off the 8-grid needs a padded output row
        out = torch.full((C, Rp), 7.0, dtype=torch.bfloat16, device=dev)
        pairs.append((src, out))
        wants.append(src.t().contiguous())
    tb = ops.TransposeBatch(pairs)
    tb.run()
    torch.cuda.synchronize()
    for (src, out), want in zip(pairs, wants):
        R = src.shape[0]
        assert torch.equal(out[:, :R], want)
        if R % 8:                                          # the rest of the last 8-group is zero, beyond it untouched
            r8 = (R + 7) // 8 * 8
            assert float(out[:, R:r8].abs().max()) == 0.0 and float(out[:, r8:].min()) == 7.0
    with pytest.raises(AssertionError):
        ops.TransposeBatch([(pairs[0][0], pairs[1][1])])   # shape mismatch is refused on the host


@pytest.mark.parametrize("B,S,nh,waves", [(5, 228, 3, 8), (3, 300, 2, 8), (4, 228, 2, 4), (2, 656, 1, 8), (5, 228, 3, 10), (3, 300, 2, 10), (5, 228, 3, 16), (4, 100, 2, 16),
                                          (5, 228, 3, 17), (4, 100, 2, 17)])
def test_attention_on_compacted_rows_equals_masked_padded_run(dev, B, S, nh, waves):
    """The *_seq_* attention entry points (rows of padded keys dropped, per-sequence start / length) against the padded
    kernels with the same keys masked: context rows, log-sum-exp and the packed q|k|v gradient of the real rows agree."""
    from visitron_amd import ops

    g = torch.Generator().manual_seed(B * 1000 + S)
    H = nh * 64
    lens = torch.randint(S // 3, S + 1, (B,), generator=g)
    lens[0] = S                                        # one full sequence, one short
    lens[-1] = max(1, S // 5)
    keep = torch.arange(S)[None, :] < lens[:, None]
    qkv = _rand((B * S, 3 * H), g, 0.8).to(dev, BF16)
    # in a training step the context gradient of a padding row is exactly zero (nothing reads that row); a padded QUERY
    # with a gradient would otherwise add to the real keys' dK / dV
    dctx = (_rand((B * S, H), g) * keep.reshape(-1, 1).float()).to(dev, BF16)
    mask = keep.float().to(dev)
    ops.set_attn_bwd_waves(waves)
    try:
        lse_p = torch.zeros((B, nh, S), dtype=torch.float32, device=dev)
        ctx_p = ops.attention_fwd(qkv, B, S, nh, mask=mask, lse=lse_p)
        dq_p = ops.attention_bwd(qkv, dctx, ctx_p, lse_p, B, S, nh, mask=mask)
        seq = ops.SeqLayout(keep.to(dev))
        assert seq.rows == int(lens.sum())
        qkv_c = qkv.index_select(0, seq.index).contiguous()
        dctx_c = dctx.index_select(0, seq.index).contiguous()
        lse_c = torch.zeros((B, nh, S), dtype=torch.float32, device=dev)
        ctx_c = ops.attention_fwd(qkv_c, B, S, nh, lse=lse_c, seq=seq)
        dq_c = ops.attention_bwd(qkv_c, dctx_c, ctx_c, lse_c, B, S, nh, seq=seq)
        torch.cuda.synchronize()
    finally:
        ops.set_attn_bwd_waves(0)       # back to the default (the persistent 16-wave kernel where it serves)
    assert ctx_c.shape == (seq.rows, H) and dq_c.shape == (seq.rows, 3 * H)
    assert maxabs(ctx_c, ctx_p.index_select(0, seq.index)) < 1e-6
    kq = keep[:, None, :].expand(B, nh, S)
    assert maxabs(lse_c.cpu()[kq], lse_p.cpu()[kq]) < 1e-5
    want = dq_p.index_select(0, seq.index).float()
    assert maxabs(dq_c, want) <= 2e-2 * float(want.abs().max())          # (dQ sums run over fewer zero terms)
    # the padded run's gradient at the padding rows: dK, dV are exactly zero there (masked keys weigh nothing)
    pad_rows = torch.nonzero(~keep.reshape(-1)).flatten().to(dev)
    if pad_rows.numel():
        assert float(dq_p.index_select(0, pad_rows)[:, H:].float().abs().max()) == 0.0


def test_attention_dropout_on_compacted_rows_matches_autograd(dev):
    """Dropout on compacted rows: the element index is q * len_b + key (the sequence's own length); torch reference on
    the kernel's own keep-mask, per sequence."""
    from visitron_amd import ops

    g = torch.Generator().manual_seed(12)
    B, S, nh = 3, 64, 2
    H = nh * 64
    lens = torch.tensor([64, 40, 17])
    keep = torch.arange(S)[None, :] < lens[:, None]
    seq = ops.SeqLayout(keep.to(dev))
    drop = (0.2, 99, ops.site_attn(1))
    qkv = _rand((seq.rows, 3 * H), g, 0.8)
    dctx = _rand((seq.rows, H), g)
    lse = torch.zeros((B, nh, S), dtype=torch.float32, device=dev)
    ctx = ops.attention_fwd(qkv.to(dev, BF16), B, S, nh, lse=lse, drop=drop, seq=seq)
    dq = ops.attention_bwd(qkv.to(dev, BF16), dctx.to(dev, BF16), ctx, lse, B, S, nh, drop=drop, seq=seq)
    torch.cuda.synchronize()
    off = 0
    for b in range(B):
        n = int(lens[b])
        x = qkv[off:off + n].to(BF16).float().requires_grad_(True)
        t = x.view(n, 3, nh, 64).permute(1, 2, 0, 3)
        p = torch.softmax(t[0] @ t[1].transpose(-1, -2) / 8.0, -1)
        km = torch.stack([ops.attn_dropout_mask(n, drop, b * nh + h, device=dev) for h in range(nh)]).float().cpu()
        o = ((p * km / (1.0 - ops.attn_drop_p(0.2))) @ t[2]).permute(1, 0, 2).reshape(n, H)
        o.backward(dctx[off:off + n].to(BF16).float())
        assert maxabs(ctx[off:off + n], o.detach()) < 3e-2
        assert maxabs(dq[off:off + n], x.grad) < 3e-2 * (1 + float(x.grad.abs().max()))
        off += n


@pytest.mark.parametrize("act,res,pre,drop_p", [(0, True, False, 0.1), (1, False, True, 0.0), (3, True, False, 0.0), (0, False, False, 0.0)])
def test_linear_tail_rows_launch(dev, act, res, pre, drop_p):
    """The persistent GEMM's half-empty last round handed to the 128x128-tile kernel (two launches over disjoint row
    ranges, the dropout seed shifted by the row offset): same values as the single launch, identical dropout mask."""
    from visitron_amd import _lib, ops

    g = torch.Generator().manual_seed(act * 10 + int(res))
    M, N, K = 256 * 200 - 37, 768, 768          # 600 tiles: two full rounds on 256 CUs + 88
    a = _rand((M, K), g).to(dev, BF16)
    w = _rand((N, K), g, 0.05).to(dev, BF16)
    b = _rand((N,), g).to(dev)
    r = _rand((M, N), g).to(dev, BF16) if (res or act == 3) else None
    drop = (drop_p, 4242, ops.site_out(1)) if drop_p else ops.NO_DROP
    lib = _lib.load()
    lib.vt_gemm_tune(M, N, K, ops.tune_kind(act, residual=res, pre_act=pre), 16)
    outs = []
    for mode in (-1, -2):          # single launch, then with the tail launch
        lib.vt_debug_set_gemm_variant(mode)
        c2 = torch.empty((M, N), dtype=BF16, device=dev) if pre else None
        out = ops.linear(a, w, b, residual=r, act=act, pre_act_out=c2, drop=drop)
        outs.append((out, c2))
    lib.vt_debug_set_gemm_variant(-1)
    torch.cuda.synchronize()
    (o0, p0), (o1, p1) = outs
    keep0, keep1 = (o0.float() != (r.float() if (res and act != 3) else 0)), (o1.float() != (r.float() if (res and act != 3) else 0))
    if drop_p:
        assert torch.equal(keep0, keep1)                      # the same elements were dropped in both runs
    assert maxabs(o0, o1) <= 2e-2 * (1 + float(o0.float().abs().max()))
    assert float((o0.float() - o1.float()).abs().mean()) < 2e-3
    if pre:
        assert maxabs(p0, p1) <= 2e-2
