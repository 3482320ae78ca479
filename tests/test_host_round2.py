"""CPU: host logic added in round 2 -- precision switch, weights generation, the tightened parity bounds, the bench's
own launcher and FLOP counts, and the alignment of the data-parallel gradient ranges (no GPU, no compute calls)."""
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_set_precision_flags_every_module_and_rejects_unknown_modes():
    from visitron_amd import set_precision
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import PreTrainOscar, _is_fp32

    m = PreTrainOscar(mini_config())
    assert not _is_fp32(m) and not _is_fp32(m.bert.encoder)
    assert set_precision(m, "fp32") is m
    assert all(_is_fp32(x) for x in m.modules())
    set_precision(m, "bf16")
    assert not any(_is_fp32(x) for x in m.modules())
    with pytest.raises(ValueError):
        set_precision(m, "fp16")
    # training in fp32 mode is refused before any tensor is touched
    set_precision(m, "fp32").train()
    ids = torch.zeros(1, 4, dtype=torch.long)
    with pytest.raises(NotImplementedError):
        m(ids, labels=ids, token_labels=ids)


def test_weights_generation_invalidates_every_cache_key():
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import PreTrainOscar, _param_key, invalidate_packed_weights

    m = PreTrainOscar(mini_config())
    k0 = _param_key(m.bert.encoder)
    assert _param_key(m.bert.encoder) == k0
    m.bert.pooler.dense.weight.data.add_(1.0)                 # a write through .data: no version bump ...
    assert _param_key(m.bert.pooler) == _param_key(m.bert.pooler)
    invalidate_packed_weights()                               # ... so writers of that kind advance the generation
    assert _param_key(m.bert.encoder) != k0
    with torch.no_grad():
        m.bert.encoder.layer[0].output.dense.bias.add_(1.0)   # an ordinary in-place update bumps the version itself
    assert _param_key(m.bert.encoder)[1:] != k0[1:]


def test_effective_parity_bound_rule(tmp_path, monkeypatch):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers

    monkeypatch.setattr(helpers, "_REC", {"a": 1e-3, "b": 4e-2, "c": 1e-9})
    assert helpers.effective_bound("a", 5e-2) == pytest.approx(5e-3)      # floor: a tenth of the stated bound
    assert helpers.effective_bound("b", 5e-2) == pytest.approx(5e-2)      # never above the stated bound
    assert helpers.effective_bound("c", 1e-3) == pytest.approx(1e-4)
    assert helpers.effective_bound("unknown", 5e-2) == 5e-2
    assert helpers.effective_bound("a", 5e-2, scalar=True) == pytest.approx(2.5e-2)   # single numbers (losses): floor at half
    assert helpers.effective_bound("b", 5e-2, scalar=True) == pytest.approx(5e-2)
    rec = json.load(open(os.path.join(ROOT, "tests", "golden", "parity_measured.json")))
    assert len(rec) > 200 and all(v >= 0 for v in rec.values())
    # the committed table is the one profiles/r02 was written from
    names = {line.split("  ")[0].strip() for line in open(os.path.join(ROOT, "profiles", "r02", "parity_measured.txt"))
             if not line.startswith("#")}
    assert len(names & set(rec)) > 200


def test_bench_flop_counts_and_self_launch_command(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench

    S = 228
    full = bench.enc_flops_per_seq(S)
    assert full == 12 * (24 * S * 768 ** 2 + 4 * S * S * 768) == 40646541312
    assert bench.enc_flops_rows([S] * 7) == pytest.approx(7 * full)
    assert bench.enc_flops_rows([100, 228]) < 2 * full
    assert bench._pct([1.0, 2.0, 3.0, 4.0, 5.0], 0.5) == 3.0 and bench._pct([1.0, 3.0], 0.1) == pytest.approx(1.2)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()                                  # must hand over to the launcher before importing torch.cuda
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and os.path.basename(cmd[-5]) == "bench.py"
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"


def test_data_parallel_gradient_ranges_are_aligned_and_tile_the_slab():
    """The chunked backward hands [start, end) ranges of the gradient slab to the communicator; with the bf16
    communication copy every range must start and end on a multiple of 8 elements, and the chunk ranges plus their
    complement must cover the slab exactly once."""
    from visitron_amd.config import mini_config
    from visitron_amd.distributed import complement_ranges
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.ops import round_up
    from visitron_amd.training import ALIGN, PretrainEngine

    for L, step in ((4, 3), (5, 2), (2, 1)):
        m = PreTrainOscar(mini_config(num_hidden_layers=L))
        e = PretrainEngine(m, grad_comm_dtype="bf16")
        assert e.g16 is None and e.world == 1          # the copy exists only with more than one rank
        f = e.flat
        done, hi = [], L
        while hi > 0:
            lo = max(0, hi - step)
            done += [(e.layer_ranges[lo][k][0], min(round_up(e.layer_ranges[hi - 1][k][1], ALIGN), f.total)) for k in (0, 1)]
            hi = lo
        cover = torch.zeros(f.total, dtype=torch.int32)
        for s, t in done + complement_ranges(f.total, done):
            assert s % 8 == 0 and t % 8 == 0 and t > s
            cover[s:t] += 1
        assert int(cover.min()) == 1 and int(cover.max()) == 1
    with pytest.raises(ValueError):
        PretrainEngine(PreTrainOscar(mini_config()), grad_comm_dtype="fp8")


def test_centered_mask_is_the_same_softmax_and_a_noop_for_plain_masks():
    """The per-key mask shifted by one constant per sequence, m - rowmax(m) + 1 (modeling._centered_mask; since round 3 one
    HIP launch, vt_center_mask, held bitwise to this expression by tests/test_gpu_round3.py::
    test_center_mask_equals_the_torch_expression): bit-identical for 0/1 masks with a kept key, 254/255 (the rollout's ~uint8
    mask, agent_models.py:267) becomes 0/1, and the oracle trunk evaluated in float64 gives the same output for the raw and
    the centred mask."""
    from oracle.modeling import BertImgModelwithLocationEmbeds as OTrunk
    from visitron_amd.config import mini_config

    def _centered_mask(m):
        m = m.to(torch.float32)
        return m - m.amax(dim=1, keepdim=True) + 1.0

    m = torch.tensor([[1, 1, 1, 0, 0], [1, 0, 1, 1, 1]], dtype=torch.float32)
    assert torch.equal(_centered_mask(m), m)
    q = torch.tensor([[255, 255, 254, 254], [255, 255, 255, 255]], dtype=torch.float32)
    assert torch.equal(_centered_mask(q), torch.tensor([[1, 1, 0, 0], [1, 1, 1, 1]], dtype=torch.float32))
    assert torch.equal(_centered_mask(torch.zeros(2, 3)), torch.ones(2, 3))          # a fully masked sequence: uniform softmax
    fr = torch.tensor([[0.25, 0.75, 0.5]])
    assert torch.allclose(_centered_mask(fr), torch.tensor([[0.5, 1.0, 0.75]]))
    torch.manual_seed(0)
    trunk = OTrunk(mini_config()).double().eval()
    ids = torch.randint(1, 500, (2, 4))
    with torch.no_grad():
        a = trunk(ids, attention_mask=q.to(torch.uint8))[0]
        b = trunk(ids, attention_mask=_centered_mask(q))[0]
    assert float((a - b).abs().max()) < 1e-6


def test_flat_params_notice_when_another_engine_took_the_parameters():
    """FlatParams.owns_params: the full model's engine and the bare trunk's engine (train.py:47 hands model.bert to the
    rollout agent) cannot both back the same parameters; the one that lost them says so and is rebuilt by the bridge."""
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.training import FlatParams, _TrunkOnly

    model = PreTrainOscar(mini_config())
    full = FlatParams(model, attach_grads=False)
    assert full.owns_params()
    w = model.bert.pooler.dense.weight
    before = w.detach().clone()
    trunk = FlatParams(_TrunkOnly(model.bert), attach_grads=False)
    assert trunk.owns_params() and not full.owns_params()
    assert torch.equal(w.detach(), before)                        # values moved with the parameters
    assert all(n.startswith("bert.") for n, *_ in trunk.entries)  # the names they have inside a PreTrainOscar
    assert trunk.total < full.total


def test_data_parallel_replicas_are_refused_with_the_way_out():
    """multi-gpu-dp (pretrain.py:93-94) is not served: a DataParallel replica says so (and names the DDP route) instead of
    failing on a missing parameter somewhere inside."""
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import PreTrainOscar

    model = PreTrainOscar(mini_config())
    replica = model._replicate_for_data_parallel()
    replica._former_parameters = {}
    with pytest.raises(NotImplementedError, match="one process per GPU"):
        replica(torch.zeros(1, 4, dtype=torch.int64))


# ---- round 3: host logic of the deferred-LayerNorm path and the autotuner's keys ---------------------------------
def test_tune_kind_matches_the_library_key():
    """ops.tune_kind must produce what VT_TUNE_KIND (csrc/gemm_bf16.hip) derives from a call's arguments."""
    from visitron_amd import ops

    assert ops.tune_kind(ops.ACT_NONE) == 0
    assert ops.tune_kind(ops.ACT_NONE, residual=True) == 16
    assert ops.tune_kind(ops.ACT_GELU, pre_act=True) == 1 + 32
    assert ops.tune_kind(ops.ACT_MUL) == 3 + 16                       # the factor operand counts as a residual
    assert ops.tune_kind(ops.ACT_NONE, out_f32=True) == 64
    assert ops.tune_kind(ops.ACT_GELU, ln_mode=1) == 1 + 256 and ops.tune_kind(ops.ACT_NONE, ln_mode=2) == 512
    src = open(os.path.join(ROOT, "visitron_amd", "csrc", "gemm_bf16.hip")).read()
    assert "((act) | ((has_r) ? 16 : 0) | ((has_c2) ? 32 : 0) | ((out_f32) ? 64 : 0) | ((ln_mode) << 8))" in src


def test_default_variant_table_lookup(tmp_path, monkeypatch):
    from visitron_amd import ops

    monkeypatch.setattr(ops, "_defaults", {(14592, 768, 768, 512): 23, (58368, 768, 768, 512): 16, (14592, 2304, 768, 256): 19})
    assert ops._default_variant((14592, 768, 768, 512)) == 23
    assert ops._default_variant((15000, 768, 768, 512)) == 23          # nearest M of the same (N, K, kind), within 25 %
    assert ops._default_variant((30000, 768, 768, 512)) is None        # too far from both
    assert ops._default_variant((14592, 768, 768, 16)) is None         # another epilogue is another entry


def test_deferred_layernorm_weight_folding_is_the_same_function():
    """_PackedEncoderLn: LN(v) W^T + b == rstd (v W'^T - mean g) + h and dense + LN(v) == acc + cb + gamma x^ for the folded
    operands, checked in fp64-ish torch arithmetic on the CPU (no kernel involved: the algebra the kernels implement)."""
    import torch

    from visitron_amd.config import mini_config
    from visitron_amd.modeling import CaptionBertEncoder, _PackedEncoderLn

    cfg = mini_config()
    torch.manual_seed(3)
    enc = CaptionBertEncoder(cfg)
    for p in enc.parameters():
        torch.nn.init.normal_(p, 0.0, 0.2) if p.dim() > 1 else torch.nn.init.uniform_(p, 0.5, 1.5)
    pk = _PackedEncoderLn(enc)
    H, eps = cfg.hidden_size, cfg.layer_norm_eps
    v = torch.randn(9, H) * 2.0 + 0.7
    mean, var = v.mean(-1, keepdim=True), v.var(-1, unbiased=False, keepdim=True)
    rstd = torch.rsqrt(var + eps)
    l0, l1 = enc.layer[0], enc.layer[1]
    # layer 1's query|key|value projection consumes LN2 of layer 0
    ln = l0.output.LayerNorm
    x = torch.nn.functional.layer_norm(v, (H,), ln.weight, ln.bias, eps)
    att = l1.attention.self
    want = torch.cat([att.query(x), att.key(x), att.value(x)], -1)
    t = pk.tensors[1]
    got = rstd * (v @ t["w_qkv"].float().t() - mean * t["g_qkv"]) + t["h_qkv"]
    assert float((got - want).abs().max()) < 3e-2 * float(want.abs().max())          # W' is rounded to bf16
    # layer 1's attention.output: dense(ctx) + LN2_0(v): bias + beta and gamma as vectors
    ctx = torch.randn(9, H)
    so = l1.attention.output
    want2 = so.dense(ctx) + x
    got2 = ctx @ so.dense.weight.t() + t["cb_ao"] + t["gamma_in"] * ((v - mean) * rstd)
    assert float((got2 - want2).abs().max()) < 1e-4
    # layer 0's input is not normalised again
    t0 = pk.tensors[0]
    assert torch.equal(t0["gamma_in"], torch.ones(H)) and float(t0["cb_ao"].sub(l0.attention.output.dense.bias).abs().max()) == 0.0
    assert pk.final_gamma is not None and torch.equal(pk.final_gamma, enc.layer[-1].output.LayerNorm.weight.detach())


def test_gelu_polynomial_of_the_epilogues_meets_its_stated_error():
    """csrc/common.hpp gelu_poly4: the degree-8 fit of (Phi(x) - 0.5) / x used by the bf16 GEMM epilogues
    (hidden_act "gelu", oscar/modeling_bert.py:104-109 through pytorch-transformers' erf form).  The coefficients are read
    from the source and evaluated here in fp32 as the kernel does: |Phi - exact| <= 2.6e-5, |GELU - exact| <= 1.2e-4 on a
    dense grid."""
    import re

    import numpy as np
    from scipy.special import erf

    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "visitron_amd", "csrc", "common.hpp")).read()
    for fn in ("gelu_poly4",):
        body = src[src.index("%s(f32x4 x" % fn):]
        body = body[:body.index("\n}\n")]
        first = re.search(r"p = t \* ([0-9.e+-]+)f ([+-]) ([0-9.e+-]+)f;", body)
        coef = [float(first.group(1)), float(first.group(2) + first.group(3))]
        coef += [float(m.group(1) + m.group(2)) for m in re.finditer(r"p = p \* t ([+-]) ([0-9.e+-]+)f;", body)]
        assert len(coef) == 9, fn
        x = np.linspace(-9, 9, 400001).astype(np.float32)
        xc = np.clip(x, np.float32(-4.5), np.float32(4.5))
        t = xc * xc
        p = np.full_like(t, np.float32(coef[0]))
        for c in coef[1:]:
            p = p * t + np.float32(c)
        phi = np.maximum(xc * p + np.float32(0.5), np.float32(0))
        x64 = x.astype(np.float64)
        exact_phi = 0.5 * (1 + erf(x64 / np.sqrt(2)))
        assert np.abs(phi - exact_phi).max() <= 2.6e-5, fn
        assert np.abs(x * phi - x64 * exact_phi).max() <= 1.2e-4, fn


def test_keep_word_count_and_layout_gather_indices():
    """Host-side helpers of round 3: the size of a dropout keep buffer (VT_KEEP_WORDS in include/visitron_hip.h) and the
    un-compaction indices of a SeqLayout (padded position -> compact row, or the caller's zero row), whole and split into the
    text / region blocks."""
    import re

    from visitron_amd import ops

    assert ops.keep_words(2, 12, 228) == 2 * 12 * 8 * 256 and ops.keep_words(1, 1, 32) == 32 and ops.keep_words(1, 1, 33) == 2 * 64
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "visitron_hip.h")).read()
    assert re.search(r"#define VT_KEEP_WORDS\(B, nh, S\) \(\(int64_t\)\(B\) \* \(nh\) \* \(\(\(S\) \+ 31\) / 32\) \* \(\(\(S\) \+ 31\) / 32 \* 32\)\)", hdr)

    keep = torch.tensor([[1, 1, 1, 0, 1, 0], [1, 0, 0, 1, 1, 1]], dtype=torch.bool)
    lay = ops.SeqLayout(keep)
    assert lay.rows == 8 and lay.start.tolist() == [0, 4] and lay.length.tolist() == [4, 4]
    assert lay.index.tolist() == [0, 1, 2, 4, 6, 9, 10, 11]
    g = lay.gather_index(8)
    assert g.tolist() == [0, 1, 2, 8, 3, 8, 4, 8, 8, 5, 6, 7]
    compact = torch.arange(1, 10, dtype=torch.float32)[:, None].repeat(1, 2)      # rows 1..8 and the zero row
    compact[8] = 0
    padded = compact.index_select(0, g)
    want = torch.zeros(12, 2)
    want.index_copy_(0, lay.index, compact[:8])
    assert torch.equal(padded, want)
    text, reg = lay.gather_index_split(8, 4)
    assert text.tolist() == [0, 1, 2, 8, 4, 8, 8, 5] and reg.tolist() == [3, 8, 6, 7]
    # a layout built from ready-made parts carries the same indices
    lay2 = ops.SeqLayout.from_parts(2, 6, 8, lay.index, lay.inverse, lay.start, lay.length)
    assert lay2.gather_index(8).tolist() == g.tolist() and lay2.rows == 8
    # known row count: no synchronising nonzero, same index
    lay3 = ops.SeqLayout(keep, rows=8)
    assert torch.equal(lay3.index, lay.index) and torch.equal(lay3.inverse, lay.inverse)
