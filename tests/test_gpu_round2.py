"""GPU: parity cases added in round 2 -- the long-dialog base config (BASELINE configs[4]) at model level, the host-side
contracts the advisor flagged (optimizers that write through p.data, caches after a fused optimizer step, validation
in eval mode with grad enabled, out-of-range ids in training), the loader's resize branch on the HIP path, the device
input pipeline bit for bit, and the data-parallel engine under two ranks."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import check_close, model_pair

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
TRUNK_KEYS = ("input_ids", "attention_mask", "img_feats", "img_location_embeddings")


def _to(b, dev):
    return {k: v.to(dev) for k, v in b.items()}


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 2e-3 * (b.numel() ** 0.5)))


# ------------------------------------------------------------------------------------------------
# BASELINE configs[4]: 512 text + 144 region tokens, 12L/768d (S = 656: three key chunks in the attention
# forward, key blocks + fp32 dQ staging in the backward) -- forward AND backward against the oracle, B = 2
# ------------------------------------------------------------------------------------------------
def test_base_config_long_dialog_cfg4_forward_backward(dev):
    from oracle.modeling import PreTrainOscar as OModel
    from visitron_amd.config import BertConfig
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import make_batch
    from visitron_amd.training import PretrainEngine

    cfg = BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    ref, prod = model_pair(OModel, PreTrainOscar, cfg, seed=0, device=dev, weight_std=0.03)
    b = make_batch(cfg, 2, text_len=512, region_len=144, seed=77)
    # forward (inference path, every position)
    with torch.no_grad():
        w_seq, w_pool = ref.bert(**{k: b[k] for k in TRUNK_KEYS})[:2]
        w_scores, _, w_act = ref.heads(w_seq, w_pool)
        g_seq, g_pool = prod.bert(**{k: b[k].to(dev) for k in TRUNK_KEYS})[:2]
        g_scores = prod.mlmhead(g_seq)
        g_act = prod.next_action(g_pool)
    check_close("base S=656 sequence_output", g_seq, w_seq, 5e-2)
    check_close("base S=656 prediction_scores", g_scores, w_scores, 5e-2)
    check_close("base S=656 pooled_output", g_pool, w_pool, 5e-2)
    check_close("base S=656 action_scores", g_act, w_act, 5e-2)
    # the REFERENCE's own outputs for the same case (tests/golden/make_golden_from_reference.py base_long)
    g = np.load(os.path.join(GOLD, "ref_base_long.npz"))
    assert np.array_equal(g["in_input_ids"], b["input_ids"].numpy())
    check_close("base S=656 golden sequence_output slice", g_seq.float().cpu()[:, ::41, ::31], g["sequence_output_slice"], 5e-2)
    check_close("base S=656 golden prediction_scores slice", g_scores.float().cpu()[:, ::41, ::1009],
                g["prediction_scores_slice"], 5e-2)
    # training step on the compacted rows (the default path) against the oracle's autograd
    prod.train()
    eng = PretrainEngine(prod)
    eng.compact_min_rows = 0
    got = eng.forward_backward(_to(b, dev))
    torch.cuda.synchronize()
    assert eng.last_layout is not None and eng.last_rows < 2 * 656
    want = ref(**b)
    want[0].backward()
    for i, n in enumerate(("loss", "mask_loss", "next_loss", "token_loss")):
        check_close("base S=656 train %s" % n, float(got[i]), float(want[i]), 5e-2)
    for i, n in ((4, "words_acc"), (5, "action_acc"), (6, "token_acc")):
        check_close("base S=656 train %s" % n, float(got[i]), float(want[i]), 1e-6)
    for i in range(4):
        check_close("base S=656 golden tuple7[%d]" % i, float(got[i]), float(g["tuple7"][i]), 5e-2)
    errs = {n: _rel(p.grad, dict(ref.named_parameters())[n].grad) for n, p in prod.named_parameters()}
    worst = max(errs, key=errs.get)
    check_close("base S=656 grads worst rel-L2 (%s)" % worst, errs[worst], 0.0, 0.035)   # measured 1.6 %


# ------------------------------------------------------------------------------------------------
# the advisor's findings
# ------------------------------------------------------------------------------------------------
class _DataWritingAdamW(object):
    """The pytorch-transformers AdamW the reference uses (pretrain.py:128-130) updates through `p.data`
    (`p.data.addcdiv_`, `p.data.add_`): parameter values change while `p._version` stays put."""

    def __init__(self, params, lr, wd=0.0, eps=1e-8, b1=0.9, b2=0.999):
        self.params, self.lr, self.wd, self.eps, self.b1, self.b2 = list(params), lr, wd, eps, b1, b2
        self.m = [torch.zeros_like(p.data) for p in self.params]
        self.v = [torch.zeros_like(p.data) for p in self.params]
        self.t = 0

    def step(self):
        self.t += 1
        ss = self.lr * (1 - self.b2 ** self.t) ** 0.5 / (1 - self.b1 ** self.t)
        for p, m, v in zip(self.params, self.m, self.v):
            if p.grad is None:
                continue
            g = p.grad.data
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            p.data.addcdiv_(m, v.sqrt().add_(self.eps), value=-ss)
            if self.wd:
                p.data.add_(p.data, alpha=-self.lr * self.wd)


def test_reference_loop_with_an_optimizer_that_writes_through_p_data(dev):
    """INTEGRATION.md's drop-in loop with the reference's own optimizer style: the bf16 copies the kernels read must
    follow updates that do not bump `_version` -- the loss must fall and track the oracle trained the same way."""
    from oracle.modeling import PreTrainOscar as OModel
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import make_batch

    cfg = mini_config()
    ref, prod = model_pair(OModel, PreTrainOscar, cfg, seed=12, device=dev)
    prod.train()
    ref.train()
    o_ref = _DataWritingAdamW(ref.parameters(), 2e-3)
    o_prod = _DataWritingAdamW(prod.parameters(), 2e-3)
    b = make_batch(cfg, 4, text_len=16, region_len=8, seed=2)
    bd = _to(b, dev)
    v0 = None
    lr_, lh = [], []
    for it in range(8):
        if it == 1:
            v0 = [p._version for p in prod.parameters()]
        ref.zero_grad()
        l_ref = ref(**b)[0]
        l_ref.backward()
        o_ref.step()
        prod.zero_grad()
        loss = prod(**bd)[0]
        loss.backward()
        o_prod.step()
        lr_.append(float(l_ref))
        lh.append(float(loss))
    assert [p._version for p in prod.parameters()] == v0, "this optimizer must not bump versions (that is the point)"
    assert lh[-1] < lh[0] - 0.5, lh                       # it learns (a stale mirror would keep the loss flat)
    check_close("p.data-optimizer loop: loss trajectory vs oracle", torch.tensor(lh), torch.tensor(lr_), 0.15)
    # ...and the inference path sees the updated weights too
    prod.eval()
    ref.eval()
    with torch.no_grad():
        check_close("p.data-optimizer loop: eval loss after training", float(prod(**bd)[0]), float(ref(**b)[0]), 5e-2)


def test_eval_forward_after_engine_steps_sees_the_new_weights(dev):
    """train -> eval -> train -> eval: the fused AdamW writes the slab through raw pointers, so the inference path's
    packed copies (encoder layers, region projection) must be invalidated explicitly."""
    from oracle.modeling import PreTrainOscar as OModel
    from oracle.optim import AdamW, grouped_parameters
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import make_batch
    from visitron_amd.training import PretrainEngine

    cfg = mini_config()
    ref, prod = model_pair(OModel, PreTrainOscar, cfg, seed=8, device=dev)
    eng = PretrainEngine(prod, lr=2e-3, weight_decay=0.05, schedule="constant", warmup_steps=0)
    opt = AdamW(grouped_parameters(ref, 0.05), lr=2e-3, eps=1e-8)
    b = make_batch(cfg, 4, text_len=16, region_len=8, seed=2)
    bd = _to(b, dev)
    evals = []
    for round_ in range(2):
        prod.eval()
        ref.eval()
        with torch.no_grad():                      # an eval forward BEFORE the steps fills the packed-weight caches
            e_p, e_r = float(prod(**bd)[0]), float(ref(**b)[0])
        check_close("train/eval interleave: eval loss, round %d" % round_, e_p, e_r, 5e-2)
        evals.append(e_p)
        prod.train()
        ref.train()
        for _ in range(4):
            ref.zero_grad()
            ref(**b)[0].backward()
            opt.step()
            eng.train_step(bd)
    prod.eval()
    ref.eval()
    with torch.no_grad():
        e_p, e_r = float(prod(**bd)[0]), float(ref(**b)[0])
    check_close("train/eval interleave: eval loss after 8 steps", e_p, e_r, 5e-2)
    # forward parity on the product's OWN current weights (the two models have trained apart by bf16 noise: copying
    # the weights over isolates "does the inference path see the weights the fused AdamW just wrote")
    ref.load_state_dict({k: v.cpu() for k, v in prod.state_dict().items()})
    with torch.no_grad():
        seq_p = prod.bert(**{k: bd[k] for k in TRUNK_KEYS})[0]
        seq_r = ref.bert(**{k: b[k] for k in TRUNK_KEYS})[0]
    check_close("train/eval interleave: sequence_output on the stepped weights", seq_p, seq_r, 5e-2)
    assert e_p < evals[0] - 0.5 and evals[1] < evals[0] - 0.2, (evals, e_p)   # a stale cache would repeat evals[0]


def test_validation_in_eval_mode_with_grad_enabled_is_forward_only(dev):
    """pretrain.val() (pretrain.py:291, 469-481) calls model(**batch) in eval() with no torch.no_grad: that must not
    build the training engine or run a backward pass; `.backward()` still works if someone calls it."""
    from oracle.modeling import PreTrainOscar as OModel
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import make_batch

    cfg = mini_config()
    ref, prod = model_pair(OModel, PreTrainOscar, cfg, seed=14, device=dev)
    b = make_batch(cfg, 3, text_len=14, region_len=6, seed=9)
    out = prod(**_to(b, dev))                              # eval mode, grad enabled
    assert getattr(prod, "_vt_engine", None) is None, "validation built the training engine"
    want = ref(**b)
    for i in range(4):
        check_close("eval+grad forward tuple[%d]" % i, float(out[i]), float(want[i]), 5e-2)
    assert out[0].requires_grad
    out[0].backward()                                      # lazily: engine + HIP forward/backward now
    assert getattr(prod, "_vt_engine", None) is not None
    want[0].backward()
    wg = dict(ref.named_parameters())
    errs = {n: _rel(p.grad, wg[n].grad) for n, p in prod.named_parameters()}
    worst = max(errs, key=errs.get)
    check_close("eval+grad lazy backward grads worst rel-L2 (%s)" % worst, errs[worst], 0.0, 0.02)   # measured 0.7 - 1.0 %


def test_training_step_raises_index_error_and_clears_unsupervised_heads(dev):
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import make_batch
    from visitron_amd.training import PretrainEngine

    cfg = mini_config()
    prod = PreTrainOscar(cfg).to(dev).train()
    eng = PretrainEngine(prod)
    b = _to(make_batch(cfg, 3, text_len=12, region_len=6, seed=3), dev)
    eng.forward_backward(b)
    tok_g = prod.token_head[0].weight.grad
    tr_g = prod.mlmhead.predictions.transform.dense.weight.grad
    assert float(tok_g.abs().sum()) > 0 and float(tr_g.abs().sum()) > 0
    c = {k: v.clone() for k, v in b.items()}
    c["token_labels"].fill_(-1)
    c["labels"].fill_(-1)
    out = eng.forward_backward(c)                          # no supervised row: NaN losses, as the reference's criterion
    assert float(out[1]) != float(out[1]) and float(out[3]) != float(out[3])
    assert float(tok_g.abs().sum()) == 0.0 and float(tr_g.abs().sum()) == 0.0, "stale head gradients survived"
    bad = {k: v.clone() for k, v in b.items()}
    bad["input_ids"][1, 3] = cfg.vocab_size + 5
    with pytest.raises(IndexError):
        eng.forward_backward(bad)
    eng.forward_backward(b)                                # the engine is still usable afterwards


# ------------------------------------------------------------------------------------------------
# a8: load_oscar_weights' resize branch (model_utils.py:101-109): +3 word embeddings (the decoder stays at the old
# vocabulary and is no longer tied), longer position table, +4 token types -- through the HIP forward and the engine
# ------------------------------------------------------------------------------------------------
def test_resized_embeddings_untied_decoder_forward_and_training(dev):
    from oracle.modeling import PreTrainOscar as OModel
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import make_batch
    from visitron_amd.training import PretrainEngine

    cfg = mini_config()
    ref, prod = model_pair(OModel, PreTrainOscar, cfg, seed=17, device="cpu")
    sizes = {"word_embeddings": cfg.vocab_size + 3, "position_embeddings": 96, "token_type_embeddings": cfg.type_vocab_size + 4}
    ref.resize_embeddings(sizes)
    prod.resize_embeddings(sizes)
    prod.load_state_dict(ref.state_dict())                 # the fresh rows are random: copy the oracle's
    prod = prod.to(dev)
    assert prod.mlmhead.predictions.decoder.weight is not prod.bert.embeddings.word_embeddings.weight
    assert prod.mlmhead.predictions.decoder.weight.shape[0] == cfg.vocab_size
    B, T, R = 3, 80, 9                                     # T beyond the old 64-position table
    b = make_batch(cfg, B, text_len=T, region_len=R, seed=6)
    b["input_ids"][0, 5] = cfg.vocab_size + 1              # one of the new tokens
    b["input_ids"][2, 7] = cfg.vocab_size + 2
    b["token_type_ids"] = torch.randint(0, cfg.type_vocab_size + 4, (B, T), generator=torch.Generator().manual_seed(1))
    b["labels"][b["labels"] >= cfg.vocab_size] = -1
    with torch.no_grad():
        want = ref(**b)
        got = prod(**_to(b, dev))
    for i in range(4):
        check_close("resized: eval tuple[%d]" % i, float(got[i]), float(want[i]), 5e-2)
    prod.train()
    ref.train()
    eng = PretrainEngine(prod, lr=1e-3)
    eng.compact_min_rows = 0
    assert "mlmhead.predictions.decoder.weight" in eng.flat.off          # its own slab entry now
    got = eng.forward_backward(_to(b, dev))
    want = ref(**b)
    want[0].backward()
    for i in range(4):
        check_close("resized: train tuple[%d]" % i, float(got[i]), float(want[i]), 5e-2)
    wg = dict(ref.named_parameters())
    errs = {n: _rel(p.grad, wg[n].grad) for n, p in prod.named_parameters()}
    worst = max(errs, key=errs.get)
    check_close("resized: grads worst rel-L2 (%s)" % worst, errs[worst], 0.0, 0.02)   # measured 0.7 - 1.0 %
    # the untied tables really got their own gradients
    assert _rel(prod.mlmhead.predictions.decoder.weight.grad, wg["mlmhead.predictions.decoder.weight"].grad) < 0.02
    assert _rel(prod.bert.embeddings.word_embeddings.weight.grad, wg["bert.embeddings.word_embeddings.weight"].grad) < 0.02
    eng.optimizer_step()
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------
# f2: the device input pipeline, bit for bit
# ------------------------------------------------------------------------------------------------
def test_device_input_pipeline_bit_exact_against_per_item_restatement(dev):
    from test_data_pipeline import check_against_per_item_restatement

    check_against_per_item_restatement(dev)


# ------------------------------------------------------------------------------------------------
# e: the ENGINE under two ranks (both on this one GPU, gloo for the collectives -- a rehearsal of the RCCL run the
# driver makes on a whole node): rank-averaged gradients, the loss /= world quirk, identical weights after the step
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("comm_dtype", ["fp32", "bf16"])
def test_engine_under_two_ranks_matches_single_rank_accumulation(dev, tmp_path, comm_dtype):
    """comm_dtype: what travels in the all-reduce -- the fp32 gradient slab itself (exact comparison) or its bf16
    communication copy (the default: half the bytes over xGMI; compared at bf16 resolution)."""
    script = os.path.join(ROOT, "tests", "dp_engine_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29671" if comm_dtype == "fp32" else "29673", script, str(tmp_path), comm_dtype]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    got = torch.load(os.path.join(str(tmp_path), "rank0.pt"))
    other = torch.load(os.path.join(str(tmp_path), "rank1.pt"))
    # one rank, same engine code, the two shards accumulated: g = (g(shard0) + g(shard1)) with the loss of each shard
    # scaled by 1/world (pretrain.py:170), then DDP's mean over ranks (another 1/world) inside the optimizer step
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import deterministic_state_dict, make_batch
    from visitron_amd.training import PretrainEngine

    from visitron_amd import ops

    cfg = mini_config(num_hidden_layers=4)
    m = PreTrainOscar(cfg)
    m.load_state_dict(deterministic_state_dict(m, seed=5))
    m.tie_weights()
    m = m.to(dev).eval()                                   # eval: no dropout, so shards are comparable across processes
    ops.force_gemm_variant(1)                              # the workers' kernel variants (same summation order)
    ops.set_wgrad_kernel(-8)
    try:
        eng = PretrainEngine(m, lr=1e-3, weight_decay=0.05, schedule="constant", warmup_steps=0)
        eng.compact_min_rows = 0
        world = 2
        outs = []
        for r_ in range(world):
            shard = _to(make_batch(cfg, 3, text_len=20, region_len=10, seed=100 + r_), dev)
            outs.append([float(v) for v in eng.forward_backward(shard, grad_scale=1.0 / world, accumulate=(r_ > 0))])
        torch.cuda.synchronize()
    finally:
        ops.force_gemm_variant(None)
        ops.set_wgrad_kernel(0)
    g_sum = eng.flat.g.clone()
    gmax = float(g_sum.abs().max())
    if comm_dtype == "fp32":
        check_close("2-rank engine (fp32 all-reduce): gradient slab vs 1-rank accumulation", got["g"], g_sum.cpu(),
                    1e-5 * gmax + 1e-9)
    else:   # each rank's contribution is rounded to bf16 (relative 2^-9 of ITS magnitude) and so is their sum
        check_close("2-rank engine (bf16 all-reduce): gradient slab, relative L2", got["g"], g_sum.cpu(), 6e-3, kind="rel_l2")
        check_close("2-rank engine (bf16 all-reduce): gradient slab, max |error| / max |g|",
                    float((got["g"] - g_sum.cpu()).abs().max()) / gmax, 0.0, 2.0 ** -7)
    assert torch.equal(got["g"], other["g"]), "ranks disagree on the all-reduced gradients"
    p_before = eng.flat.p.detach().cpu().clone()
    eng.optimizer_step(grad_scale=1.0 / world)
    torch.cuda.synchronize()
    if comm_dtype == "fp32":
        check_close("2-rank engine (fp32 all-reduce): weights after AdamW vs 1-rank", got["p"], eng.flat.p.cpu(), 2e-6)
    else:
        # Adam's first step is lr * sign-like (m / sqrt(v) = +-1): where a gradient is all but zero its bf16 rounding can
        # flip the step, so single weights may differ by 2 lr; compared as the update's relative L2
        check_close("2-rank engine (bf16 all-reduce): AdamW update vs 1-rank, relative L2", got["p"] - p_before,
                    eng.flat.p.cpu() - p_before, 5e-2, kind="rel_l2")
    assert torch.equal(got["p"], other["p"]), "ranks diverged after the optimizer step"
    for r_, rec in enumerate((got, other)):
        for i in range(4):
            assert abs(rec["out"][i] - outs[r_][i]) < 1e-5, (r_, i, rec["out"], outs[r_])
    # the 7 logged scalars: mean over ranks, one message
    for i in range(7):
        assert abs(got["metrics"][i] - 0.5 * (outs[0][i] + outs[1][i])) < 1e-5


# ------------------------------------------------------------------------------------------------
# f1: the inference 7-tuple from the labelled rows only, chunked (no [B, S, vocab] logits tensor)
# ------------------------------------------------------------------------------------------------
def test_eval_tuple_from_labelled_rows_in_chunks(dev):
    from oracle.modeling import PreTrainOscar as OModel
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import make_batch

    cfg = mini_config()
    ref, prod = model_pair(OModel, PreTrainOscar, cfg, seed=19, device=dev)
    b = make_batch(cfg, 6, text_len=30, region_len=12, seed=4)
    n_lab = int((b["labels"] != -1).sum())
    assert n_lab > 7
    prod.LOSS_ROWS_PER_CHUNK = 5                          # several ragged chunks
    with torch.no_grad():
        want = ref(**b)
        got = prod(**_to(b, dev))
    for i in range(4):
        check_close("eval 7-tuple from labelled rows [%d]" % i, float(got[i]), float(want[i]), 5e-2)
    for i in range(4, 7):
        check_close("eval 7-tuple from labelled rows [%d]" % i, float(got[i]), float(want[i]), 1e-6)
    # corners: no label at all (NaN like the criterion), a target outside the vocabulary (IndexError like the criterion)
    c = {k: v.clone() for k, v in b.items()}
    c["labels"].fill_(-1)
    with torch.no_grad():
        g2, w2 = prod(**_to(c, dev)), ref(**c)
    assert float(g2[1]) != float(g2[1]) and float(w2[1]) != float(w2[1])
    assert float(g2[4]) != float(g2[4]) and float(w2[4]) != float(w2[4])
    c = {k: v.clone() for k, v in b.items()}
    c["labels"][0, 3] = cfg.vocab_size
    with torch.no_grad(), pytest.raises(IndexError):
        prod(**_to(c, dev))


# ------------------------------------------------------------------------------------------------
# training with head_mask and with 3-D (per-query) attention masks (oscar/modeling_bert.py:65-66; encoder.py:226-229,
# 248-265): losses and every gradient against the oracle's autograd
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("what", ["head_mask", "mask3d", "both_with_dropout"])
def test_training_with_head_mask_and_per_query_masks(dev, what):
    from helpers import inject_dropout_masks
    from oracle.modeling import PreTrainOscar as OModel
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import make_batch
    from visitron_amd.training import PretrainEngine

    p = 0.1 if what == "both_with_dropout" else 0.0
    cfg = mini_config(num_hidden_layers=3, hidden_dropout_prob=p, attention_probs_dropout_prob=p)
    ref, prod = model_pair(OModel, PreTrainOscar, cfg, seed=23, device=dev)
    prod.train()
    eng = PretrainEngine(prod)
    eng.compact_min_rows = 0
    B, T, R = 3, 22, 9
    S = T + R
    b = make_batch(cfg, B, text_len=T, region_len=R, seed=15)
    g = torch.Generator().manual_seed(5)
    hm = None
    if what != "mask3d":
        hm = torch.tensor([[1.0, 0.5], [0.0, 1.0], [1.5, 0.25]])           # [layers, heads]
    if what != "head_mask":
        m3 = (torch.rand(B, S, S, generator=g) > 0.25).float() * b["attention_mask"][:, None, :].float()
        m3[:, :, 0] = 1.0
        b["attention_mask"] = m3
    got = eng.forward_backward(_to(b, dev), head_mask=None if hm is None else hm.to(dev))
    torch.cuda.synchronize()
    assert eng.last_layout is None                                        # these forms keep the padded rows
    ref.train()
    if p > 0:
        inject_dropout_masks(ref, p, p, eng.last_drop_seed, B, T, R, device=dev)
    want = ref(**b, head_mask=hm)
    want[0].backward()
    for i in range(4):
        check_close("train %s tuple[%d]" % (what, i), float(got[i]), float(want[i]), 5e-2)
    wg = dict(ref.named_parameters())
    errs = {n: _rel(p_.grad, wg[n].grad) for n, p_ in prod.named_parameters()}
    worst = max(errs, key=errs.get)
    check_close("train %s grads worst rel-L2 (%s)" % (what, worst), errs[worst], 0.0, 0.02)   # measured 0.7 - 1.0 %
    # ... and through the module's own forward (the autograd bridge)
    if what == "head_mask":
        prod.zero_grad()
        out = prod(**_to(b, dev), head_mask=hm.to(dev))
        out[0].backward()
        errs = {n: _rel(p_.grad, wg[n].grad) for n, p_ in prod.named_parameters()}
        assert max(errs.values()) < 0.02, max(errs, key=errs.get)


# ------------------------------------------------------------------------------------------------
# ragged and degenerate inputs (encoder.py:215-241; data_loader_pretrain.py:666-690): one position, no regions, a
# sequence whose mask is zero everywhere (the reference then spreads the softmax evenly: every key carries the same
# -10000), masks with holes, text at the position table's maximum -- inference and one training step
# ------------------------------------------------------------------------------------------------
def _ragged_cases(cfg):
    from visitron_amd.synth import make_batch

    cases = []
    b = make_batch(cfg, 1, text_len=1, region_len=0, seed=1, with_labels=False)
    cases.append(("one_position", b))
    b = make_batch(cfg, 3, text_len=9, region_len=0, seed=2)
    cases.append(("text_only", b))
    b = make_batch(cfg, 4, text_len=14, region_len=6, seed=3)
    b["attention_mask"][2].zero_()
    cases.append(("one_sequence_fully_masked", b))
    b = make_batch(cfg, 3, text_len=14, region_len=6, seed=4)
    b["attention_mask"][:, 3] = 0
    b["attention_mask"][1, 10:16] = 0
    b["attention_mask"][1, 18] = 1
    cases.append(("masks_with_holes", b))
    b = make_batch(cfg, 2, text_len=cfg.max_position_embeddings, region_len=5, seed=5)
    cases.append(("text_at_position_table_maximum", b))
    b = make_batch(cfg, 2, text_len=10, region_len=4, seed=6)
    b["attention_mask"] = b["attention_mask"].float() * 0.5 + 0.25          # arbitrary numeric masks are taken literally
    cases.append(("fractional_mask_values", b))
    return cases


def test_ragged_and_degenerate_inputs_inference(dev):
    from oracle.modeling import PreTrainOscar as OModel
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import PreTrainOscar

    cfg = mini_config()
    ref, prod = model_pair(OModel, PreTrainOscar, cfg, seed=29, device=dev)
    for name, b in _ragged_cases(cfg):
        trunk = {k: b[k] for k in TRUNK_KEYS if k in b}
        with torch.no_grad():
            want = ref.bert(**trunk)
            got = prod.bert(**_to(trunk, dev))
            want7 = ref(**{**b, **trunk}) if "img_feats" in trunk else None
            got7 = prod(**_to({**b, **trunk}, dev)) if "img_feats" in trunk else None
        torch.cuda.synchronize()
        assert got[0].shape == want[0].shape, name
        check_close("ragged %s sequence_output" % name, got[0], want[0], 5e-2)
        check_close("ragged %s pooled_output" % name, got[1], want[1], 5e-2)
        if want7 is not None:
            for i in range(4):
                assert _both_nan_or_close(got7[i], want7[i], 5e-2), (name, i, float(got7[i]), float(want7[i]))
            for i in range(4, 7):
                assert _both_nan_or_close(got7[i], want7[i], 1e-6), (name, i, float(got7[i]), float(want7[i]))


def _both_nan_or_close(a, b, tol):
    a, b = float(torch.as_tensor(a).detach()), float(torch.as_tensor(b).detach())
    return (a != a and b != b) or abs(a - b) < tol


def test_ragged_and_degenerate_inputs_training_step(dev):
    from oracle.modeling import PreTrainOscar as OModel
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.training import PretrainEngine

    cfg = mini_config()
    ref, prod = model_pair(OModel, PreTrainOscar, cfg, seed=37, device=dev)
    prod.train()
    eng = PretrainEngine(prod)
    wg = dict(ref.named_parameters())
    for compact_min in (0, 1 << 30):                                        # compacted rows / padded rows
        eng.compact_min_rows = compact_min
        for name, b in _ragged_cases(cfg):
            if "img_feats" not in b:
                continue                                                    # PreTrainOscar's callers always pass regions
            ref.zero_grad()
            want = ref(**b)
            want[0].backward()
            got = eng.forward_backward(_to(b, dev))
            torch.cuda.synchronize()
            tag = "ragged train %s%s" % (name, "" if compact_min == 0 else " padded")
            for i in range(4):
                assert _both_nan_or_close(got[i], want[i], 5e-2), (tag, i, float(got[i]), float(want[i]))
            for i in range(4, 7):
                assert _both_nan_or_close(got[i], want[i], 1e-6), (tag, i, float(got[i]), float(want[i]))
            errs = {n: _rel(p_.grad, wg[n].grad) for n, p_ in prod.named_parameters()}
            worst = max(errs, key=errs.get)
            check_close(tag + " grads worst rel-L2", errs[worst], 0.0, 0.03)


def test_train_mode_under_no_grad_runs_the_forward_with_dropout(dev):
    """model.train() inside torch.no_grad() (dropout on, no graph): the engine's forward alone -- a 7-tuple that differs
    from call to call (two dropout draws), no gradient anywhere; with dropout 0 it is the eval-mode result."""
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import make_batch

    cfg = mini_config(hidden_dropout_prob=0.2, attention_probs_dropout_prob=0.2)
    torch.manual_seed(0)
    m = PreTrainOscar(cfg).to(dev).train()
    b = _to(make_batch(cfg, 4, text_len=20, region_len=8, seed=3), dev)
    with torch.no_grad():
        o1, o2 = m(**b), m(**b)
        s1, s2 = m.bert(b["input_ids"], attention_mask=b["attention_mask"][:, :20])[0], \
            m.bert(b["input_ids"], attention_mask=b["attention_mask"][:, :20])[0]
    assert all(float(v) == float(v) for v in o1[:4]) and float(o1[0]) != float(o2[0])
    assert not o1[0].requires_grad and all(p.grad is None for p in m.parameters())
    assert not s1.requires_grad and float((s1 - s2).abs().max()) > 0
    m.eval()
    with torch.no_grad():
        e = m(**b)
    assert abs(float(e[0]) - float(o1[0])) < 1.0          # same model, dropout noise only
    cfg0 = mini_config()
    m0 = PreTrainOscar(cfg0).to(dev)
    with torch.no_grad():
        a = m0.train()(**b)
        c = m0.eval()(**b)
    check_close("train() under no_grad, dropout 0, equals eval()", float(a[0]), float(c[0]), 1e-5)


def test_engine_state_dict_resumes_the_run(dev):
    """PretrainEngine.state_dict / load_state_dict (AdamW moments by parameter name, step / scheduler / dropout counters):
    two steps, checkpoint, two more steps == a fresh model + engine restored from the checkpoint taking the same two steps
    (dropout on: the seeds continue too)."""
    from visitron_amd.config import mini_config
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.synth import make_batch
    from visitron_amd.training import PretrainEngine

    cfg = mini_config(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    torch.manual_seed(11)
    a = PreTrainOscar(cfg).to(dev).train()
    ea = PretrainEngine(a, lr=2e-3, warmup_steps=3, t_total=10)
    batches = [_to(make_batch(cfg, 4, text_len=16, region_len=6, seed=40 + i), dev) for i in range(4)]
    for i in range(2):
        ea.train_step(batches[i])
    ck_model = {k: v.detach().clone() for k, v in a.state_dict().items()}
    ck_opt = ea.state_dict()
    la = [float(ea.train_step(batches[i])[0]) for i in (2, 3)]
    torch.manual_seed(99)                                   # a different process: different initial seed, fresh init
    b = PreTrainOscar(cfg).to(dev).train()
    b.load_state_dict(ck_model)
    b.tie_weights()
    eb = PretrainEngine(b, lr=1.0)                          # hyper-parameters come from the checkpoint
    eb.load_state_dict(ck_opt)
    assert eb.step_count == 2 and eb.lr == 2e-3 and eb.warmup_steps == 3
    lb = [float(eb.train_step(batches[i])[0]) for i in (2, 3)]
    for x, y in zip(la, lb):
        check_close("resumed engine: loss", x, y, 1e-5)
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    worst = max(float((pa[n].detach() - pb[n].detach()).abs().max()) for n in pa)
    check_close("resumed engine: weights after two more steps", worst, 0.0, 1e-6)
    with pytest.raises(KeyError):
        eb.load_state_dict(dict(ck_opt, state={}))
