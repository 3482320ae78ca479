"""CPU tests of the round-4 host logic: the GEMM policy under more than one rank, the early all-reduce ranges of the head
gradients (tasks/viewpoint_select/pretrain.py:96-102,191: DDP's bucketed all-reduce overlapped with the backward)."""
import pytest
import torch


def test_multi_rank_gemm_policy_is_opt_in():
    from visitron_amd.ops import multi_rank_gemm_policy

    k, text = multi_rank_gemm_policy({})
    assert k == 0 and "off" in text
    k, _ = multi_rank_gemm_policy({"NCCL_MAX_NCHANNELS": "16"})
    assert k == 0                        # a pinned channel count no longer switches the reservation on by inference
    k, text = multi_rank_gemm_policy({"VT_GEMM_RESERVE_CUS": "0", "NCCL_MAX_NCHANNELS": "16"})
    assert k == 0 and "VT_GEMM_RESERVE_CUS=0" in text
    k, text = multi_rank_gemm_policy({"VT_GEMM_RESERVE_CUS": "24"})
    assert k == 24 and "CUs - 24" in text
    k, _ = multi_rank_gemm_policy({"VT_GEMM_RESERVE_CUS": "4096"})
    assert k == 0                        # not a plausible CU count: nothing is reserved on its word
    k, _ = multi_rank_gemm_policy({"VT_GEMM_RESERVE_CUS": "many"})
    assert k == 0


@pytest.mark.parametrize("tied", [True, False])
def test_head_gradient_ranges_are_aligned_disjoint_from_the_layers_and_cover_their_parameters(tied):
    from visitron_amd.config import mini_config
    from visitron_amd.distributed import complement_ranges
    from visitron_amd.modeling import PreTrainOscar
    from visitron_amd.training import PretrainEngine

    m = PreTrainOscar(mini_config(num_hidden_layers=3))
    if not tied:
        m.resize_embeddings({"word_embeddings": m.config.vocab_size + 3})
    e = PretrainEngine(m)
    pr, lin = m.mlmhead.predictions, m.token_head[0]
    dec_tied = pr.decoder.weight is m.bert.embeddings.word_embeddings.weight
    assert dec_tied == tied
    params = [pr.transform.dense.weight, pr.transform.dense.bias, pr.transform.LayerNorm.weight, pr.transform.LayerNorm.bias,
              pr.bias, lin.weight, lin.bias, m.next_action.linear.weight, m.next_action.linear.bias,
              m.bert.pooler.dense.weight, m.bert.pooler.dense.bias] + ([] if tied else [pr.decoder.weight])
    rng = e._param_ranges(params)
    f = e.flat
    cover = torch.zeros(f.total, dtype=torch.int32)
    for s, t in rng:
        assert s % 8 == 0 and t % 8 == 0 and t > s
        cover[s:t] += 1
    assert int(cover.max()) == 1
    for prm in params:                                   # every element of every head parameter is inside a range
        o, cnt, _ = f.off[e._name_of(prm)]
        assert int(cover[o:o + cnt].min()) == 1
    owned = torch.zeros(f.total, dtype=torch.bool)
    for n, p_, o, cnt, _ in f.entries:
        if not any(p_ is q for q in params):
            owned[o:o + cnt] = True
    assert not bool((owned & (cover > 0)).any())        # ... and no element of any other parameter
    # together with the layer chunks and the complement the slab is covered exactly once
    done = list(rng)
    for l in range(3):
        done += [(e.layer_ranges[l][k][0], e.layer_ranges[l][k][1]) for k in (0, 1)]
    cover2 = torch.zeros(f.total, dtype=torch.int32)
    for s, t in done + complement_ranges(f.total, done):
        cover2[s:t] += 1
    assert int(cover2.min()) == 1 and int(cover2.max()) == 1
