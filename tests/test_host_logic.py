"""CPU: host-side logic of the drop-in modules (no kernels are launched)."""
import os

import pytest
import torch

from visitron_amd.config import BertConfig, mini_config
from visitron_amd.modeling import (
    MODEL_CLASS, BertImgModelwithLocationEmbeds, CaptionBertEncoder, PreTrainOscar, _additive_mask_2d, _head_scale,
)
from visitron_amd.synth import deterministic_state_dict, make_batch, viewpoint_loc_embedding


def test_state_dict_keys_are_the_reference_checkpoint_keys():
    cfg = BertConfig(num_hidden_layers=2, use_img_layernorm=True, img_layer_norm_eps=1e-12)
    keys = set(PreTrainOscar(cfg).state_dict())
    expected = {
        "bert.embeddings.word_embeddings.weight", "bert.embeddings.position_embeddings.weight",
        "bert.embeddings.token_type_embeddings.weight", "bert.embeddings.LayerNorm.weight", "bert.embeddings.LayerNorm.bias",
        "bert.pooler.dense.weight", "bert.pooler.dense.bias", "bert.img_embedding.weight", "bert.img_embedding.bias",
        "bert.location_embeds.weight", "bert.location_embeds.bias", "bert.LayerNorm.weight", "bert.LayerNorm.bias",
        "next_action.linear.weight", "next_action.linear.bias", "mlmhead.predictions.bias",
        "mlmhead.predictions.transform.dense.weight", "mlmhead.predictions.transform.dense.bias",
        "mlmhead.predictions.transform.LayerNorm.weight", "mlmhead.predictions.transform.LayerNorm.bias",
        "mlmhead.predictions.decoder.weight", "token_head.0.weight", "token_head.0.bias",
    }
    for i in range(2):
        p = "bert.encoder.layer.%d." % i
        for leaf in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense",
                     "attention.output.LayerNorm", "intermediate.dense", "output.dense", "output.LayerNorm"):
            expected |= {p + leaf + ".weight", p + leaf + ".bias"}
    assert keys == expected


def test_oracle_and_product_share_keys_shapes_and_tying():
    from oracle.modeling import PreTrainOscar as OModel

    cfg = mini_config()
    o, p = OModel(cfg), PreTrainOscar(cfg)
    so, sp = o.state_dict(), p.state_dict()
    assert list(so) == list(sp)
    assert all(so[k].shape == sp[k].shape for k in so)
    assert p.mlmhead.predictions.decoder.weight is p.bert.embeddings.word_embeddings.weight
    assert p.bert.img_embedding.weight.shape == (cfg.hidden_size, cfg.img_feature_dim)
    assert p.bert.location_embeds.weight.shape == (cfg.hidden_size, 128)
    assert p.next_action.linear.weight.shape == (cfg.action_space, cfg.hidden_size)
    assert p.token_head[0].weight.shape == (cfg.detector_classes, cfg.hidden_size)


def test_no_decay_split_names_exist():
    """pretrain.py:109-127 splits parameters on 'bias' / 'LayerNorm.weight' substrings."""
    names = [n for n, _ in PreTrainOscar(mini_config()).named_parameters()]
    assert any("LayerNorm.weight" in n for n in names) and any(n.endswith("bias") for n in names)
    assert "mlmhead.predictions.decoder.weight" not in names  # tied: reported once, under the embedding


def test_init_weights_statistics():
    torch.manual_seed(0)
    m = PreTrainOscar(BertConfig(num_hidden_layers=1))
    w = m.bert.encoder.layer[0].intermediate.dense.weight
    assert abs(float(w.std()) - 0.02) < 1e-3 and abs(float(w.mean())) < 1e-3
    assert float(m.bert.encoder.layer[0].intermediate.dense.bias.abs().max()) == 0.0
    ln = m.bert.encoder.layer[0].output.LayerNorm
    assert bool((ln.weight == 1).all()) and bool((ln.bias == 0).all())


def test_save_and_from_pretrained_round_trip(tmp_path):
    cfg = mini_config()
    m = PreTrainOscar(cfg)
    m.load_state_dict(deterministic_state_dict(m, seed=9))
    m.save_pretrained(str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == ["config.json", "pytorch_model.bin"]
    m2 = PreTrainOscar.from_pretrained(str(tmp_path))
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    assert m2.config.detector_classes == cfg.detector_classes and m2.config.img_feature_dim == cfg.img_feature_dim
    assert m2.mlmhead.predictions.decoder.weight is m2.bert.embeddings.word_embeddings.weight
    # the fine-tune entry point pulls the trunk out of a full-model checkpoint (train.py:47)
    trunk = BertImgModelwithLocationEmbeds.from_pretrained(str(tmp_path), config=BertConfig.from_pretrained(str(tmp_path)))
    assert torch.equal(trunk.pooler.dense.weight, m.bert.pooler.dense.weight)


def test_config_round_trip_and_code_set_fields(tmp_path):
    cfg = BertConfig(hidden_dropout_prob=0.3, img_feature_dim=2054, action_space=36, detector_classes=1601, foo=7)
    cfg.save_pretrained(str(tmp_path))
    c2 = BertConfig.from_pretrained(str(tmp_path))
    assert c2.to_dict() == cfg.to_dict() and c2.foo == 7
    assert c2.layer_norm_eps == 1e-12 and c2.hidden_act == "gelu" and c2.type_vocab_size == 2
    assert MODEL_CLASS["PreTrainOscar"][1] is PreTrainOscar


def test_resize_embeddings_semantics():
    cfg = mini_config()
    m = PreTrainOscar(cfg)
    old = m.bert.embeddings.word_embeddings.weight.detach().clone()
    m.resize_embeddings({"word_embeddings": cfg.vocab_size + 3, "position_embeddings": 80, "token_type_embeddings": 6})
    e = m.bert.embeddings
    assert e.word_embeddings.weight.shape[0] == cfg.vocab_size + 3
    assert torch.equal(e.word_embeddings.weight[: cfg.vocab_size], old)
    assert e.position_embeddings.weight.shape[0] == 80 and e.token_type_embeddings.weight.shape[0] == 6
    assert m.mlmhead.predictions.decoder.weight.shape[0] == cfg.vocab_size  # reference does not re-tie
    with pytest.raises(AssertionError):
        m.resize_embeddings({"nonsense": 3})


def test_packed_qkv_layout_and_cache_invalidation():
    cfg = mini_config()
    enc = CaptionBertEncoder(cfg)
    att = enc.layer[0].attention.self
    w, b = att.packed_qkv()
    H = cfg.hidden_size
    assert w.shape == (3 * H, H) and w.dtype == torch.bfloat16 and b.shape == (3 * H,) and b.dtype == torch.float32
    assert torch.equal(w[H : 2 * H], att.key.weight.detach().to(torch.bfloat16))
    assert torch.equal(b[2 * H :], att.value.bias.detach())
    from visitron_amd.modeling import _param_key

    k0 = _param_key(enc)
    with torch.no_grad():
        att.key.weight.add_(1.0)  # an optimizer step bumps the version
    assert _param_key(enc) != k0


def test_mask_and_head_mask_helpers():
    ext = torch.zeros(2, 1, 1, 5)
    ext[1, 0, 0, 3:] = -10000.0
    m = _additive_mask_2d(ext, 2, 5)
    assert m.shape == (2, 5) and m.dtype == torch.float32 and float(m[1, 4]) == -10000.0
    with pytest.raises(RuntimeError):
        _additive_mask_2d(torch.zeros(2, 1, 1, 4), 2, 5)
    m3 = _additive_mask_2d(torch.zeros(2, 1, 5, 5), 2, 5)      # per-query mask -> [B,S,S]
    assert m3.shape == (2, 5, 5) and m3.is_contiguous()
    with pytest.raises(NotImplementedError):
        _additive_mask_2d(torch.zeros(2, 3, 5, 5), 2, 5)       # per-head masks are not served
    assert _head_scale([None, None], 2, 3, "cpu") is None
    hs = _head_scale(torch.tensor([1.0, 0.0, 0.5]), 2, 3, "cpu")
    assert hs.shape == (2, 3) and float(hs[1, 2]) == 0.5
    hs2 = _head_scale([None, torch.tensor([0.0, 1.0, 1.0]).view(1, 3, 1, 1)], 2, 3, "cpu")
    assert hs2.tolist() == [[1.0, 1.0, 1.0], [0.0, 1.0, 1.0]]


def test_no_cpu_fallback():
    cfg = mini_config()
    m = PreTrainOscar(cfg).eval()
    b = make_batch(cfg, 2, text_len=8, region_len=4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(**b)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.bert(b["input_ids"])


def test_training_mode_dropout_is_refused_not_silently_skipped():
    m = BertImgModelwithLocationEmbeds(mini_config(hidden_dropout_prob=0.1)).train()
    with pytest.raises((NotImplementedError, RuntimeError)):
        m(torch.ones(1, 4, dtype=torch.long))


def test_synthetic_batch_matches_the_dataset_contract():
    cfg = BertConfig()
    b = make_batch(cfg, 5, seed=7)
    assert set(b) == {"input_ids", "labels", "token_labels", "attention_mask", "img_feats", "img_location_embeddings", "next_action"}
    assert b["input_ids"].shape == (5, 128) and b["attention_mask"].shape == (5, 228)
    assert b["img_feats"].shape == (5, 100, 2054) and b["img_location_embeddings"].shape == (5, 100, 128)
    assert b["labels"].shape == (5, 228) and bool((b["labels"][:, 128:] == -1).all())
    assert bool((b["input_ids"][:, 0] == 101).all())
    real = b["attention_mask"][:, :128].bool()
    assert bool((b["input_ids"][~real] == 0).all()) and bool((b["input_ids"][real] >= 101).all())
    assert float(b["img_feats"].min()) >= 0.0 and float(b["img_location_embeddings"].abs().max()) <= 1.0
    masked_regions = ~b["attention_mask"][:, 128:].bool()
    assert float(b["img_feats"][masked_regions].abs().max()) == 0.0
    assert int(b["next_action"].min()) >= 0 and int(b["next_action"].max()) < 36
    b2 = make_batch(cfg, 5, seed=7)
    assert all(torch.equal(b[k], b2[k]) for k in b)


def test_location_embedding_table():
    """Values of build_viewpoint_loc_embedding (data_loader_pretrain.py:25-43) for viewIndex 0 and 13."""
    import math

    e = viewpoint_loc_embedding(0)
    assert e.shape == (36, 128)
    assert abs(float(e[0, 0])) < 1e-7 and abs(float(e[0, 32]) - 1.0) < 1e-7          # heading 0
    assert abs(float(e[0, 64]) - math.sin(-math.pi / 6)) < 1e-6                      # elevation -30 deg (row 0)
    assert abs(float(e[3, 0]) - 1.0) < 1e-6                                           # heading 90 deg
    assert abs(float(e[12, 64])) < 1e-7 and abs(float(e[24, 64]) - 0.5) < 1e-6        # elevation 0 / +30 deg
    e13 = viewpoint_loc_embedding(13)
    assert abs(float(e13[13, 0])) < 1e-7 and abs(float(e13[14, 0]) - 0.5) < 1e-6      # relative heading
