"""Config namespace for the oracle (TEST INFRASTRUCTURE).

Field names are the ones the reference reads; base values
are BERT-base as shipped with Oscar ``base-no-labels`` plus the fields
``tasks/viewpoint_select/model_utils.py:75-83`` sets in code.
"""
from types import SimpleNamespace

BASE = dict(
    vocab_size=30522,
    hidden_size=768,
    num_hidden_layers=12,
    num_attention_heads=12,
    intermediate_size=3072,
    hidden_act="gelu",
    hidden_dropout_prob=0.1,
    attention_probs_dropout_prob=0.1,
    max_position_embeddings=512,
    type_vocab_size=2,
    initializer_range=0.02,
    layer_norm_eps=1e-12,
    output_attentions=False,
    output_hidden_states=False,
    torchscript=False,
    img_feature_dim=2054,
    img_feature_type="faster_r-cnn",
    action_space=36,
    detector_classes=1601,
    classifier="linear",
    loss_type="CrossEntropy",
    cls_hidden_scale=2,
)

TINY = dict(
    BASE,
    vocab_size=97,
    hidden_size=64,
    num_hidden_layers=2,
    num_attention_heads=4,
    intermediate_size=128,
    max_position_embeddings=32,
    img_feature_dim=22,
    action_space=36,
    detector_classes=11,
    hidden_dropout_prob=0.0,
    attention_probs_dropout_prob=0.0,
)


def make_config(base=BASE, **overrides):
    d = dict(base)
    d.update(overrides)
    return SimpleNamespace(**d)
