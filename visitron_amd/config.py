"""``BertConfig`` for the hot path: the attribute names the reference reads.

The reference builds its config with ``BertConfig.from_pretrained(path)`` (a class
of the un-vendored pytorch-transformers dependency) and then sets extra fields in
code (tasks/viewpoint_select/model_utils.py:47,75-83).  This class keeps that
surface: attribute access by the same names, ``from_pretrained`` /
``from_json_file`` / ``from_dict`` / ``save_pretrained`` / ``to_dict``; unknown
keys found in a ``config.json`` are kept as attributes, as upstream does.
"""
import copy
import json
import os

CONFIG_NAME = "config.json"


class BertConfig(object):
    def __init__(
        self,
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
        **kwargs
    ):
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.intermediate_size = intermediate_size
        self.hidden_act = hidden_act
        self.hidden_dropout_prob = hidden_dropout_prob
        self.attention_probs_dropout_prob = attention_probs_dropout_prob
        self.max_position_embeddings = max_position_embeddings
        self.type_vocab_size = type_vocab_size
        self.initializer_range = initializer_range
        self.layer_norm_eps = layer_norm_eps
        # PretrainedConfig-level fields the reference reads
        self.output_attentions = kwargs.pop("output_attentions", False)
        self.output_hidden_states = kwargs.pop("output_hidden_states", False)
        self.torchscript = kwargs.pop("torchscript", False)
        self.num_labels = kwargs.pop("num_labels", 2)
        self.pruned_heads = kwargs.pop("pruned_heads", {})
        # fields set in code by load_oscar_weights (model_utils.py:75-83) and read by encoder.py:170-185,317-324
        self.img_feature_dim = kwargs.pop("img_feature_dim", 2054)
        self.img_feature_type = kwargs.pop("img_feature_type", "faster_r-cnn")
        self.action_space = kwargs.pop("action_space", 36)
        self.detector_classes = kwargs.pop("detector_classes", 1601)
        self.classifier = kwargs.pop("classifier", "linear")
        self.loss_type = kwargs.pop("loss_type", "CrossEntropy")
        self.cls_hidden_scale = kwargs.pop("cls_hidden_scale", 2)
        for k, v in kwargs.items():  # e.g. use_img_layernorm / img_layer_norm_eps (optional, encoder.py:173-185)
            setattr(self, k, v)

    # ---- construction -------------------------------------------------
    @classmethod
    def from_dict(cls, d):
        return cls(**dict(d))

    @classmethod
    def from_json_file(cls, path):
        with open(path, "r", encoding="utf-8") as f:
            return cls.from_dict(json.load(f))

    @classmethod
    def from_pretrained(cls, path, **kwargs):
        """``path`` is a directory holding ``config.json`` or the json file itself."""
        f = os.path.join(path, CONFIG_NAME) if os.path.isdir(path) else path
        cfg = cls.from_json_file(f)
        for k, v in kwargs.items():
            setattr(cfg, k, v)
        return cfg

    # ---- serialisation ------------------------------------------------
    def to_dict(self):
        return copy.deepcopy(self.__dict__)

    def to_json_string(self):
        return json.dumps(self.to_dict(), indent=2, sort_keys=True) + "\n"

    def save_pretrained(self, save_directory):
        assert os.path.isdir(save_directory)
        with open(os.path.join(save_directory, CONFIG_NAME), "w", encoding="utf-8") as f:
            f.write(self.to_json_string())

    def __repr__(self):
        return "BertConfig " + self.to_json_string()


def tiny_config(**overrides):
    """Small config used by tests: L=2, H=64 (4 heads x 16) does NOT hit the fused
    dh=64 kernels; ``mini_config`` does."""
    d = dict(
        vocab_size=97, hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
        intermediate_size=128, max_position_embeddings=32, img_feature_dim=22,
        action_space=36, detector_classes=11, hidden_dropout_prob=0.0,
        attention_probs_dropout_prob=0.0,
    )
    d.update(overrides)
    return BertConfig(**d)


def mini_config(**overrides):
    """Smallest config the HIP kernels serve (head size 64): L=2, H=128, 2 heads."""
    d = dict(
        vocab_size=512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
        intermediate_size=512, max_position_embeddings=64, img_feature_dim=70,
        action_space=36, detector_classes=40, hidden_dropout_prob=0.0,
        attention_probs_dropout_prob=0.0,
    )
    d.update(overrides)
    return BertConfig(**d)
