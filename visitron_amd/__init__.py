"""visitron_amd -- MI355X-native (gfx950) encoder hot path of alexa/visitron.

Public surface = the reference's module API for this path (see ``visitron_amd.modeling``):
``PreTrainOscar``, ``BertImgModelwithLocationEmbeds``, ``CaptionBertEncoder`` and friends,
``BertConfig``, ``MODEL_CLASS``; ``set_precision(model, "fp32")`` selects the fp32 parity kernels (default: bf16);
``check_errors()`` waits for the asynchronous error flags of the calls issued so far (ids outside an embedding table).
Arithmetic runs in ``lib/libvisitron_hip.so`` (C ABI in
``include/visitron_hip.h``); there is no CPU fallback.
"""
from .config import BertConfig, mini_config, tiny_config  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):  # lazy: importing the package must not require torch+HIP for config-only users
    if name in ("PreTrainOscar", "BertImgModelwithLocationEmbeds", "CaptionBertEncoder", "CaptionBertLayer",
                "CaptionBertAttention", "CaptionBertSelfAttention", "NextActionPrediction", "MODEL_CLASS", "set_precision",
                "invalidate_packed_weights", "check_errors"):
        from . import modeling

        return getattr(modeling, name)
    raise AttributeError(name)
