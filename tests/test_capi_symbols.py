"""CPU: the C-ABI library loads and exports every function include/visitron_hip.h declares, and the
ctypes binding lists exactly those (no compute calls without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "visitron_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vt_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_entry_points():
    names = _declared()
    for must in ("vt_linear_bf16", "vt_attention_fwd_bf16", "vt_layernorm_bf16", "vt_embed_layernorm",
                 "vt_pack_concat_bf16", "vt_encoder_forward_bf16", "vt_error_string", "vt_abi_version"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from visitron_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    assert sorted(_lib.SIGNATURES) == _declared()
    loaded = _lib.load()
    assert loaded.vt_abi_version() >= 1
    assert loaded.vt_error_string(0) == b"ok" and loaded.vt_error_string(-4) == b"unsupported configuration"


def test_struct_layouts_match_the_header():
    from visitron_amd import _lib

    assert ctypes.sizeof(_lib.LayerWeights) == 12 * ctypes.sizeof(ctypes.c_void_p)
    # vt_layer_acts: 16 pointers, then two int32 (ln_residual_mode, reserved0: ABI 10)
    assert ctypes.sizeof(_lib.LayerActs) == 16 * ctypes.sizeof(ctypes.c_void_p) + 8
    assert _lib.LayerActs.ln_residual_mode.offset == 16 * ctypes.sizeof(ctypes.c_void_p)
    src = open(os.path.join(ROOT, "include", "visitron_hip.h")).read()
    for struct, cls in (("vt_layer_weights", _lib.LayerWeights), ("vt_layer_acts", _lib.LayerActs)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), src, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = re.findall(r"(?:\*|int32_t)\s*([a-z0-9_]+)\s*;", body)
        assert fields == [f[0] for f in cls._fields_], struct
