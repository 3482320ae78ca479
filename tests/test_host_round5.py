"""CPU: round-5 host logic -- the attention dropout's quantised probability at its edges, the bench's kernel-tree stamp."""
import pytest


def test_attention_dropout_probability_edges():
    from visitron_amd import ops

    before = ops.attn_dropout_bits()
    try:
        ops.set_attn_dropout_bits(16)                                 # the default: the reference's 0.1 runs as 0.100006
        assert ops.attn_drop_p(0.0) == 0.0
        assert ops.attn_drop_p(0.1) == 6554.0 / 65536.0 and abs(ops.attn_drop_p(0.1) - 0.1) < 1e-5
        assert ops.attn_drop_p(1e-6) == 1.0 / 65536.0                 # below half a step: one step, never silently off
        assert ops.attn_drop_p(65535.4 / 65536.0) == 65535.0 / 65536.0
        for p in (65535.5 / 65536.0, 1.0):
            with pytest.raises(ValueError):
                ops.attn_drop_p(p)
        ops.set_attn_dropout_bits(8)                                  # rounds 4-5's form: steps of 1/256
        assert ops.attn_drop_p(0.0) == 0.0
        assert ops.attn_drop_p(0.1) == 26.0 / 256.0                  # the reference's 0.1 runs as 0.1016
        assert ops.attn_drop_p(0.001) == 1.0 / 256.0                  # below 1/512: the smallest step, never silently off
        assert ops.attn_drop_p(0.3) == 77.0 / 256.0
        assert ops.attn_drop_p(255.4 / 256.0) == 255.0 / 256.0
        for p in (255.5 / 256.0, 0.999, 1.0):                         # the quantised value would be 1: scale 1 / (1 - p) infinite
            with pytest.raises(ValueError):
                ops.attn_drop_p(p)
    finally:
        ops.set_attn_dropout_bits(before)


def test_kernel_tree_stamp_is_stable_and_names_the_kernel_sources():
    import bench

    a, b = bench.kernel_tree_stamp(), bench.kernel_tree_stamp()
    assert a == b and len(a) == 16 and int(a, 16) >= 0
