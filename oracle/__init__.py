"""CPU oracle for the VISITRON encoder hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-torch fp32/fp64 restatement of the reference's
algorithm for the one hot path this repo accelerates:

  * ``oracle.bert_blocks``  -- the BERT building blocks the reference imports
    from the un-vendored ``transformers.pytorch_transformers`` submodule
    (huggingface/transformers, pytorch-transformers 1.x layout; commit SHA not
    recoverable from /root/reference, see ``.gitmodules:1-3``).  Restated from
    the published algorithm; every block is flagged "upstream, not in mount".
  * ``oracle.modeling``     -- ``oscar/modeling_bert.py:26-169`` and
    ``tasks/viewpoint_select/encoder.py:142-441`` op for op.
  * ``oracle.optim``        -- the pytorch-transformers AdamW rule and warm-up
    schedules used by ``tasks/viewpoint_select/pretrain.py:108-139``.

PARITY STATUS.  The reference ships no tests, fixtures or golden vectors.  Since round 5 the
oracle is pinned BY EXECUTION of the reference's own source in the build container:
``tests/golden/make_golden_from_reference.py`` puts /root/reference on sys.path, imports ``oscar/modeling_bert.py``,
``tasks/viewpoint_select/encoder.py``, ``agent_models.py`` and ``data_loader_pretrain.py`` as they lie, runs them on
deterministic weights (mini, configs[0] B=2 128+100, S=656, the shipped 511+256 and text-only T=511 B=8; forward and
``loss.backward()``; head_mask, 3-D / float / uint8 masks, history states, text_only, loss corners; OscarEncoder,
SoftDotAttention, AttnDecoderLSTM; _mask_tokens / _preprocess_item) and writes ``tests/golden/ref_*.npz``.  In that
process ``oracle.modeling`` / ``oracle.rollout`` / ``oracle.data`` agree with the reference BITWISE on every output and
gradient (``tests/golden/ref_pin_report.json``); ``tests/test_oracle_reference_pin.py`` re-checks the oracle against
the fixtures on every CPU run and the GPU tests read the same fixtures.

WHAT REMAINS UNPINNED by the reference: ``oracle.bert_blocks`` and ``oracle.optim``.  The package they restate
(``transformers.pytorch_transformers``) is an empty submodule, so the generator ran the reference over
``oracle.bert_blocks`` as a flagged STAND-IN for it: the fixtures pin every line of the reference's files, not the
arithmetic inside LayerNorm / erf-GELU / embeddings / pooler / MLM head / init / AdamW / schedules.  Those rest on
``oracle/crosscheck_hf.py`` (every block, forward and gradients, against the independent transformers 5.x installed
in this image) and the analytic known-answer tests in ``tests/test_oracle_kat.py``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker.  The product package
``visitron_amd`` never imports it and has no CPU fallback.
"""
