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

PARITY UNPINNED by the reference's own tests: the reference ships no tests,
fixtures or golden vectors and its encoder cannot be
imported in this container (``ModuleNotFoundError: transformers.
pytorch_transformers`` -- an ordinary Python error, nothing was denied).  The
strongest pin available is used instead: ``oracle/crosscheck_hf.py`` checks
every block against the *independent* third-party implementation installed in
this image (transformers 5.x ``BertEncoder``/``BertEmbeddings``/``BertPooler``/
``BertOnlyMLMHead``, eager attention), plus analytic known-answer tests in
``tests/test_oracle_kat.py``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker.  The product package
``visitron_amd`` never imports it and has no CPU fallback.
"""
