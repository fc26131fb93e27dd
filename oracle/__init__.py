"""oracle/ -- CPU restatement of the reference's streaming-ST inference path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import anything from this package, and only as the *checker* (or the
timed CPU baseline).  ``simulst_amd`` never imports it: the product path is the
HIP library behind ``include/simulst_hip.h`` and fails loudly without it.

Every function cites the reference file:line (under /root/reference/codebase
unless noted) whose arithmetic it restates.  Plain torch fp32 on CPU / numpy,
no fairseq, no SimulEval.

Parity status (see DESIGN.md "Oracle pinning"):
* pinned against the reference itself (golden vectors recorded by
  tests/golden/gen_golden.py importing the reference modules by path):
  functions.py, monotonic.py, causal_conv.py, emformer.py, and the
  READ/WRITE control flow of decoder.py / cif.py (CIFLayer.infer).
* PARITY UNPINNED (third-party arithmetic absent from /root/reference):
  - ``cif.cif_function``: restates the published algorithm of
    George0828Zhang/torch_cif (git submodule, directory empty, SHA unknown);
    anchored only on the reference's call sites and SURVEY appendix C.
  - fairseq @4a7835b ``TransformerDecoderLayer``/``MultiheadAttention``/
    sinusoidal positions: restated from the in-repo near-copy
    models/cif_transformer.py:391-537.
  - SimulEval's Average Lagging (latency.py): restated from the published
    definition.
"""
