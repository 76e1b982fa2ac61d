"""CPU oracle for the WSI-HGNN hot path — TEST INFRASTRUCTURE ONLY.

This package restates, in plain PyTorch (CPU, fp32 or fp64), the arithmetic of the reference's
message-passing path so the HIP kernels can be checked against it.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the product
package ``wsi-hgnn_amd/`` never does (tests/test_boundary.py enforces that by grepping the sources).

PARITY UNPINNED.  The reference's arithmetic lives in DGL (``dgl.function``, ``dgl.nn.edge_softmax``,
``dgl.readout``, ``dgl.nn.pytorch.GraphConv``), an un-vendored, un-pinned third-party dependency
that is not installed in the build container (no network), and the reference ships no tests or
golden vectors.  The oracle therefore follows the reference sources line by line
(each function cites file:line) together with DGL's documented semantics written down in
SURVEY.md Appendix A.  What pins it instead:
  * the one fragment of the reference that imports without DGL — ``LinearAttentionBlock``
    (models/HEATNet4.py:20-42) — was executed from /root/reference to produce
    tests/golden/linear_attention_block.npz (script: tests/golden/make_reference_fixture.py);
  * two independent formulations (scatter/index_add vs dense masked softmax, oracle/dense.py)
    cross-checked in tests/test_oracle.py;
  * fp64 gradcheck of the scatter formulation on tiny graphs;
  * analytic known-answer cases (uniform attention, single in-edge, empty relation, passthrough).
"""
