"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (torch-CPU / numpy, fp32) of the reference's hot path
`HOTFormerLoc.forward` (octree -> global descriptor) and of the subset of the
un-vendored third-party dependency `ocnn==2.2.2` (reference `requirements.txt:6`)
that the path relies on.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import anything from this package -- as the checker / the timed CPU baseline,
never as the thing shipped.  `hotformerloc_amd/` must never import it.

Pinning status (see DESIGN.md "Oracle"):
  * `oracle.ocnn_ref` (octree build / merge / 27-neighbour tables / feature
    averaging) is pinned bit-exactly against ocnn's own golden vectors that the
    reference vendors under `libs/dwconv/test/data/` (copied as data fixtures to
    `tests/golden/ocnn/`) -- `tests/test_oracle_ocnn.py`.
  * `oracle.hotformer_ref` (the model forward) is pinned against outputs of the
    reference's own Python model files imported in the build container over
    `oracle.ocnn_ref` (script `oracle/gen_golden.py`, fixtures
    `tests/golden/model_*.npz`) -- `tests/test_oracle_model.py`.
  * Unpinned by any reference test (ocnn source absent): `OctreeConv` weight
    layout, stride-2 `get_neigh`, `InputFeature('P')` scale, key wrap-around of
    out-of-range coordinates.  These follow ocnn's published behaviour as
    restated in SURVEY.md Appendix A.
"""
