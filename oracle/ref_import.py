"""Import the reference's own Python model files (BUILD CONTAINER ONLY).

TEST INFRASTRUCTURE (see oracle/__init__.py).  `/root/reference` exists only in
the build container; nothing that runs on the GPU box may call this module.  It
is used by `oracle/gen_golden.py` (fixture generation) and by the optional
`tests/test_oracle_model.py::test_oracle_matches_reference_live` (skipped when the
reference tree is absent).

What it does:
  * serves `import ocnn` from `oracle.ocnn_ref` (ocnn==2.2.2 is not installable),
  * serves `import dwconv` with `OctreeDWConv` == `ocnn.nn.OctreeDWConv` semantics
    (the reference's CUDA op is asserted equal to it to 1e-6 by
    `libs/dwconv/test/test_octree_dwconv.py:44-47`),
  * points `sys.modules['datasets']` at the reference's `datasets/` directory,
    which is otherwise shadowed by the HuggingFace `datasets` wheel,
  * puts the reference root on `sys.path` (`README.md:63-66`).
"""

import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get('HOTFORMERLOC_REFERENCE', '/root/reference')


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'models'))


def install():
    """Make `from models.model_factory import model_factory` importable."""
    if not reference_available():
        raise RuntimeError('reference tree not found at ' + REFERENCE_ROOT)
    from oracle import ocnn_ref

    sys.modules['ocnn'] = ocnn_ref
    sys.modules['ocnn.octree'] = ocnn_ref.octree
    sys.modules['ocnn.nn'] = ocnn_ref.nn
    sys.modules['ocnn.modules'] = ocnn_ref.modules

    dw = types.ModuleType('dwconv')

    class OctreeDWConv(ocnn_ref.nn.OctreeDWConv):
        def __init__(self, channels, kernel_size=[3], nempty=False, use_bias=False):
            super().__init__(in_channels=channels, kernel_size=kernel_size, stride=1,
                             nempty=nempty, use_bias=use_bias)
    dw.OctreeDWConv = OctreeDWConv
    sys.modules['dwconv'] = dw

    ds = types.ModuleType('datasets')
    ds.__path__ = [os.path.join(REFERENCE_ROOT, 'datasets')]
    sys.modules['datasets'] = ds
    for name in list(sys.modules):
        if name.startswith('datasets.'):
            del sys.modules[name]

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    importlib.invalidate_caches()


def reference_model(cfg_path: str):
    """Build the reference `HOTFormerLoc` from one of its model cfg files."""
    install()
    from misc.utils import ModelParams                # noqa: E402  (reference)
    from models.model_factory import model_factory     # noqa: E402  (reference)
    params = ModelParams(cfg_path)
    model = model_factory(params)
    model.eval()
    return model, params
