import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'timeout: per-test time limit (pytest-timeout)')


def pytest_collection_modifyitems(config, items):
    """A GPU test that stops making progress should fail with a stack dump, not stall the run
    (pytest-timeout is part of the image; without it the marker is inert)."""
    import torch
    if not torch.cuda.is_available():
        # a box without a GPU skips the GPU tests instead of failing them with "No HIP GPUs are available"
        skip = pytest.mark.skip(reason='needs a real MI355X (torch.cuda.is_available() is False)')
        for item in items:
            if 'gpu' in item.keywords:
                item.add_marker(skip)
    if not config.pluginmanager.hasplugin('timeout'):
        return
    for item in items:
        if 'gpu' in item.keywords and item.get_closest_marker('timeout') is None:
            item.add_marker(pytest.mark.timeout(600))


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _reset_kernel_knobs(request):
    """The library's probe knobs (`hfl_set_variant`) and the package's mode switches are process-global: put them back to
    their defaults around every GPU test, so that a test that fails between a set and its `finally` cannot leak a variant
    into the tests after it."""
    if 'gpu' not in request.keywords:
        yield
        return
    from hotformerloc_amd import _native, model
    lib = _native.load()
    lib.hfl_set_variant(b'reset', 0)
    yield
    lib.hfl_set_variant(b'reset', 0)
    model.set_gemm_mode(os.environ.get('HFL_GEMM', 'x3'))
