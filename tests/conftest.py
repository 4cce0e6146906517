import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, e.g. plain `pytest tests/`
    in the build container; `-m "not gpu"` deselects them entirely."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


class Golden:
    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)

    def __contains__(self, k):
        return k in self._z.files

    def np(self, k):
        return self._z[k]

    def t(self, k):
        import torch
        return torch.from_numpy(np.ascontiguousarray(self._z[k]))

    def keys(self):
        return self._z.files


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return load


def bits_equal(a, b):
    """Bitwise equality of two fp32 arrays (NaN == NaN, +0 != -0)."""
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    b = np.ascontiguousarray(np.asarray(b, dtype=np.float32))
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.fixture(autouse=True)
def _group_workspace_zero_at_rest(request):
    """After every -m gpu test: every group workspace this process has handed to a kernel is zero at rest (header word 0, the
    sticky status, aside) - the invariant the slot meeting and the sum exchange rest on, checked where it could first break."""
    yield
    if request.node.get_closest_marker('gpu') is None:
        return
    import sys
    ops = sys.modules.get('cnn_quantization_amd.ops')
    if ops is None or not getattr(ops, '_GROUP_WS', None):
        return
    import ctypes
    import torch
    from cnn_quantization_amd import _lib
    torch.cuda.synchronize()
    for key, ws in list(ops._GROUP_WS.items()):
        nz = ctypes.c_uint64()
        rc = _lib.load().cnnq_group_ws_at_rest(ws, ctypes.byref(nz))
        assert rc == 0 and nz.value == 0, 'group workspace %r: %d non-zero words at rest after %s' % (key, nz.value, request.node.nodeid)
