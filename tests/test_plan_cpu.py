"""Host logic of the kernels, checked without a GPU: the launch plan (cnnq_plan_describe) must cut
x[N][C][HW] into workgroups that cover every element exactly once, keep every workgroup inside
channel boundaries as its mode promises, and give every (group, channel) partial exactly one writer.
The block -> range arithmetic below restates `blk_of` of csrc/cnnq_common.hip.h."""
import ctypes
import itertools

import numpy as np
import pytest

from cnn_quantization_amd import _lib as L


def describe(N, C, HW, aligned, fine):
    out = (ctypes.c_int32 * 12)()
    rc = L.load().cnnq_plan_describe(N, C, HW, aligned, fine, out)
    assert rc == 0, rc
    keys = ('vec', 'A', 'J', 'mode', 'nb', 'w', 'k', 'ncb', 'S', 'tpb', 'G')
    return dict(zip(keys, list(out)))


def blocks(N, C, HW, p):
    """Yield (n0, n1, elem0, elem1, c0, c1, grp) per workgroup: samples, plane element range, channels."""
    vec = p['vec']
    for bid in range(p['S'] * p['ncb']):
        cb, s = bid % p['ncb'], bid // p['ncb']
        n0, n1 = s * N // p['S'], (s + 1) * N // p['S']
        if p['mode'] == 1:
            cpc = HW // vec
            c, bb = cb // p['nb'], cb % p['nb']
            col0 = c * cpc + bb * p['w']
            col1 = min(col0 + p['w'], (c + 1) * cpc)
            yield n0, n1, col0 * vec, col1 * vec, c, c + 1, s * p['nb'] + bb
        else:
            c0 = cb * p['k']
            c1 = min(C, c0 + p['k'])
            col0, col1 = c0 * HW // vec, c1 * HW // vec
            assert (c0 * HW) % vec == 0 and (c1 * HW) % vec == 0          # vector loads never split
            yield n0, n1, col0 * vec, col1 * vec, c0, c1, s


SHAPES = [(512, 64, 112 * 112), (512, 256, 56 * 56), (512, 512, 28 * 28), (512, 1024, 14 * 14), (512, 2048, 7 * 7),
          (512, 512, 7 * 7), (512, 256, 14 * 14), (512, 128, 28 * 28), (64, 64, 224 * 224), (32, 64, 112 * 112),
          (4, 8, 49), (3, 16, 45), (5, 6, 3), (2, 20, 144), (1, 1100, 4), (5, 7, 1029), (1, 512, 4608), (1, 1000, 2048),
          (1, 1, 25690112), (7, 3, 1), (1, 1, 1), (9, 5, 2), (2, 4097, 5), (3, 2, 1000001)]


def check_plans(shape):
    N, C, HW = shape
    for aligned, fine in itertools.product((1, 0), (0, 1)):
        p = describe(N, C, HW, aligned, fine)
        assert p['tpb'] == 256 and p['vec'] in (1, 4) and p['J'] in (1, 2, 4)
        if p['vec'] == 4 and p['A'] == 1:
            assert HW % 4 == 0 and aligned
        if p['A'] == 4:
            assert HW % 4 != 0 and (C * HW) % 4 == 0 and aligned
        cap = p['tpb'] * p['J'] * p['vec']                         # elements one workgroup can load per sample
        cover = np.zeros(C * HW, dtype=np.int32)                   # plane coverage for the FIRST batch split
        rows = np.zeros(N, dtype=np.int32)
        writers = {}
        for n0, n1, e0, e1, c0, c1, grp in blocks(N, C, HW, p):
            assert 0 <= n0 <= n1 <= N and 0 <= e0 < e1 <= C * HW, (p, n0, n1, e0, e1)
            assert e1 - e0 <= cap
            assert c0 * HW <= e0 and e1 <= c1 * HW                # stays inside its channels
            assert c1 - c0 <= 256                                  # MAXCH: LDS parameter tables
            if p['mode'] == 1:
                assert c1 - c0 == 1
            if n0 == 0:
                cover[e0:e1] += 1
            if e0 == 0:
                rows[n0:n1] += 1
            for c in range(c0, c1):
                writers[(grp, c)] = writers.get((grp, c), 0) + 1
        assert (cover == 1).all(), (shape, aligned, fine, p)
        assert (rows == 1).all(), (shape, aligned, fine, p)
        assert set(writers.values()) == {1}                        # one writer per partial record
        assert len(writers) == p['G'] * C
        if not fine:
            assert p['G'] == L.load().cnnq_pc_groups(N, C, HW, aligned)
            assert p['S'] <= 64
        else:
            per_wg = (N / p['S']) * min(cap, (HW if p['mode'] == 1 else p['k'] * HW)) * 4
            assert per_wg <= 64 * 1024 or p['S'] == N              # short workgroups (about 14 KB)


@pytest.mark.parametrize('shape', SHAPES)
def test_plans_partition_the_tensor(shape):
    check_plans(shape)


def test_plans_partition_random_shapes():
    """The same partition properties on random geometries (hypothesis, derandomised): odd H*W, channel counts
    around the workgroup capacity, single samples, planes from a few elements to ~100 K."""
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st

    @settings(max_examples=150, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(N=st.integers(1, 70), C=st.integers(1, 600),
           HW=st.one_of(st.integers(1, 64), st.sampled_from([49, 81, 121, 169, 196, 289, 784, 1024, 3136, 12544, 12545])))
    def run(N, C, HW):
        if C * HW <= 400000:
            check_plans((N, C, HW))
    run()


def test_fine_geometry_targets_short_workgroups():
    for (N, C, HW) in [(512, 64, 12544), (512, 256, 3136), (512, 1024, 196), (512, 2048, 49)]:
        coarse, fine = describe(N, C, HW, 1, 0), describe(N, C, HW, 1, 1)
        assert fine['S'] > coarse['S']
        assert fine['S'] * fine['ncb'] >= 16 * coarse['S'] * coarse['ncb'] // 8
