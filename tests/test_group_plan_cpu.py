"""Host logic of the single-launch kernels, checked without a GPU: the plan cnnq_pc_group_describe reports (the one
k_mmq_group / k_mmq_flat, the fused clipping kernels and the single-read statistics kernels are launched with) must cut
x[N][C][HW] into groups x members that cover every float4 of every channel exactly once, keep every group co-resident
(<= 512 members of <= 128 KB, or <= 704 of 160 KB for channels no smaller tiling fits), and need no more exchange workspace
than ops.GROUP_WS_BYTES.  The tile arithmetic below restates rblk_of (csrc/cnnq_group.hip.h) and the flat tiles of k_mmq_flat."""
import ctypes

import numpy as np
import pytest

from cnn_quantization_amd import _lib as L

TPB = 256


def describe(N, C, HW):
    out = (ctypes.c_int32 * 8)()
    rc = L.load().cnnq_pc_group_describe(N, C, HW, out)
    if rc == L.ENOTSUP:
        return None
    assert rc == 0, rc
    return dict(zip(('A', 'K', 'mode', 'S', 'ncb', 'Gs', 'groups', 'wgs'), list(out)))


RESNET = [(64, 112), (256, 56), (128, 56), (512, 28), (64, 56), (256, 28), (1024, 14), (128, 28), (512, 14), (2048, 7), (256, 14), (512, 7)]
VGG = [(64, 224), (128, 112), (256, 56), (512, 28), (512, 14)]
SHAPES = ([(512, c, hw * hw) for c, hw in RESNET + VGG] + [(64, c, hw * hw) for c, hw in RESNET] +
          [(8, 4, 112 * 112), (40, 6, 56 * 56), (33, 5, 28 * 28), (70, 12, 196), (40, 44, 49), (6, 3, 64 * 64), (300, 8, 49),
           (3, 16, 45), (5, 6, 3), (2, 20, 144), (1, 512, 4608), (9, 5, 2), (512, 3, 224 * 224), (16, 3, 28 * 28), (1000, 1, 4096)])


@pytest.mark.parametrize('shape', SHAPES)
def test_single_launch_plan_covers_the_tensor(shape):
    N, C, HW = shape
    p = describe(N, C, HW)
    if p is None:                                          # no 16-byte tiling of the rows, or a channel 704 tiles of 160 KB cannot hold: the chain takes it
        m = 4 // int(np.gcd(HW % 4, 4)) if HW % 4 else 1
        assert (HW % 4 != 0 and ((C * HW) % 4 != 0 or m * HW > TPB * 4)) or N * (HW // 4) > 704 * TPB * 40 or -(-N // 32) * max(1, -(-(HW // 4) // TPB)) > 512
        return
    assert p['wgs'] == p['groups'] * p['Gs'] and p['Gs'] >= 1
    assert p['K'] in (4, 8, 16, 32, 40)
    assert p['Gs'] <= (704 if p['K'] == 40 else 512)       # a group's members are co-resident: 768 slots of 128 KB, or one channel of 160 KB tiles
    if p['mode'] == 3:
        # flat tiles: a group is one channel, member m holds float4 [m * 256 K, (m + 1) * 256 K) of the channel's N * HW / 4
        assert HW % 4 == 0 and p['A'] == 1 and p['groups'] == C
        total = N * (HW // 4)
        assert (p['Gs'] - 1) * TPB * p['K'] < total <= p['Gs'] * TPB * p['K']
        if p['K'] == 40:
            assert total > 512 * TPB * 32                   # eight more rows in LDS only for channels 512 plain tiles cannot hold
        return
    # row pieces: S batch splits of <= K samples x column blocks of <= 256 float4 columns
    assert p['S'] == -(-N // p['K']) or p['K'] == 4
    plane4 = C * HW // 4
    cover = np.zeros(plane4, dtype=np.int32)
    if p['mode'] == 1:                                      # a group is one channel: nb column blocks x S splits
        cpc = HW // 4
        nb = p['ncb'] // C
        assert p['A'] == 1 and p['groups'] == C and p['Gs'] == p['S'] * nb
        w = -(-cpc // nb)
        assert w <= TPB
        for c in range(C):
            for bb in range(nb):
                c0 = c * cpc + bb * w
                cover[c0:min(c0 + w, (c + 1) * cpc)] += 1
    else:                                                   # a group is a block of k whole channels: S members
        assert p['groups'] == p['ncb'] and p['Gs'] == p['S']
        if p['A'] == 4:                                     # straddling channels: k whole channels with k * HW % 4 == 0 (make_geo)
            m = 4 // int(np.gcd(HW % 4, 4))
            k = min(256, TPB * 4 // HW)
            k -= k % m
        else:
            k = min(256, TPB // (HW // 4))
        assert p['ncb'] == -(-C // k)
        for g in range(p['ncb']):
            c0, c1 = g * k, min(C, (g + 1) * k)
            assert (c0 * HW) % 4 == 0 and (c1 * HW) % 4 == 0 and (c1 * HW - c0 * HW) // 4 <= TPB and c1 - c0 <= 256
            cover[c0 * HW // 4:c1 * HW // 4] += 1
        if p['A'] == 4:
            assert HW % 4 != 0 and (C * HW) % 4 == 0
        else:
            assert HW % 4 == 0
    assert (cover == 1).all(), (shape, p)
    rows = np.zeros(N, dtype=np.int32)
    for s in range(p['S']):
        n0, n1 = s * N // p['S'], (s + 1) * N // p['S']
        assert 1 <= n1 - n0 <= p['K']
        rows[n0:n1] += 1
    assert (rows == 1).all()


@pytest.mark.parametrize('shape', SHAPES)
def test_exchange_workspace_fits_the_one_python_allocates(shape):
    from cnn_quantization_amd import ops
    N, C, HW = shape
    if describe(N, C, HW) is None:
        return
    need = L.load().cnnq_pc_group_workspace(N, C, HW)
    assert 0 < need <= ops.GROUP_WS_BYTES, (shape, need)


def test_planner_rules_of_round_5():
    # the 64-sample shard: flat tiles prefer <= 8 members per channel ([64,64,56,56]: K = 32, 7 members)
    p = describe(64, 64, 56 * 56)
    assert p['mode'] == 3 and p['K'] == 32 and p['Gs'] == 7
    # VGG-16's 224x224 layers: 103 MB per channel -> 160 KB tiles (32 rows in registers + 8 in LDS), one channel on the chip at a time
    p = describe(512, 64, 224 * 224)
    assert p['mode'] == 3 and p['K'] == 40 and p['Gs'] == -(-512 * 224 * 224 // 4 // (TPB * 40)) == 628
    # the headline's big layers: 128 KB flat tiles
    for c, hw in ((64, 112), (256, 56), (512, 28)):
        p = describe(512, c, hw * hw)
        assert p['mode'] == 3 and p['K'] == 32 and p['Gs'] == -(-512 * hw * hw // 4 // (TPB * 32))
    # short rows: whole channels per workgroup (14x14), channels straddling the loads (7x7)
    assert describe(512, 1024, 196)['A'] == 1 and describe(512, 1024, 196)['mode'] == 2
    assert describe(512, 2048, 49)['A'] == 4 and describe(512, 2048, 49)['mode'] == 2
