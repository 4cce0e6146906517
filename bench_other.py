"""bench_other.py - the `other_configs` object of bench.py's JSON line: BASELINE.json configs 1, 3, 4, 5 (and config 2 with
its optional outputs) on one GPU, each timed like the headline and checked after its timed region.  Imported by bench.py
only when the default single-GPU run asks for it (`--no-other-configs` skips it)."""
import math
import time

import torch

import json
import os

from bench import HBM_PEAK_GBS, RESNET50_CONV_OUTPUTS, VGG16_CONV_OUTPUTS, laplace_activation, timed_best

ROOT = os.path.dirname(os.path.abspath(__file__))


def pmc_traffic(name):
    """HBM bytes per launch of a configuration's kernels from the PMC counters (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, collected in
    separate rocprofv3 --pmc passes and committed under profiles/; never measured inside a timed run): the newest round's file."""
    for rnd in ('r06', 'r05'):
        f = os.path.join(ROOT, 'profiles', '%s_pmc_traffic_%s.json' % (rnd, name))
        if os.path.exists(f):
            try:
                with open(f) as fh:
                    rec = json.load(fh)
                return {'bytes_per_launch': rec.get('bytes_per_launch'), 'box': rec.get('box'), 'source': 'profiles/' + os.path.basename(f)}
            except (OSError, ValueError):
                return None
    return None


def leg_object(elems, t, bpe, what, verified=None, moved=None, pmc=None):
    """One entry of `other_configs`.  roofline.achieved / frac price every element at the bytes the launches of THIS round read
    and write (`bytes_moved_per_element`: the single-launch forms keep their tile in registers between the statistics and the
    quantization, one read of x fewer than SURVEY 8(d3) counts) - achieved bandwidth, never above what HBM delivered.
    SURVEY 8(d3)'s accounting (12 / 16 / 8 bytes per element: statistics read(s) + Q/DQ read + write), the figure earlier rounds
    called `frac`, is kept next to it as an EQUIVALENCE - the rate a chain of separate passes would need for the same time -
    under `frac_survey_accounting`."""
    m = bpe if moved is None else moved
    gbs = elems * m / t / 1e9
    r = {'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
         'bytes_moved_per_element': m, 'traffic': pmc,
         'survey_bytes_per_element': bpe, 'achieved_survey_accounting': elems * bpe / t / 1e9,
         'frac_survey_accounting': elems * bpe / t / 1e9 / HBM_PEAK_GBS}
    return {'workload': what, 'ms': t * 1e3, 'value': elems / t, 'unit': 'elements/s', 'verified': verified, 'roofline': r}


def other_configs(ops, device, batch):
    """BASELINE.json configs 1, 3, 4, 5 (and config 2 with its optional outputs) on one GPU, inputs resident, best of 3
    wall-clock passes bracketed by synchronisation; algorithmic bytes per element as SURVEY.md 8(d3).  After the timing
    of each entry ONE layer of it is checked (`verified`): properties of the result that need no oracle, on the
    largest tensor of the set."""
    from cnn_quantization_amd import _lib as Lb

    def obj(elems, t, bpe, what, verified=None, moved=None, pmc=None):
        return leg_object(elems, t, bpe, what, verified, moved, pmc if batch == 512 else None)

    def codes_consistent(y, codes, qp, C):
        sc, zp, qm = (qp[r].view(1, C, 1, 1) for r in (Lb.QP_SCALE, Lb.QP_ZP, Lb.QP_QMAX))
        cf = codes.float()
        return bool((cf <= qm).all()) and bool(torch.equal((cf - zp) * sc, y))

    out = {}
    # ---- config 1
    xs = [laplace_activation((32, 64, 112, 112), 1 + i, device) for i in range(16)]
    t = timed_best(lambda: [ops.minmax_qdq_per_tensor(x, 8, avg_over_batch=True) for x in xs])
    x = xs[0]
    y = ops.minmax_qdq_per_tensor(x, 8, avg_over_batch=True)
    flat = x.view(32, -1)
    mn, mx = float(flat.min(1)[0].mean()), float(flat.max(1)[0].mean())        # iq.py:361-379: batch mean of the extrema
    step = (mx - mn) / 255.
    inside = (x >= mn) & (x <= mx)
    ok1 = int(torch.unique(y).numel()) <= 256 and float(((x - y).abs() * inside).max()) <= 0.5001 * step + 1e-6
    out['config1'] = obj(xs[0].numel() * 16, t, 12, 'per-tensor int8 GEMMLOWP Q/DQ with dynamic min/max on 16 distinct '
                         '[32,64,112,112] tensors (%.1f us per tensor)' % (t / 16 * 1e6), bool(ok1))
    del xs, x, y, flat, inside
    layers, seed = [], 100
    for (C, hw, half, count) in RESNET50_CONV_OUTPUTS:
        for _ in range(count):
            layers.append((laplace_activation((batch, C, hw, hw), seed, device), half))
            seed += 1
    elems = sum(x.numel() for x, _ in layers)
    big = max(range(len(layers)), key=lambda i: layers[i][0].numel())
    xb, hb = layers[big]
    Cb = xb.shape[1]
    ys = [torch.empty_like(x) for x, _ in layers]
    # ---- config 2 with the entropy of the codes (-me), and with the packed codes as the stored result
    def step_me():
        # the entropies of the forward's 53 tensors in ONE launch at its end (ops.entropy_batch, round 6)
        with ops.entropy_batch():
            for (x, half), y in zip(layers, ys):
                ops.act_qdq_per_channel(x, 4, positive=half, want_entropy=True, out=y)
    t = timed_best(step_me)
    yb, cb, eb, pb = ops.act_qdq_per_channel(xb, 4, positive=hb, want_codes=True, want_entropy=True, want_parts=True)
    cnt = torch.bincount(cb.flatten()[:1 << 28].long(), minlength=16) if cb.numel() <= (1 << 28) else None
    if cnt is None:
        cnt = torch.zeros(16, dtype=torch.int64, device=device)
        for n0 in range(0, xb.shape[0], 64):
            cnt += torch.bincount(cb[n0:n0 + 64].flatten().long(), minlength=16)
    pr = cnt[cnt > 0].double() / xb.numel()
    ent_ref = float(-(pr * torch.log2(pr)).sum())
    ok2 = (bool(torch.equal(yb, ys[big])) and codes_consistent(yb, cb, pb['qp'], Cb)
           and abs(float(eb) - ent_ref) <= 2e-5 * max(1., ent_ref))
    out['config2_entropy'] = obj(elems, t, 8, 'ResNet-50 b%d, config 2 plus the Shannon entropy of the integer codes (-me, '
                                 'iq.py:586-587): one launch per tensor with the code histogram fused in, ONE entropy '
                                 'launch for all 53 tensors at the end of the step (ops.entropy_batch)' % batch, bool(ok2))
    del cb, cnt
    pbufs = [torch.empty(x.numel() // 2, dtype=torch.uint8, device=device) for x, _ in layers]
    t = timed_best(lambda: [ops.minmax_quantize_pack4(x, 4, half, out=b) for (x, half), b in zip(layers, pbufs)])
    pk, qpk = ops.minmax_quantize_pack4(xb, 4, hb)
    ok2p = bool(torch.equal(pk, pbufs[big])) and bool(torch.equal(qpk, pb['qp']))
    if xb[0, 0].numel() % 4 == 0:
        ok2p = ok2p and bool(torch.equal(ops.dequantize_pack4(pk, xb.shape, qpk), yb))
    out['config2_packed_single_launch'] = obj(elems, t, 4.5, 'ResNet-50 b%d, config 2 with the packed 4-bit codes as the STORED '
                                              'result instead of the dequantized floats: one launch, 4 B read + 0.5 B written '
                                              'per element (SURVEY 8 f3)' % batch, bool(ok2p))
    del pbufs, pk, yb, pb
    # ---- config 3
    t = timed_best(lambda: [ops.act_qdq_per_channel(x, 4, positive=half, clip='laplace', bit_alloc=True, out=y)
                            for (x, half), y in zip(layers, ys)])
    y3, c3, p3 = ops.act_qdq_per_channel(xb, 4, positive=hb, clip='laplace', bit_alloc=True, want_codes=True, want_parts=True)
    bits3 = p3['diag'][Lb.DIAG_BITS]
    ok3 = (bool(torch.equal(y3, ys[big])) and codes_consistent(y3, c3, p3['qp'], Cb) and float(bits3.min()) >= 0
           and float(bits3.max()) <= 8 and abs(float(bits3.mean()) - 4.) <= 0.011 + 1. / Cb)      # iq.py:403, in steps of 1/C
    out['config3'] = obj(elems, t, 16, 'ResNet-50 b%d, per-channel int4 + ACIQ laplace + bit allocation, dynamic statistics '
                         '(-c laplace -baa): pass A, merge, bit allocation, then ONE launch for pass B + parameters + Q/DQ '
                         '(cnnq_pc_aciq_qdq_single)' % batch, bool(ok3), moved=12, pmc=pmc_traffic('config3'))
    del ys, c3
    # the same configuration with the bit-allocated integer codes as the STORED result
    # (sum(bits)/8 bytes per position instead of 4 B/elem of dequantized floats); the packing pass alone is timed: its
    # parameters - scale / zero point / width per channel and the row layout that follows from the widths - come from
    # one untimed statistics + parameter run per tensor
    pk = []
    for (x, half) in layers:
        _, parts = ops.act_qdq_per_channel(x, 4, positive=half, clip='laplace', bit_alloc=True, want_parts=True)
        bits = parts['diag'][Lb.DIAG_BITS].contiguous()
        pk.append((x, parts['qp'], bits, ops.packed_layout(bits, x.shape[2] * x.shape[3])))
    del _
    stored = [ops.quantize_packed(x, qp, bits) for x, qp, bits, _ in pk]
    nbytes = sum(p.numel() for p, _ in stored)
    del stored
    torch.cuda.empty_cache()
    bufs = [torch.empty(ops.packed_capacity(x.shape), dtype=torch.uint8, device=device) for x, _, _, _ in pk]
    t = timed_best(lambda: [ops.quantize_packed(x, qp, bits, out=b, rowoff=ro) for (x, qp, bits, ro), b in zip(pk, bufs)])
    t_with_layout = timed_best(lambda: [ops.quantize_packed(x, qp, bits, out=b) for (x, qp, bits, _), b in zip(pk, bufs)])
    bpe = 4 + nbytes / elems
    # ... and the way back: stored codes -> fp32 (what the next layer's kernel would fuse into its load)
    ys = [torch.empty_like(x) for x, _, _, _ in pk]
    t_load = timed_best(lambda: [ops.dequantize_packed(b, x.shape, qp, bits, ro, out=yy)
                                 for (x, qp, bits, ro), b, yy in zip(pk, bufs, ys)])
    ok3p = bool(torch.equal(ys[big], y3))                 # stored codes -> fp32 == the fused Q/DQ of config 3, bit for bit
    del bufs, ys, y3
    out['config3_packed_storage'] = obj(elems, t, bpe, 'ResNet-50 b%d, the quantize+pack pass of config 3 with the bit-allocated '
                                        'codes as the stored format: %.3f bytes per element written (fp32 dequantized: 4), into '
                                        'preallocated buffers (no host read); one launch per tensor, the row layout comes with '
                                        'the parameters' % (batch, nbytes / elems), ok3p)
    out['config3_packed_storage']['ms_with_layout_launch'] = t_with_layout * 1e3      # the layout recomputed in front of every pass
    out['config3_packed_load'] = obj(elems, t_load, bpe, 'ResNet-50 b%d, the inverse pass: bit-allocated stored codes -> '
                                     'dequantized fp32 (%.3f bytes per element read, 4 written)' % (batch, nbytes / elems), ok3p)
    del pk
    # ---- config 4
    t = timed_best(lambda: [ops.pc_stats(x, x.shape[0], x.shape[1], x.shape[2] * x.shape[3], need_b=True, need_kurt=True,
                                         need_relu=True) for x, _ in layers])
    st4, _ = ops.pc_stats(xb, xb.shape[0], Cb, xb.shape[2] * xb.shape[3], need_b=True, need_kurt=True, need_relu=True)
    sub = xb[:, :8].double()
    m8, s8 = sub.mean(dim=(0, 2, 3)), sub.transpose(0, 1).reshape(8, -1).std(1, unbiased=True)
    ok4 = (bool(torch.equal(st4[Lb.STAT_MAX], xb.amax(dim=(0, 2, 3)))) and bool(torch.equal(st4[Lb.STAT_MIN], xb.amin(dim=(0, 2, 3))))
           and float(((st4[Lb.STAT_MEAN][:8].double() - m8).abs() / m8.abs().clamp(min=1e-3)).max()) < 1e-5
           and float(((st4[Lb.STAT_STD][:8].double() - s8).abs() / s8).max()) < 1e-5)
    # the layers the library routes to ONE launch that reads x once (cnnq_pc_stats_route: flat tiles of at most 256 members per
    # channel, row pieces where they beat the chain)
    lib = Lb.load()
    one_read = sum(x.numel() for x, _ in layers
                   if lib.cnnq_pc_stats_route(x.shape[0], x.shape[1], x.shape[2] * x.shape[3], 1, ops.GROUP_WS_BYTES, 0) > 0)
    out['config4'] = obj(elems, t, 8, 'ResNet-50 b%d, -sm collect: the seven per-channel statistics; %.0f %% of the elements in one '
                         'launch and one read of x (cnnq_pc_stats_single), the rest in the three-launch chain' % (batch, 100. * one_read / elems),
                         bool(ok4), moved=8 - 4. * one_read / elems, pmc=pmc_traffic('config4'))
    del layers, xb, sub
    torch.cuda.empty_cache()
    # ---- config 5
    vl, seed = [], 500
    for (C, hw, count) in VGG16_CONV_OUTPUTS:
        for _ in range(count):
            vl.append(laplace_activation((batch, C, hw, hw), seed, device))
            seed += 1
    elems = sum(x.numel() for x in vl)
    def step5():
        with ops.entropy_batch():                          # the 13 tensors' entropies in one launch at the end of the forward
            for x in vl:
                ops.mid_tread_qdq(x, 4, clip=True, sym=False, want_entropy=True)
    t = timed_best(step5, reps=2)
    xv = vl[2]                                             # [b, 128, 112, 112]
    y5, e5, c5, p5 = ops.mid_tread_qdq(xv, 4, clip=True, sym=False, want_entropy=True, want_codes=True, want_parts=True)
    Cv = xv.shape[1]
    d5, lo5, hi5 = (p5['mt'][r].view(1, Cv, 1, 1) for r in (Lb.MT_DELTA, Lb.MT_CMIN, Lb.MT_CMAX))
    ok5 = (bool(torch.equal(c5 * d5, y5)) and bool((c5 >= lo5).all()) and bool((c5 <= hi5).all())
           and int(p5['hist'][:-1].sum()) == xv.numel() and math.isfinite(float(e5)))
    # the timed route (no codes output) is the single launch; it is checked against the chain's outputs on the same tensor
    y5s, e5s, p5s = ops.mid_tread_qdq(xv, 4, clip=True, sym=False, want_entropy=True, want_parts=True)
    same5 = (p5s['mt'][Lb.MT_DELTA] == p5['mt'][Lb.MT_DELTA])
    ok5 = (ok5 and int(same5.sum()) >= Cv - 1 and bool(torch.equal(y5s[:, same5], y5[:, same5]))
           and abs(float(e5s) - float(e5)) <= 1e-3)
    del y5s
    out['config5'] = obj(elems, t, 16, 'VGG-16 b%d, mid-tread per-channel W4A4 + ACIQ + bin allocation + entropy (-mtq -me): pass A, '
                         'merge, bin allocation, then ONE launch for pass B + step sizes + quantization + code histogram '
                         '(cnnq_pc_midtread_qdq_single; the two 224x224 layers on 160 KB tiles, one channel on the chip at a time)'
                         % batch, bool(ok5), moved=12, pmc=pmc_traffic('config5'))
    del vl, y5, c5
    torch.cuda.empty_cache()
    return out


def shard_configs(ops, device, per_rank, group, world, rank, exchange_name):
    """BASELINE configs 3 / 4 / 5 at the SHARD of a batch-sharded run (round 6): every rank quantizes / reduces its per_rank
    samples of each tensor, the statistics are the GLOBAL batch's.  COLLECTIVE: every rank of the job calls it (also a forced
    1-rank exchange).  With the in-launch exchange in force the ranks' sums meet inside the single launch
    (cnnq_pc_aciq_fused_xrank / _midtread_fused_xrank / _stats_xrank: 12 / 12 / 4 bytes per element moved); otherwise the chain runs
    around the collective (16 / 16 / 8).  Each leg is timed like the headline (barrier + synchronise on both sides, max over the
    ranks, best of 3) and followed by distributed.xrank_checkpoint: a leg during which a wait for a peer expired is timed again
    through the collective, on every rank.  Returns the entries, keyed config3 / config4 / config5."""
    import torch.distributed as dist
    from cnn_quantization_amd import _lib as Lb, distributed as D

    def barrier():
        if world > 1:
            dist.barrier(group=group)
        torch.cuda.synchronize()

    def timed(fn, reps=3):
        fn()
        best = None
        for _ in range(reps):
            barrier()
            t0 = time.perf_counter()
            fn()
            barrier()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        if world > 1:
            on_dev = dist.get_backend(group) == 'nccl'
            tt = torch.tensor([best], dtype=torch.float64, device=device if on_dev else 'cpu')
            dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=group)
            best = float(tt[0].item())
        return best

    def leg(fn, reps=3):
        """(seconds, in_launch): the leg through whatever exchange is in force; again through the collective if a wait expired"""
        in_launch = D.xrank_exchange(group) is not None
        t = timed(fn, reps)
        if in_launch and not D.xrank_checkpoint(group):
            in_launch = False
            t = timed(fn, reps)
        return t, in_launch

    out = {}
    layers, seed = [], 100 + 1000 * rank
    for (C, hw, half, count) in RESNET50_CONV_OUTPUTS:
        for _ in range(count):
            layers.append((laplace_activation((per_rank, C, hw, hw), seed, device), half))
            seed += 1
    elems = sum(x.numel() for x, _ in layers) * world        # the job's elements (shards are equal up to one sample)
    ys = [torch.empty_like(x) for x, _ in layers]
    big = max(range(len(layers)), key=lambda i: layers[i][0].numel())
    xb, hb = layers[big]
    Cb = xb.shape[1]
    # ---- config 3
    t, inl = leg(lambda: [ops.act_qdq_per_channel(x, 4, positive=half, clip='laplace', bit_alloc=True, group=group, out=y)
                          for (x, half), y in zip(layers, ys)])
    y3, p3 = ops.act_qdq_per_channel(xb, 4, positive=hb, clip='laplace', bit_alloc=True, group=group, want_parts=True)
    bits3 = p3['diag'][Lb.DIAG_BITS]
    sc, zp, qm = (p3['qp'][r].view(1, Cb, 1, 1) for r in (Lb.QP_SCALE, Lb.QP_ZP, Lb.QP_QMAX))
    codes = torch.round(y3 / sc + zp)
    ok3 = (bool(torch.equal(y3, ys[big])) and float(codes.min()) >= 0 and bool((codes <= qm).all()) and bool(torch.equal((codes - zp) * sc, y3))
           and abs(float(bits3.mean()) - 4.) <= 0.011 + 1. / Cb and bool(torch.isfinite(p3['stats'][Lb.STAT_B]).all()))
    del codes
    out['config3'] = leg_object(elems, t, 16, 'ResNet-50 b%d sharded %d ways (%d samples per GPU), per-channel int4 + ACIQ laplace + bit '
                                'allocation (-c laplace -baa): pass A, all_gather of the moment records, merge, bit allocation, %s'
                                % (per_rank * world, world, per_rank, 'ONE launch for pass B + the ranks\' sums through the windows + '
                                   'parameters + Q/DQ (cnnq_pc_aciq_fused_xrank)' if inl else 'pass B, all_gather, merge, parameters, Q/DQ (the chain)'),
                                bool(ok3), moved=12 if inl else 16)
    out['config3']['exchange'] = 'in-launch (hipIpc windows) for the sums of |x - mean|, collective for the moment records' if inl else exchange_name
    del ys, y3
    # ---- config 4
    t, inl = leg(lambda: [ops.pc_stats(x, x.shape[0], x.shape[1], x.shape[2] * x.shape[3], need_b=True, need_kurt=True,
                                       need_relu=True, group=group) for x, _ in layers])
    st4, mom4 = ops.pc_stats(xb, xb.shape[0], Cb, xb.shape[2] * xb.shape[3], need_b=True, need_kurt=True, need_relu=True, group=group)
    mx, mn = xb.amax(dim=(0, 2, 3)), xb.amin(dim=(0, 2, 3))
    if world > 1:
        on_dev = dist.get_backend(group) == 'nccl'
        mxr, mnr = (mx, mn) if on_dev else (mx.cpu(), mn.cpu())
        dist.all_reduce(mxr, op=dist.ReduceOp.MAX, group=group)
        dist.all_reduce(mnr, op=dist.ReduceOp.MIN, group=group)
        mx, mn = mxr.to(device), mnr.to(device)
    ok4 = (bool(torch.equal(st4[Lb.STAT_MAX], mx)) and bool(torch.equal(st4[Lb.STAT_MIN], mn)) and bool(torch.isfinite(st4).all())
           and float(mom4[Lb.MOM_COUNT][0]) == float(xb.shape[0] * world * xb.shape[2] * xb.shape[3]))
    lib = Lb.load()
    one_read = sum(x.numel() for x, _ in layers
                   if lib.cnnq_pc_stats_route(x.shape[0], x.shape[1], x.shape[2] * x.shape[3], 1, ops.GROUP_WS_BYTES, 0) > 0)
    frac1 = one_read * world / elems if inl else 0.
    out['config4'] = leg_object(elems, t, 8, 'ResNet-50 b%d sharded %d ways, -sm collect: the seven per-channel statistics of the GLOBAL '
                                'batch on every rank; %s' % (per_rank * world, world, ('%.0f %% of the elements in one launch and one read of '
                                                             'x with both phases\' folds exchanged inside it (cnnq_pc_stats_xrank), the rest in the '
                                                             'chain\'s passes around the same window slots' % (100. * frac1)) if inl
                                                             else 'two passes around two all_gathers of the fp64 records'),
                                bool(ok4), moved=8 - 4. * frac1)
    out['config4']['exchange'] = 'in-launch (hipIpc windows)' if inl else exchange_name
    del layers, xb
    torch.cuda.empty_cache()
    # ---- config 5
    vl, seed = [], 500 + 1000 * rank
    for (C, hw, count) in VGG16_CONV_OUTPUTS:
        for _ in range(count):
            vl.append(laplace_activation((per_rank, C, hw, hw), seed, device))
            seed += 1
    elems = sum(x.numel() for x in vl) * world
    t, inl = leg(lambda: [ops.mid_tread_qdq(x, 4, clip=True, sym=False, group=group, want_entropy=True) for x in vl], reps=2)
    xv = vl[2]
    y5, e5, p5 = ops.mid_tread_qdq(xv, 4, clip=True, sym=False, group=group, want_entropy=True, want_parts=True)
    Cv = xv.shape[1]
    d5, lo5, hi5 = (p5['mt'][r].view(1, Cv, 1, 1) for r in (Lb.MT_DELTA, Lb.MT_CMIN, Lb.MT_CMAX))
    # y = code * delta with the code an integer or exactly a clamp bound (iq.py:202-224): every y inside [c_min, c_max] * delta,
    # every count accounted for (the all-reduced table holds the GLOBAL batch's codes), a finite entropy
    c5 = y5 / d5
    ok5 = (bool((c5 >= lo5 - 1e-3).all()) and bool((c5 <= hi5 + 1e-3).all()) and math.isfinite(float(e5))
           and bool(((c5 - torch.round(c5)).abs() < 1e-3).float().mean() > 0.9)
           and int(p5['hist'][:-1].sum()) == xv.numel() * world)
    out['config5'] = leg_object(elems, t, 16, 'VGG-16 b%d sharded %d ways, mid-tread per-channel + ACIQ + bin allocation + entropy of the '
                                'GLOBAL batch\'s codes (-mtq -me): pass A, all_gather, merge, bin allocation, %s, all-reduce of the code '
                                'counts' % (per_rank * world, world, 'ONE launch for pass B + the ranks\' sums through the windows + step sizes + '
                                            'quantization + code histogram (cnnq_pc_midtread_fused_xrank)' if inl
                                            else 'pass B, all_gather, merge, parameters, quantization (the chain)'),
                                bool(ok5), moved=12 if inl else 16)
    out['config5']['exchange'] = 'in-launch (hipIpc windows) for the sums of |x - mean|, collective for the moment records and the code counts' if inl else exchange_name
    del vl, y5, c5
    torch.cuda.empty_cache()
    return out
