#!/usr/bin/env python3
"""Write-only streaming rate on this GPU (torch fill_ / zero_ on 1.64 GB and 205 MB tensors): the ceiling of any pass
whose traffic is mostly stores (the packed-codes -> fp32 pass writes 4 of its 4.5 bytes per element)."""
import torch

for n in (512 * 64 * 112 * 112, 64 * 64 * 112 * 112):
    bufs = [torch.empty(n, dtype=torch.float32, device='cuda') for _ in range(4)]
    for name, fn in (('fill_', lambda t: t.fill_(1.5)), ('zero_', lambda t: t.zero_())):
        for b in bufs:
            fn(b)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        best = 1e9
        for _ in range(5):
            ev[0].record()
            for b in bufs:
                fn(b)
            ev[1].record()
            torch.cuda.synchronize()
            best = min(best, ev[0].elapsed_time(ev[1]) / len(bufs))
        print('%-6s %5.0f MB: %.3f ms = %.2f TB/s' % (name, n * 4 / 1e6, best, n * 4 / best / 1e9))
