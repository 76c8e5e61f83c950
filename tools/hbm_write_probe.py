#!/usr/bin/env python3
"""Development: streaming write, read and copy rates of this GPU (torch fill_/sum/copy_ on 1 GiB)."""
import torch
n = 1 << 28
a = torch.empty(n, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
def t(f, reps=20):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
gb = n * 4 / 1e9
print("fill  %.2f TB/s" % (gb / t(lambda: a.fill_(1.0)) / 1e3))
print("read  %.2f TB/s" % (gb / t(lambda: a.sum()) / 1e3))
print("copy  %.2f TB/s (read+write)" % (2 * gb / t(lambda: b.copy_(a)) / 1e3))
for mb in (64, 256):
    m = mb << 18
    c = a[:m]
    print("fill %d MB  %.2f TB/s" % (mb, m * 4 / 1e9 / t(lambda: c.fill_(2.0), 200) / 1e3))
