#!/usr/bin/env python3
"""HBM bandwidth probes with torch ops (fill = write only, sum = read only, copy = read+write)."""
import torch
n = 4 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda")
b = torch.empty(n, dtype=torch.uint8, device="cuda")
af, bf = a.view(torch.float32), b.view(torch.float32)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
ms = t(lambda: af.fill_(1.0)); print("fill  (write only): %.0f GB/s" % (n / ms / 1e6))
ms = t(lambda: bf.copy_(af)); print("copy  (read+write): %.0f GB/s (bytes moved both ways)" % (2 * n / ms / 1e6))
ms = t(lambda: af.sum()); print("sum   (read only) : %.0f GB/s" % (n / ms / 1e6))
