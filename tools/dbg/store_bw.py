#!/usr/bin/env python3
"""HBM write bandwidth seen by plain store kernels (torch fill_ / copy_) at the sizes of conv1's activation stash."""
import torch
dev = "cuda"
for mb in (115, 460, 1840):
    x = torch.empty(mb * 1024 * 1024 // 4, device=dev)
    y = torch.empty_like(x)
    for name, fn, nbytes in (("fill_", lambda: x.fill_(1.0), x.numel() * 4), ("copy_", lambda: y.copy_(x), 2 * x.numel() * 4)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{name} {mb:5d} MB: {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s")
