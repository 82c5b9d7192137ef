#!/usr/bin/env python
"""Does a read of freshly WRITTEN data come out of the 256 MB Infinity Cache?  Times a read-only
pass over a buffer right after a kernel wrote it ("hot") and after 2 GB of other traffic ("cold"),
for several buffer sizes.  Diagnostic for the slab-ordered line passes idea."""
import json
import torch

dev = "cuda"
big = torch.empty(512 * 1024 * 1024, dtype=torch.float32, device=dev)  # 2 GB
res = {}
for mb in (32, 64, 128, 192, 256, 512):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, dtype=torch.float32, device=dev)
    out = {}
    for mode in ("hot_after_write", "hot_after_read", "cold"):
        ts = []
        for rep in range(6):
            a.fill_(1.0)
            if mode == "cold":
                big.fill_(2.0)
            elif mode == "hot_after_read":
                big.fill_(2.0)
                a.sum()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            s = a.sum()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = sorted(ts)[len(ts) // 2]
        out[mode] = {"ms": round(t, 4), "GBs": round(mb / 1024 / (t * 1e-3), 1)}
    res[f"{mb}MB"] = out
print(json.dumps(res, indent=1))
