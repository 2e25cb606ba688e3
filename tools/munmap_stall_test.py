#!/usr/bin/env python3
"""Does returning pageable host memory to the kernel stall the GPU's queues?  (tools/: diagnosis of the "slow mode" of the setup,
DESIGN.md section 7.)  An anonymous mapping is filled, optionally copied to the device from a SIDE stream that is then left
idle, unmapped, and a tiny kernel on the default stream is timed."""
import mmap
import time
import numpy as np
import torch

dev = torch.device("cuda:0")
x = torch.zeros(1 << 20, device=dev)
side = torch.cuda.Stream()


def tiny():
    t0 = time.perf_counter()
    x.add_(1.0)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)


for _ in range(5):
    tiny()
print("baseline tiny kernel + sync: %.3f ms" % tiny())
for mb in (1, 4, 16):
    for kind in ("untouched", "copied on the default stream", "copied on a side stream", "copied back on a side stream"):
        res = []
        for rep in range(4):
            m = mmap.mmap(-1, mb << 20)
            a = np.frombuffer(m, dtype=np.uint8)
            a[:] = 1
            if kind != "untouched":
                t = torch.from_numpy(a)
                d = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
                if kind == "copied on the default stream":
                    d.copy_(t)
                    torch.cuda.synchronize()
                else:
                    with torch.cuda.stream(side):
                        if kind == "copied on a side stream":
                            d.copy_(t, non_blocking=True)
                        else:
                            t.copy_(d, non_blocking=True)
                    side.synchronize()
                del t, d
            del a
            torch.cuda.current_stream().synchronize()
            t0 = time.perf_counter()
            m.close()
            tc = 1e3 * (time.perf_counter() - t0)
            res.append((tc, tiny()))
        print("%3d MB %-30s: %s" % (mb, kind, "  ".join("munmap %.2f -> %.2f ms" % r for r in res)), flush=True)
