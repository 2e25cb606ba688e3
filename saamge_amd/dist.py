"""Process-group plumbing (one process per GPU, torch.distributed; nccl == RCCL on ROCm).

Multi-GPU state of the hot path: the per-agglomerate spectral problems (the dominant setup
cost) are sharded -- every level's AEs are split into `world` contiguous ranges, each rank
solves its range and the eigenvectors are all-gathered IN PLACE through the callback built
here (the reference's exchange of MIS-restricted eigenvectors, amg/src/contrib.cpp:519-548, is
a subset of it).  Topology, P, RAP and the solve are still replicated on every rank; the
row-partitioned operators with halo exchange are the next step.
The collectives are written against torch.distributed only, so the same code is rehearsed
with gloo (tests) and runs on RCCL in bench.py."""
import os


class Group(object):
    def __init__(self, backend=None, device=None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            kw = {}
            if backend == "nccl" and device is not None:
                import torch
                kw["device_id"] = torch.device(device)
            dist.init_process_group(backend or "gloo", **kw)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_time(self, dt):
        """MAX over ranks of a python float."""
        if self.dist is None:
            return float(dt)
        import torch
        t = torch.tensor([float(dt)], dtype=torch.float64, device=self.device or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def allgather_bytes(self, parts):
        """all-gather variable-size uint8 tensors (one per rank); returns the list of all parts.
        Equal-size collective on padded buffers (nccl and gloo both support it)."""
        import torch
        dist = self.dist
        mine = parts
        n = torch.tensor([mine.numel()], dtype=torch.int64, device=mine.device)
        sizes = [torch.zeros_like(n) for _ in range(self.world)]
        dist.all_gather(sizes, n)
        sizes = [int(t.item()) for t in sizes]
        mx = max(sizes + [1])
        send = torch.zeros(mx, dtype=torch.uint8, device=mine.device)
        send[:mine.numel()] = mine
        recv = [torch.empty(mx, dtype=torch.uint8, device=mine.device) for _ in range(self.world)]
        dist.all_gather(recv, send)
        return [recv[r][:sizes[r]] for r in range(self.world)]

    def allgather_callback(self):
        """ctypes callback for saamge_amd_params.allgather: in-place all-gather of a device
        buffer whose rank-r part is [off[r], off[r+1]) bytes."""
        import torch
        from . import capi
        lib = capi.load()
        use_cuda = self.dist.get_backend() == "nccl"
        dev = self.device if use_cuda else "cpu"

        def cb(ctx, buf, off):
            try:
                lo, hi = int(off[self.rank]), int(off[self.rank + 1])
                mine = torch.empty(max(hi - lo, 0), dtype=torch.uint8, device=dev)
                if hi > lo:
                    rc = lib.saamge_amd_memcpy(capi.C.c_void_p(mine.data_ptr()),
                                               capi.C.c_void_p(buf + lo), capi.C.c_longlong(hi - lo))
                    if rc:
                        return rc
                parts = self.allgather_bytes(mine)
                for r in range(self.world):
                    a, b = int(off[r]), int(off[r + 1])
                    if r == self.rank or b <= a:
                        continue
                    if parts[r].numel() != b - a:
                        return 3
                    src = parts[r].contiguous()
                    rc = lib.saamge_amd_memcpy(capi.C.c_void_p(buf + a), capi.C.c_void_p(src.data_ptr()),
                                               capi.C.c_longlong(b - a))
                    if rc:
                        return rc
                if use_cuda:
                    torch.cuda.synchronize()
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                import sys
                print("saamge_amd allgather callback failed: %r" % (e,), file=sys.stderr)
                return 4

        return capi.ALLGATHER_FN(cb)

    def aggregate_rate(self, units_per_rank, steps, dt):
        """whole-job throughput of N replicas: sum of units / slowest rank's time"""
        return self.world * units_per_rank * steps / self.max_time(dt)

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None
