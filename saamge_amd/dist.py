"""Process-group plumbing for bench.py (one process per GPU, torch.distributed).

The row-partitioned multi-GPU hierarchy is not built yet: with N > 1
every rank runs an independent replica of the workload (no data-path collective) and the
job-level number is  N * dofs * steps / max_over_ranks(time).  This module holds the only
collectives involved -- the barrier and the max-reduction of the timing -- so that they can
be exercised on CPU with the gloo backend (tests/test_dist_gloo.py)."""
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

    def aggregate_rate(self, units_per_rank, steps, dt):
        """whole-job throughput of N replicas: sum of units / slowest rank's time"""
        return self.world * units_per_rank * steps / self.max_time(dt)

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None
