"""Process-group plumbing (one process per GPU, torch.distributed; nccl == RCCL on ROCm).

Multi-GPU state of the hot path:
  * setup: the per-agglomerate spectral problems (the dominant setup cost) are sharded -- every
    level's AEs are split into `world` contiguous ranges, each rank solves its range and the
    eigenvectors are all-gathered IN PLACE through `allgather_callback` (the reference's exchange
    of MIS-restricted eigenvectors, amg/src/contrib.cpp:519-548, is a subset of it).  Topology, P
    and RAP are still replicated on every rank.
  * solve: large levels are row-partitioned (csrc/dist.hip): halo exchange before every SpMV
    (`alltoallv` = grouped send/recv), summed inner products and restricted residuals
    (`allreduce_sum`), all-gathered corrections -- `solve_callbacks`.
The collectives are written against torch.distributed only, so the same code is rehearsed
with gloo (tests, staged through host memory) and runs on RCCL in bench.py (zero-copy on the
library's device buffers, enqueued on the library's stream)."""
import os


def _fail_fast(what, e):
    """A collective callback failed on this rank: the C side would throw here while the peers block in the
    next collective.  End the process instead, so that the launcher sees a non-zero exit and the peers'
    collectives time out / abort."""
    import sys
    print("saamge_amd %s callback failed on rank %s: %r -- aborting this rank" % (what, os.environ.get("RANK", "0"), e),
          file=sys.stderr, flush=True)
    os._exit(17)


class Group(object):
    def __init__(self, backend=None, device=None, host_buffers=False, native=False):
        """native: the collectives of setup and solve run inside the library on its own RCCL communicator
        (csrc/comm.hip: no Python in the solve loop); torch.distributed is then only the launcher's rendezvous
        (unique-id broadcast, barrier, max-over-ranks timing).  Otherwise the library calls back into
        torch.distributed (gloo rehearsals, MPI-style host codes)."""
        self.native = bool(native)
        self._comm = None
        self._comm_stream = None
        self.host_buffers = bool(host_buffers)
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            import datetime
            # (a rank that dies inside a collective must not leave its peers waiting forever)
            kw = {"timeout": datetime.timedelta(seconds=int(os.environ.get("SAAMGE_AMD_DIST_TIMEOUT", "600")))}
            if backend == "nccl" and device is not None:
                import torch
                kw["device_id"] = torch.device(device)
            dist.init_process_group(backend or "gloo", **kw)
            self.dist = dist

    def native_comm(self, stream=0):
        """saamge_amd_comm of this rank on `stream` (created once; rank 0's unique id travels through
        torch.distributed)."""
        import ctypes as C
        import torch  # noqa: F401  (first: csrc/comm.hip binds to the RCCL the process already maps, and that must be torch's)
        from . import capi
        if self._comm is not None:
            assert self._comm_stream == int(stream or 0), "one communicator per stream"
            return self._comm
        lib = capi.load()
        lib.saamge_amd_comm_last_error.restype = C.c_char_p
        buf = C.create_string_buffer(128)
        if self.rank == 0 and lib.saamge_amd_comm_unique_id(buf):
            raise RuntimeError("saamge_amd: " + lib.saamge_amd_comm_last_error().decode())
        obj = [buf.raw]
        if self.dist is not None:
            self.dist.broadcast_object_list(obj, src=0)
        comm = C.c_void_p()
        if lib.saamge_amd_comm_create(C.c_int(self.rank), C.c_int(self.world), obj[0], C.c_void_p(int(stream or 0)),
                                      C.byref(comm)):
            raise RuntimeError("saamge_amd: " + lib.saamge_amd_comm_last_error().decode())
        self._comm, self._comm_stream = comm, int(stream or 0)
        return comm

    def try_native(self, stream=0):
        """Bring up the library's own RCCL communicator and run its self-test (all-reduce, all-gather, all-to-all of
        known data); every step is agreed on by all ranks through torch.distributed, so that a rank that cannot load
        RCCL or fails the self-test sends the whole job to the torch.distributed callbacks instead of leaving the
        others inside a collective.  Returns True when the native communicator is in use."""
        import ctypes as C
        import sys
        import torch
        from . import capi
        if not self.native or self.world <= 1 or self.dist is None:
            return bool(self.native)

        def all_ok(flag):
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=self.device or "cpu")
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
            return bool(int(t.item()))

        lib = capi.load()
        lib.saamge_amd_comm_last_error.restype = C.c_char_p
        why = ""
        probe = C.create_string_buffer(128)
        ok = lib.saamge_amd_comm_unique_id(probe) == 0          # local: are the RCCL entry points there?
        if not ok:
            why = lib.saamge_amd_comm_last_error().decode()
        if all_ok(ok):
            try:
                comm = self.native_comm(stream)
                ok = True
            except Exception as e:      # noqa: BLE001  (reported below; the job continues on the callbacks)
                ok, why = False, str(e)
            if all_ok(ok):
                ok = lib.saamge_amd_comm_selftest(comm) == 0
                if not ok:
                    why = "self-test failed: " + lib.saamge_amd_comm_last_error().decode()
                if all_ok(ok):
                    return True
        if self.rank == 0 or why:
            print("saamge_amd: native RCCL communicator not used (%s); collectives through torch.distributed"
                  % (why or "another rank failed"), file=sys.stderr)
        if self._comm is not None:
            lib.saamge_amd_comm_destroy(self._comm)
            self._comm = None
        self.native = False
        return False

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

    def _memcpy(self):
        """hipMemcpy(hipMemcpyDefault) through the library, or -- for the CPU-only tests of the
        collective logic (`host_buffers=True`) -- a plain memmove."""
        from . import capi
        C = capi.C
        if self.host_buffers:
            def mv(dst, src, nbytes):
                C.memmove(C.c_void_p(int(dst)), C.c_void_p(int(src)), int(nbytes))
                return 0
            return mv
        lib = capi.load()
        return lambda dst, src, nbytes: lib.saamge_amd_memcpy(C.c_void_p(int(dst)), C.c_void_p(int(src)),
                                                                C.c_longlong(int(nbytes)))

    def allgather_callback(self):
        """ctypes callback for saamge_amd_params.allgather: in-place all-gather of a device
        buffer whose rank-r part is [off[r], off[r+1]) bytes."""
        import torch
        from . import capi
        memcpy = self._memcpy()
        use_cuda = self.dist.get_backend() == "nccl"
        dev = self.device if use_cuda else "cpu"

        def cb(ctx, buf, off):
            try:
                lo, hi = int(off[self.rank]), int(off[self.rank + 1])
                mine = torch.empty(max(hi - lo, 0), dtype=torch.uint8, device=dev)
                if hi > lo:
                    rc = memcpy(mine.data_ptr(), buf + lo, hi - lo)
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
                    rc = memcpy(buf + a, src.data_ptr(), b - a)
                    if rc:
                        return rc
                if use_cuda:
                    torch.cuda.synchronize()
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                _fail_fast("allgather", e)
                return 4

        return capi.ALLGATHER_FN(cb)

    # ---- solve-phase collectives ------------------------------------------------------
    def stream_ordered(self, stream=0):
        """True when the callbacks enqueue on the library's stream (RCCL on torch's current
        stream == the hierarchy's stream), so that no host synchronisation is needed."""
        if self.dist is None or self.dist.get_backend() != "nccl":
            return False
        import torch
        return int(torch.cuda.current_stream().cuda_stream) == int(stream or 0)

    def _wrap(self, ptr, count, dtype):
        """zero-copy torch view of library-owned device memory (cached: the library reuses its
        communication buffers)."""
        import torch
        key = (int(ptr), int(count), dtype)
        t = self._views.get(key)
        if t is None:
            typestr = {torch.float64: "<f8", torch.uint8: "|u1"}[dtype]

            class _Mem(object):
                pass
            m = _Mem()
            m.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr,
                                          "data": (int(ptr), False), "version": 2}
            t = torch.as_tensor(m, device=self.device)
            if len(self._views) > 256:
                self._views.clear()
            self._views[key] = t
        return t

    def solve_callbacks(self, stream=0):
        """(allreduce_sum, alltoallv) ctypes callbacks for saamge_amd_params."""
        import sys
        import torch
        from . import capi
        dist = self.dist
        use_cuda = dist.get_backend() == "nccl"
        ordered = self.stream_ordered(stream)
        self._views = {}
        world, rank = self.world, self.rank
        memcpy = self._memcpy()

        def allreduce(ctx, buf, count):
            try:
                if use_cuda:
                    dist.all_reduce(self._wrap(buf, count, torch.float64))
                    if not ordered:
                        torch.cuda.current_stream().synchronize()
                    return 0
                t = torch.empty(int(count), dtype=torch.float64)
                rc = memcpy(t.data_ptr(), buf, 8 * count)
                if rc:
                    return rc
                dist.all_reduce(t)
                return memcpy(buf, t.data_ptr(), 8 * count)
            except Exception as e:
                _fail_fast("allreduce", e)
                return 4

        plans = {}   # RCCL: (send ptr, recv ptr, sizes) -> prepared send/recv list (the library reuses its buffers)

        def alltoallv(ctx, send, soff, recv, roff):
            try:
                if use_cuda:
                    so = tuple(int(soff[r]) for r in range(world + 1))
                    ro = tuple(int(roff[r]) for r in range(world + 1))
                    key = (int(send or 0), int(recv or 0), so, ro)     # (the full per-peer layout, not only the totals)
                    ops = plans.get(key)
                    if ops is None:
                        st = self._wrap(send, so[world], torch.uint8) if so[world] else None
                        rt = self._wrap(recv, ro[world], torch.uint8) if ro[world] else None
                        ops = []
                        for r in range(world):
                            if r == rank:
                                continue
                            if so[r + 1] > so[r]:
                                ops.append(dist.P2POp(dist.isend, st[so[r]:so[r + 1]], r))
                            if ro[r + 1] > ro[r]:
                                ops.append(dist.P2POp(dist.irecv, rt[ro[r]:ro[r + 1]], r))
                        if len(plans) > 64:
                            plans.clear()
                        plans[key] = ops
                    if ops:
                        for w in dist.batch_isend_irecv(ops):
                            w.wait()
                    if not ordered:
                        torch.cuda.current_stream().synchronize()
                    return 0
                so = [int(soff[r]) for r in range(world + 1)]
                ro = [int(roff[r]) for r in range(world + 1)]
                if use_cuda:
                    st = self._wrap(send, so[world], torch.uint8) if so[world] else None
                    rt = self._wrap(recv, ro[world], torch.uint8) if ro[world] else None
                else:
                    st = torch.empty(so[world], dtype=torch.uint8)
                    rt = torch.empty(ro[world], dtype=torch.uint8)
                    if so[world]:
                        rc = memcpy(st.data_ptr(), send, so[world])
                        if rc:
                            return rc
                ops = []
                for r in range(world):
                    if r == rank:
                        continue
                    if so[r + 1] > so[r]:
                        ops.append(dist.P2POp(dist.isend, st[so[r]:so[r + 1]], r))
                    if ro[r + 1] > ro[r]:
                        ops.append(dist.P2POp(dist.irecv, rt[ro[r]:ro[r + 1]], r))
                if ops:
                    for w in dist.batch_isend_irecv(ops):
                        w.wait()
                if use_cuda:
                    if not ordered:
                        torch.cuda.current_stream().synchronize()
                elif ro[world]:
                    return memcpy(recv, rt.data_ptr(), ro[world])
                return 0
            except Exception as e:
                _fail_fast("alltoallv", e)
                return 4

        return capi.ALLREDUCE_FN(allreduce), capi.ALLTOALLV_FN(alltoallv)

    def aggregate_rate(self, units_per_rank, steps, dt):
        """whole-job throughput of N replicas: sum of units / slowest rank's time"""
        return self.world * units_per_rank * steps / self.max_time(dt)

    def close(self):
        if self._comm is not None:
            from . import capi
            capi.load().saamge_amd_comm_destroy(self._comm)
            self._comm = None
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None
