"""CPU test of the N > 1 control path (gloo, world_size 2): barrier, max-over-ranks timing and
the replica aggregation bench.py uses."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import sys, time
    sys.path.insert(0, %r)
    from saamge_amd.dist import Group
    g = Group(backend="gloo")
    assert g.world == 2
    g.barrier()
    dt = 0.25 if g.rank == 0 else 0.5          # rank 1 is the slow one
    m = g.max_time(dt)
    assert abs(m - 0.5) < 1e-12, m
    rate = g.aggregate_rate(1000, 3, dt)        # 2 replicas * 1000 units * 3 steps / 0.5 s
    assert abs(rate - 12000.0) < 1e-9, rate
    g.barrier()
    g.close()
    print("rank", g.rank, "ok")
""" % ROOT)


COLLECTIVES = textwrap.dedent("""
    import sys, ctypes as C
    sys.path.insert(0, %r)
    import numpy as np
    from saamge_amd.dist import Group
    g = Group(backend="gloo", host_buffers=True)     # the collective logic on plain host memory
    world, rank = g.world, g.rank
    off_t = C.c_longlong * (world + 1)
    # in-place all-gather of ragged parts: rank r owns r + 2 doubles
    sizes = [r + 2 for r in range(world)]
    offs = np.concatenate([[0], np.cumsum(sizes)]) * 8
    buf = np.full(sum(sizes), -1.0)
    lo = offs[rank] // 8
    buf[lo:lo + sizes[rank]] = 100 * rank + np.arange(sizes[rank])
    ag = g.allgather_callback()
    assert ag(None, buf.ctypes.data, off_t(*[int(o) for o in offs])) == 0
    want = np.concatenate([100 * r + np.arange(sizes[r]) for r in range(world)])
    assert np.array_equal(buf, want), buf
    allreduce, alltoallv = g.solve_callbacks(0)
    # sum over ranks
    v = np.arange(5, dtype=np.float64) * (rank + 1)
    assert allreduce(None, v.ctypes.data, 5) == 0
    assert np.array_equal(v, np.arange(5) * sum(r + 1 for r in range(world))), v
    # halo-style exchange: rank r sends (r + 1) + q values to rank q, nothing to itself
    scnt = [0 if q == rank else rank + 1 + q for q in range(world)]
    rcnt = [0 if q == rank else q + 1 + rank for q in range(world)]
    send = np.concatenate([1000 * rank + 10 * q + np.arange(scnt[q], dtype=np.float64) for q in range(world)])
    recv = np.full(sum(rcnt), -1.0)
    so = np.concatenate([[0], np.cumsum(scnt)]) * 8
    ro = np.concatenate([[0], np.cumsum(rcnt)]) * 8
    assert alltoallv(None, send.ctypes.data, off_t(*[int(o) for o in so]), recv.ctypes.data,
                     off_t(*[int(o) for o in ro])) == 0
    want = np.concatenate([1000 * q + 10 * rank + np.arange(rcnt[q], dtype=np.float64) for q in range(world)])
    assert np.array_equal(recv, want), (recv, want)
    g.barrier()
    g.close()
    print("rank", rank, "ok")
""" % ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_group(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(rank), LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % rank in o


def _run_ranks(tmp_path, source, world):
    script = tmp_path / "worker.py"
    script.write_text(source)
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % rank in o


def test_collective_callbacks_two_and_three_ranks(tmp_path):
    """The three callbacks the library drives (in-place ragged all-gather, summed all-reduce,
    halo-style all-to-all-v) on host buffers, world sizes 2 and 3."""
    _run_ranks(tmp_path, COLLECTIVES, 2)
    _run_ranks(tmp_path, COLLECTIVES, 3)


def test_single_process_group_is_a_noop():
    sys.path.insert(0, ROOT)
    from saamge_amd.dist import Group
    env_backup = {k: os.environ.pop(k, None) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    try:
        g = Group()
        assert g.world == 1 and g.max_time(1.5) == 1.5
        assert g.aggregate_rate(10, 2, 4.0) == 5.0
        g.barrier()
        g.close()
    finally:
        for k, v in env_backup.items():
            if v is not None:
                os.environ[k] = v
