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
