"""bench.py --gpus N without a launcher starts N rank processes itself (fresh children, before the parent
touches the GPU); here with a stand-in rank program on the CPU: environment of every rank, rank 0's JSON line
relayed, a failing rank fails the run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK_PROG = r"""
import json, os, sys
env = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
open(os.path.join(sys.argv[1], "rank%s.json" % env["RANK"]), "w").write(json.dumps(env))
if sys.argv[2] == "fail" and env["RANK"] == "1":
    sys.exit(3)
if env["RANK"] == "0":
    print(json.dumps({"n_gpus": int(env["WORLD_SIZE"])}))
"""

DRIVER = r"""
import sys
sys.path.insert(0, %r)
import bench
bench.spawn_ranks(int(sys.argv[1]), sys.argv[3:], script=sys.argv[2])
"""


def _run(tmp_path, mode, n=3):
    prog = tmp_path / "rank_prog.py"
    prog.write_text(RANK_PROG)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    return subprocess.run([sys.executable, "-c", DRIVER % ROOT, str(n), str(prog), str(tmp_path), mode], env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)


def test_launcher_starts_n_ranks_and_relays_rank0(tmp_path):
    p = _run(tmp_path, "ok")
    assert p.returncode == 0, p.stderr
    assert json.loads(p.stdout.strip())["n_gpus"] == 3
    ports = set()
    for r in range(3):
        e = json.loads((tmp_path / ("rank%d.json" % r)).read_text())
        assert e["RANK"] == e["LOCAL_RANK"] == str(r) and e["WORLD_SIZE"] == "3" and e["MASTER_ADDR"] == "127.0.0.1"
        ports.add(e["MASTER_PORT"])
    assert len(ports) == 1


def test_launcher_fails_when_a_rank_fails(tmp_path):
    p = _run(tmp_path, "fail")
    assert p.returncode != 0
