"""cpu_baseline leg of bench.py: the CPU restatement (TEST infrastructure) timed on the host cores of
the GPU box, in its own GPU-free process.

Default: oracle/cpu_ref.cpp -- threaded C++, LAPACK dsygvx / dgesvd, one agglomerate per core: the shape
of the reference's own parallelism (MPI ranks over AEs, one per core).  `--python`: the numpy/scipy oracle
with the per-AE work on a fork()ed process pool (round 1's baseline, kept for comparison).

    python oracle/baseline_worker.py NX NY NZ LEVELS CORES [--python]   ->  one JSON line
"""
import json
import os
import sys
import time

for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[_v] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

COARSE_BLK = (8, 8, 4)      # the same coarse blocks as the GPU run of bench.py


def main():
    n = tuple(int(v) for v in sys.argv[1:4])
    levels, cores = int(sys.argv[4]), int(sys.argv[5])
    use_python = "--python" in sys.argv
    from saamge_amd import problems
    prob = problems.poisson3d_problem(n, blk=(8, 8, 4), coarse_blk=[COARSE_BLK] * (levels - 2))
    if use_python:
        from oracle import saamge_oracle as oracle
        oracle.PARALLEL_CORES = cores
        t0 = time.perf_counter()
        H = oracle.ml_produce_data(prob.A, prob.elem_to_dof, prob.elmat, prob.bdr, prob.partitions, theta=0.003, nu_relax=3)
        t1 = time.perf_counter()
        x, it, conv, hist = oracle.solve(H, prob.b, rel_tol=1e-8)
        t2 = time.perf_counter()
        dims = [lv.A.shape[0] for lv in H.levels] + [H.levels[-1].Ac.shape[0]]
        setup_s, solve_s = t1 - t0, t2 - t1
    else:
        from oracle import cpu_ref
        h = cpu_ref.Hierarchy(prob, num_coarsenings=levels - 1, theta=0.003, nu_relax=3, threads=cores)
        x, it, conv, hist = h.pcg(prob.b, rel_tol=1e-8)
        dims, setup_s, solve_s = h.level_dims(), h.setup_s, h.solve_s
    print(json.dumps({"dofs": int(prob.A.shape[0]), "setup_s": setup_s, "solve_s": solve_s, "iters": int(it),
                      "converged": bool(conv), "cores": cores, "level_dims": [int(v) for v in dims],
                      "hist": [float(v) for v in hist], "impl": "python" if use_python else "c++",
                      "coarse_blk": list(COARSE_BLK)}))


if __name__ == "__main__":
    main()
