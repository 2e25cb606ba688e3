"""cpu_baseline leg of bench.py: the oracle (TEST infrastructure, CPU restatement of the
reference) timed on the host cores of the GPU box.  Runs in its own GPU-free process so that the
per-agglomerate eigenproblems can be spread over a fork()ed process pool, one LAPACK thread each
-- the shape of the reference's own parallelism (MPI ranks over AEs, one per core).

    python oracle/baseline_worker.py NX NY NZ LEVELS CORES   ->  one JSON line
"""
import json
import os
import sys
import time

for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[_v] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = tuple(int(v) for v in sys.argv[1:4])
    levels, cores = int(sys.argv[4]), int(sys.argv[5])
    from saamge_amd import problems
    from oracle import saamge_oracle as oracle
    cb = [(2, 2, 2)] * (levels - 2)
    prob = problems.poisson3d_problem(n, blk=(8, 8, 4), coarse_blk=cb)
    oracle.PARALLEL_CORES = cores
    t0 = time.perf_counter()
    H = oracle.ml_produce_data(prob.A, prob.elem_to_dof, prob.elmat, prob.bdr, prob.partitions,
                               theta=0.003, nu_relax=3)
    t1 = time.perf_counter()
    x, it, conv, hist = oracle.solve(H, prob.b, rel_tol=1e-8)
    t2 = time.perf_counter()
    print(json.dumps({"dofs": int(prob.A.shape[0]), "setup_s": t1 - t0, "solve_s": t2 - t1,
                      "iters": int(it), "converged": bool(conv), "cores": cores}))


if __name__ == "__main__":
    main()
