#!/usr/bin/env python3
"""bench.py -- AMG setup+solve throughput on synthetic 3-D Poisson (BASELINE.json metric).

One "step" = one full pass of the hot path over one problem that is already resident in
HBM: ml_produce_data (topology -> local spectral problems -> P -> RAP, all levels) followed
by PCG to sqrt((B r_k,r_k)/(B r_0,r_0)) < 1e-8.  value = DoF/s = N * steps / time.
Default workload = the configuration BASELINE.json quotes the metric on: 3-D Poisson 256^3,
3-level SAAMGE (it fits one MI355X).

    python bench.py --gpus 1 --steps K --warmup W [--n 256] [--levels 3]

Prints ONE JSON line (rank 0).  Extra legs, outside the timed region:
  * roofline: one more step with the library's HIP-event pair around every kernel launch
    (events recorded on the launch stream); reported for the kernel with the largest total
    time (the SpMV family is listed per operator, i.e. per level).  `achieved` / `frac` price the
    bytes the kernel has to move IN THE FORMAT IT RUNS (coded SELL slices + slice tables + tile
    descriptors + vectors: the library's census of the operator, csrc/sparse.hip build_sell) --
    a fraction of the 8 TB/s HBM peak that cannot exceed 1.  The SURVEY 8(d) figure (the CSR
    stream of the reference, 12 B per stored entry) is kept beside it as `csr_model_*`.
    `traffic` = HBM bytes per launch from committed rocprofv3 --pmc passes over the same
    kernel at the same size (profiles/r03_pmc_traffic.json, tools/pmc_lab.sh; FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950; `traffic_source` says so).
  * general_coefficient: the same workload with a coefficient without any symmetry
    (problems.poisson3d_device(coef="skew")): no slice of the operator can be pair-coded, the
    smoother streams 9 - 12 B per entry -- what a general MFEM operator gets.
  * cpu_baseline: the CPU restatement (oracle/cpu_ref.cpp: threaded C++, LAPACK dsygvx/dgesvd, one
    agglomerate per core) on a bounded sample of the same workload, timed on this box's host cores;
    the GPU path is then run on the same sample and must give the same level dimensions, iteration
    count and (B r,r) history.

N > 1: `python bench.py --gpus N` starts N fresh rank processes itself (one per GPU, before this
process touches the GPU; under torchrun / the driver's torch.distributed.run the ranks already exist
and WORLD_SIZE must equal --gpus).  STRONG scaling on the same global problem -- the
per-agglomerate spectral problems of every level (the dominant setup cost) are sharded over the
ranks and their eigenvectors all-gathered in place, likewise the Galerkin product and the coarse
element matrices; the PCG solve is row-partitioned on the large levels (halo exchange before every
SpMV, all-reduced inner products and restricted residuals); topology, MIS SVD and P are replicated.
value = dofs * steps / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # MI355X datasheet fp64 vector == matrix

# profiler label -> (kernel symbol in rocprofv3 output, roofline bound)
KERNELS = {
    "smooth_step": ("sell_staged2_kernel<3> / sell_staged_kernel<3> / sell_spmv_kernel<3>", "hbm"),
    "spmv": ("sell_staged2_kernel<0> / sell_staged_kernel<0> / sell_spmv_kernel<0>", "hbm"),
    "spmv_residual": ("sell_staged2_kernel<1> / sell_staged_kernel<1> / sell_spmv_kernel<1>", "hbm"),
    "eig_sbr_symm": ("sbr_symm_kernel", "hbm"),
    "eig_sbr_syr2k": ("sbr_fused_kernel<false, 1>", "hbm"),
    "eig_sbr_fused": ("sbr_fused_kernel<true, 2>", "hbm"),
    "eig_sbr_fused1": ("sbr_fused_kernel<true, 1>", "hbm"),
    "eig_ss_solve": ("ss_solve_lds_pf_kernel<128>", "hbm"),
    "eig_ss_chol_lds": ("chol_band_lds2_kernel<67, false>", "hbm"),
    "eig_ss_inertia_lds": ("chol_band_lds2_kernel<67, true>", "hbm"),
    "coarse_inverse": ("gj_update_kernel", "mfma"),
    "eig_ss_solve_g": ("ss_trsolve_kernel<false, 1024>", "hbm"),
    "eig_ss_update": ("sbr_fused_kernel<false, 2, 3>", "hbm"),
    "eig_ss_update1": ("sbr_fused_kernel<false, 1, 3>", "hbm"),
    "eig_ss_panel": ("chol_panel_kernel<1024>", "hbm"),
    "eig_ss_rr": ("ss_rr_kernel", "hbm"),
    "ae_build": ("ae_build_kernel<true, 8, true>", "hbm"),
    "eig_band_chase": ("band_chase_kernel", "mfma"),
    "eig_sbr_qr": ("sbr_qr_kernel<256, true>", "mfma"),
    "ae_assemble": ("ae_assemble_kernel", "hbm"),
    "ae_scale": ("ae_scale_kernel", "hbm"),
    "rap": ("rap_numeric_kernel", "hbm"),
}


# BASELINE.json configs as bench workloads (config 1 is the reference's CPU-only ctest case: tests/test_oracle_kat.py).
# aniso128: theta = 1e-4 keeps exactly the four z-constant modes of an interior 8x8x4-element agglomerate
# (generalised eigenvalues 0, 0.371e-4 (twice), 0.723e-4, then 1.444e-4: "4 eigenvectors/aggregate"); three
# levels, so that the coarsest operator stays small enough for the direct coarsest solve; the second coarsening
# uses 4x4x2-AE blocks and theta = 1e-5 (2-4 vectors per level-1 agglomerate); aniso128_c884 below: 8x8x4 blocks.
WORKLOADS = {
    "poisson256": {"n": 256, "levels": 3, "theta": 0.003, "aniso": 1.0, "theta2": None},
    "poisson128": {"n": 128, "levels": 2, "theta": 0.003, "aniso": 1.0, "theta2": None},
    "aniso128": {"n": 128, "levels": 3, "theta": 1e-4, "aniso": 1000.0, "theta2": 1e-5, "coarse_blk": "4,4,2"},
    # the same with the other configs' 8x8x4-AE coarse blocks: 32 level-1 agglomerates of ~7 900 rows, 24 of them with
    # EIGHT wanted pairs -- more than the six a block of the few-eigenpairs path holds: the first six are locked and the
    # iteration goes on (csrc/eig2.hip, ss_lock_kernel); before round 3's last part the level fell back to the dense path
    "aniso128_c884": {"n": 128, "levels": 3, "theta": 1e-4, "aniso": 1000.0, "theta2": 1e-5, "coarse_blk": "8,8,4"},
    # BASELINE config 5: 3-D linear elasticity, Q2 hexes (81 dofs per element), 4x4x4-element agglomerates of
    # 9^3 nodes x 3 = 2187 dofs carrying the six rigid-body modes, clamped on x = 0.  96^3 elements: 21.6 M dofs,
    # 4.17e9 stored entries (64-bit row offsets: csrc/common.h roff_t), ~280 GB of HBM in use.  elasticity_q2_64:
    # the 64^3 case (6.44 M dofs, 1.24e9 entries) for a quicker run.
    "elasticity_q2": {"n": 96, "levels": 4, "theta": 0.003, "aniso": 1.0, "theta2": None, "blk": "4,4,4", "coarse_blk": "2,2,2"},
    "elasticity_q2_64": {"n": 64, "levels": 4, "theta": 0.003, "aniso": 1.0, "theta2": None, "blk": "4,4,4", "coarse_blk": "2,2,2"},
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_problem(n, levels, dev, aniso=1.0, blk=(8, 8, 4), coarse_blk=(8, 8, 4), workload="poisson", coef=None):
    from saamge_amd import problems
    cb = [tuple(coarse_blk)] * (levels - 2)
    if workload == "elasticity_q2":
        return problems.elasticity3d_q2_device(n, blk=blk, coarse_blk=cb, device=dev)
    return problems.poisson3d_device(n, blk=blk, coarse_blk=cb, K=(1.0, 1.0, aniso), device=dev, coef=coef)


def one_step(capi, prob, params, rel_tol=1e-8, group=None):
    import torch
    # the library works on torch's current stream
    t0 = time.perf_counter()
    c0 = time.process_time()
    h = capi.Hierarchy(prob.rowptr, prob.col, prob.val, prob.n, prob.elem_to_dof, prob.elmat,
                       prob.bdr, prob.partitions, prob.nparts, params, prob.NE_, getattr(prob, "nde_", 8),
                       stream=torch.cuda.current_stream().cuda_stream, group=group)
    t1 = time.perf_counter()
    c1 = time.process_time()
    x = torch.zeros_like(prob.b)
    _, it, conv, hist = h.pcg(prob.b, x, rel_tol=rel_tol, max_iter=200)
    # (host wall: the setup call, the solve call; CPU time of all threads of the process in the setup call)
    one_step.last_split = (1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t1), 1e3 * (c1 - c0))
    return h, x, it, conv, hist


def cpu_baseline(n_sample, levels):
    """The CPU restatement (oracle/cpu_ref.cpp: threaded C++, LAPACK dsygvx / dgesvd, one agglomerate per
    core like the reference's one-MPI-rank-per-core runs) on a bounded sample: the same discretisation, AE
    shape and coarse blocks on a smaller box, reported per dof.  Runs in a GPU-free child process."""
    import subprocess
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "baseline_worker.py")] + \
          [str(v) for v in n_sample] + [str(levels), str(cores)]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    if out.returncode != 0:
        raise RuntimeError("cpu_baseline worker failed: " + out.stderr.decode()[-2000:])
    r = json.loads(out.stdout.decode().strip().splitlines()[-1])
    return r, {"value": r["dofs"] / (r["setup_s"] + r["solve_s"]), "unit": "DoF/s", "cores": r["cores"],
               "kind": "port",
               "sample": "oracle/cpu_ref.cpp (C++ restatement of the reference's setup + solve: LAPACK dsygvx / dgesvd per "
                         "agglomerate / MIS on %d threads, one agglomerate per core; threaded SpMV / RAP) on 3-D Poisson %s "
                         "(%d dofs), %d-level, 8x8x4-element AEs, %s-AE coarse blocks (the GPU run's shapes): setup %.2f s, "
                         "solve %.2f s, %d PCG its, level dims %s"
                         % (r["cores"], "x".join(str(v) for v in n_sample), r["dofs"], levels,
                            "x".join(str(v) for v in r["coarse_blk"]), r["setup_s"], r["solve_s"], r["iters"], r["level_dims"])}


def gpu_check_on_sample(capi, n_sample, levels, dev, theta, cpu):
    """The GPU path on the cpu_baseline sample: level dimensions and iteration count must equal the CPU
    restatement's, the (B r,r) history must agree to 1e-8 (the oracle's answer is already in hand)."""
    import numpy as np
    import torch
    prob = build_problem(n_sample, levels, dev)
    params = capi.default_params(num_coarsenings=levels - 1, theta=theta, nu_relax=3)
    h, x, its, conv, hist = one_step(capi, prob, params)
    infos = [h.level_info(l) for l in range(levels - 1)]
    dims = [i["n"] for i in infos] + [infos[-1]["ncoarse"]]
    h.close()
    ch = np.array(cpu["hist"])
    dev_hist = float(np.max(np.abs(hist - ch) / ch)) if len(hist) == len(ch) else float("inf")
    ok = dims == cpu["level_dims"] and its == cpu["iters"] and dev_hist <= 1e-8
    res = {"level_dims_gpu": dims, "level_dims_cpu": cpu["level_dims"], "pcg_iterations_gpu": its,
           "pcg_iterations_cpu": cpu["iters"], "max_rel_history_deviation": dev_hist, "identical": bool(ok)}
    if not ok:
        raise SystemExit("bench.py: the GPU path and the CPU restatement disagree on the sample: %s" % json.dumps(res))
    return res


PMC_TRAFFIC_FILE = "profiles/r04_pmc_traffic.json"


def pmc_traffic(label, rows, fmt):
    """Per-launch HBM bytes of the kernel behind profiler label `label` on an operator of `rows` rows, from the committed
    PMC passes (PMC_TRAFFIC_FILE: {"<label>@<rows>": {"fetch_bytes_raw": .., "write_bytes": .., "launches": .., "format":
    {...}}}).  Only when the operator's FORMAT CENSUS (slices per format, staged tiles, dictionary pairs: the library's
    level_format) equals the one the passes were taken on -- the same number of rows in another format runs another kernel
    over other bytes (--coef skew, other theta / block settings): then None."""
    try:
        d = json.load(open(os.path.join(ROOT, PMC_TRAFFIC_FILE)))
    except Exception:
        return None
    k = d.get("%s@%d" % (label, rows))
    if not k or not k.get("launches") or fmt is None:
        return None
    want = k.get("format")
    have = {"slices": fmt["slices"], "staged_tiles": fmt["staged_tiles"], "dictionary_pairs": fmt["dictionary_pairs"]}
    if want != have:
        return None
    return (2.0 * k["fetch_bytes_raw"] + k["write_bytes"]) / k["launches"]


def side_workload(capi, name, dev, coarse_solver=None, steps=2):
    """One more BASELINE config as a short leg OUTSIDE the timed region (one warm-up step, `steps` timed ones)."""
    import torch
    w = WORKLOADS[name]
    levels = w["levels"]
    cb = tuple(int(v) for v in w.get("coarse_blk", "8,8,4").split(","))
    prob = build_problem(w["n"], levels, dev, w["aniso"], (8, 8, 4), cb)
    params = capi.default_params(num_coarsenings=levels - 1, theta=w["theta"], nu_relax=3, coarse_solver=coarse_solver)
    if w["theta2"] is not None:
        for l in range(1, capi.MAX_LEVELS):
            params.theta[l] = w["theta2"]
    torch.cuda.synchronize()
    h, x, its, conv, _ = one_step(capi, prob, params)
    h.close()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        h, x, its, conv, _ = one_step(capi, prob, params)
        if i < steps - 1:
            h.close()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    infos = [h.level_info(l) for l in range(levels - 1)]
    out = {"ms_per_step": 1e3 * dt, "value": prob.n / dt, "unit": "DoF/s", "steps": steps, "dofs": prob.n,
           "pcg_iterations": its, "converged": bool(conv), "level_dims": [i["n"] for i in infos] + [infos[-1]["ncoarse"]],
           "eigenvectors_per_AE": [round(i["nvec"] / max(i["nparts"], 1), 2) for i in infos]}
    h.close()
    del prob, x
    torch.cuda.empty_cache()
    return out


def spawn_ranks(n, argv, script=None):
    """`python bench.py --gpus N` without a launcher: N child processes, one per GPU, started before
    this process imports torch or touches the GPU (a process that initialised the GPU must never be
    re-executed).  Rank 0's JSON line is relayed; any failing rank fails the run."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, WORLD_SIZE=str(n), RANK=str(r), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # poll every child: the first one that dies takes the others with it (a rank that fails during start-up would
    # otherwise leave rank 0 in the rendezvous until the process-group timeout, the survivors holding their GPUs)
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = r
                break
        time.sleep(0.2)
    if failed is None:
        failed = next((r for r, p in enumerate(procs) if p.returncode != 0), None)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except Exception:
                p.kill()
    reader.join(timeout=5)
    rcs = [p.returncode for p in procs]
    sys.stdout.write(b"".join(c for c in chunks if c).decode())
    sys.stdout.flush()
    if any(rcs):
        log("bench.py: rank exit codes %s" % rcs)
        sys.exit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="poisson256", choices=sorted(WORKLOADS),
                    help="BASELINE.json configs: poisson256 = config 3 (the metric's configuration, default), poisson128 = "
                         "config 2, aniso128 = config 4; --n / --levels / --theta / --aniso override the preset")
    ap.add_argument("--size", "--n", dest="n", type=int, default=None)
    ap.add_argument("--levels", type=int, default=None)
    ap.add_argument("--theta", type=float, default=None)
    ap.add_argument("--theta2", type=float, default=None, help="spectral tolerance of the coarsenings after the first (default: --theta)")
    ap.add_argument("--aniso", type=float, default=None, help="K = diag(1, 1, aniso) (BASELINE config 4: 1000)")
    ap.add_argument("--blk", type=str, default=None, help="elements per AE along x,y,z (default 8,8,4; elasticity_q2: 4,4,4)")
    ap.add_argument("--coarse-blk", type=str, default=None, help="AEs per coarse AE along x,y,z (default 8,8,4)")
    ap.add_argument("--correct-nullspace", action="store_true",
                    help="extra scaling_P level under the coarsest spectral operator (reference drivers' default)")
    ap.add_argument("--nu-pro", type=int, default=0, help="prolongator smoothing degree (0 = tentative, the reference default)")
    ap.add_argument("--eigensolver", default="subspace", choices=["subspace", "dense"],
                    help="local eigensolver: few-eigenpairs path with certified count (default) or the dense two-stage path only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-general", action="store_true", help="skip the general-coefficient leg of the default workload")
    ap.add_argument("--no-others", action="store_true", help="skip the BASELINE config 2 / config 4 legs of the default workload")
    ap.add_argument("--coarse-solver", default="auto", choices=["auto", "direct", "pcg", "blocktri"],
                    help="coarsest solve: auto (dense inverse up to 8192 rows, inner PCG beyond), direct (the reference's "
                         "coarse_direct: dense inverse / block-tridiagonal elimination), pcg, blocktri")
    ap.add_argument("--coef", default=None, choices=["skew"], help="Poisson workloads: variable coefficient instead of the constant one "
                    "(the timed workload itself becomes the general-coefficient problem; used to profile it)")
    args = ap.parse_args()
    for k, v in WORKLOADS[args.workload].items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    if args.theta2 is None:
        args.theta2 = args.theta
    if args.coarse_blk is None:
        args.coarse_blk = "8,8,4"
    if args.blk is None:
        args.blk = "8,8,4"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus, sys.argv[1:])
    elasticity = args.workload.startswith("elasticity_q2")
    if elasticity and args.n >= 80:
        # 96^3 fills the card (operator 50 GB, its SELL copy 50 GB, element matrices 46 GB, eigensolver workspace):
        # the library's cache of freed blocks is kept small so that torch's own allocations find room
        os.environ.setdefault("SAAMGE_AMD_POOL_MAX_GB", "16")

    import torch
    from saamge_amd import capi
    from saamge_amd.dist import Group
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_dev = local_rank % max(torch.cuda.device_count(), 1)   # (rehearsals put 2 ranks on one GPU)
    torch.cuda.set_device(local_dev)
    dev = "cuda:%d" % local_dev
    backend = os.environ.get("SAAMGE_AMD_DIST_BACKEND", "nccl")
    # collectives: the library's own RCCL communicator (SAAMGE_AMD_DIST_COMM=torch: callbacks into torch.distributed)
    native = backend == "nccl" and os.environ.get("SAAMGE_AMD_DIST_COMM", "native") != "torch"
    grp = Group(backend=backend, device=dev, native=native)
    world, rank = grp.world, grp.rank
    if native and world > 1:
        native = grp.try_native(0)       # (falls back to the torch.distributed callbacks, on all ranks together)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with `python bench.py --gpus N` or "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (args.gpus, world))

    prob = build_problem(args.n, args.levels, dev, args.aniso, tuple(int(v) for v in args.blk.split(",")),
                         tuple(int(v) for v in args.coarse_blk.split(",")),
                         "elasticity_q2" if elasticity else "poisson", coef=args.coef)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()        # the generator's temporaries go back to the device before the library allocates
    params = capi.default_params(num_coarsenings=args.levels - 1, theta=args.theta, nu_relax=3, nu_pro=args.nu_pro,
                                 eigensolver=args.eigensolver,
                                 correct_nullspace=args.correct_nullspace,
                                 coarse_solver={"auto": None, "direct": 1, "pcg": 2, "blocktri": 3}[args.coarse_solver])
    if args.theta2 is not None:      # first_theta / theta of the reference's MultilevelParameters (inc/ml.hpp:66-70)
        for l in range(1, capi.MAX_LEVELS):
            params.theta[l] = args.theta2

    its = conv = None
    for _ in range(args.warmup):
        h, x, its, conv, hist = one_step(capi, prob, params, group=grp)
        h.close()
    grp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = [t0]
    splits = []
    closes = []
    for i in range(args.steps):
        h, x, its, conv, hist = one_step(capi, prob, params, group=grp)
        tc = time.perf_counter()
        if i < args.steps - 1:
            h.close()
        closes.append(1e3 * (time.perf_counter() - tc))
        marks.append(time.perf_counter())      # (a step ends with the iteration count on the host: no extra synchronisation)
        splits.append(getattr(one_step, "last_split", (0.0, 0.0, 0.0)) + capi.pool_counts(reset=True)[:3:2])
    if rank == 0:
        sys.stderr.write("bench: wall ms of the timed steps: %s\n" % " ".join("%.1f" % (1e3 * (b - a)) for a, b in zip(marks, marks[1:])))
        sys.stderr.write("bench: release of the hierarchy after those steps (ms): %s\n" % " ".join("%.1f" % c for c in closes))
        sys.stderr.write("bench: setup / solve calls of those steps (ms; (CPU ms of the process in the setup call) + hipMalloc / hipFree calls "
                         "of the block cache): %s\n" % " ".join("%.0f/%.0f(%.0f)+%d/%d" % sp for sp in splits))
    grp.barrier()
    torch.cuda.synchronize()
    dt = grp.max_time(time.perf_counter() - t0)
    infos = [h.level_info(l) for l in range(args.levels - 1 + int(args.correct_nullspace))]
    # size-independent check on the full-size problem: true residual of the computed solution
    # (torch's CSR product, by row blocks of at most 2^28 entries: its int64 copy of the column indices
    # of a 4e9-entry operator would not fit beside the hierarchy)
    A_rowptr, A_col, A_val = prob.rowptr.long(), prob.col, prob.val
    res2 = 0.0
    r0 = 0
    while r0 < prob.n:
        e0 = int(A_rowptr[r0])
        r1 = int(torch.searchsorted(A_rowptr, torch.tensor(e0 + (1 << 28), device=A_rowptr.device)))
        r1 = min(prob.n, max(r1 - 1, r0 + 1))
        e1 = int(A_rowptr[r1])
        Ablk = torch.sparse_csr_tensor(A_rowptr[r0:r1 + 1] - e0, A_col[e0:e1].long(), A_val[e0:e1], size=(r1 - r0, prob.n))
        res2 += float(torch.sum((Ablk @ x - prob.b[r0:r1]) ** 2))
        del Ablk
        r0 = r1
    relres = float(res2 ** 0.5 / torch.linalg.norm(prob.b))
    op_formats = [dict(h.level_format(l)["slices"], dictionary_pairs=h.level_format(l)["dictionary_pairs"]) for l in range(args.levels - 1)]
    eig_solved = [[h.level_format(l)["eigenproblems_solved"], infos[l]["nparts"]] for l in range(args.levels - 1)]
    h.close()

    res = {
        "metric": "AMG setup+solve DoF/s (%s, PCG to 1e-8)" % ("3D elasticity Q2" if elasticity else "3D Poisson"),
        "value": prob.n * args.steps / dt,
        "unit": "DoF/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "strong",     # the same global problem at every N (BASELINE config 3: "256^3 ... 1 -> 8 MI355X")
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload_id": args.workload,
                   "workload": ("3D linear elasticity %d^3 Q2 hexes (81 dofs per element, lambda = mu = 1, clamped on x = 0), "
                                "%d-level SAAMGE, theta=%g, nu_relax=3, %s-element AEs, %s-AE coarse blocks"
                                % (args.n, args.levels, args.theta, args.blk.replace(",", "x"), args.coarse_blk.replace(",", "x")))
                   if elasticity else "3D Poisson %d^3 Q1 hexes%s, %d-level SAAMGE, theta=%s, nu_relax=3, "
                               "8x8x4-element AEs, %s-AE coarse blocks%s" % (args.n, (" coefficient=" + args.coef if args.coef else "") if args.aniso == 1.0 else
                                                       " K=diag(1,1,%g)" % args.aniso, args.levels,
                                                       ("%g" % args.theta) if args.theta2 == args.theta else "%g / %g (first / later coarsenings)" % (args.theta, args.theta2),
                                                       args.coarse_blk.replace(",", "x"),
                                                       ("" if args.nu_pro == 0 else ", nu_pro=%d" % args.nu_pro) +
                                                       (", corrected null-space level" if args.correct_nullspace else "")),
                   "eigenvectors_per_AE": [round(i["nvec"] / max(i["nparts"], 1), 2) for i in infos],
                   "dofs": prob.n, "pcg_iterations": its, "converged": bool(conv),
                   "true_relative_residual": relres,
                   "level_dims": [i["n"] for i in infos] + [infos[-1]["ncoarse"]],
                   "operator_formats": op_formats,
                   # local eigenproblems solved / agglomerates per level (rank 0's share): the agglomerates of a structured mesh with
                   # piecewise constant coefficients fall into a few classes of bitwise identical matrices, each solved once
                   # (saamge_amd_options.eig_dedupe; general_coefficient below has no such classes)
                   "eigenproblems_solved": eig_solved,
                   "collectives": ("RCCL inside the library (csrc/comm.hip)" if native else "torch.distributed callbacks (%s)" % backend)
                   if world > 1 else None,
                   "parallelism": ("%d ranks: per-AE spectral problems, RAP and coarse element matrices sharded + "
                                   "all-gathered; levels %s solved row-partitioned (halo exchange per SpMV, "
                                   "all-reduced dots); topology/MIS SVD/P and smaller levels replicated"
                                   % (world, [l for l, i in enumerate(infos) if i["row_partitioned"]]))
                   if world > 1 else "single GPU"},
    }

    if rank == 0 and not args.no_roofline:
        capi.profile(True)
        capi.profile_reset()
        h, x, its2, conv2, hist2 = one_step(capi, prob, params)      # (single rank: no group)
        h_formats = [h.level_format(l) for l in range(args.levels - 1)]
        h.close()
        capi.profile(False)
        stats = sorted(capi.profile_stats(), key=lambda s: -s["ms"])
        tot = sum(s["ms"] for s in stats)
        for s in stats:
            log("  %-28s %9.3f ms %6d launches  %8.1f GB/s (format)  %8.1f GB/s (CSR model)  %8.2f TFLOP/s"
                % (s["name"], s["ms"], s["launches"], s["fmt_bytes"] / max(s["ms"], 1e-9) / 1e6,
                   s["bytes"] / max(s["ms"], 1e-9) / 1e6, s["flops"] / max(s["ms"], 1e-9) / 1e9))
        log("  kernel total %.3f ms (profiled step)" % tot)
        d = stats[0]
        label, _, rows = d["name"].partition("@")
        symbol, bound = KERNELS.get(label, (label, "hbm"))
        avg_ms = d["ms"] / d["launches"]
        lev_of = [l for l, i in enumerate(infos) if rows and i["n"] == int(rows)]
        traffic = pmc_traffic(label, int(rows), h_formats[lev_of[0]] if lev_of and lev_of[0] < len(h_formats) else None) if rows else None
        if bound == "hbm":
            ach = d["fmt_bytes"] / d["ms"] / 1e6
            res["roofline"] = {"kernel": symbol, "profiler_label": d["name"], "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                               "traffic_source": (PMC_TRAFFIC_FILE + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this kernel "
                                                  "on an operator of this size AND format census, committed; not collected in this "
                                                  "run)") if traffic else None,
                               "format_bytes_per_launch": d["fmt_bytes"] / d["launches"],
                               "csr_model_bytes_per_launch": d["bytes"] / d["launches"],
                               "csr_model_GBps": d["bytes"] / d["ms"] / 1e6,
                               "avg_launch_ms": avg_ms, "launches": d["launches"]}
            if label in ("smooth_step", "spmv", "spmv_residual") and rows:
                lev = [l for l, i in enumerate(infos) if i["n"] == int(rows)]
                if lev:
                    res["roofline"]["operator_format"] = h_formats[lev[0]]
                    if h_formats[lev[0]]["dictionary_pairs"]:      # plain slices replaced by the operator-level dictionary
                        mode = symbol[symbol.index("<"):symbol.index(">") + 1]
                        res["roofline"]["kernel"] = ("sell_gpair3_kernel%s (+ sell_gpair3_fix_kernel)" if h_formats[lev[0]]["node_blocks"]
                                                     else "sell_gpair_kernel%s") % mode
        else:
            ach = d["flops"] / d["ms"] / 1e9
            res["roofline"] = {"kernel": symbol, "profiler_label": d["name"], "bound": "mfma", "achieved": ach,
                               "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / FP64_PEAK_TFLOPS, "traffic": traffic,
                               "avg_launch_ms": avg_ms, "launches": d["launches"]}
        res["kernels"] = [{"name": s["name"], "ms": round(s["ms"], 3), "launches": s["launches"],
                           "format_GBps": round(s["fmt_bytes"] / max(s["ms"], 1e-9) / 1e6, 1),
                           "csr_model_GBps": round(s["bytes"] / max(s["ms"], 1e-9) / 1e6, 1)}
                          for s in stats[:10]]
    if rank == 0 and world == 1 and not args.no_general and args.workload == "poisson256" and args.aniso == 1.0 and args.coef is None:
        # the same workload on a GENERAL operator (outside the timed region): a coefficient without any symmetry, so that
        # no 64-row slice repeats its (offset, value) pairs -- the pair-coded format of the constant-coefficient headline
        # does not apply and the smoother streams the values
        del prob
        torch.cuda.empty_cache()
        gprob = build_problem(args.n, args.levels, dev, args.aniso, tuple(int(v) for v in args.blk.split(",")),
                              tuple(int(v) for v in args.coarse_blk.split(",")), "poisson", coef="skew")
        torch.cuda.synchronize()
        hg, xg, itg, convg, _ = one_step(capi, gprob, params)
        hg.close()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gsteps = 2
        for i in range(gsteps):
            hg, xg, itg, convg, _ = one_step(capi, gprob, params)
            if i < gsteps - 1:
                hg.close()
        torch.cuda.synchronize()
        gdt = (time.perf_counter() - t0) / gsteps
        ginfos = [hg.level_info(l) for l in range(args.levels - 1)]
        res["config"]["general_coefficient"] = {
            "coefficient": "exp(0.7x + 0.4y - 0.3z) (1 + 0.3 sin(5x + 3y + 7z)) per element (problems.poisson3d_device(coef='skew'))",
            "ms_per_step": 1e3 * gdt, "value": gprob.n / gdt, "unit": "DoF/s", "steps": gsteps,
            "pcg_iterations": itg, "converged": bool(convg),
            "level_dims": [i["n"] for i in ginfos] + [ginfos[-1]["ncoarse"]],
            "slice_formats": [hg.level_format(l)["slices"] for l in range(args.levels - 1)]}
        hg.close()
        del gprob, xg
        torch.cuda.empty_cache()
    if (rank == 0 and world == 1 and not args.no_others and args.workload == "poisson256" and args.n == 256 and args.aniso == 1.0
            and args.coef is None and args.nu_pro == 0 and not args.correct_nullspace):
        # BASELINE configs 2 and 4 in the driver's line (outside the timed region; config 5 fills the card and takes ~10 s per
        # step: it stays a separate run, `--workload elasticity_q2`, profiles/r04_bench_elasticity_q2.json)
        try:
            del prob
        except NameError:
            pass
        torch.cuda.empty_cache()
        res["config"]["other_workloads"] = {
            "poisson128": dict(side_workload(capi, "poisson128", dev), config="BASELINE config 2: 3D Poisson 128^3 Q1, theta=0.003, 2-level; "
                               "coarsest solve: inner PCG (the default beyond 8192 rows)"),
            "poisson128_coarse_direct": dict(side_workload(capi, "poisson128", dev, coarse_solver=1), config="the same with the reference's "
                                             "coarse_direct: block-tridiagonal elimination of the 67 975-row coarsest operator (csrc/blocktri.hip)"),
            "aniso128": dict(side_workload(capi, "aniso128", dev), config="BASELINE config 4: K=diag(1,1,1000) 128^3, theta=1e-4 / 1e-5, "
                             "3-level, 4x4x2-AE coarse blocks"),
            "elasticity_q2": "not run here (BASELINE config 5 fills the card, ~10 s per step): python bench.py --workload elasticity_q2"}
    if rank == 0 and not args.no_cpu_baseline and args.workload != "poisson256":
        log("bench.py: cpu_baseline is timed on the default workload only (poisson256)")
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "poisson256" and args.blk == "8,8,4" and args.coef is None:
        sample = (96, 96, 64)       # ~15 s on the GPU box's 16 cores
        cpu_raw, res["cpu_baseline"] = cpu_baseline(sample, args.levels)
        if args.aniso == 1.0 and args.nu_pro == 0 and not args.correct_nullspace and args.blk == "8,8,4":
            res["cpu_baseline"]["gpu_check"] = gpu_check_on_sample(capi, sample, args.levels, dev, args.theta, cpu_raw)
    if rank == 0:
        print(json.dumps(res), flush=True)
    grp.close()


if __name__ == "__main__":
    main()
