"""Multi-rank rehearsals on ONE GPU (gloo, ranks share the card).

Setup: the ranks split the per-AE eigenproblems of every level and all-gather the eigenvectors;
the resulting hierarchy must be bit-identical to the single-process one (every AE is computed by
exactly one rank with the same kernels).
Solve: with dist_min_local_rows = 1 every level is row-partitioned (halo exchange before each
SpMV, summed inner products / restricted residuals, all-gathered corrections); iteration counts
must match the single-process run and vectors agree to summation-order round-off."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import sys, json
    sys.path.insert(0, %r)
    import numpy as np
    from saamge_amd import capi, problems as pr
    from saamge_amd.dist import Group
    grp = Group(backend="gloo")
    prob = pr.poisson3d_problem((16, 16, 16), blk=(8, 8, 4), coarse_blk=[(2, 2, 2)], coef="checkerboard")
    params = capi.default_params(num_coarsenings=2, keep_debug=True, coarse_rtol=1e-28,
                                 dist_min_local_rows=int(sys.argv[2]))
    h = capi.Hierarchy.from_problem(prob, params, group=grp if grp.world > 1 else None)
    out_info = [h.level_info(l) for l in range(2)]
    out = {}
    for l in range(2):
        P = h.get_csr(l, "P"); Ac = h.get_csr(l, "Ac")
        out["P%%d_data" %% l] = P.data; out["P%%d_idx" %% l] = P.indices
        out["Ac%%d_data" %% l] = Ac.data; out["Ac%%d_idx" %% l] = Ac.indices
        m, ev, X, Ds = h.get_ae_eigens(l)
        out["m%%d" %% l] = m
        out["ev%%d" %% l] = np.concatenate(ev)
    x, it, conv, hist = h.pcg(prob.b, rel_tol=1e-8)
    out["hist"] = hist; out["x"] = x; out["it"] = np.array([it])
    bb = np.cos(np.arange(prob.ND) * 0.13) * (~prob.ess)
    out["vc"] = h.vcycle(bb)
    out["sm"] = h.smoother(0, bb, np.sin(np.arange(prob.ND) * 0.7))
    out["part"] = np.array([i["row_partitioned"] for i in out_info])
    out["own"] = np.array([i["own_rows"] for i in out_info])
    np.savez(sys.argv[1], **out)
    h.close()
    grp.barrier()
    grp.close()
    print("rank", grp.rank, "done")
""" % ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(tmp_path, world, min_rows=262144, extra_env=None, tag=""):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs, outs = [], []
    for rank in range(world):
        env = dict(os.environ, WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **(extra_env or {}))
        out = str(tmp_path / ("w%d_r%d%s.npz" % (world, rank, tag)))
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, str(script), out, str(min_rows)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, lg in zip(procs, logs):
        assert p.returncode == 0, lg[-3000:]
    return [np.load(o) for o in outs]


def test_two_rank_setup_matches_single_rank(tmp_path):
    ref, = _run(tmp_path, 1)
    r0, r1 = _run(tmp_path, 2)
    for r in (r0, r1):
        for k in ref.files:
            assert np.array_equal(ref[k], r[k]), k      # bit-identical hierarchy and PCG history
    assert int(ref["it"][0]) > 0 and not r0["part"].any()


@pytest.mark.parametrize("world", [2, 3])
def test_row_partitioned_solve_matches_single_rank(tmp_path, world):
    ref, = _run(tmp_path, 1)
    ranks = _run(tmp_path, world, min_rows=1)
    n = [4913, None]
    for r in ranks:
        assert r["part"].all()                          # both levels row-partitioned
        assert int(r["it"][0]) == int(ref["it"][0])
        assert np.allclose(r["hist"], ref["hist"], rtol=1e-8)
        for k in ("x", "vc", "sm"):
            assert np.linalg.norm(r[k] - ref[k]) <= 1e-10 * np.linalg.norm(ref[k]), k
        for k in ref.files:                             # the setup is untouched by the solve mode
            if k.startswith(("P", "Ac", "m", "ev")):
                assert np.array_equal(ref[k], r[k]), k
    assert sum(int(r["own"][0]) for r in ranks) == n[0]
    for k in ("x", "vc", "sm"):                         # every rank returns the full vectors
        assert np.array_equal(ranks[0][k], ranks[-1][k]), k


def test_halo_overlap_does_not_change_the_arithmetic(tmp_path):
    """Interior rows beside the halo exchange (csrc/dist.hip: halo_then) against exchange-then-apply: the same
    kernels on the same rows, so every vector must be bit-identical."""
    on = _run(tmp_path, 2, min_rows=1)
    off = _run(tmp_path, 2, min_rows=1, extra_env={"SAAMGE_AMD_TEST_OPTIONS": "overlap=13"}, tag="_off")
    for a, b in zip(on, off):
        for k in ("x", "vc", "sm", "hist", "it"):
            assert np.array_equal(a[k], b[k]), k


RCCL_WORKER = textwrap.dedent("""
    import sys, os
    sys.path.insert(0, %r)
    import torch, ctypes as C
    import torch.distributed as dist
    from saamge_amd import capi
    from saamge_amd.dist import Group
    os.environ.update(WORLD_SIZE="1", RANK="0")
    g = Group(backend="nccl", device="cuda:0")          # world 1: no process group yet
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    g.dist = dist
    assert g.stream_ordered(0)
    allreduce, alltoallv = g.solve_callbacks(0)
    t = torch.arange(1000, dtype=torch.float64, device="cuda:0")
    w = g._wrap(t.data_ptr(), t.numel(), torch.float64)  # zero-copy view of raw device memory
    assert w.data_ptr() == t.data_ptr() and w.device == t.device
    w += 1.0
    assert float(t[0]) == 1.0 and float(t[999]) == 1000.0
    assert allreduce(None, t.data_ptr(), t.numel()) == 0  # RCCL all-reduce on the raw pointer
    torch.cuda.synchronize()
    assert float(t.sum()) == 1000 * 1001 / 2
    off = (C.c_longlong * 2)(0, 0)
    assert alltoallv(None, t.data_ptr(), off, t.data_ptr(), off) == 0
    dist.destroy_process_group()
    print("ok")
""" % ROOT)


def test_rccl_callbacks_on_raw_device_pointers(tmp_path):
    """The RCCL flavour of the solve-phase callbacks works on library-owned device memory
    through a zero-copy __cuda_array_interface__ view (one rank: the multi-rank logic is the
    gloo-tested one, two RCCL ranks cannot share a GPU)."""
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0 and b"ok" in p.stdout, p.stdout.decode()[-3000:]


NATIVE_WORKER = textwrap.dedent("""
    import sys, os
    sys.path.insert(0, %r)
    import torch, ctypes as C
    import torch.distributed as dist
    from saamge_amd import capi, problems
    from saamge_amd.dist import Group
    torch.cuda.set_device(0)
    os.environ.update(WORLD_SIZE="1", RANK="0")
    # the sequence of a bench.py rank: torch.distributed (RCCL) for the rendezvous, then the library's own communicator
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    g = Group(backend="nccl", device="cuda:0", native=True)
    g.dist = dist
    comm = g.native_comm(0)
    lib = capi.load()
    lib.saamge_amd_comm_last_error.restype = C.c_char_p
    assert lib.saamge_amd_comm_selftest(comm) == 0, lib.saamge_amd_comm_last_error()
    p = capi.default_params(num_coarsenings=1)
    assert lib.saamge_amd_params_set_comm(C.byref(p), comm) == 0
    assert p.world == 1 and p.rank == 0 and p.comm_stream_ordered == 1 and bool(p.allreduce_sum) and bool(p.alltoallv)
    # a hierarchy next to the communicator, like the bench's roofline leg
    prob = problems.poisson3d_device((16, 16, 8), blk=(8, 8, 4), device="cuda:0")
    h = capi.Hierarchy(prob.rowptr, prob.col, prob.val, prob.n, prob.elem_to_dof, prob.elmat, prob.bdr, prob.partitions,
                       prob.nparts, capi.default_params(num_coarsenings=1), prob.NE_, 8)
    x = torch.zeros_like(prob.b)
    _, it, conv, hist = h.pcg(prob.b, x, rel_tol=1e-8)
    assert conv
    h.close()
    g.close()
    print("ok")
""" % ROOT)


def test_native_rccl_communicator_single_rank(tmp_path):
    """csrc/comm.hip on one rank, in the process layout of a bench.py rank (torch.distributed's RCCL for the
    rendezvous + the library's own communicator): RCCL is found at run time (the copy the process already maps),
    the three primitives (all-reduce, all-gather, all-to-all) return what the arithmetic says, and the process
    exits cleanly.  More ranks need more GPUs than the test box has (two RCCL ranks cannot share a device): the
    multi-rank data path is covered by the gloo tests above, which drive the same library code through the
    callback plug."""
    script = tmp_path / "native_worker.py"
    script.write_text(NATIVE_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0 and b"ok" in p.stdout, p.stdout.decode()[-3000:]


PARCSR_WORKER = textwrap.dedent("""
    import sys, json
    sys.path.insert(0, %r)
    import numpy as np
    from saamge_amd import capi, problems as pr
    from saamge_amd.dist import Group
    grp = Group(backend="gloo")
    mode = sys.argv[2]
    prob = pr.poisson3d_problem((16, 16, 24), blk=(8, 8, 4), coarse_blk=[(2, 2, 2)], coef="checkerboard")
    params = capi.default_params(num_coarsenings=2, keep_debug=True, coarse_rtol=1e-28, dist_min_local_rows=int(sys.argv[3]))
    if mode == "parcsr":      # every rank passes ITS row block (diag + offd + col_map_offd), ITS elements and ITS partitions
        piece = pr.split_parcsr(prob, grp.world, 3)[grp.rank]
        h = capi.Hierarchy.from_parcsr(piece, params, group=grp if grp.world > 1 else None)
    else:
        h = capi.Hierarchy.from_problem(prob, params, group=grp if grp.world > 1 else None)
    out = {}
    for l in range(2):
        P = h.get_csr(l, "P"); Ac = h.get_csr(l, "Ac"); A = h.get_csr(l, "A")
        out["P%%d_data" %% l] = P.data; out["P%%d_idx" %% l] = P.indices
        out["Ac%%d_data" %% l] = Ac.data; out["Ac%%d_idx" %% l] = Ac.indices
        out["A%%d_data" %% l] = A.data; out["A%%d_idx" %% l] = A.indices; out["A%%d_ptr" %% l] = A.indptr
        m, ev, X, Ds = h.get_ae_eigens(l)
        out["m%%d" %% l] = m
        out["ev%%d" %% l] = np.concatenate(ev)
        mises, k, ncols, flags = h.get_mis(l)
        out["mises%%d" %% l] = mises; out["k%%d" %% l] = k
    x, it, conv, hist = h.pcg(prob.b, rel_tol=1e-8)
    out["hist"] = hist; out["x"] = x; out["it"] = np.array([it])
    out["own_ae"] = np.array([h.level_info(l)["own_rows"] for l in range(2)])
    np.savez(sys.argv[1], **out)
    h.close()
    grp.barrier()
    grp.close()
    print("rank", grp.rank, "done")
""" % ROOT)


def _run_parcsr(tmp_path, world, mode, min_rows=262144):
    script = tmp_path / "parcsr_worker.py"
    script.write_text(PARCSR_WORKER)
    port = _free_port()
    procs, outs = [], []
    for rank in range(world):
        env = dict(os.environ, WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        out = str(tmp_path / ("p%d_r%d_%s_%d.npz" % (world, rank, mode, min_rows)))
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, str(script), out, mode, str(min_rows)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join("--- rank %d (rc %s) ---\n%s" % (i, p.returncode, lg[-2500:]) for i, (p, lg) in enumerate(zip(procs, logs)))
    return [np.load(o) for o in outs]


@pytest.mark.parametrize("world", [1, 2, 3])
def test_per_rank_parcsr_inputs_give_the_single_rank_hierarchy(tmp_path, world):
    """saamge_amd_ml_produce_data_parcsr: every rank hands over what a rank of the reference's pmltest holds -- its row block
    in hypre's ParCSR split (diag with the diagonal entry first and unsorted rows, offd, col_map_offd), its own elements
    (global dof ids) with THEIR matrices only, the flags of its own rows, its own agglomerate partitions with local ids.
    The hierarchy (operators of both levels, eigenvector counts and eigenvalues, MIS tables, P, Ac) and the PCG history
    must be bit-identical to the single-rank hierarchy built from the assembled global problem."""
    ref, = _run_parcsr(tmp_path, 1, "global")
    ranks = _run_parcsr(tmp_path, world, "parcsr")
    for r in ranks:
        for k in ref.files:
            if k != "own_ae":
                assert np.array_equal(ref[k], r[k]), k
    assert int(ref["it"][0]) > 0


def test_per_rank_parcsr_inputs_with_a_row_partitioned_solve(tmp_path):
    ref, = _run_parcsr(tmp_path, 1, "global")
    ranks = _run_parcsr(tmp_path, 2, "parcsr", min_rows=1)
    for r in ranks:
        assert int(r["it"][0]) == int(ref["it"][0])
        assert np.allclose(r["hist"], ref["hist"], rtol=1e-8)
        assert np.linalg.norm(r["x"] - ref["x"]) <= 1e-10 * np.linalg.norm(ref["x"])
        for k in ref.files:
            if k.startswith(("P", "Ac", "m", "ev", "A", "k")):
                assert np.array_equal(ref[k], r[k]), k


MEMORY_WORKER = textwrap.dedent("""
    import sys, json, gc
    sys.path.insert(0, %r)
    import numpy as np, torch
    from saamge_amd import capi, problems as pr
    from saamge_amd.dist import Group
    grp = Group(backend="gloo", device="cuda:0")
    n = int(sys.argv[2])
    prob = pr.poisson3d_device(n, blk=(8, 8, 4), coarse_blk=[(8, 8, 4)], device="cuda:0")
    params = capi.default_params(num_coarsenings=2, theta=0.003)
    b = prob.b
    if grp.world > 1:
        piece = pr.split_parcsr_device(prob, grp.world, grp.rank, 3)
        del prob
        gc.collect(); torch.cuda.empty_cache()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    capi.memory_stats(reset_peak=True)
    if grp.world > 1:
        h = capi.Hierarchy.from_parcsr(piece, params, group=grp)
    else:
        h = capi.Hierarchy(prob.rowptr, prob.col, prob.val, prob.n, prob.elem_to_dof, prob.elmat, prob.bdr, prob.partitions,
                           prob.nparts, params, prob.NE_, 8)
    x = torch.zeros_like(b)
    _, it, conv, hist = h.pcg(b, x, rel_tol=1e-8, max_iter=200)
    torch.cuda.synchronize()
    live, peak = capi.memory_stats()
    infos = [h.level_info(l) for l in range(2)]
    out = {"it": it, "conv": bool(conv), "dims": [i["n"] for i in infos] + [infos[-1]["ncoarse"]], "lib_peak": peak,
           "torch_peak": int(torch.cuda.max_memory_allocated()), "hist_last": float(hist[-1]), "row_partitioned": [bool(i["row_partitioned"]) for i in infos]}
    h.close()
    grp.barrier()
    grp.close()
    json.dump(out, open(sys.argv[1], "w"))
""" % ROOT)


def test_per_rank_inputs_cut_the_device_memory_of_a_rank(tmp_path):
    """3-D Poisson 128^3, three levels (the bench's agglomerate shapes): device memory of ONE rank -- the caller's input arrays
    (torch) plus everything the library holds (saamge_amd_memory_stats: operators, topology, eigensolver workspace), high-water
    marks over set-up + solve.  With two ranks and per-rank inputs (each rank holds its slab of the operator and of the
    element matrices; the library gathers the operator's rows and the integer topology, never the element matrices; the
    eigenproblems of a rank's own agglomerates only) a rank must need at most 0.6 of what the single rank needs; same level
    dimensions, same iteration count."""
    script = tmp_path / "mem_worker.py"
    script.write_text(MEMORY_WORKER)

    def run(world):
        port = _free_port()
        procs, outs = [], []
        for rank in range(world):
            env = dict(os.environ, WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            out = str(tmp_path / ("mem%d_%d.json" % (world, rank)))
            outs.append(out)
            procs.append(subprocess.Popen([sys.executable, str(script), out, "128"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        logs = [p.communicate(timeout=900)[0].decode() for p in procs]
        for p, lg in zip(procs, logs):
            assert p.returncode == 0, lg[-3000:]
        return [json.load(open(o)) for o in outs]
    one, = run(1)
    two = run(2)
    tot1 = one["lib_peak"] + one["torch_peak"]
    tot2 = max(r["lib_peak"] + r["torch_peak"] for r in two)
    print("one rank: library %.2f GB + inputs %.2f GB; two ranks, per rank: library %.2f GB + inputs %.2f GB; ratio %.3f"
          % (one["lib_peak"] / 1e9, one["torch_peak"] / 1e9, max(r["lib_peak"] for r in two) / 1e9, max(r["torch_peak"] for r in two) / 1e9, tot2 / tot1))
    for r in two:
        assert r["dims"] == one["dims"] and r["it"] == one["it"] and r["conv"]
    assert tot2 <= 0.6 * tot1, (tot1, tot2)
