"""GPU parity AT SCALE: the HIP path against oracle goldens generated at the headline agglomerate
shapes (tests/golden/make_golden_scale.py: 8x8x4-element AEs, 8x8x4-AE coarse blocks, 3 levels,
up to 128^3 = 2.1 M dofs).  What `north_star` wants identical is required identical: level
dimensions, eigenvectors per agglomerate, coarse dofs per MIS, PCG iteration count; the (B r,r)
history to HIST_TOL relative.  Each case runs through the default eigensolver (few-eigenpairs path,
certified count) and through the dense path (dsygvx's algorithm).  Sizes above the small fixtures
exercise what only exists at scale: several workspace chunks, the wide-band level-1 factorisation
and inertia pass, multi-pass RAP."""
import os

import numpy as np
import pytest

from saamge_amd import problems as pr

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HIST_TOL = 1e-8     # relative, every entry of the (B r_k, r_k) history (3 levels, 10-15 iterations)


def _run(name, eigensolver, workspace_bytes=None):
    import torch
    from saamge_amd import capi
    g = np.load(os.path.join(GOLD, name + ".npz"))
    n = tuple(int(v) for v in g["dims"])
    skew = name.endswith("_skew")
    params = capi.default_params(num_coarsenings=2, theta=float(g["theta"][0]), nu_relax=3,
                                 eigensolver=eigensolver, workspace_bytes=workspace_bytes)
    if skew:
        prob = pr.poisson3d_problem(n, blk=(8, 8, 4), coarse_blk=[(8, 8, 4)], coef="skew")
        h = capi.Hierarchy.from_problem(prob, params)
        b = prob.b
        x, it, conv, hist = h.pcg(b, rel_tol=1e-8, max_iter=100)
        nrm = float(np.linalg.norm(x))
    else:
        prob = pr.poisson3d_device(n, blk=(8, 8, 4), coarse_blk=[(8, 8, 4)], device="cuda:0")
        h = capi.Hierarchy(prob.rowptr, prob.col, prob.val, prob.n, prob.elem_to_dof, prob.elmat, prob.bdr,
                           prob.partitions, prob.nparts, params, prob.NE_, 8)
        x = torch.zeros_like(prob.b)
        _, it, conv, hist = h.pcg(prob.b, x, rel_tol=1e-8, max_iter=100)
        nrm = float(torch.linalg.norm(x))
    infos = [h.level_info(l) for l in range(2)]
    dims = [i["n"] for i in infos] + [infos[-1]["ncoarse"]]
    out = {"dims": dims, "iters": it, "conv": conv, "hist": hist, "x_norm": nrm}
    for l in range(2):
        _, k, _, _ = h.get_mis(l)
        m = np.zeros(infos[l]["nparts"], dtype=np.int32)
        import ctypes as C
        capi._check(capi.load().saamge_amd_get_ae_eigens(h.h, C.c_int(l), capi._ptr(m), None, None, None))
        out["l%d_ae_m" % l] = m
        out["l%d_mis_k" % l] = k
    h.close()
    return g, out


def _check(g, out, tag):
    for l in range(2):
        gm, gk = g["l%d_ae_m" % l].astype(np.int32), g["l%d_mis_k" % l].astype(np.int32)
        bad_m = np.nonzero(out["l%d_ae_m" % l] != gm)[0]
        bad_k = np.nonzero(out["l%d_mis_k" % l] != gk)[0]
        assert bad_m.size == 0, "%s level %d: eigenvector counts differ on AEs %s (gpu %s, oracle %s)" % (
            tag, l, bad_m[:8], out["l%d_ae_m" % l][bad_m[:8]], gm[bad_m[:8]])
        assert bad_k.size == 0, ("%s level %d: coarse dofs differ on MISes %s (gpu %s, oracle %s; oracle's smallest kept / "
                                 "largest dropped sigma ratio there: %s / %s)" % (
                                     tag, l, bad_k[:8], out["l%d_mis_k" % l][bad_k[:8]], gk[bad_k[:8]],
                                     g["l%d_sv_min_kept" % l][bad_k[:8]], g["l%d_sv_max_dropped" % l][bad_k[:8]]))
    assert out["dims"] == [int(v) for v in g["level_dims"]], (tag, out["dims"], g["level_dims"])
    assert out["conv"] and out["iters"] == int(g["pcg_iters"][0]), (tag, out["iters"], g["pcg_iters"])
    gh = g["pcg_hist"]
    rel = np.abs(out["hist"] - gh) / gh
    print("%s: dims %s, %d iterations, max relative history deviation %.2e, |x| deviation %.2e"
          % (tag, out["dims"], out["iters"], rel.max(), abs(out["x_norm"] - float(g["x_norm"][0])) / float(g["x_norm"][0])))
    assert rel.max() <= HIST_TOL, (tag, rel)


@pytest.mark.parametrize("eigensolver", ["subspace", "dense"])
@pytest.mark.parametrize("name", ["scale_64x64x32", "scale_64x64x32_skew", "scale_96x96x64", "scale_96x96x64_skew"])
def test_scale_golden(name, eigensolver):
    g, out = _run(name, eigensolver)
    _check(g, out, "%s/%s" % (name, eigensolver))


def test_scale_golden_with_single_matrices_on_the_dense_path():
    """A few matrices of a chunk leave the few-eigenpairs path (no certificate, a non-positive pivot, no
    convergence ...): they alone are redone by the dense path and the chunk's results are put together again
    (csrc/hierarchy.hip, post()).  saamge_amd_options.eig_force_fallback = 50 marks every 50th agglomerate: 10 of the 512 level-0
    agglomerates go alone (in runs of one; with 1 GiB chunks also across chunk boundaries), the one marked
    agglomerate of the 8 on level 1 exceeds the 10 % limit and takes the whole level with it."""
    from saamge_amd import capi
    old = capi.set_options(eig_force_fallback=50)
    try:
        g, out = _run("scale_64x64x32_skew", "subspace")
        _check(g, out, "scale_64x64x32_skew/subspace/forced-bad")
        g, out = _run("scale_96x96x64", "subspace", workspace_bytes=1 << 30)
        _check(g, out, "scale_96x96x64/subspace/forced-bad/1GiB-chunks")
    finally:
        capi.set_options(eig_force_fallback=old.eig_force_fallback)


def test_scale_golden_chunked():
    """The same answers when the level-0 agglomerates go through several workspace chunks."""
    g, out = _run("scale_96x96x64", "subspace", workspace_bytes=1 << 30)
    _check(g, out, "scale_96x96x64/subspace/1GiB-chunks")


@pytest.mark.parametrize("eigensolver", ["subspace", "dense"])
def test_scale_golden_128(eigensolver):
    """128^3 3-level (BASELINE config 2's size with config 3's depth): the oracle (LAPACK) gives a coarsest
    dimension of 151; round 1's default path gave 153 here."""
    g, out = _run("scale_128", eigensolver)
    _check(g, out, "scale_128/%s" % eigensolver)


def test_coarse_level_switches_give_the_same_hierarchy():
    """The round-2 switches of the coarse-level eigenproblems -- band-limited assembly / scaling of the few large
    agglomerates (saamge_amd_options.band_assembly), kept inertia factor (eig_keep_inertia_factor), subspace iteration beside
    the next chunk (overlap bit 0, with the level-0 agglomerates split over several chunks) -- each switched
    off in a process of its own: identical level dimensions, eigenvector counts and iteration counts, history to 1e-9
    (the golden is the oracle's; this is the library against itself)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, json; sys.path.insert(0, %r)\n"
        "import numpy as np, torch, ctypes as C\n"
        "from saamge_amd import capi, problems as pr\n"
        "prob = pr.poisson3d_device((64, 64, 32), blk=(8, 8, 4), coarse_blk=[(8, 8, 4)], device='cuda:0')\n"
        "params = capi.default_params(num_coarsenings=2, theta=0.003, nu_relax=3, workspace_bytes=1 << 28)\n"
        "h = capi.Hierarchy(prob.rowptr, prob.col, prob.val, prob.n, prob.elem_to_dof, prob.elmat, prob.bdr,\n"
        "                   prob.partitions, prob.nparts, params, prob.NE_, 8)\n"
        "x = torch.zeros_like(prob.b)\n"
        "_, it, conv, hist = h.pcg(prob.b, x, rel_tol=1e-8, max_iter=100)\n"
        "infos = [h.level_info(l) for l in range(2)]\n"
        "ms = []\n"
        "for l in range(2):\n"
        "    m = np.zeros(infos[l]['nparts'], dtype=np.int32)\n"
        "    capi._check(capi.load().saamge_amd_get_ae_eigens(h.h, C.c_int(l), capi._ptr(m), None, None, None))\n"
        "    ms.append(m.tolist())\n"
        "print(json.dumps({'dims': [i['n'] for i in infos] + [infos[-1]['ncoarse']], 'it': it, 'conv': bool(conv),\n"
        "                  'hist': list(hist), 'm': ms}))\n" % root)
    outs = []
    for extra in ({}, {"SAAMGE_AMD_TEST_OPTIONS": "band_assembly=0"}, {"SAAMGE_AMD_TEST_OPTIONS": "eig_keep_inertia_factor=0"},
                  {"SAAMGE_AMD_TEST_OPTIONS": "overlap=14"}):
        o = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=600)
        assert o.returncode == 0, o.stderr[-2000:]
        outs.append(json.loads(o.stdout.strip().splitlines()[-1]))
    g = np.load(os.path.join(GOLD, "scale_64x64x32.npz"))
    assert outs[0]["dims"] == [int(v) for v in g["level_dims"]] and outs[0]["it"] == int(g["pcg_iters"][0])
    for other in outs[1:]:
        assert other["conv"] and other["dims"] == outs[0]["dims"] and other["it"] == outs[0]["it"] and other["m"] == outs[0]["m"]
        assert np.allclose(other["hist"], outs[0]["hist"], rtol=1e-9, atol=0.0)


@pytest.mark.parametrize("flip", [
    {"eig_min_n": 0}, {"eig_nullcheck": 0}, {"eig_keep_inertia_factor": 0}, {"band_assembly": 0}, {"overlap": 0},
    {"eig_outer_panels": 2}, {"eig_outer_panels": 4}, {"eig_dedupe": 0}, {"host_heap_pad_mb": 0},
    {"sell": 0}, {"sell": 1}, {"sell": 3}, {"sell": 31 & ~4}, {"sell": 63}, {"eig_force_fallback": 7},
    {"eig_dense_one_stage": 1, "_eigensolver": "dense"}, {"eig_certify": 0}])
def test_every_remaining_option_flipped_gives_the_golden(flip):
    """saamge_amd_options (include/saamge_amd.h) is what is left of the environment switches of rounds 1-3: every field that
    selects a code path is flipped here, one at a time, on the 64 x 64 x 32 three-level golden -- the oracle's level
    dimensions, eigenvector counts, coarse dofs per MIS, iteration count and history (1e-8) must come out whichever path ran.
    (eig_strict, spmv_sell and debug select no arithmetic: tests/test_gpu_parity.py, test_gpu_sell.py use them.)"""
    from saamge_amd import capi
    flip = dict(flip)
    eigensolver = flip.pop("_eigensolver", "subspace")
    old = capi.get_options()
    capi.set_options(**flip)
    try:
        g, out = _run("scale_64x64x32", eigensolver)
        _check(g, out, "scale_64x64x32/" + ",".join("%s=%s" % kv for kv in flip.items()))
    finally:
        capi.load().saamge_amd_set_options(__import__("ctypes").byref(old))


def test_identical_agglomerates_solved_once_give_the_bitwise_same_hierarchy():
    """saamge_amd_options.eig_dedupe (csrc/eig.hip "Duplicate agglomerate matrices"): on a structured mesh most agglomerates
    are translates of one another, their local matrices identical bit for bit; one member per class is solved, the others get
    copies.  Required: the prolongators and the coarse operators of BOTH levels are bitwise those of the per-agglomerate
    computation (eig_dedupe = 0), with the level-0 agglomerates in several chunks (classes are carried from chunk to chunk),
    and far fewer eigenproblems are solved; with a coefficient without symmetry nothing is a duplicate and every agglomerate
    is solved on its own.  Reference semantics: every AE's eigenproblem is the reference's (src/spectral.cpp:124-237) --
    which ones share a solve is not observable in the result."""
    import hashlib
    import torch
    from saamge_amd import capi

    def build(dedupe, coef):
        capi.set_options(eig_dedupe=dedupe)
        try:
            if coef == "q2":      # Q2 elasticity 16^3, 4x4x4-element agglomerates of 2 187 rows: the GENERIC assembly, whose classes
                # are found on its inputs (element matrices, local numbering, rows of the global matrix), in two chunks
                prob = pr.elasticity3d_q2_device(16, blk=(4, 4, 4), coarse_blk=[(2, 2, 2)], device="cuda:0")
                params = capi.default_params(num_coarsenings=2, theta=0.003, nu_relax=3, workspace_bytes=2 << 30)
                nde = 81
            else:
                prob = pr.poisson3d_device((64, 64, 32), blk=(8, 8, 4), coarse_blk=[(8, 8, 4)], device="cuda:0", coef=coef)
                params = capi.default_params(num_coarsenings=2, theta=0.003, nu_relax=3, workspace_bytes=1 << 28)
                nde = 8
            h = capi.Hierarchy(prob.rowptr, prob.col, prob.val, prob.n, prob.elem_to_dof, prob.elmat, prob.bdr,
                               prob.partitions, prob.nparts, params, prob.NE_, nde)
            dig = []
            for l in range(2):
                for which in ("P", "Ac"):
                    M = h.get_csr(l, which).tocsr()
                    M.sort_indices()
                    dig.append(hashlib.sha256(M.indptr.tobytes() + M.indices.tobytes() + M.data.tobytes()).hexdigest())
            solved = [h.level_format(l)["eigenproblems_solved"] for l in range(2)]
            nparts = [h.level_info(l)["nparts"] for l in range(2)]
            x = torch.zeros_like(prob.b)
            _, it, conv, _ = h.pcg(prob.b, x, rel_tol=1e-8, max_iter=100)
            h.close()
            return dig, solved, nparts, it, bool(conv)
        finally:
            capi.reset_options()

    on, off = build(1, None), build(0, None)
    print("constant coefficient: eigenproblems solved", on[1], "of", on[2], "(without classes:", off[1], ")")
    assert on[0] == off[0] and on[3] == off[3] and on[4] and off[4]
    assert off[1] == off[2]                                   # every agglomerate on its own
    assert on[1][0] * 4 < on[2][0]                            # (512 fine agglomerates: a few dozen classes)
    assert on[1][1] <= on[2][1]
    gen_on, gen_off = build(1, "skew"), build(0, "skew")
    print("general coefficient: eigenproblems solved", gen_on[1], "of", gen_on[2])
    assert gen_on[0] == gen_off[0] and gen_on[1] == gen_on[2]
    q2_on, q2_off = build(1, "q2"), build(0, "q2")
    print("Q2 elasticity: eigenproblems solved", q2_on[1], "of", q2_on[2])
    assert q2_on[0] == q2_off[0] and q2_on[3] == q2_off[3] and q2_on[4]
    assert q2_on[1][0] <= 4 and q2_on[2][0] == 64            # (clamped on one face: two classes)
