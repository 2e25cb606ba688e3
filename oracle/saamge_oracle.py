"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A numpy/scipy restatement of the SAAMGE setup+solve hot path.
It exists to *check* the HIP implementation; nothing under ``saamge_amd/`` may import
it.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it.

Pinning: the reference cannot be compiled here (every translation unit needs MFEM +
hypre, neither is installed), so this oracle is pinned by the reference's own
known-answer tests: the ctest PCG iteration counts for ``mltest`` (3), ``mltest2``
(4) and ``threelevel`` (3) on ``amg/test/mltest.mesh`` with the hard-coded partitions
(amg/CMakeLists.txt:191-217, amg/test/mltest/mltest.cpp:224-228) -- see
tests/test_oracle_kat.py.  Dense eigen/SVD arithmetic uses the *same LAPACK routines*
as the reference (dsygvx, dgesvd) through scipy.  Deviation: the coarsest solve is an
exact (dense LU) solve where the reference defaults to one BoomerAMG V-cycle
(third-party, unpinned; amg/src/tg.cpp:1005-1011) -- the reference's own
``--coarse-direct`` path (amg/src/tg.cpp:989-997).

Each function cites the reference file:line it follows (paths relative to
/root/reference/amg).  Single MPI rank throughout (Dof == TrueDof).
"""
import math

import numpy as np
import scipy.sparse as sp
from scipy.linalg import lapack

BETWEEN = 0x01   # AGG_BETWEEN_AES_FLAG          inc/aggregates.hpp:102
ESS = 0x02       # AGG_ON_ESS_DOMAIN_BORDER_FLAG  inc/aggregates.hpp:103
SVD_EPS = 1e-10  # ContribTent::svd_eps           src/contrib.cpp:61
DIFF_EPS = 1e-10  # GLOBAL.diff_eps               inc/config.hpp:68


# ---------------------------------------------------------------------------
# mfem::Table algebra (third-party semantics restated)
# ---------------------------------------------------------------------------
class Table(object):
    """CSR table of ints (mfem::Table)."""

    def __init__(self, I, J, ncols):
        self.I = np.asarray(I, dtype=np.int64)
        self.J = np.asarray(J, dtype=np.int64)
        self.ncols = int(ncols)

    @property
    def nrows(self):
        return self.I.size - 1

    def row(self, i):
        return self.J[self.I[i]:self.I[i + 1]]

    def row_size(self, i):
        return int(self.I[i + 1] - self.I[i])

    @staticmethod
    def from_rows(rows, ncols):
        I = np.zeros(len(rows) + 1, dtype=np.int64)
        for i, r in enumerate(rows):
            I[i + 1] = I[i] + len(r)
        J = np.concatenate([np.asarray(r, dtype=np.int64) for r in rows]) if rows else np.zeros(0, np.int64)
        return Table(I, J, ncols)

    @staticmethod
    def from_fixed(arr2d, ncols):
        arr2d = np.asarray(arr2d)
        n, k = arr2d.shape
        return Table(np.arange(n + 1) * k, arr2d.ravel(), ncols)


def table_transpose(T):
    """mfem::Transpose(Table): row j of the result lists the i with j in row i,
    in ascending i."""
    rows = np.repeat(np.arange(T.nrows), np.diff(T.I))
    order = np.argsort(T.J, kind="stable")
    counts = np.bincount(T.J, minlength=T.ncols)
    I = np.concatenate([[0], np.cumsum(counts)])
    return Table(I, rows[order], T.nrows)


def table_mult(A, B):
    """mfem::Mult(Table A, Table B): row i = union of B's rows j for j in A's row i,
    in first-encounter order."""
    rows = []
    for i in range(A.nrows):
        js = A.row(i)
        if js.size == 0:
            rows.append(np.zeros(0, np.int64))
            continue
        cat = np.concatenate([B.row(j) for j in js])
        _, first = np.unique(cat, return_index=True)
        rows.append(cat[np.sort(first)])
    return Table.from_rows(rows, B.ncols)


# ---------------------------------------------------------------------------
# a1/a2: partitioning relations (topology)
# ---------------------------------------------------------------------------
class Relations(object):
    """agg_partitioning_relations_t (inc/aggregates.hpp:120-179), single rank."""
    pass


def construct_mises(dof_to_AE):
    """agg_construct_mises_local (src/aggregates.cpp:501-653): a MIS is the set of
    dofs contained in exactly the same set of AEs.  MIS ids are assigned in order of
    first appearance scanning dofs upward (:541-606); dofs inside a MIS are sorted by
    true-dof id (:602) == ascending dof on one rank.  The reference's O(#MIS x ND)
    counting loop is replaced by a signature dictionary with identical output."""
    ND = dof_to_AE.nrows
    mises = np.empty(ND, dtype=np.int64)
    sig_to_mis = {}
    rows = []
    for i in range(ND):
        sig = tuple(sorted(dof_to_AE.row(i).tolist()))
        m = sig_to_mis.get(sig)
        if m is None:
            m = len(rows)
            sig_to_mis[sig] = m
            rows.append([])
        rows[m].append(i)
        mises[i] = m
    return mises, Table.from_rows(rows, ND)


def construct_aggregate_mises(A, dof_to_AE, nparts):
    """agg_construct_aggregate_mises (src/aggregates.cpp:324-487) + Arbitrator::suggest
    (src/arbitrator.cpp:93-204), one rank: on the LAST coarsening the "MISes" are aggregates,
    one per AE.  Dofs of a single AE go to it; the others are distributed greedily in ascending
    dof order: to the aggregate of the already distributed neighbour with the strongest
    connection |a_ij| / sqrt(a_ii a_jj) among the aggregates whose AE contains the dof (first
    maximum in stored -- here ascending column -- order), else to the smallest of its AEs' aggregates
    (first minimum in dof_to_AE order).  Returns mises[dof], mis_to_dof and mises_size."""
    A = sp.csr_matrix(A)
    A.sort_indices()
    ND = dof_to_AE.nrows
    indptr, indices, data = A.indptr, A.indices, A.data
    diag = A.diagonal()
    mises = np.full(ND, -2, dtype=np.int64)
    size = np.zeros(nparts, dtype=np.int64)
    for i in range(ND):
        row = dof_to_AE.row(i)
        if row.size == 1:
            mises[i] = row[0]
            size[row[0]] += 1
    for i in range(ND):
        if mises[i] != -2:
            continue
        parts = dof_to_AE.row(i)
        agg, max_stren = -1, -1.0
        if indptr[i + 1] - indptr[i] > 1:
            for k in range(indptr[i], indptr[i + 1]):
                nb = int(indices[k])
                if nb != i and mises[nb] >= 0 and mises[nb] in parts:
                    strength = abs(data[k]) / np.sqrt(diag[i] * diag[nb])
                    if strength > max_stren:
                        max_stren, agg = strength, int(mises[nb])
        if max_stren < 0.0:
            agg = int(parts[0])
            for p in parts[1:]:
                if size[agg] > size[p]:
                    agg = int(p)
        mises[i] = agg
        size[agg] += 1
    rows = [[] for _ in range(nparts)]
    for i in range(ND):
        rows[mises[i]].append(i)
    return mises, Table.from_rows(rows, ND), size


def build_relations(elem_to_dof, partitioning, nparts, ND, bdr=None, aggregates_A=None):
    """agg_create_partitioning_tables (src/aggregates.cpp:1357-1443) and
    agg_produce_mises / agg_construct_mises_parallel (:712-853) on one rank.  `aggregates_A`
    (the level matrix): `do_aggregates`, aggregates with arbitration instead of MISes."""
    r = Relations()
    r.ND = ND
    r.nparts = int(nparts)
    r.elem_to_dof = elem_to_dof
    r.partitioning = np.asarray(partitioning, dtype=np.int64)
    r.dof_to_elem = table_transpose(elem_to_dof)
    NE = elem_to_dof.nrows
    elem_to_AE = Table(np.arange(NE + 1), r.partitioning, nparts)
    r.elem_to_AE = elem_to_AE
    r.AE_to_elem = table_transpose(elem_to_AE)                  # :1381
    r.AE_to_dof = table_mult(r.AE_to_elem, elem_to_dof)         # :1383
    r.dof_to_AE = table_transpose(r.AE_to_dof)                  # :1385
    # dof_id_inAE / agg_map_id_glob_to_AE (:1202-1244): local index of dof in AE
    r.loc_in_AE = [dict((int(d), j) for j, d in enumerate(r.AE_to_dof.row(p)))
                   for p in range(nparts)]
    if aggregates_A is not None:
        r.mises, r.mis_to_dof, r.mises_size = construct_aggregate_mises(aggregates_A, r.dof_to_AE, r.nparts)
        r.num_mises = r.nparts
        r.mis_to_AE = Table(np.arange(r.nparts + 1), np.arange(r.nparts), r.nparts)   # :770-771
    else:
        r.mises, r.mis_to_dof = construct_mises(r.dof_to_AE)
        r.num_mises = r.mis_to_dof.nrows
        r.mises_size = np.diff(r.mis_to_dof.I)
        r.mis_to_AE = table_mult(r.mis_to_dof, r.dof_to_AE)     # :776
    r.AE_to_mis = table_transpose(r.mis_to_AE)                  # :777
    # agg_construct_agg_flags (:198-216)
    flags = np.zeros(ND, dtype=np.int64) if bdr is None else np.asarray(bdr, dtype=np.int64).copy()
    multi = np.diff(r.dof_to_AE.I) > 1
    flags[multi] |= BETWEEN
    r.agg_flags = flags
    return r


# ---------------------------------------------------------------------------
# a3/a4: AE stiffness matrices
# ---------------------------------------------------------------------------
def build_AE_stiffm_with_global(A, part, rel, elmats):
    """agg_build_AE_stiffm_with_global (src/aggregates.cpp:855-945) with
    bdr_cond_imposed = assemble_ess_diag = true (src/elmat.cpp:50-51) and
    agg_assemble_value (:68-184).  `elmats[e]` is the raw element matrix.
    Returns the dense n x n AE matrix (the reference stores it sparse)."""
    dofs = rel.AE_to_dof.row(part)
    n = dofs.size
    loc = rel.loc_in_AE[part]
    flags = rel.agg_flags
    out = np.zeros((n, n))
    # locally assembled matrix: sum over this AE's elements in ascending element id
    Mloc = np.zeros((n, n))
    for e in rel.AE_to_elem.row(part):
        ed = rel.elem_to_dof.row(e)
        li = np.array([loc[int(d)] for d in ed])
        Mloc[np.ix_(li, li)] += elmats[e]
    indptr, indices, data = A.indptr, A.indices, A.data
    for i in range(n):
        g = int(dofs[i])
        for k in range(indptr[g], indptr[g + 1]):
            c = int(indices[k])
            j = loc.get(c)
            if j is None:
                continue
            both_between = (flags[g] & BETWEEN) and (flags[c] & BETWEEN)
            ess_pair = (flags[g] & ESS) or (flags[c] & ESS)
            if both_between and not (ess_pair and not (c == g)):
                v = Mloc[i, j]          # agg_assemble_value
            else:
                v = data[k]             # copied from the global matrix
            if v != 0.0:
                out[i, j] = v
    return out


def build_AE_stiffm_algebraic(A, part, rel):
    """ExtractSubMatrices (src/tg.cpp:579-672), the element-free mode: the principal submatrix of A
    on the AE's dofs (non-zero entries), then every row with more than one stored entry gets its
    row sum subtracted from the diagonal (constants in the kernel); a non-positive diagonal is
    reset to 1; a single-dof AE is the 1 x 1 identity."""
    dofs = rel.AE_to_dof.row(part)
    n = dofs.size
    if n == 1:
        return np.ones((1, 1))
    loc = rel.loc_in_AE[part]
    out = np.zeros((n, n))
    stored = np.zeros(n, dtype=np.int64)
    indptr, indices, data = A.indptr, A.indices, A.data
    for i in range(n):
        g = int(dofs[i])
        for k in range(indptr[g], indptr[g + 1]):
            j = loc.get(int(indices[k]))
            if j is None or data[k] == 0.0:
                continue
            out[i, j] = data[k]
            stored[i] += 1
    for i in range(n):
        if stored[i] > 1:
            out[i, i] += -float(np.sum(out[i, :]))
        if out[i, i] <= 0.0:
            out[i, i] = 1.0
    return out


def build_AE_stiffm_window(A, part, rel):
    """WindowSubMatrices (src/tg.cpp:741-858): A_TT + A_TX E for the AE's dof set T and its
    outside neighbours X, with the extension E[x, i] = a_ix / sum_{k in T} a_xk (every outside
    value is replaced by that weighted average of the inside ones); a single-dof AE is [1]."""
    dofs = rel.AE_to_dof.row(part)
    n = dofs.size
    if n == 1:
        return np.ones((1, 1))
    loc = rel.loc_in_AE[part]
    indptr, indices, data = A.indptr, A.indices, A.data
    denom, xnum = {}, {}
    for i in range(n):
        g = int(dofs[i])
        for k in range(indptr[g], indptr[g + 1]):
            x = int(indices[k])
            if x in denom or x in loc:
                continue
            value = 0.0
            for kk in range(indptr[x], indptr[x + 1]):
                if int(indices[kk]) in loc:
                    value += data[kk]
            assert abs(value) > 0.0
            denom[x] = value
            xnum[x] = len(xnum)
    nx = len(xnum)
    ext = np.zeros((nx, n))
    ATX = np.zeros((n, nx))
    ATT = np.zeros((n, n))
    for i in range(n):
        g = int(dofs[i])
        for k in range(indptr[g], indptr[g + 1]):
            x = int(indices[k])
            if x in denom:
                ATX[i, xnum[x]] += data[k]
                ext[xnum[x], i] += data[k] / denom[x]
            else:
                ATT[i, loc[x]] += data[k]
    return ATT + (ATX @ ext if nx else 0.0)


def coarse_element_matrix(e, rel_f, rel_c, level_f):
    """ElementMatrixParallelCoarse::GetMatrix (src/elmat.cpp:105-195):
    P_loc^T * AEs_stiffm[e] * P_loc with P_loc built from mis_tent_interps."""
    A_e = level_f.AEs_stiffm[e]
    nf = A_e.shape[0]
    mis_in_AE = np.sort(rel_f.AE_to_mis.row(e))
    edofs = rel_c.elem_to_dof.row(e)
    pos = dict((int(d), j) for j, d in enumerate(edofs))
    nc = sum(int(level_f.mis_numcoarsedof[m]) for m in mis_in_AE)
    Ploc = np.zeros((nf, nc))
    loc = rel_f.loc_in_AE[e]
    for m in mis_in_AE:
        k = int(level_f.mis_numcoarsedof[m])
        if k == 0:
            continue
        rows = np.array([loc[int(d)] for d in rel_f.mis_to_dof.row(m)])
        cols = np.array([pos[int(rel_c.mis_coarsedofoffsets[m]) + i] for i in range(k)])
        Ploc[np.ix_(rows, cols)] += level_f.mis_tent_interps[m]
    return Ploc.T @ (A_e @ Ploc)


def build_AE_stiffm(part, rel, elmats):
    """agg_build_AE_stiffm (src/aggregates.cpp:959-1086): plain sum of the AE's
    element matrices, no boundary treatment."""
    dofs = rel.AE_to_dof.row(part)
    n = dofs.size
    loc = rel.loc_in_AE[part]
    out = np.zeros((n, n))
    for e in rel.AE_to_elem.row(part):
        ed = rel.elem_to_dof.row(e)
        li = np.array([loc[int(d)] for d in ed])
        out[np.ix_(li, li)] += elmats[e]
    return out


# ---------------------------------------------------------------------------
# a5/a6: local spectral problems
# ---------------------------------------------------------------------------
def snd_D_from_dense(A):
    """mbox_snd_D_sparse_from_sparse (src/mbox.cpp:913-949):
    D_ii = sum_j |a_ij| sqrt(a_ii / a_jj)."""
    d = np.diag(A)
    assert np.all(d > 0.0)
    return (np.abs(A) * np.sqrt(d[:, None] / d[None, :])).sum(axis=1)


def lower_eigens_dense(A, D, upper):
    """xpacks_calc_lower_eigens_dense (src/xpacks.cpp:222-314): LAPACK dsygvx,
    itype 1, range 'V' on (-1, upper], abstol = 2*dlamch('S'), uplo 'U'; if nothing
    is found take the single smallest (range 'I', il=iu=1)."""
    n = A.shape[0]
    abstol = 2.0 * lapack.dlamch("S")
    B = np.diag(D)
    w, z, m, ifail, info = lapack.dsygvx(A, B, itype=1, jobz="V", range="V", uplo="U",
                                         vl=-1.0, vu=upper, abstol=abstol)
    assert info == 0
    if m <= 0:
        w, z, m, ifail, info = lapack.dsygvx(A, B, itype=1, jobz="V", range="I", uplo="U",
                                             il=1, iu=1, abstol=abstol)
        assert info == 0 and m == 1
    return w[:m].copy(), np.array(z[:, :m], order="F", copy=True)


# Optional process pool for the per-AE work -- fine-level AE assembly and all eigenproblems --
# (bench.py's cpu_baseline times the oracle on the host cores: the reference is MPI-parallel over
# AEs, one rank per core).  PARALLEL_CORES > 1 makes ml_produce_data fork a pool once the
# relations exist, so the workers inherit the matrix and the tables instead of unpickling them.
PARALLEL_CORES = 1
PARALLEL_MAP = None
_SHARED = None


def _stiff_task(p):
    A, rel, elmat = _SHARED
    return build_AE_stiffm_with_global(A, p, rel, elmat)


def _eig_task(args):
    A_i, theta = args
    D_i = snd_D_from_dense(A_i)
    w, z = lower_eigens_dense(A_i, D_i, theta * 1.0)
    return w, z, D_i


# Test hook (tests/test_oracle_mis_rotation.py): callable (AE, w, Z) -> Z' applied to the eigenvectors dsygvx returned for
# an agglomerate.  Inside a group of EQUAL eigenvalues (the six rigid-body modes of an elasticity agglomerate, all at
# zero) dsygvx returns one D-orthonormal basis of the eigenspace among all of them.
EVECTS_HOOK = None


def compute_vectors(rel, AEs_stiffm, theta, testmesh=False):
    """interp_compute_vectors + Eigensolver::SolveDirect
    (src/interp.cpp:387-556, src/spectral.cpp:124-237)."""
    evals, evects, Ds = [], [], []
    mapper = PARALLEL_MAP or map
    results = list(mapper(_eig_task, [(AEs_stiffm[i], theta) for i in range(rel.nparts)]))
    for i in range(rel.nparts):
        w, z, D_i = results[i]
        if EVECTS_HOOK is not None:
            z = EVECTS_HOOK(i, w, z)
        if testmesh and i == 0:
            # extra all-ones vector on AE 0 of rank 0 (src/interp.cpp:510-524)
            z = np.concatenate([z, np.ones((z.shape[0], 1))], axis=1)
        evals.append(w)
        evects.append(z)
        Ds.append(D_i)
    return evals, evects, Ds


# ---------------------------------------------------------------------------
# a7/a8: MIS gather + SVD -> tentative prolongator
# ---------------------------------------------------------------------------
def svd_dense_normalized(M):
    """xpack_svd_dense_arr (src/xpacks.cpp:494-589): normalise columns to unit
    2-norm, drop columns with norm <= diff_eps, dgesvd('S','N')."""
    cols = []
    for j in range(M.shape[1]):
        nrm = math.sqrt(float(np.dot(M[:, j], M[:, j])))
        if nrm <= 0.0 + DIFF_EPS:
            continue
        cols.append(M[:, j] / nrm)
    if not cols:
        return np.zeros((M.shape[0], 0)), np.zeros(0)
    a = np.asfortranarray(np.stack(cols, axis=1))
    u, s, vt, info = lapack.dgesvd(a, compute_uv=1, full_matrices=0)
    assert info == 0
    return u, s


# Test hook (tests/test_oracle_mis_rotation.py): callable (mis, U, s) -> U' applied to the kept left singular vectors of a
# MIS before they are inserted.  dgesvd's basis inside a group of EQUAL singular values is one orthonormal basis of
# that group's space among all of them; the hook lets a test pick another one and watch what depends on the choice.
MIS_BASIS_HOOK = None


def contrib_mises(rel, evects, avoid_ess=True, extra=None):
    """ContribTent::contrib_mises -> CommunicateEigenvectors + SVDInsert
    (src/contrib.cpp:492-687) with contrib_filter_boundary (:102-163),
    xpack_orth_set (src/xpacks.cpp:591-620) and contrib_tent_insert_simple
    (src/contrib.cpp:170-194).  Returns P_tent (CSR), mis_tent_interps,
    mis_numcoarsedof, per-MIS singular values."""
    ND = rel.ND
    rows_i, cols_i, vals_i = [], [], []
    mis_tent = [None] * rel.num_mises
    mis_nc = np.zeros(rel.num_mises, dtype=np.int64)
    mis_svals = [None] * rel.num_mises
    filled = 0
    for mis in range(rel.num_mises):
        mdofs = rel.mis_to_dof.row(mis)
        dim = mdofs.size
        on_ess = (rel.agg_flags[mdofs] & ESS) != 0
        if avoid_ess and np.all(on_ess):                       # :578-605
            mis_tent[mis] = np.zeros((dim, 0))
            continue
        if dim == 1:                                           # :607-612
            U = np.ones((1, 1))
            mis_svals[mis] = np.ones(1)
        else:
            blocks = []
            for AE in rel.mis_to_AE.row(mis):                  # :525-542
                loc = rel.loc_in_AE[int(AE)]
                li = np.array([loc[int(d)] for d in mdofs])
                blocks.append(evects[int(AE)][li, :])          # agg_restrict_to_agg_enforce
            M = np.concatenate(blocks, axis=1)
            if extra is not None:
                # ContribTent::ExtendWithPolynomials / ExtendWithRBMs (src/contrib.cpp:302-436):
                # extra per-dof modes (constants, coordinates, rigid-body modes) restricted to
                # the MIS are appended after the spectral columns, before filter and SVD
                M = np.concatenate([M, extra[mdofs, :]], axis=1)
            # contrib_filter_boundary: zero essential rows, drop all-zero columns
            M = M.copy()
            if avoid_ess:
                M[on_ess, :] = 0.0
            keep = np.any(M != 0.0, axis=0)
            M = M[:, keep]
            if M.shape[1] == 0:
                mis_tent[mis] = np.zeros((dim, 0))
                continue
            u, s = svd_dense_normalized(M)
            if s.size == 0:
                mis_tent[mis] = np.zeros((dim, 0))
                continue
            eps = SVD_EPS * s[0]
            k = 0
            while k < s.size and s[k] > eps:                   # xpack_orth_set
                k += 1
            assert k > 0
            U = u[:, :k]
            if MIS_BASIS_HOOK is not None:
                U = MIS_BASIS_HOOK(mis, U, s)
            mis_svals[mis] = s
        mis_tent[mis] = np.array(U, copy=True)
        k = U.shape[1]
        for c in range(k):                                     # contrib_tent_insert_simple
            for j in range(dim):
                if abs(U[j, c]) > 0.0:
                    rows_i.append(int(mdofs[j]))
                    cols_i.append(filled + c)
                    vals_i.append(float(U[j, c]))
        mis_nc[mis] = k
        filled += k
    P = sp.csr_matrix((vals_i, (rows_i, cols_i)), shape=(ND, filled))
    P.sort_indices()
    return P, mis_tent, mis_nc, mis_svals


# ---------------------------------------------------------------------------
# a9-a12: prolongator smoothing, smoother data, polynomial smoother
# ---------------------------------------------------------------------------
def build_Dinv_neg(A):
    """mbox_build_Dinv_neg_parallel_matrix (src/mbox.cpp:1839-1861):
    d_i = sqrt(|a_ii|) * sum_j |a_ij| / sqrt(|a_jj|);  returns -1/d."""
    Aabs = abs(A).tocsr()
    diag = Aabs.diagonal()
    d1 = 1.0 / np.sqrt(diag)
    y = Aabs @ d1
    return -1.0 / (np.sqrt(diag) * y)


def sa_poly_roots(nu):
    """smpr_sa_poly_roots (src/smpr.cpp:266-280)."""
    denom = float(2 * nu + 1)
    return np.array([math.sin(i * math.pi / denom) ** 2 for i in range(1, nu + 1)])


def sas_poly_roots(nu):
    """smpr_sas_poly_roots (src/smpr.cpp:282-306): cos^2(i pi/(2nu+1)), i=0..2nu,
    then sin^2(j pi/(2nu+1)), j=1..nu."""
    denom = float(2 * nu + 1)
    r = [math.cos(i * math.pi / denom) ** 2 for i in range(0, 2 * nu + 1)]
    r += [math.sin(i * math.pi / denom) ** 2 for i in range(1, nu + 1)]
    return np.array(r)


def compute_poly(A, b, x, roots, Dinv_neg):
    """smpr_compute_poly (inc/smpr.hpp:320-339): x += (1/tau) * Dinv_neg * (A x - b)."""
    for tau in roots:
        tmp = A @ x - b
        tmp *= Dinv_neg
        x += (1.0 / tau) * tmp
    return x


def interp_smooth(A, tent, Dinv_neg, nu_pro, drop_tol=0.0):
    """interp_smooth (src/interp.cpp:172-229): P = prod_k (I + (1/tau_k) Dinv_neg A) tent, then
    AltThreshold (src/interp.cpp:89-170): only entries with fabs(v) > drop_tol are kept."""
    P = tent.copy()
    S = sp.diags(Dinv_neg) @ A
    for tau in sa_poly_roots(nu_pro):
        P = P + (1.0 / tau) * (S @ P)
    P = P.tocsr()
    if drop_tol != 0.0:
        P.data[np.abs(P.data) <= drop_tol] = 0.0
        P.eliminate_zeros()
    return P


# ---------------------------------------------------------------------------
# hierarchy
# ---------------------------------------------------------------------------
class Level(object):
    """One tg_data_t + interp_data_t + relations (inc/tg_data.hpp:47-83)."""
    pass


def build_level(A, rel, AEs_stiffm, theta, nu_relax, nu_pro=0, testmesh=False, extra=None,
                drop_tol=0.0):
    """tg_init_data + tg_build_hierarchy + tg_update_coarse_operator
    (src/tg.cpp:402-430, :502-540, :979-1014)."""
    lv = Level()
    lv.A = A.tocsr()
    lv.rel = rel
    lv.AEs_stiffm = AEs_stiffm
    lv.Dinv_neg = build_Dinv_neg(lv.A)
    lv.roots = sas_poly_roots(nu_relax)
    lv.evals, lv.evects, lv.Ds = compute_vectors(rel, AEs_stiffm, theta, testmesh)
    lv.tent, lv.mis_tent_interps, lv.mis_numcoarsedof, lv.mis_svals = contrib_mises(rel, lv.evects, extra=extra)
    lv.P = interp_smooth(lv.A, lv.tent, lv.Dinv_neg, nu_pro, drop_tol) if nu_pro > 0 else lv.tent.copy()
    lv.R = lv.P.T.tocsr()
    lv.Ac = (lv.R @ lv.A @ lv.P).tocsr()       # tg_coarse_matr == RAP, inc/tg.hpp:696-709
    return lv


def coarse_relations(rel_f, level_f, partitioning, nparts, aggregates_A=None):
    """agg_create_partitioning_coarse + agg_build_coarse_Dof_TrueDof +
    agg_create_rels_except_elem_coarse (src/aggregates.cpp:1610-1832,:1481-1602):
    coarse elements = fine AEs; coarse dofs numbered MIS by MIS; elem_to_dof =
    AE_to_dof_fine x pattern(P_tent); no essential flags on coarse levels."""
    nc = level_f.tent.shape[1]
    offs = np.concatenate([[0], np.cumsum(level_f.mis_numcoarsedof)])
    T = level_f.tent.tocsr()
    finedof_to_dof = Table(T.indptr, T.indices, nc)
    elem_to_dof = table_mult(rel_f.AE_to_dof, finedof_to_dof)
    rel_c = build_relations(elem_to_dof, partitioning, nparts, nc, bdr=None, aggregates_A=aggregates_A)
    rel_c.mis_coarsedofoffsets = offs
    return rel_c


class Hierarchy(object):
    pass


def scaling_P_from_level(lv):
    """ContribTent::SVDInsert with scaling_P (src/contrib.cpp:655-668) + interp_scaling_P_assemble
    (src/interp.cpp:842-909): one column per MIS with coarse dofs; its entries are the normalised
    least-squares coefficients of the constant vector in that MIS's basis (xpack_solve_lls)."""
    rows, cols, vals = [], [], []
    row = col = 0
    for mis, k in enumerate(lv.mis_numcoarsedof):
        if k == 0:
            continue
        U = lv.mis_tent_interps[mis]
        x, *_ = np.linalg.lstsq(U, np.ones(U.shape[0]), rcond=None)
        x = x / math.sqrt(float(np.dot(x, x)))
        for v in range(k):
            rows.append(row + v)
            cols.append(col)
            vals.append(x[v])
        row += k
        col += 1
    return sp.csr_matrix((vals, (rows, cols)), shape=(row, col))


def nullspace_level(lv_last):
    """CorrectNullspace (src/solve.cpp:52-164) as configured by ml_produce_hierarchy_from_level
    (src/ml.cpp:225-236: smoother_steps = 3, smooth_phat = false): one more two-grid level under
    the coarsest spectral operator, interp = scaling_P, SAS polynomial smoother of nu = 3."""
    lv = Level()
    lv.A = lv_last.Ac.tocsr()
    lv.P = scaling_P_from_level(lv_last)
    lv.R = lv.P.T.tocsr()
    lv.Ac = (lv.R @ lv.A @ lv.P).tocsr()
    lv.Dinv_neg = build_Dinv_neg(lv.A)
    lv.roots = sas_poly_roots(3)
    return lv


def ml_produce_data(A, elem_to_dof, elmat, bdr, partitions, theta=0.003, nu_relax=3,
                    nu_pro=0, testmesh=False, correct_nullspace=False, extra_modes=None,
                    algebraic=False, smooth_drop_tol=0.0, do_aggregates=False):
    """ml_produce_data + ml_produce_hierarchy_from_level (src/ml.cpp:379-472,:111-236).
    `partitions[k]` maps level-k elements to level-k AEs.  Exact coarsest solve.
    `do_aggregates`: aggregates instead of MISes on the LAST coarsening (src/ml.cpp:149)."""
    A = sp.csr_matrix(A)
    ND = A.shape[0]
    if algebraic:
        # tg_produce_data_algebraic (src/tg.cpp:862-886) on the tables of TestWindowSubMatrices /
        # fem_create_partitioning_from_matrix: elements = dofs, partitions[0] maps dofs to AEs
        elem_to_dof = np.arange(ND, dtype=np.int64).reshape(ND, 1)
        bdr = None
    e2d = Table.from_fixed(elem_to_dof, ND)
    H = Hierarchy()
    H.levels = []
    nparts0 = int(np.max(partitions[0])) + 1
    last = len(partitions) - 1
    rel = build_relations(e2d, partitions[0], nparts0, ND, bdr=bdr,
                          aggregates_A=A if (do_aggregates and last == 0) else None)
    global PARALLEL_MAP, _SHARED
    pool = None
    if PARALLEL_CORES > 1 and not algebraic:
        import multiprocessing
        _SHARED = (A, rel, elmat)
        pool = multiprocessing.get_context("fork").Pool(PARALLEL_CORES)
        PARALLEL_MAP = lambda f, it: pool.map(f, it, chunksize=2)
        stiff = pool.map(_stiff_task, range(rel.nparts), chunksize=2)
    elif algebraic:
        build = build_AE_stiffm_window if algebraic == "window" else build_AE_stiffm_algebraic
        stiff = [build(A, p, rel) for p in range(rel.nparts)]
    else:
        stiff = [build_AE_stiffm_with_global(A, p, rel, elmat) for p in range(rel.nparts)]
    lv = build_level(A, rel, stiff, theta, nu_relax, nu_pro, testmesh,
                     extra=None if extra_modes is None else np.asarray(extra_modes, dtype=float).reshape(ND, -1),
                     drop_tol=smooth_drop_tol)
    H.levels.append(lv)
    for k in range(1, len(partitions)):
        prev = H.levels[-1]
        nparts = int(np.max(partitions[k])) + 1
        rel_c = coarse_relations(prev.rel, prev, partitions[k], nparts,
                                 aggregates_A=prev.Ac if (do_aggregates and k == last) else None)
        cel = [coarse_element_matrix(e, prev.rel, rel_c, prev) for e in range(prev.rel.nparts)]
        stiff = [build_AE_stiffm(p, rel_c, cel) for p in range(rel_c.nparts)]
        lv = build_level(prev.Ac, rel_c, stiff, theta, nu_relax, nu_pro, False, drop_tol=smooth_drop_tol)
        lv.coarse_elmats = cel
        H.levels.append(lv)
    if pool is not None:
        pool.close()
        pool.join()
        PARALLEL_MAP = None
        _SHARED = None
    if correct_nullspace:
        # (the reference solves the null-space level with one BoomerAMG V-cycle -- third party;
        # here, as for the plain coarsest level, exactly)
        H.levels.append(nullspace_level(H.levels[-1]))
    H.coarse_dense = H.levels[-1].Ac.toarray()
    H.coarse_lu = None
    return H


def coarse_solve(H, rc):
    import scipy.linalg as sla
    if H.coarse_lu is None:
        H.coarse_lu = sla.lu_factor(H.coarse_dense)
    return sla.lu_solve(H.coarse_lu, rc)


def vcycle(H, b, level=0, smoothers=None):
    """VCycleSolver::Mult (x = 0 start, src/solve.cpp:309-323) -> tg_cycle_atb
    (src/tg.cpp:91-132), recursing through ml_impose_cycle (src/ml.cpp:361-377).
    `smoothers`: {level: (pre, post)} replacing the polynomial smoother in either place (None keeps it) -- the
    reference's smpr_ft plug, tg_data_t::pre_smoother / post_smoother (src/tg.cpp:113,131,411-414); each is called
    as fn(A, b, x) and updates x in place, x += M^-1 (b - A x)."""
    lv = H.levels[level]
    pre, post = (smoothers or {}).get(level, (None, None))
    x = np.zeros_like(b)
    if pre is not None:
        pre(lv.A, b, x)
    else:
        compute_poly(lv.A, b, x, lv.roots, lv.Dinv_neg)      # pre_smoother
    res = b - lv.A @ x
    rc = lv.R @ res
    if level + 1 < len(H.levels):
        xc = vcycle(H, rc, level + 1, smoothers)
    else:
        xc = coarse_solve(H, rc)
    x += lv.P @ xc
    if post is not None:
        post(lv.A, b, x)
    else:
        compute_poly(lv.A, b, x, lv.roots, lv.Dinv_neg)      # post_smoother
    return x


def vcycle_iterative(H, b, x0, coarse=None):
    """VCycleSolver::Mult with iterative_mode = true (src/solve.cpp:309-323): tg_cycle_atb (src/tg.cpp:91-132)
    started from the caller's x on the finest level; coarser levels start from zero as always.
    `coarse`: replacement for the coarsest solve (tg_data_t::coarse_solver plug)."""
    lv = H.levels[0]
    x = x0.copy()
    compute_poly(lv.A, b, x, lv.roots, lv.Dinv_neg)
    rc = lv.R @ (b - lv.A @ x)
    if len(H.levels) > 1:
        xc = vcycle(H, rc, 1)
    else:
        xc = coarse(rc) if coarse is not None else coarse_solve(H, rc)
    x += lv.P @ xc
    compute_poly(lv.A, b, x, lv.roots, lv.Dinv_neg)
    return x


def pcg(A, prec, b, x0=None, rel_tol=1e-6, abs_tol=0.0, max_iter=1000, squared_tol=True):
    """MFEM CGSolver::Mult as driven by amg/test/mltest/mltest.cpp:773-781
    (``SetRelTol(1e-6)``, "MFEM squares this") -- the same loop as kalchev_pcg
    (src/mfem_addons.cpp:106-248), which compares (B r, r) against
    max(rtol * (B r0, r0), atol) un-squared (squared_tol=False).
    Returns (x, iterations, converged, [ (B r_k, r_k) ])."""
    x = np.zeros_like(b) if x0 is None else x0.copy()
    r = b - A @ x
    z = prec(r)
    d = z.copy()
    nom0 = nom = float(d @ r)
    hist = [nom]
    if squared_tol:
        r0 = max(nom * rel_tol * rel_tol, abs_tol * abs_tol)
    else:
        r0 = max(nom * rel_tol, abs_tol)
    if nom <= r0:
        return x, 0, True, hist
    z = A @ d
    den = float(z @ d)
    if den == 0.0:
        return x, 0, False, hist
    i = 1
    converged = False
    final_iter = max_iter
    while True:
        alpha = nom / den
        x += alpha * d
        r -= alpha * z
        z = prec(r)
        betanom = float(r @ z)
        hist.append(betanom)
        if betanom < r0:
            converged = True
            final_iter = i
            break
        i += 1
        if i > max_iter:
            break
        beta = betanom / nom
        d = z + beta * d
        z = A @ d
        den = float(d @ z)
        nom = betanom
    return x, final_iter, converged, hist


def solve(H, b, **kw):
    A = H.levels[0].A
    return pcg(A, lambda r: vcycle(H, r), b, **kw)
