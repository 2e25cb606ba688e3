"""Synthetic problem generators: the *inputs* of the hot path.

These stand in for the MFEM-side problem setup of the reference (which is out of
scope, SURVEY.md section 2 row "fem"): they produce exactly the arrays the reference
hands to ``agg_create_partitioning_fine`` / ``ml_produce_data``
(reference: amg/src/fem.cpp:687-717, amg/inc/fem.hpp:427-448,
amg/test/mltest/mltest.cpp:481-621):

* ``A``            global CSR stiffness matrix, essential rows/cols eliminated with
                   the diagonal kept, explicit zeros left in place (``Finalize(0)``)
* ``elem_to_dof``  NE x nde int32 (fixed number of dofs per element)
* ``elmat``        NE x nde x nde fp64 raw (un-eliminated) element matrices
* ``bdr``          ND int8 flags, bit 0x02 = on essential boundary
                   (reference: amg/inc/aggregates.hpp:102-105)
* ``b``            right-hand side with essential entries zeroed
* ``partitions``   list of element->AE maps, one per coarsening

Everything is numpy on the host; nothing here is timed by bench.py.
"""
import numpy as np
import scipy.sparse as sp

AGG_BETWEEN_AES_FLAG = 0x01
AGG_ON_ESS_DOMAIN_BORDER_FLAG = 0x02
AGG_ON_PROC_IFACE_FLAG = 0x04
AGG_OWNED_FLAG = 0x08


class Problem(object):
    """Plain container of the hot-path inputs."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    @property
    def ND(self):
        return self.A.shape[0]

    @property
    def NE(self):
        return self.elem_to_dof.shape[0]


# --------------------------------------------------------------------------
# 1-D building blocks
# --------------------------------------------------------------------------
def _lagrange_1d(order):
    """1-D mass and stiffness matrices on [0,1] for equispaced Lagrange nodes."""
    nodes = np.linspace(0.0, 1.0, order + 1)
    # Gauss-Legendre with enough points to be exact for degree 2*order
    xg, wg = np.polynomial.legendre.leggauss(order + 2)
    xg = 0.5 * (xg + 1.0)
    wg = 0.5 * wg
    n = order + 1
    phi = np.ones((n, xg.size))
    dphi = np.zeros((n, xg.size))
    for i in range(n):
        for j in range(n):
            if j != i:
                phi[i] *= (xg - nodes[j]) / (nodes[i] - nodes[j])
        for k in range(n):
            if k == i:
                continue
            t = np.ones_like(xg) / (nodes[i] - nodes[k])
            for j in range(n):
                if j != i and j != k:
                    t *= (xg - nodes[j]) / (nodes[i] - nodes[j])
            dphi[i] += t
    M = (phi * wg) @ phi.T
    K = (dphi * wg) @ dphi.T
    L = phi @ wg
    return M, K, L


def _assemble(ND, elem_to_dof, elmat):
    """Sum element matrices in ascending element order (MFEM Assemble order)."""
    NE, nde = elem_to_dof.shape
    rows = np.repeat(elem_to_dof, nde, axis=1).ravel()
    cols = np.tile(elem_to_dof, (1, nde)).ravel()
    A = sp.coo_matrix((elmat.ravel(), (rows, cols)), shape=(ND, ND)).tocsr()
    A.sort_indices()
    return A


def _eliminate(A, b, ess):
    """EliminateEssentialBCFromDofs(ess, x=0, b, keep_diag=true) then Finalize(0):
    zero rows/cols of essential dofs, keep the diagonal, keep explicit zeros
    (reference: amg/inc/fem.hpp:438-448)."""
    A = A.tocsr().copy()
    ess = np.asarray(ess, dtype=bool)
    row_of = np.repeat(np.arange(A.shape[0]), np.diff(A.indptr))
    kill = (ess[row_of] | ess[A.indices]) & (row_of != A.indices)
    A.data[kill] = 0.0
    b = b.copy()
    b[ess] = 0.0
    return A, b


# --------------------------------------------------------------------------
# mltest.mesh fixture (reference: amg/test/mltest.mesh, mltest.cpp:196-295)
# --------------------------------------------------------------------------
MLTEST_PARTITION = np.array([0, 0, 1, 1, 0, 0, 2, 2, 3, 3, 3, 2], dtype=np.int32)
MLTEST_COARSE_PARTITION = np.array([0, 0, 1, 1], dtype=np.int32)
MLTEST_VERTEX_Y = (0.0, 0.333333333, 0.666666667, 1.0)        # amg/test/mltest.mesh, vertex rows
# 2-rank fixture `pmltest`: element -> rank and per-rank AE maps (mltest.cpp:230-241,279-286)
MLTEST_RANK_OF_ELEM = np.array([0] * 6 + [1] * 6, dtype=np.int32)
MLTEST_PARTITION_2RANKS = [np.array([0, 0, 1, 1, 0, 0], dtype=np.int32),
                           np.array([0, 0, 1, 1, 1, 0], dtype=np.int32)]


def checkerboard_coef(x, y, z=None):
    """Reference: amg/test/mltest/mltest.cpp:156-175."""
    d = 10.0
    cx = np.ceil(x * d).astype(np.int64) & 1
    cy = np.ceil(y * d).astype(np.int64) & 1
    if z is None:
        hi = cx == cy
    else:
        cz = np.ceil(z * d).astype(np.int64) & 1
        hi = np.where(cz == 1, cx == cy, cx != cy)
    return np.where(hi, 1e6, 1e0)


def quad_mesh_problem(nx, ny, lx=1.0, ly=1.0, order=1, coef="checkerboard",
                      ess_sides=("left",), partition=None, vertex_y=None):
    """2-D tensor grid of rectangles with Q1/Q2 Lagrange elements.

    Vertex numbering is row-major (x fastest), which is exactly mltest.mesh's
    vertex numbering for nx=4, ny=3.  Element vertex order is MFEM's
    counter-clockwise (v00, v10, v11, v01).  For order 2 the extra dofs are
    numbered after the vertices: x-edges, y-edges, then element interiors
    (MFEM's own edge numbering differs; the solver is invariant to it).
    """
    hx, hy = lx / nx, ly / ny
    NE = nx * ny
    ex, ey = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    ex = ex.ravel()
    ey = ey.ravel()
    # `vertex_y`: the y coordinates of the vertex rows as a mesh FILE stores them (mltest.mesh keeps
    # 0.333333333 / 0.666666667): the element heights are then what MFEM computes from the file
    yv = np.arange(ny + 1) * hy if vertex_y is None else np.asarray(vertex_y, dtype=float)
    assert yv.size == ny + 1
    hy_e = (yv[1:] - yv[:-1])[ey]                # height of every element
    nvx, nvy = nx + 1, ny + 1
    vid = lambda i, j: j * nvx + i
    M1, K1, L1 = _lagrange_1d(order)
    if order == 1:
        ND = nvx * nvy
        elem_to_dof = np.stack([vid(ex, ey), vid(ex + 1, ey),
                                vid(ex + 1, ey + 1), vid(ex, ey + 1)], axis=1)
        # tensor index (ix, iy) of each local dof in MFEM order
        loc = [(0, 0), (1, 0), (1, 1), (0, 1)]
        xs = np.tile(np.arange(nvx) * hx, nvy)
        ys = np.repeat(yv, nvx)
    elif order == 2:
        nV = nvx * nvy
        nEx = nx * nvy          # horizontal edges
        nEy = nvx * ny          # vertical edges
        ND = nV + nEx + nEy + NE
        xedge = lambda i, j: nV + j * nx + i
        yedge = lambda i, j: nV + nEx + j * nvx + i
        inter = nV + nEx + nEy + (ey * nx + ex)
        elem_to_dof = np.stack([vid(ex, ey), vid(ex + 1, ey), vid(ex + 1, ey + 1),
                                vid(ex, ey + 1), xedge(ex, ey), yedge(ex + 1, ey),
                                xedge(ex, ey + 1), yedge(ex, ey), inter], axis=1)
        loc = [(0, 0), (2, 0), (2, 2), (0, 2), (1, 0), (2, 1), (1, 2), (0, 1), (1, 1)]
        xs = np.zeros(ND)
        ys = np.zeros(ND)
        for k, (ix, iy) in enumerate(loc):
            xs[elem_to_dof[:, k]] = (ex + 0.5 * ix) * hx
            ys[elem_to_dof[:, k]] = yv[ey] + 0.5 * iy * hy_e
    else:
        raise ValueError("order must be 1 or 2")
    nde = len(loc)
    n1 = order + 1
    # reference element matrix in tensor ordering (ix fastest): K = Kx(x)My + Mx(x)Ky
    perm = np.array([iy * n1 + ix for (ix, iy) in loc])
    KxMy, MxKy = np.kron(M1, K1)[np.ix_(perm, perm)], np.kron(K1, M1)[np.ix_(perm, perm)]
    Kel = (hy_e / hx)[:, None, None] * KxMy[None] + (hx / hy_e)[:, None, None] * MxKy[None]
    Lel = (hx * hy_e)[:, None] * np.kron(L1, L1)[perm][None]
    cx = (ex + 0.5) * hx
    cy = yv[ey] + 0.5 * hy_e
    if coef == "checkerboard":
        c = checkerboard_coef(cx, cy)
    else:
        c = np.full(NE, float(coef))
    elmat = c[:, None, None] * Kel
    elem_to_dof = elem_to_dof.astype(np.int32)
    A0 = _assemble(ND, elem_to_dof, elmat)
    b0 = np.zeros(ND)
    np.add.at(b0, elem_to_dof.ravel(), Lel.ravel())
    eps = 1e-12
    ess = np.zeros(ND, dtype=bool)
    for s in ess_sides:
        if s == "left":
            ess |= xs < eps
        elif s == "right":
            ess |= xs > lx - eps
        elif s == "bottom":
            ess |= ys < eps
        elif s == "top":
            ess |= ys > ly - eps
    A, b = _eliminate(A0, b0, ess)
    bdr = np.where(ess, AGG_ON_ESS_DOMAIN_BORDER_FLAG, 0).astype(np.int8) | AGG_OWNED_FLAG
    bdr = bdr.astype(np.int8)
    return Problem(A=A, b=b, elem_to_dof=elem_to_dof, elmat=np.ascontiguousarray(elmat),
                   bdr=bdr, ess=ess, partitions=partition, dims=(nx, ny), order=order,
                   coords=np.stack([xs, ys], axis=1))


def mltest_problem(order=1, levels=2):
    """The reference's ctest fixture `mltest` / `mltest2` / `threelevel`
    (amg/CMakeLists.txt:191-217): mltest.mesh (4x3 rectangles on the unit square with the vertex
    rows at y = 0, 0.333333333, 0.666666667, 1 exactly as the file stores them),
    checkerboard 1e6/1 coefficient sampled at element centres, essential boundary
    = attribute 4 (x = 0), f = 1, hard-coded AE maps."""
    parts = [MLTEST_PARTITION.copy()]
    if levels >= 3:
        parts.append(MLTEST_COARSE_PARTITION.copy())
    return quad_mesh_problem(4, 3, order=order, coef="checkerboard",
                             ess_sides=("left",), partition=parts, vertex_y=MLTEST_VERTEX_Y)


def quad_elasticity_matrix(hx, hy, lam=1.0, mu=1.0):
    """8 x 8 stiffness of isotropic plane elasticity (lambda div u div v + 2 mu eps(u):eps(v),
    MFEM's ElasticityIntegrator form) on an hx x hy bilinear rectangle, 2x2 Gauss (exact),
    vertex order (v00, v10, v11, v01), dofs 2*vertex + component."""
    g = np.array([-1.0, 1.0]) / np.sqrt(3.0)
    sgn = np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]], dtype=float)
    C = np.array([[lam + 2 * mu, lam, 0.0], [lam, lam + 2 * mu, 0.0], [0.0, 0.0, mu]])
    Ke = np.zeros((8, 8))
    detJ = hx * hy / 4.0
    for xi in g:
        for eta in g:
            B = np.zeros((3, 8))
            for a in range(4):
                dx = sgn[a, 0] * (1.0 + sgn[a, 1] * eta) / 4.0 * (2.0 / hx)
                dy = sgn[a, 1] * (1.0 + sgn[a, 0] * xi) / 4.0 * (2.0 / hy)
                B[0, 2 * a] = dx
                B[1, 2 * a + 1] = dy
                B[2, 2 * a], B[2, 2 * a + 1] = dy, dx
            Ke += detJ * (B.T @ C @ B)
    return 0.5 * (Ke + Ke.T)


def mltest_elasticity_problem(levels=2):
    """The reference's ctest `elasticity` (amg/CMakeLists.txt:226-233): mltest.mesh, two
    displacement components per vertex, constant coefficient (lambda = mu = 1,
    `ElasticityIntegrator(q, 1.0, 1.0)`, amg/test/mltest/mltest.cpp:581), clamped on
    attribute 4 (x = 0), ZERO right-hand side (the driver starts PCG from a random vector,
    mltest.cpp:752-758), the fixture's AE maps."""
    nx, ny = 4, 3
    hx, hy = 1.0 / nx, 1.0 / ny
    nvx, nvy = nx + 1, ny + 1
    NV, NE = nvx * nvy, nx * ny
    ND = 2 * NV
    ex, ey = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    ex, ey = ex.ravel(), ey.ravel()
    vid = lambda i, j: j * nvx + i
    e2v = np.stack([vid(ex, ey), vid(ex + 1, ey), vid(ex + 1, ey + 1), vid(ex, ey + 1)], axis=1)
    elem_to_dof = (2 * e2v[:, :, None] + np.arange(2)[None, None, :]).reshape(NE, 8).astype(np.int32)
    yv = np.asarray(MLTEST_VERTEX_Y)             # the file's vertex rows: element heights as MFEM sees them
    Kref = quad_elasticity_matrix(hx, hy)
    elmat = np.ascontiguousarray(np.stack([quad_elasticity_matrix(hx, yv[j + 1] - yv[j]) for j in ey]))
    A0 = _assemble(ND, elem_to_dof, elmat)
    ess = np.repeat((np.arange(NV) % nvx) == 0, 2)
    A, b = _eliminate(A0, np.zeros(ND), ess)
    bdr = (np.where(ess, AGG_ON_ESS_DOMAIN_BORDER_FLAG, 0) | AGG_OWNED_FLAG).astype(np.int8)
    parts = [MLTEST_PARTITION.copy()]
    if levels >= 3:
        parts.append(MLTEST_COARSE_PARTITION.copy())
    return Problem(A=A, b=b, elem_to_dof=elem_to_dof, elmat=elmat, bdr=bdr, ess=ess,
                   partitions=parts, dims=(nx, ny), order=1, Kref=Kref)


# --------------------------------------------------------------------------
# structured 3-D hexahedral Poisson
# --------------------------------------------------------------------------
_HEX_LOC = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0),
            (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]


def hex_element_matrix(h, K=(1.0, 1.0, 1.0)):
    """Closed-form trilinear stiffness on an h[0] x h[1] x h[2] box, MFEM vertex order."""
    M1, K1, _ = _lagrange_1d(1)
    hx, hy, hz = h
    Kt = (K[0] * (hy * hz / hx) * np.kron(M1, np.kron(M1, K1)) +
          K[1] * (hx * hz / hy) * np.kron(M1, np.kron(K1, M1)) +
          K[2] * (hx * hy / hz) * np.kron(K1, np.kron(M1, M1)))
    perm = np.array([(iz * 2 + iy) * 2 + ix for (ix, iy, iz) in _HEX_LOC])
    return Kt[np.ix_(perm, perm)]


def block_partition(n, blk):
    """element (x fastest) -> AE id for axis-aligned blocks of blk elements; AE ids x fastest."""
    nx, ny, nz = n
    bx, by, bz = blk
    nbx, nby = -(-nx // bx), -(-ny // by)
    ez, eyy, exx = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    part = ((ez // bz) * nby + (eyy // by)) * nbx + (exx // bx)
    return part.ravel().astype(np.int32), (nbx, nby, -(-nz // bz))


def poisson3d_problem(n, blk=(8, 8, 4), K=(1.0, 1.0, 1.0), coarse_blk=None,
                      coef=None, with_elmat=True):
    """Unit cube, n^3 (or n = (nx,ny,nz)) Q1 hexes, Dirichlet on all six faces
    (diagonal kept), f = 1.  AE partition = blocks of `blk` elements; optional
    further coarsenings = blocks of `coarse_blk` AEs (list of triples).
    `coef`: None (constant 1) or "checkerboard"."""
    if np.isscalar(n):
        n = (int(n),) * 3
    nx, ny, nz = n
    h = (1.0 / nx, 1.0 / ny, 1.0 / nz)
    nvx, nvy, nvz = nx + 1, ny + 1, nz + 1
    ND = nvx * nvy * nvz
    NE = nx * ny * nz
    ez, ey, ex = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    ex, ey, ez = ex.ravel(), ey.ravel(), ez.ravel()
    vid = lambda i, j, k: (k * nvy + j) * nvx + i
    elem_to_dof = np.stack([vid(ex + a, ey + b_, ez + c) for (a, b_, c) in _HEX_LOC],
                           axis=1).astype(np.int32)
    Kref = hex_element_matrix(h, K)
    if coef == "checkerboard":
        c = checkerboard_coef((ex + 0.5) * h[0], (ey + 0.5) * h[1], (ez + 0.5) * h[2])
    elif coef == "skew":
        # smooth coefficient without any mirror / permutation symmetry: no two agglomerates are
        # congruent and no local eigenspace is degenerate (parity tests at tight tolerances)
        cx, cy, cz = (ex + 0.5) * h[0], (ey + 0.5) * h[1], (ez + 0.5) * h[2]
        c = np.exp(0.7 * cx + 0.4 * cy - 0.3 * cz) * (1.0 + 0.3 * np.sin(5.0 * cx + 3.0 * cy + 7.0 * cz))
    else:
        c = np.ones(NE)
    elmat = c[:, None, None] * Kref[None, :, :]
    A0 = _assemble(ND, elem_to_dof, elmat)
    b0 = np.zeros(ND)
    np.add.at(b0, elem_to_dof.ravel(), h[0] * h[1] * h[2] / 8.0)
    iz, iy, ix = np.meshgrid(np.arange(nvz), np.arange(nvy), np.arange(nvx), indexing="ij")
    ess = ((ix == 0) | (ix == nx) | (iy == 0) | (iy == ny) | (iz == 0) | (iz == nz)).ravel()
    A, b = _eliminate(A0, b0, ess)
    bdr = (np.where(ess, AGG_ON_ESS_DOMAIN_BORDER_FLAG, 0) | AGG_OWNED_FLAG).astype(np.int8)
    part0, nb = block_partition(n, blk)
    parts = [part0]
    for cb in (coarse_blk or []):
        p, nb = block_partition(nb, cb)
        parts.append(p)
    return Problem(A=A, b=b, elem_to_dof=elem_to_dof,
                   elmat=np.ascontiguousarray(elmat) if with_elmat else None,
                   bdr=bdr, ess=ess, partitions=parts, dims=n, order=1, Kref=Kref, coefs=c)


def _lagrange_1d_mixed(order):
    """C[i, j] = int_0^1 phi_i'(x) phi_j(x) dx for the equispaced Lagrange basis of _lagrange_1d."""
    nodes = np.linspace(0.0, 1.0, order + 1)
    xg, wg = np.polynomial.legendre.leggauss(order + 2)
    xg, wg = 0.5 * (xg + 1.0), 0.5 * wg
    n = order + 1
    phi = np.ones((n, xg.size))
    dphi = np.zeros((n, xg.size))
    for i in range(n):
        for j in range(n):
            if j != i:
                phi[i] *= (xg - nodes[j]) / (nodes[i] - nodes[j])
        for k in range(n):
            if k == i:
                continue
            t = np.ones_like(xg) / (nodes[i] - nodes[k])
            for j in range(n):
                if j != i and j != k:
                    t *= (xg - nodes[j]) / (nodes[i] - nodes[j])
            dphi[i] += t
    return (dphi * wg) @ phi.T


def hex_elasticity_matrix_tensor(h, order, lam=1.0, mu=1.0):
    """Stiffness of isotropic linear elasticity (lambda div u div v + 2 mu eps(u):eps(v)) on an
    h[0] x h[1] x h[2] box with tensor-product Lagrange elements of the given order (order 2:
    the 27-node hex, 81 dofs, of BASELINE config 5), exact Gauss integration.  Local numbering:
    node = (iz * n1 + iy) * n1 + ix, dof = 3 * node + component."""
    M1, K1, _ = _lagrange_1d(order)
    C1 = _lagrange_1d_mixed(order)
    n1 = order + 1
    nn = n1 ** 3
    # 1-D factors per direction for "derivative on i / on j / on both / on none"
    f = [{"00": M1 * hd, "11": K1 / hd, "10": C1, "01": C1.T} for hd in h]

    def integral(da, db):
        """int d_a phi_i d_b phi_j over the box, node-by-node (z slowest)."""
        key = []
        for d in range(3):
            key.append(("1" if d == da else "0") + ("1" if d == db else "0"))
        return np.kron(f[2][key[2]], np.kron(f[1][key[1]], f[0][key[0]]))

    G = [[integral(a, b) for b in range(3)] for a in range(3)]
    lap = G[0][0] + G[1][1] + G[2][2]
    Ke = np.zeros((3 * nn, 3 * nn))
    for a in range(3):
        for b in range(3):
            blk = lam * G[a][b] + mu * G[b][a]
            if a == b:
                blk = blk + mu * lap
            Ke[a::3, b::3] = blk
    return 0.5 * (Ke + Ke.T)


def elasticity3d_q2_problem(n, blk=(2, 2, 2), lam=1.0, mu=1.0, order=2):
    """BASELINE config 5 in small: unit cube of n hexes with tensor-product elements of `order`
    (2: 27 nodes, 81 dofs per element), 3 displacement components per node (byVDIM), clamped
    on x = 0, body force (0, 0, -1), AEs = blocks of blk elements."""
    if np.isscalar(n):
        n = (int(n),) * 3
    nx, ny, nz = n
    h = (1.0 / nx, 1.0 / ny, 1.0 / nz)
    n1 = order + 1
    gx, gy, gz = order * nx + 1, order * ny + 1, order * nz + 1
    NV = gx * gy * gz
    ND, NE = 3 * NV, nx * ny * nz
    ez, ey, ex = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    ex, ey, ez = ex.ravel(), ey.ravel(), ez.ravel()
    nid = lambda i, j, k: (k * gy + j) * gx + i
    loc = [(a, b_, c) for c in range(n1) for b_ in range(n1) for a in range(n1)]
    e2n = np.stack([nid(order * ex + a, order * ey + b_, order * ez + c) for (a, b_, c) in loc], axis=1)
    nde = 3 * len(loc)
    elem_to_dof = (3 * e2n[:, :, None] + np.arange(3)[None, None, :]).reshape(NE, nde).astype(np.int32)
    Kref = hex_elasticity_matrix_tensor(h, order, lam, mu)
    elmat = np.ascontiguousarray(np.broadcast_to(Kref, (NE, nde, nde)))
    A0 = _assemble(ND, elem_to_dof, elmat)
    _, _, L1 = _lagrange_1d(order)
    load = -h[0] * h[1] * h[2] * np.kron(L1, np.kron(L1, L1))
    b0 = np.zeros(ND)
    np.add.at(b0, elem_to_dof[:, 2::3].ravel(), np.tile(load, NE))
    ix = np.arange(NV) % gx
    ess = np.repeat(ix == 0, 3)
    A, b = _eliminate(A0, b0, ess)
    bdr = (np.where(ess, AGG_ON_ESS_DOMAIN_BORDER_FLAG, 0) | AGG_OWNED_FLAG).astype(np.int8)
    part0, nb = block_partition(n, blk)
    return Problem(A=A, b=b, elem_to_dof=elem_to_dof, elmat=elmat, bdr=bdr, ess=ess,
                   partitions=[part0], dims=n, order=order, Kref=Kref, coefs=np.ones(NE))


# --------------------------------------------------------------------------
# device-resident generator (torch is plumbing: it only allocates/fills HBM)
# --------------------------------------------------------------------------
def hex_elasticity_matrix(h, lam=1.0, mu=1.0):
    """24 x 24 stiffness of isotropic linear elasticity on an h[0] x h[1] x h[2] trilinear hex
    (2x2x2 Gauss, exact for the box), vertex order _HEX_LOC, dofs ordered byVDIM
    (3*vertex + component; the ordering of the reference's elasticity driver,
    amg/src/fem.cpp:493-504)."""
    g = np.array([-1.0, 1.0]) / np.sqrt(3.0)
    sgn = np.array([[2 * a - 1, 2 * b - 1, 2 * c - 1] for (a, b, c) in _HEX_LOC], dtype=float)
    C = np.zeros((6, 6))
    C[:3, :3] = lam
    C[np.arange(3), np.arange(3)] += 2 * mu
    C[np.arange(3, 6), np.arange(3, 6)] = mu
    Ke = np.zeros((24, 24))
    detJ = h[0] * h[1] * h[2] / 8.0
    for xi in g:
        for eta in g:
            for zeta in g:
                pt = np.array([xi, eta, zeta])
                dN = np.zeros((8, 3))
                for a in range(8):
                    f = 1.0 + sgn[a] * pt
                    dN[a, 0] = sgn[a, 0] * f[1] * f[2] / 8.0 * (2.0 / h[0])
                    dN[a, 1] = sgn[a, 1] * f[0] * f[2] / 8.0 * (2.0 / h[1])
                    dN[a, 2] = sgn[a, 2] * f[0] * f[1] / 8.0 * (2.0 / h[2])
                B = np.zeros((6, 24))
                for a in range(8):
                    dx, dy, dz = dN[a]
                    B[0, 3 * a] = dx
                    B[1, 3 * a + 1] = dy
                    B[2, 3 * a + 2] = dz
                    B[3, 3 * a], B[3, 3 * a + 1] = dy, dx
                    B[4, 3 * a + 1], B[4, 3 * a + 2] = dz, dy
                    B[5, 3 * a], B[5, 3 * a + 2] = dz, dx
                Ke += detJ * (B.T @ C @ B)
    return 0.5 * (Ke + Ke.T)


def elasticity3d_problem(n, blk=(4, 4, 4), lam=1.0, mu=1.0, coarse_blk=None):
    """Unit cube of n Q1 hexes, 3 displacement components per vertex (byVDIM), clamped on the
    face x = 0, body force (0, 0, -1): the vector-dof workload of the reference's elasticity
    drivers (lambda = mu = 1, amg/test/mltest/mltest.cpp:581).  AEs away from the clamped face
    carry the six rigid-body modes as exact zero eigenvalues."""
    if np.isscalar(n):
        n = (int(n),) * 3
    nx, ny, nz = n
    h = (1.0 / nx, 1.0 / ny, 1.0 / nz)
    nvx, nvy, nvz = nx + 1, ny + 1, nz + 1
    NV = nvx * nvy * nvz
    ND = 3 * NV
    NE = nx * ny * nz
    ez, ey, ex = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    ex, ey, ez = ex.ravel(), ey.ravel(), ez.ravel()
    vid = lambda i, j, k: (k * nvy + j) * nvx + i
    e2v = np.stack([vid(ex + a, ey + b_, ez + c) for (a, b_, c) in _HEX_LOC], axis=1)
    elem_to_dof = (3 * e2v[:, :, None] + np.arange(3)[None, None, :]).reshape(NE, 24).astype(np.int32)
    Kref = hex_elasticity_matrix(h, lam, mu)
    elmat = np.ascontiguousarray(np.broadcast_to(Kref, (NE, 24, 24)))
    A0 = _assemble(ND, elem_to_dof, elmat)
    b0 = np.zeros(ND)
    np.add.at(b0, elem_to_dof[:, 2::3].ravel(), -h[0] * h[1] * h[2] / 8.0)
    iz, iy, ix = np.meshgrid(np.arange(nvz), np.arange(nvy), np.arange(nvx), indexing="ij")
    ess = np.repeat((ix == 0).ravel(), 3)
    A, b = _eliminate(A0, b0, ess)
    bdr = (np.where(ess, AGG_ON_ESS_DOMAIN_BORDER_FLAG, 0) | AGG_OWNED_FLAG).astype(np.int8)
    part0, nb = block_partition(n, blk)
    parts = [part0]
    for cb in (coarse_blk or []):
        p, nb = block_partition(nb, cb)
        parts.append(p)
    return Problem(A=A, b=b, elem_to_dof=elem_to_dof, elmat=elmat, bdr=bdr, ess=ess,
                   partitions=parts, dims=n, order=1, Kref=Kref, coefs=np.ones(NE))


def poisson3d_device(n, blk=(8, 8, 4), coarse_blk=None, K=(1.0, 1.0, 1.0), device="cuda", coef=None):
    """Same problem as poisson3d_problem, generated directly in HBM with torch so that 128^3 / 256^3 inputs never
    touch the host.  Returns a Problem whose arrays are torch tensors on `device` (A as rowptr/col/val tensors).
    coef: None (constant) or "skew" (poisson3d_problem's smooth coefficient without any symmetry: no two
    agglomerates congruent, every stored entry a different value -- a GENERAL operator for the SpMV formats)."""
    import torch
    if np.isscalar(n):
        n = (int(n),) * 3
    nx, ny, nz = n
    h = (1.0 / nx, 1.0 / ny, 1.0 / nz)
    nvx, nvy, nvz = nx + 1, ny + 1, nz + 1
    ND = nvx * nvy * nvz
    NE = nx * ny * nz
    dev = torch.device(device)
    Kref = torch.tensor(hex_element_matrix(h, K), dtype=torch.float64, device=dev)
    # local index of the vertex with offset (ox,oy,oz) in {0,1}^3 inside an element
    lidx = {}
    for a, (ox, oy, oz) in enumerate(_HEX_LOC):
        lidx[(ox, oy, oz)] = a
    iz = torch.arange(nvz, device=dev).view(-1, 1, 1)
    iy = torch.arange(nvy, device=dev).view(1, -1, 1)
    ix = torch.arange(nvx, device=dev).view(1, 1, -1)
    ess = ((ix == 0) | (ix == nx) | (iy == 0) | (iy == ny) | (iz == 0) | (iz == nz)).reshape(-1)
    node = ((iz * nvy + iy) * nvx + ix).reshape(-1)
    cel = None
    if coef == "skew":
        cz = ((torch.arange(nz, device=dev, dtype=torch.float64) + 0.5) * h[2]).view(-1, 1, 1)
        cy = ((torch.arange(ny, device=dev, dtype=torch.float64) + 0.5) * h[1]).view(1, -1, 1)
        cx = ((torch.arange(nx, device=dev, dtype=torch.float64) + 0.5) * h[0]).view(1, 1, -1)
        cel = torch.exp(0.7 * cx + 0.4 * cy - 0.3 * cz) * (1.0 + 0.3 * torch.sin(5.0 * cx + 3.0 * cy + 7.0 * cz))
        cpad = torch.zeros((nz + 2, ny + 2, nx + 2), dtype=torch.float64, device=dev)      # zero outside the mesh
        cpad[1:-1, 1:-1, 1:-1] = cel
    elif coef is not None:
        raise ValueError("poisson3d_device: coef must be None or 'skew'")
    vals = torch.zeros((ND, 27), dtype=torch.float64, device=dev)
    cols = torch.zeros((ND, 27), dtype=torch.int32, device=dev)
    valid = torch.zeros((ND, 27), dtype=torch.bool, device=dev)
    o = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                jx, jy, jz = ix + dx, iy + dy, iz + dz
                ok = ((jx >= 0) & (jx <= nx) & (jy >= 0) & (jy <= ny) & (jz >= 0) & (jz <= nz))
                acc = torch.zeros((nvz, nvy, nvx), dtype=torch.float64, device=dev)
                # elements containing both nodes, ascending element id (ez, ey, ex)
                for sz in (-1, 0):
                    for sy in (-1, 0):
                        for sx in (-1, 0):
                            ex, ey, ez = ix + sx, iy + sy, iz + sz
                            a = (-sx, -sy, -sz)                     # this node inside the element
                            b = (dx - sx, dy - sy, dz - sz)         # neighbour inside the element
                            if min(b) < 0 or max(b) > 1:
                                continue
                            inside = ((ex >= 0) & (ex < nx) & (ey >= 0) & (ey < ny) &
                                      (ez >= 0) & (ez < nz))
                            if cel is None:
                                acc = acc + torch.where(inside & ok, Kref[lidx[a], lidx[b]], 0.0)
                            else:      # coefficient of element (ex, ey, ez): the padded array, shifted views
                                ce = cpad[1 + sz:1 + sz + nvz, 1 + sy:1 + sy + nvy, 1 + sx:1 + sx + nvx]
                                acc = acc + torch.where(inside & ok, ce * Kref[lidx[a], lidx[b]], 0.0)
                vals[:, o] = acc.reshape(-1)
                cols[:, o] = ((jz * nvy + jy) * nvx + jx).reshape(-1).to(torch.int32)
                valid[:, o] = ok.expand(nvz, nvy, nvx).reshape(-1)
                o += 1
    # eliminate essential rows / columns, keep the diagonal and the explicit zeros
    colc = cols.clamp(0, ND - 1).long()
    kill = (ess.view(-1, 1) | ess[colc]) & (colc != node.view(-1, 1))
    vals = torch.where(kill, torch.zeros_like(vals), vals)
    counts = valid.sum(dim=1)
    rowptr = torch.zeros(ND + 1, dtype=torch.int32, device=dev)
    rowptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    flat = valid.reshape(-1)
    A_col = cols.reshape(-1)[flat].contiguous()
    A_val = vals.reshape(-1)[flat].contiguous()
    del vals, cols, valid, colc, kill
    # elements
    ez = torch.arange(nz, device=dev).view(-1, 1, 1)
    ey = torch.arange(ny, device=dev).view(1, -1, 1)
    ex = torch.arange(nx, device=dev).view(1, 1, -1)
    e2d = torch.stack([(((ez + c) * nvy + (ey + b_)) * nvx + (ex + a)).reshape(-1)
                       for (a, b_, c) in _HEX_LOC], dim=1).to(torch.int32).contiguous()
    if cel is None:
        elmat = Kref.reshape(1, 64).expand(NE, 64).contiguous()
    else:
        elmat = (cel.reshape(-1, 1) * Kref.reshape(1, 64)).contiguous()
    # right-hand side f = 1
    wx = torch.full((nvx,), 1.0, dtype=torch.float64, device=dev); wx[0] = wx[-1] = 0.5
    wy = torch.full((nvy,), 1.0, dtype=torch.float64, device=dev); wy[0] = wy[-1] = 0.5
    wz = torch.full((nvz,), 1.0, dtype=torch.float64, device=dev); wz[0] = wz[-1] = 0.5
    b = (h[0] * h[1] * h[2]) * (wz.view(-1, 1, 1) * wy.view(1, -1, 1) * wx.view(1, 1, -1)).reshape(-1)
    b = torch.where(ess, torch.zeros_like(b), b).contiguous()
    bdr = torch.where(ess, AGG_ON_ESS_DOMAIN_BORDER_FLAG | AGG_OWNED_FLAG, AGG_OWNED_FLAG).to(torch.int8)
    # partitions
    def blockpart(dims, bl):
        mx, my, mz = dims
        nbx, nby, nbz = -(-mx // bl[0]), -(-my // bl[1]), -(-mz // bl[2])
        kz = torch.arange(mz, device=dev).view(-1, 1, 1) // bl[2]
        ky = torch.arange(my, device=dev).view(1, -1, 1) // bl[1]
        kx = torch.arange(mx, device=dev).view(1, 1, -1) // bl[0]
        return ((kz * nby + ky) * nbx + kx).reshape(-1).to(torch.int32).contiguous(), (nbx, nby, nbz)
    part0, nb = blockpart(n, blk)
    parts, nparts = [part0], [nb[0] * nb[1] * nb[2]]
    for cb in (coarse_blk or []):
        p, nb = blockpart(nb, cb)
        parts.append(p)
        nparts.append(nb[0] * nb[1] * nb[2])
    return Problem(rowptr=rowptr, col=A_col, val=A_val, n=ND, b=b, elem_to_dof=e2d, elmat=elmat,
                   bdr=bdr, partitions=parts, nparts=nparts, dims=n, NE_=NE, ess=ess)


def elasticity3d_q2_device(n, blk=(4, 4, 4), coarse_blk=None, lam=1.0, mu=1.0, device="cuda", slab_nodes=1 << 20,
                           index_dtype=None):
    """Same problem as elasticity3d_q2_problem (BASELINE config 5: Q2 hexes, 3 displacement components per node
    byVDIM, clamped on x = 0, body force (0, 0, -1)), generated directly in HBM with torch.

    A uniform mesh with a constant coefficient has only 4^3 = 64 kinds of rows -- per direction a node sits on
    the low end, inside, on the high end (element vertices: two, one or two elements) or at an element midpoint
    -- so the row stencils (offsets, 3 x 3 blocks summed over the elements that contain both nodes, in
    ascending element order like the assembly) are tabulated on the host and every stored entry is a table
    lookup; rows are produced slab by slab to bound the working set.  rowptr is int64 when nnz >= 2^31.
    Returns a Problem whose arrays are torch tensors on `device`."""
    import torch
    if np.isscalar(n):
        n = (int(n),) * 3
    nx, ny, nz = n
    h = (1.0 / nx, 1.0 / ny, 1.0 / nz)
    gx, gy, gz = 2 * nx + 1, 2 * ny + 1, 2 * nz + 1
    NV, NE = gx * gy * gz, nx * ny * nz
    ND = 3 * NV
    dev = torch.device(device)
    Ke = hex_elasticity_matrix_tensor(h, 2, lam, mu)
    _, _, L1 = _lagrange_1d(2)
    load = -h[0] * h[1] * h[2] * np.kron(L1, np.kron(L1, L1))          # node = (iz * 3 + iy) * 3 + ix
    # per-direction kinds: 0 low end, 1 inside vertex, 2 high end, 3 midpoint -> [(element shift, local index)]
    # in ascending element order
    opts = {0: [(0, 0)], 1: [(-1, 2), (0, 0)], 2: [(-1, 2)], 3: [(0, 1)]}
    noff = np.zeros(64, dtype=np.int64)
    offl = np.zeros((64, 125), dtype=np.int64)                          # linear node offset of every stencil slot
    vtab = np.zeros((64, 3, 125, 3))
    ltab = np.zeros(64)
    for cz in range(4):
        for cy in range(4):
            for cx in range(4):
                cls = (cz * 4 + cy) * 4 + cx
                blocks = {}
                for (sz, lz) in opts[cz]:                               # ascending element id: z slowest
                    for (sy, ly) in opts[cy]:
                        for (sx, lx) in opts[cx]:
                            la = (lz * 3 + ly) * 3 + lx
                            ltab[cls] += load[la]
                            for dz in range(-lz, 3 - lz):
                                for dy in range(-ly, 3 - ly):
                                    for dx in range(-lx, 3 - lx):
                                        lb = ((lz + dz) * 3 + (ly + dy)) * 3 + (lx + dx)
                                        blk3 = Ke[3 * la:3 * la + 3, 3 * lb:3 * lb + 3]
                                        key = (dz, dy, dx)
                                        blocks[key] = blocks[key] + blk3 if key in blocks else blk3.copy()
                keys = sorted(blocks)
                noff[cls] = len(keys)
                for s, (dz, dy, dx) in enumerate(keys):
                    offl[cls, s] = (dz * gy + dy) * gx + dx
                    vtab[cls, :, s, :] = blocks[(dz, dy, dx)]
    t_noff = torch.tensor(noff, device=dev)
    t_offl = torch.tensor(offl, device=dev)
    t_vtab = torch.tensor(vtab, device=dev)
    t_ltab = torch.tensor(ltab, device=dev)

    def kind(c, g):
        return torch.where(c % 2 == 1, torch.full_like(c, 3), torch.where(c == 0, torch.zeros_like(c),
                           torch.where(c == g - 1, torch.full_like(c, 2), torch.ones_like(c))))

    node_all = torch.arange(NV, device=dev)
    ci, cj, ck = node_all % gx, (node_all // gx) % gy, node_all // (gx * gy)
    cls_all = (kind(ck, gz) * 4 + kind(cj, gy)) * 4 + kind(ci, gx)
    ess_node = ci == 0
    row_len = (3 * t_noff[cls_all]).repeat_interleave(3)                  # per dof row
    nnz = int(row_len.sum())
    big = nnz >= 2 ** 31
    it = torch.int64 if (big or index_dtype == "int64") else torch.int32
    rowptr = torch.zeros(ND + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(row_len, 0)
    A_col = torch.empty(nnz, dtype=torch.int32, device=dev)
    A_val = torch.empty(nnz, dtype=torch.float64, device=dev)
    for n0 in range(0, NV, slab_nodes):
        n1 = min(NV, n0 + slab_nodes)
        rows = torch.arange(3 * n0, 3 * n1, device=dev)
        lens = row_len[3 * n0:3 * n1]
        e0, e1 = int(rowptr[3 * n0]), int(rowptr[3 * n1])
        erow = rows.repeat_interleave(lens)                             # row of every entry of the slab
        slot = torch.arange(e0, e1, device=dev) - rowptr[erow]
        enode, ecomp = erow // 3, erow % 3
        o, c2 = slot // 3, slot % 3
        ecls = cls_all[enode]
        cnode = enode + t_offl[ecls, o]
        col = 3 * cnode + c2
        val = t_vtab[ecls, ecomp, o, c2]
        kill = (ess_node[enode] | ess_node[cnode]) & (col != erow)
        A_col[e0:e1] = col.to(torch.int32)
        A_val[e0:e1] = torch.where(kill, torch.zeros_like(val), val)
        del erow, slot, enode, ecomp, o, c2, ecls, cnode, col, val, kill
    # elements
    ez = torch.arange(nz, device=dev).view(-1, 1, 1)
    ey = torch.arange(ny, device=dev).view(1, -1, 1)
    ex = torch.arange(nx, device=dev).view(1, 1, -1)
    loc = [(a, b_, c) for c in range(3) for b_ in range(3) for a in range(3)]
    e2n = torch.stack([(((2 * ez + c) * gy + (2 * ey + b_)) * gx + (2 * ex + a)).reshape(-1) for (a, b_, c) in loc], dim=1)
    e2d = (3 * e2n.unsqueeze(2) + torch.arange(3, device=dev).view(1, 1, 3)).reshape(NE, 81).to(torch.int32).contiguous()
    elmat = torch.tensor(Ke, device=dev).reshape(1, 81 * 81).expand(NE, 81 * 81).contiguous()
    b = torch.zeros(ND, dtype=torch.float64, device=dev)
    b[2::3] = torch.where(ess_node, torch.zeros(NV, dtype=torch.float64, device=dev), t_ltab[cls_all])
    ess = ess_node.repeat_interleave(3)
    bdr = torch.where(ess, AGG_ON_ESS_DOMAIN_BORDER_FLAG | AGG_OWNED_FLAG, AGG_OWNED_FLAG).to(torch.int8)

    def blockpart(dims, bl):
        mx, my, mz = dims
        nbx, nby, nbz = -(-mx // bl[0]), -(-my // bl[1]), -(-mz // bl[2])
        kz = torch.arange(mz, device=dev).view(-1, 1, 1) // bl[2]
        ky = torch.arange(my, device=dev).view(1, -1, 1) // bl[1]
        kx = torch.arange(mx, device=dev).view(1, 1, -1) // bl[0]
        return ((kz * nby + ky) * nbx + kx).reshape(-1).to(torch.int32).contiguous(), (nbx, nby, nbz)
    part0, nb = blockpart(n, blk)
    parts, nparts = [part0], [nb[0] * nb[1] * nb[2]]
    for cb in (coarse_blk or []):
        p, nb = blockpart(nb, cb)
        parts.append(p)
        nparts.append(nb[0] * nb[1] * nb[2])
    return Problem(rowptr=rowptr.to(it), col=A_col, val=A_val, n=ND, b=b, elem_to_dof=e2d, elmat=elmat, bdr=bdr,
                   partitions=parts, nparts=nparts, dims=n, NE_=NE, ess=ess, nde_=81, nnz_=nnz)


def split_parcsr(prob, world, levels):
    """Per-rank inputs of saamge_amd_ml_produce_data_parcsr from a global host Problem: what the reference's multi-rank
    drivers hold (pmltest, amg/CMakeLists.txt:198-203) -- the row block of A in hypre's ParCSR split (diag / offd /
    col_map_offd), the rank's own elements (global dof ids), their matrices, the flags of its own rows and its own
    agglomerate partitions with LOCAL agglomerate ids.

    The agglomerates of the coarsest partition are dealt to the ranks in contiguous id ranges; the problem must be
    numbered so that this makes the elements and the agglomerates of every level contiguous per rank (structured meshes
    split into slabs along the slowest direction are) -- checked.  A dof belongs to the lowest rank one of whose elements
    touches it; those sets must be contiguous row ranges -- checked.  Returns a list of `world` dicts."""
    import scipy.sparse as sp
    nco = levels - 1
    parts = [np.asarray(p, dtype=np.int64) for p in prob.partitions[:nco]]
    ntop = int(parts[-1].max()) + 1
    assert ntop >= world, "fewer coarsest agglomerates (%d) than ranks (%d)" % (ntop, world)
    top_owner = (np.arange(ntop) * world) // ntop                       # contiguous ranges of the coarsest agglomerates
    owner = [None] * nco                                                # owner[l][a] = rank of level-l agglomerate a
    owner[nco - 1] = top_owner
    for l in range(nco - 2, -1, -1):
        owner[l] = owner[l + 1][parts[l + 1]]
    elem_owner = owner[0][parts[0]]
    for arr, what in [(elem_owner, "elements")] + [(owner[l], "level-%d agglomerates" % l) for l in range(nco)]:
        assert np.all(np.diff(arr) >= 0), "%s of a rank are not contiguous in the global numbering" % what
    e2d = np.asarray(prob.elem_to_dof, dtype=np.int64)
    n = prob.A.shape[0]
    dof_owner = np.full(n, world, dtype=np.int64)
    np.minimum.at(dof_owner, e2d.ravel(), np.repeat(elem_owner, e2d.shape[1]))
    assert dof_owner.max() < world and np.all(np.diff(dof_owner) >= 0), "the dofs of a rank are not a contiguous row range"
    row_starts = np.searchsorted(dof_owner, np.arange(world + 1)).astype(np.int64)
    A = prob.A.tocsr()
    elmat = np.asarray(prob.elmat, dtype=np.float64).reshape(e2d.shape[0], -1)
    bdr = np.asarray(prob.bdr, dtype=np.int8)
    out = []
    for r in range(world):
        r0, r1 = int(row_starts[r]), int(row_starts[r + 1])
        rows = A[r0:r1].tocoo()
        inside = (rows.col >= r0) & (rows.col < r1)
        diag = sp.csr_matrix((rows.data[inside], (rows.row[inside], rows.col[inside] - r0)), shape=(r1 - r0, r1 - r0))
        cmap = np.unique(rows.col[~inside]).astype(np.int64)
        offd = sp.csr_matrix((rows.data[~inside], (rows.row[~inside], np.searchsorted(cmap, rows.col[~inside]))),
                             shape=(r1 - r0, max(len(cmap), 1)))
        # hypre keeps the diagonal entry FIRST in every row of diag and the rest unsorted: reproduce that order
        dI, dJ, dV = diag.indptr.astype(np.int32), diag.indices.astype(np.int32).copy(), diag.data.copy()
        for i in range(r1 - r0):
            a, b = dI[i], dI[i + 1]
            k = np.nonzero(dJ[a:b] == i)[0]
            if k.size and k[0] != 0:
                kk = a + int(k[0])
                dJ[a + 1:kk + 1], dJ[a] = dJ[a:kk].copy(), dJ[kk]
                dV[a + 1:kk + 1], dV[a] = dV[a:kk].copy(), dV[kk]
        mine = np.nonzero(elem_owner == r)[0]
        p_loc, np_loc = [], []
        for l in range(nco):
            src = mine if l == 0 else np.nonzero(owner[l - 1] == r)[0]
            ae = np.nonzero(owner[l] == r)[0]
            p_loc.append(np.ascontiguousarray(parts[l][src] - ae[0], dtype=np.int32))
            np_loc.append(int(ae.size))
        out.append({"global_rows": n, "row_starts": row_starts.copy(), "nrows": r1 - r0,
                    "diag_i": dI, "diag_j": dJ, "diag_a": dV,
                    "offd_i": offd.indptr.astype(np.int32), "offd_j": offd.indices.astype(np.int32), "offd_a": offd.data.copy(),
                    "num_cols_offd": int(len(cmap)), "col_map_offd": cmap if len(cmap) else np.zeros(1, dtype=np.int64),
                    "elem_to_dof": np.ascontiguousarray(e2d[mine], dtype=np.int32), "elmat": np.ascontiguousarray(elmat[mine]),
                    "bdr": np.ascontiguousarray(bdr[r0:r1]), "partitions": p_loc, "nparts": np_loc})
    return out


def split_parcsr_device(prob, world, rank, levels):
    """split_parcsr for a Problem whose arrays are torch tensors in HBM (poisson3d_device / elasticity3d_q2_device): the
    piece of ONE rank, built on the device without touching the host (the rows of a slab are a contiguous range of the CSR
    arrays: diag / offd are masked selections of it, in the caller's column order)."""
    import torch
    nco = levels - 1
    parts = [p.long() for p in prob.partitions[:nco]]
    ntop = int(prob.nparts[nco - 1])
    assert ntop >= world
    dev = parts[0].device
    owner = [None] * nco
    owner[nco - 1] = (torch.arange(ntop, device=dev) * world) // ntop
    for l in range(nco - 2, -1, -1):
        owner[l] = owner[l + 1][parts[l + 1]]
    elem_owner = owner[0][parts[0]]
    for arr in [elem_owner] + owner:
        assert bool((arr[1:] >= arr[:-1]).all()), "a rank's elements / agglomerates are not contiguous in the global numbering"
    e2d = prob.elem_to_dof.long()
    n = int(prob.n)
    dof_owner = torch.full((n,), world, dtype=torch.long, device=dev)
    dof_owner.scatter_reduce_(0, e2d.reshape(-1), elem_owner.repeat_interleave(e2d.shape[1]), reduce="amin")
    assert int(dof_owner.max()) < world and bool((dof_owner[1:] >= dof_owner[:-1]).all())
    row_starts = torch.searchsorted(dof_owner, torch.arange(world + 1, device=dev)).cpu().numpy().astype(np.int64)
    r0, r1 = int(row_starts[rank]), int(row_starts[rank + 1])
    nl = r1 - r0
    rp = prob.rowptr[r0:r1 + 1].long()
    e0, e1 = int(rp[0]), int(rp[-1])
    col = prob.col[e0:e1].long()
    val = prob.val[e0:e1]
    row_of = torch.repeat_interleave(torch.arange(nl, device=dev), rp[1:] - rp[:-1])
    inside = (col >= r0) & (col < r1)

    def csr(mask, cols):
        cnt = torch.bincount(row_of[mask], minlength=nl)
        ptr = torch.zeros(nl + 1, dtype=torch.int32, device=dev)
        ptr[1:] = torch.cumsum(cnt, 0).to(torch.int32)
        return ptr, cols.to(torch.int32).contiguous(), val[mask].contiguous()
    di, dj, da = csr(inside, col[inside] - r0)
    cmap = torch.unique(col[~inside])
    oi, oj, oa = csr(~inside, torch.searchsorted(cmap, col[~inside]))
    mine = torch.nonzero(elem_owner == rank).reshape(-1)
    nde2 = prob.elmat.shape[-1] if prob.elmat.dim() == 2 else int(np.prod(prob.elmat.shape[1:]))
    elm = prob.elmat.reshape(e2d.shape[0], nde2)
    p_loc, np_loc = [], []
    for l in range(nco):
        src = mine if l == 0 else torch.nonzero(owner[l - 1] == rank).reshape(-1)
        ae = torch.nonzero(owner[l] == rank).reshape(-1)
        p_loc.append((parts[l][src] - ae[0]).to(torch.int32).contiguous())
        np_loc.append(int(ae.numel()))
    m0, m1 = int(mine[0]), int(mine[-1]) + 1
    return {"global_rows": n, "row_starts": row_starts, "nrows": nl, "diag_i": di, "diag_j": dj, "diag_a": da,
            "offd_i": oi, "offd_j": oj, "offd_a": oa, "num_cols_offd": int(cmap.numel()),
            "col_map_offd": cmap.contiguous() if cmap.numel() else torch.zeros(1, dtype=torch.long, device=dev),
            # (clones: a slice that is contiguous already would be a VIEW that keeps the whole global array alive)
            "elem_to_dof": prob.elem_to_dof[m0:m1].clone(), "elmat": elm[m0:m1].clone(),
            "bdr": prob.bdr[r0:r1].clone(), "partitions": p_loc, "nparts": np_loc}
