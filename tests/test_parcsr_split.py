"""Host logic of the per-rank inputs: problems.split_parcsr produces hypre's ParCSR split (diag / offd / col_map_offd) of the
global operator; put together again the pieces must give the global problem back."""
import numpy as np
import scipy.sparse as sp

from saamge_amd import problems as pr


def test_split_parcsr_reassembles_to_the_global_problem():
    prob = pr.poisson3d_problem((8, 8, 12), blk=(4, 4, 2), coarse_blk=[(2, 2, 2)], coef="checkerboard")
    world = 3
    pieces = pr.split_parcsr(prob, world, 3)
    n = prob.A.shape[0]
    rs = pieces[0]["row_starts"]
    assert rs[0] == 0 and rs[-1] == n and sum(p["nrows"] for p in pieces) == n
    blocks, e2d, elmat, parts0, parts1 = [], [], [], [], []
    off0 = off1 = 0
    for r, p in enumerate(pieces):
        nl = p["nrows"]
        assert nl == rs[r + 1] - rs[r]
        d = sp.csr_matrix((p["diag_a"], p["diag_j"], p["diag_i"]), shape=(nl, nl)).tocoo()
        # hypre's order: the diagonal entry first in every row
        for i in range(nl):
            assert p["diag_j"][p["diag_i"][i]] == i
        o = sp.csr_matrix((p["offd_a"], p["offd_j"], p["offd_i"]), shape=(nl, max(p["num_cols_offd"], 1))).tocoo()
        cm = p["col_map_offd"]
        assert np.all(np.diff(cm[:p["num_cols_offd"]]) > 0) and not np.any((cm[:p["num_cols_offd"]] >= rs[r]) & (cm[:p["num_cols_offd"]] < rs[r + 1]))
        rows = np.concatenate([d.row, o.row]) + rs[r]
        cols = np.concatenate([d.col + rs[r], cm[o.col]])
        blocks.append(sp.csr_matrix((np.concatenate([d.data, o.data]), (rows, cols)), shape=(n, n)))
        e2d.append(p["elem_to_dof"]); elmat.append(p["elmat"])
        parts0.append(p["partitions"][0] + off0); parts1.append(p["partitions"][1] + off1)
        off0 += p["nparts"][0]; off1 += p["nparts"][1]
        assert np.array_equal(p["bdr"], prob.bdr[rs[r]:rs[r + 1]])
    assert abs(sum(blocks) - prob.A).max() == 0.0
    assert np.array_equal(np.concatenate(e2d), prob.elem_to_dof)
    assert np.array_equal(np.concatenate(elmat).reshape(prob.elmat.shape), prob.elmat)
    assert np.array_equal(np.concatenate(parts0), prob.partitions[0]) and np.array_equal(np.concatenate(parts1), prob.partitions[1])
