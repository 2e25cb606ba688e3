"""The device generators of bench.py's workloads against the host generators (torch on the CPU here):
same sparsity, flags, right-hand side and partitions; values to one ulp (the assembly order of scipy's
COO -> CSR conversion is not the table's ascending element order)."""
import numpy as np
import pytest

from saamge_amd import problems as pr


@pytest.mark.parametrize("n", [(2, 2, 2), (3, 2, 4)])
def test_q2_elasticity_device_generator_matches_host(n):
    hp = pr.elasticity3d_q2_problem(n, blk=(2, 2, 2))
    dp = pr.elasticity3d_q2_device(n, blk=(2, 2, 2), device="cpu", slab_nodes=50)
    A = hp.A.tocsr()
    A.sort_indices()
    assert dp.nnz_ == A.nnz and dp.n == A.shape[0] and dp.nde_ == 81
    assert np.array_equal(A.indptr, dp.rowptr.numpy()) and np.array_equal(A.indices, dp.col.numpy())
    assert np.abs(A.data - dp.val.numpy()).max() <= 4 * np.finfo(float).eps * np.abs(A.data).max()
    assert np.array_equal(hp.elem_to_dof, dp.elem_to_dof.numpy())
    assert np.allclose(hp.b, dp.b.numpy(), rtol=0, atol=1e-18) and np.array_equal(hp.bdr, dp.bdr.numpy())
    assert np.array_equal(hp.partitions[0], dp.partitions[0].numpy())
    assert np.allclose(hp.elmat[0].ravel(), dp.elmat[0].numpy(), rtol=1e-15)


def test_poisson_device_generator_matches_host():
    hp = pr.poisson3d_problem((5, 4, 3), blk=(2, 2, 2), K=(1.0, 1.0, 7.0))
    dp = pr.poisson3d_device((5, 4, 3), blk=(2, 2, 2), K=(1.0, 1.0, 7.0), device="cpu")
    A = hp.A.tocsr()
    A.sort_indices()
    assert np.array_equal(A.indptr, dp.rowptr.numpy()) and np.array_equal(A.indices, dp.col.numpy())
    assert np.abs(A.data - dp.val.numpy()).max() <= 4 * np.finfo(float).eps * np.abs(A.data).max()
    assert np.array_equal(hp.elem_to_dof, dp.elem_to_dof.numpy()) and np.array_equal(hp.bdr, dp.bdr.numpy())
    assert np.allclose(hp.b, dp.b.numpy(), rtol=1e-15, atol=0)
