"""CPU tests (no GPU): the C-ABI library loads and exports every symbol the header
declares; host-side generators are consistent.  No compute calls."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from saamge_amd import capi
    lib = capi.load()
    hdr = open(os.path.join(ROOT, "include", "saamge_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(saamge_amd_[a-z_0-9]+)\s*\(", hdr)))
    assert declared, "no declarations found"
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert sorted(capi.SYMBOLS) == declared


def test_params_struct_matches_header_defaults():
    from saamge_amd import capi
    p = capi.default_params()
    assert p.num_coarsenings == 1 and abs(p.theta[0] - 0.003) < 1e-15 and p.nu_relax[0] == 3
    assert p.avoid_ess_bdr_dofs == 1 and p.nu_pro[0] == 0 and p.workspace_bytes == 32 << 30
    assert p.world == 1 and p.rank == 0


def test_options_struct_matches_the_header_field_by_field():
    """saamge_amd_options: the ctypes mirror lists the header's fields in the header's order, and the library's defaults are the
    documented ones (the struct is embedded in saamge_amd_params: a field out of place shifts everything behind it)."""
    import ctypes as C
    from saamge_amd import capi
    hdr = open(os.path.join(ROOT, "include", "saamge_amd.h")).read()
    body = hdr[hdr.index("typedef struct saamge_amd_options {"):hdr.index("} saamge_amd_options;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\bint\s+([a-z_0-9]+)\s*;", body)
    assert fields == [f for f, _ in capi.Options._fields_]
    o = capi.Options()
    capi.load().saamge_amd_options_default(C.byref(o))
    got = {f: getattr(o, f) for f in fields}
    assert got["eig_dedupe"] == 1 and got["eig_outer_panels"] == 8 and got["overlap"] == 15 and got["sell"] == 31
    assert got["host_heap_pad_mb"] == 256 and got["eig_min_n"] == 64 and got["debug"] == 0 and got["eig_strict"] == 0


def test_product_has_no_oracle_dependency():
    """The shipped package must not import the oracle (test infrastructure only)."""
    pkg = os.path.join(ROOT, "saamge_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("no CPU fallback", ""), os.path.join(dirpath, f)


def test_generators_consistent():
    import torch
    from saamge_amd import problems as pr
    for n in [(4, 4, 4), (5, 3, 2)]:
        p = pr.poisson3d_problem(n, blk=(2, 2, 2))
        d = pr.poisson3d_device(n, blk=(2, 2, 2), device="cpu")
        assert np.array_equal(p.A.indptr, d.rowptr.numpy())
        assert np.array_equal(p.A.indices, d.col.numpy())
        assert np.allclose(p.A.data, d.val.numpy(), rtol=0, atol=1e-15)
        assert np.array_equal(p.elem_to_dof, d.elem_to_dof.numpy())
        assert np.array_equal(p.partitions[0], d.partitions[0].numpy())
        assert np.allclose(p.b, d.b.numpy(), atol=1e-18)
        assert abs(p.A - p.A.T).max() < 1e-15
        # nnz = (3n+1)^3 for a cube
    p = pr.poisson3d_problem(4, blk=(2, 2, 2))
    assert p.A.nnz == 13 ** 3
    # element matrices sum to the un-eliminated operator: row sums vanish (Neumann)
    assert np.abs(p.elmat.sum(axis=2)).max() < 1e-14
