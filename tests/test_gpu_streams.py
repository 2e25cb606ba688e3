"""Lifetime of the caller's stream vs the library's cache of freed device blocks (round-2 advisor finding): a hierarchy
is built and solved on a stream created by the caller, freed, the stream is DESTROYED, and another hierarchy is built
on a new stream.  The frees of the first hierarchy are filed in a batch keyed by its stream; saamge_amd_ml_free_data
closes that batch (records its event) while the stream is alive, so nothing refers to the dead handle afterwards and
the second build reuses the cached blocks."""
import ctypes as C

import numpy as np
import pytest

from saamge_amd import problems as pr

pytestmark = pytest.mark.gpu


def test_free_destroy_stream_rebuild_on_a_new_stream():
    import torch
    from saamge_amd import capi
    torch.cuda.init()
    torch.zeros(1, device="cuda:0")
    lib = capi.load()
    # the one HIP runtime of the process (torch's), already mapped: open it by the path it was loaded from
    path = next(line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line)
    hip = C.CDLL(path)
    for fn in ("hipStreamCreate", "hipStreamDestroy", "hipStreamSynchronize"):
        assert hasattr(hip, fn), fn
    prob = pr.poisson3d_problem((12, 12, 8), blk=(4, 4, 4), coarse_blk=[(2, 2, 2)], coef="skew")
    params = capi.default_params(num_coarsenings=2, theta=0.003, nu_relax=3)
    results = []
    cached = []
    for rep in range(3):
        s = C.c_void_p()
        assert hip.hipStreamCreate(C.byref(s)) == 0
        h = capi.Hierarchy.from_problem(prob, params, stream=s.value)
        x, it, conv, hist = h.pcg(prob.b, rel_tol=1e-8)
        results.append((it, np.array(hist)))
        h.close()
        cached.append(capi.cached_memory_bytes())
        assert hip.hipStreamDestroy(s) == 0          # the handle is dead from here on
    assert all(r[0] == results[0][0] for r in results)
    assert all(np.array_equal(r[1], results[0][1]) for r in results)      # same arithmetic whatever blocks were reused
    assert cached[0] > 0 and cached[2] <= 2 * cached[0] + (64 << 20)       # blocks are reused, not piled up
    capi.release_cached_memory()
    assert capi.cached_memory_bytes() == 0
