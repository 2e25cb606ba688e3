"""ctypes wrapper of oracle/cpu_ref.cpp -- TEST INFRASTRUCTURE + bench.py's cpu_baseline only.

The threaded C++ restatement of the reference's setup + solve (LAPACK dsygvx / dgesvd, one
agglomerate per core).  Nothing under ``saamge_amd/`` may import this module."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libcpu_ref.so")
_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        lib = C.CDLL(LIB_PATH)
        lib.cpu_ref_setup.restype = C.c_void_p
        lib.cpu_ref_setup2.restype = C.c_void_p
        lib.cpu_ref_Ac_trace.restype = C.c_double
        lib.cpu_ref_Ac_trace.argtypes = [C.c_void_p, C.c_int]
        lib.cpu_ref_Ac_fro2.restype = C.c_double
        lib.cpu_ref_Ac_fro2.argtypes = [C.c_void_p, C.c_int]
        lib.cpu_ref_error.restype = C.c_char_p
        lib.cpu_ref_error.argtypes = [C.c_void_p]
        lib.cpu_ref_free.argtypes = [C.c_void_p]
        lib.cpu_ref_setup_seconds.restype = C.c_double
        lib.cpu_ref_setup_seconds.argtypes = [C.c_void_p]
        lib.cpu_ref_num_levels.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


def _p(a):
    return C.c_void_p(0) if a is None else C.c_void_p(a.ctypes.data)


class Hierarchy(object):
    """ml_produce_data on the host cores; same inputs as saamge_amd.capi.Hierarchy.from_problem."""

    def __init__(self, prob, num_coarsenings=1, theta=0.003, nu_relax=3, threads=1, lean=False):
        """`theta`: one value or one per coarsening.  `prob`: a host Problem (scipy `A`) or the output of
        problems.poisson3d_device(device="cpu") (`rowptr` / `col` / `val` tensors).  `lean`: rebuild the fine level's
        dense AE matrices on demand (256^3: 86 GB otherwise)."""
        lib = load()
        def host(a, dt):
            return np.ascontiguousarray(a.numpy() if hasattr(a, "numpy") else a, dtype=dt)
        if hasattr(prob, "A"):
            A = prob.A.tocsr()
            self.n = A.shape[0]
            rowptr, col, val = host(A.indptr, np.int32), host(A.indices, np.int32), host(A.data, np.float64)
        else:
            self.n = int(prob.n)
            rowptr, col, val = host(prob.rowptr, np.int32), host(prob.col, np.int32), host(prob.val, np.float64)
        e2d = host(prob.elem_to_dof, np.int32)
        self._elmat = elmat = host(prob.elmat, np.float64).reshape(e2d.shape[0], -1)    # (lean mode reads it during the setup only)
        bdr = host(prob.bdr, np.int8)
        parts = [host(p, np.int32) for p in prob.partitions[:num_coarsenings]]
        nparts = (C.c_int * len(parts))(*[int(p.max()) + 1 for p in parts])
        pp = (C.c_void_p * len(parts))(*[p.ctypes.data for p in parts])
        thetas = np.ascontiguousarray(np.broadcast_to(np.asarray(theta, dtype=np.float64), (len(parts),)))
        self.h = C.c_void_p(lib.cpu_ref_setup2(
            C.c_int(self.n), _p(rowptr), _p(col), _p(val), C.c_int(e2d.shape[0]), C.c_int(e2d.shape[1]), _p(e2d),
            _p(elmat), _p(bdr), C.c_int(len(parts)), pp, nparts, _p(thetas), C.c_int(nu_relax),
            C.c_int(threads), C.c_int(int(lean))))
        err = lib.cpu_ref_error(self.h)
        if err:
            raise RuntimeError("cpu_ref: " + err.decode())
        self.setup_s = lib.cpu_ref_setup_seconds(self.h)
        self.num_levels = lib.cpu_ref_num_levels(self.h)

    def level_info(self, l):
        info = (C.c_longlong * 8)()
        load().cpu_ref_level_info(self.h, C.c_int(l), info)
        return dict(zip(["n", "nnz", "nparts", "num_mises", "ncoarse", "nnzP", "nnzAc"], [int(v) for v in info]))

    def level_dims(self):
        infos = [self.level_info(l) for l in range(self.num_levels)]
        return [i["n"] for i in infos] + [infos[-1]["ncoarse"]]

    def ae_m(self, l):
        out = np.zeros(self.level_info(l)["nparts"], dtype=np.int32)
        load().cpu_ref_get_ints(self.h, C.c_int(l), C.c_int(0), _p(out))
        return out

    def mis_k(self, l):
        out = np.zeros(self.level_info(l)["num_mises"], dtype=np.int32)
        load().cpu_ref_get_ints(self.h, C.c_int(l), C.c_int(1), _p(out))
        return out

    def mises(self, l):
        out = np.zeros(self.level_info(l)["n"], dtype=np.int32)
        load().cpu_ref_get_ints(self.h, C.c_int(l), C.c_int(2), _p(out))
        return out

    def evals_max(self, l):
        out = np.zeros(self.level_info(l)["nparts"])
        load().cpu_ref_get_evals_max(self.h, C.c_int(l), _p(out))
        return out

    def evals(self, l):
        """all kept eigenvalues of level l, agglomerate after agglomerate (split with np.cumsum(ae_m))"""
        out = np.zeros(int(self.ae_m(l).sum()))
        load().cpu_ref_get_evals(self.h, C.c_int(l), _p(out))
        return out

    def mis_to_AE(self, l):
        nm = self.level_info(l)["num_mises"]
        I = np.zeros(nm + 1, dtype=np.int32)
        load().cpu_ref_get_mis_to_AE(self.h, C.c_int(l), _p(I), None)
        J = np.zeros(int(I[-1]), dtype=np.int32)
        load().cpu_ref_get_mis_to_AE(self.h, C.c_int(l), _p(I), _p(J))
        return I, J

    def sv_ratios(self, l):
        """per MIS: (smallest kept, largest dropped) singular value over the largest (the 1e-10 cut, src/xpacks.cpp:591-620)"""
        nm = self.level_info(l)["num_mises"]
        kept, dropped = np.zeros(nm), np.zeros(nm)
        load().cpu_ref_get_sv_ratios(self.h, C.c_int(l), C.c_int(0), _p(kept))
        load().cpu_ref_get_sv_ratios(self.h, C.c_int(l), C.c_int(1), _p(dropped))
        return kept, dropped

    def Ac_trace(self, l):
        return float(load().cpu_ref_Ac_trace(self.h, C.c_int(l)))

    def Ac_fro(self, l):
        return float(load().cpu_ref_Ac_fro2(self.h, C.c_int(l))) ** 0.5

    def vcycle(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros_like(b)
        load().cpu_ref_vcycle(self.h, _p(b), _p(x))
        return x

    def pcg(self, b, rel_tol=1e-6, max_iter=1000, squared_tol=True):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros_like(b)
        hist = np.zeros(max_iter + 2)
        conv = C.c_int(0)
        secs = C.c_double(0.0)
        lib = load()
        it = lib.cpu_ref_pcg(self.h, _p(b), _p(x), C.c_double(rel_tol), C.c_int(max_iter), C.c_int(int(squared_tol)),
                             C.byref(conv), _p(hist), C.byref(secs))
        self.solve_s = secs.value
        return x, it, bool(conv.value), hist[:it + 1].copy()

    def close(self):
        if self.h:
            load().cpu_ref_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
