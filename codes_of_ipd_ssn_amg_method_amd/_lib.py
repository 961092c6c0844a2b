"""ctypes binding of libipdamg.so (the C ABI declared in include/ipd_amg.h).

There is no CPU fallback: importing this module fails loudly when the shared
library is missing, and creating a context fails loudly when no gfx950 device
is usable.
"""
from __future__ import annotations

import ctypes as C
import os
from ctypes import POINTER, Structure, byref, c_char_p, c_double, c_int, c_int32, c_int64, c_size_t, \
    c_uint8, c_uint32, c_void_p

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libipdamg.so")


class IpdError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libipdamg error {code}: {msg}")
        self.code = code
        self.msg = msg


IPD_E_ARG, IPD_E_HIP, IPD_E_NOMEM, IPD_E_LIMIT, IPD_E_NUMERIC, IPD_E_UNSUPPORTED, IPD_E_COMM = \
    -1, -2, -3, -4, -5, -6, -7
COMM_ID_BYTES = 128


class ipd_csc(Structure):
    _fields_ = [("nrows", c_int64), ("ncols", c_int64), ("nnz", c_int64),
                ("jc", POINTER(c_int64)), ("ir", POINTER(c_int64)), ("pr", POINTER(c_double))]


class ipd_csc_out(Structure):
    _fields_ = ipd_csc._fields_


class ipd_amg_opts(Structure):
    _fields_ = [("retol", c_double), ("bigph", c_int32), ("maxit", c_int32), ("theta", c_double),
                ("smoth", c_int32), ("cycle", c_int32), ("isnsp", c_int32), ("inter", c_int32),
                ("fnode", c_int64)]


class ipd_pcg_opts(Structure):
    _fields_ = [("retol", c_double), ("maxit", c_int64), ("precd", c_int32), ("nf", c_int64)]


class ipd_prob(Structure):
    _fields_ = [("m", c_int64), ("n", c_int64), ("bk1", c_double), ("tk", c_double),
                ("p", POINTER(c_double)), ("q", POINTER(c_double)), ("t", POINTER(c_double)),
                ("H0", POINTER(ipd_csc)), ("z", POINTER(c_double)), ("s", POINTER(c_uint8)),
                ("phi", POINTER(c_double))]


class ipd_apd_data(Structure):
    _fields_ = [("cls", c_int32), ("m", c_int64), ("n", c_int64), ("c", POINTER(c_double)),
                ("r", POINTER(c_double)), ("l", POINTER(c_double)), ("p", POINTER(c_double)),
                ("q", POINTER(c_double)), ("gama", POINTER(c_double)), ("gama_scalar", c_double),
                ("mu", c_double), ("phi", POINTER(c_double))]


class ipd_apd_opts(Structure):
    _fields_ = [("maxit", c_int32), ("kkt_tol", c_double), ("ssn_it", c_int32),
                ("ssn_tol1", c_double), ("nu", c_double), ("delta", c_double),
                ("ll_max", c_int32), ("prob", c_int32), ("inner_solver", c_int32),
                ("pcg_retol", c_double), ("pcg_maxit", c_int64)]


class ipd_ssn_rec(Structure):
    _fields_ = [("k", c_int32), ("ssn_it", c_int32), ("ll", c_int32), ("itamg", c_int32),
                ("E", c_int64), ("info0", c_int64), ("info1", c_int64), ("Fk_norm", c_double),
                ("resamg", c_double), ("bk1", c_double), ("tk", c_double)]


class ipd_apd_result(Structure):
    _fields_ = [("converged", c_int32), ("k", c_int32), ("fval", c_double),
                ("kkt", c_double * 4), ("rr", c_double), ("sum_amg", c_int64),
                ("total_amg", c_int64), ("fail_amg", c_int64), ("max_amg", c_int64),
                ("restarts", c_int32), ("nrec", c_int64)]


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
            "(python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback.")
    # RTLD_LOCAL on purpose: torch bundles ROCm libraries with the same SONAMEs; exporting ours
    # globally and importing torch afterwards aborts at interpreter exit (double free)
    return C.CDLL(LIB_PATH)


lib = _load()
lib.ipd_last_error.restype = c_char_p
lib.ipd_rng_consumed.restype = c_int64
lib.ipd_csc_free.restype = None
lib.ipd_ctx_destroy.restype = None
lib.ipd_rng_destroy.restype = None
lib.ipd_amg_destroy.restype = None
lib.ipd_dmat_destroy.restype = None
lib.ipd_amg_opts_init.restype = None
lib.ipd_pcg_opts_init.restype = None
lib.ipd_apd_opts_init.restype = None
lib.ipd_apd_destroy.restype = None

# every symbol the header declares (tests check that they all resolve)
EXPORTS = [
    "ipd_version", "ipd_last_error", "ipd_device_count", "ipd_ctx_create", "ipd_ctx_destroy", "ipd_ctx_sync",
    "ipd_csc_free", "ipd_amg_opts_init", "ipd_pcg_opts_init", "ipd_rng_create",
    "ipd_rng_create_replay", "ipd_rng_destroy", "ipd_rng_rand", "ipd_rng_consumed", "ipd_ax",
    "ipd_aty", "ipd_asat", "ipd_inv_aat", "ipd_inv_hht", "ipd_strength", "ipd_cf_split",
    "ipd_mis_set", "ipd_transfer", "ipd_amg_setup", "ipd_amg_destroy", "ipd_amg_num_levels",
    "ipd_amg_level_dims", "ipd_amg_get_A", "ipd_amg_get_P", "ipd_amg_get_cmask", "ipd_amg_solve",
    "ipd_amg_vcycle", "ipd_amg_wcycle", "ipd_class_amg", "ipd_pcg", "ipd_components",
    "ipd_hybrid_amg", "ipd_amg4pot", "ipd_dmalloc", "ipd_dfree", "ipd_h2d", "ipd_d2h",
    "ipd_dmat_upload", "ipd_dmat_download", "ipd_dmat_dims", "ipd_dmat_destroy",
    "ipd_dmat_multiply", "ipd_spmv_dev",
    "ipd_ax_dev", "ipd_aty_dev", "ipd_asat_dev", "ipd_amg_setup_dev", "ipd_amg_solve_dev",
    "ipd_hybrid_amg_dev", "ipd_amg_bench_cycles", "ipd_amg_bench_sweeps", "ipd_amg_cycle_bytes", "ipd_amg_solve_mode", "ipd_amg_bench_resident", "ipd_comm_get_unique_id",
    "ipd_comm_init", "ipd_comm_finalize", "ipd_comm_stats", "ipd_ctx_set_component_order", "ipd_amg_bench_cycles_sharded",
    "ipd_apd_opts_init", "ipd_apd_create", "ipd_apd_destroy", "ipd_apd_dims", "ipd_apd_warmup",
    "ipd_apd_set_state", "ipd_apd_get_state", "ipd_apd_run", "ipd_apd_history",
    "ipd_apd_records", "ipd_apd_reuse_stats", "ipd_apd_begin", "ipd_apd_eval", "ipd_apd_bench_eval", "ipd_prof_read", "ipd_amg_bench_subcycle",
    "ipd_amg_attach_mask_operator", "ipd_amg_attach_mask_transfers", "ipd_amg_attach_level2_poly", "ipd_twogrid_bigph", "ipd_twogrid", "ipd_hybrid_twogrid", "ipd_amg4pot_twogrid",
    "ipd_aug_pcg", "ipd_pcg4pot", "ipd_spd_solve", "ipd_amg_resident_levels", "ipd_amg_resident_kernel", "ipd_amg_level_forms", "ipd_amg_poly_operator",
]


def check(rc: int) -> None:
    if rc != 0:
        raise IpdError(rc, (lib.ipd_last_error() or b"").decode("utf-8", "replace"))


# ---------------------------------------------------------------------------
# marshalling helpers
# ---------------------------------------------------------------------------
def f64(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))


def u8(a) -> np.ndarray:
    return np.ascontiguousarray((np.asarray(a).reshape(-1) != 0).astype(np.uint8))


def dptr(a: np.ndarray):
    return a.ctypes.data_as(POINTER(c_double))


def bptr(a: np.ndarray):
    return a.ctypes.data_as(POINTER(c_uint8))


def iptr(a: np.ndarray):
    return a.ctypes.data_as(POINTER(c_int64))


class CscIn:
    """Keeps the numpy arrays behind an ipd_csc alive."""

    def __init__(self, A):
        A = sp.csc_matrix(A, dtype=np.float64)
        A.sum_duplicates()
        A.eliminate_zeros()
        A.sort_indices()
        self.jc = np.ascontiguousarray(A.indptr, dtype=np.int64)
        self.ir = np.ascontiguousarray(A.indices, dtype=np.int64)
        self.pr = np.ascontiguousarray(A.data, dtype=np.float64)
        if self.ir.size == 0:
            self.ir = np.zeros(1, np.int64)
            self.pr = np.zeros(1, np.float64)
        self.struct = ipd_csc(A.shape[0], A.shape[1], A.nnz, iptr(self.jc), iptr(self.ir),
                              dptr(self.pr))

    def ref(self):
        return byref(self.struct)


def csc_out_to_scipy(out: ipd_csc_out) -> sp.csc_matrix:
    """Copy a library-owned output matrix into SciPy and free it."""
    nr, nc, nnz = int(out.nrows), int(out.ncols), int(out.nnz)
    jc = np.ctypeslib.as_array(out.jc, shape=(nc + 1,)).copy()
    if nnz:
        ir = np.ctypeslib.as_array(out.ir, shape=(nnz,)).copy()
        pr = np.ctypeslib.as_array(out.pr, shape=(nnz,)).copy()
    else:
        ir = np.zeros(0, np.int64)
        pr = np.zeros(0, np.float64)
    lib.ipd_csc_free(byref(out))
    return sp.csc_matrix((pr, ir, jc), shape=(nr, nc))


# ---------------------------------------------------------------------------
# context / rng singletons
# ---------------------------------------------------------------------------
class Context:
    def __init__(self, device: int = 0):
        self.handle = c_void_p()
        check(lib.ipd_ctx_create(c_int(device), byref(self.handle)))
        self.device = device

    def sync(self):
        check(lib.ipd_ctx_sync(self.handle))

    def close(self):
        if self.handle:
            lib.ipd_ctx_destroy(self.handle)
            self.handle = c_void_p()


_default_ctx: Context | None = None


def get_ctx() -> Context:
    """Process-wide context on ``LOCAL_RANK`` (one process per GPU)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get("IPD_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    return _default_ctx


class MatlabRand:
    """MATLAB-compatible ``rand`` stream (mt19937ar; seed 5489 == MATLAB default)."""

    def __init__(self, seed: int = 5489, replay=None):
        self.handle = c_void_p()
        if replay is not None:
            self._vals = f64(replay)
            check(lib.ipd_rng_create_replay(dptr(self._vals), c_int64(self._vals.size),
                                            byref(self.handle)))
        else:
            check(lib.ipd_rng_create(c_uint32(seed), byref(self.handle)))

    def rand(self, n: int) -> np.ndarray:
        out = np.empty(int(n), np.float64)
        check(lib.ipd_rng_rand(self.handle, c_int64(int(n)), dptr(out)))
        return out

    @property
    def consumed(self) -> int:
        return int(lib.ipd_rng_consumed(self.handle))

    def __del__(self):
        try:
            if self.handle:
                lib.ipd_rng_destroy(self.handle)
                self.handle = c_void_p()
        except Exception:
            pass


class DeviceBuffer:
    """Raw device allocation on the context's GPU (ipd_dmalloc / ipd_h2d / ipd_d2h)."""

    def __init__(self, nbytes: int, ctx: Context | None = None):
        self.ctx = ctx or get_ctx()
        self.nbytes = int(nbytes)
        self.ptr = c_void_p()
        check(lib.ipd_dmalloc(self.ctx.handle, c_size_t(self.nbytes), byref(self.ptr)))

    @classmethod
    def from_array(cls, a: np.ndarray, ctx: Context | None = None) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        buf = cls(a.nbytes, ctx)
        check(lib.ipd_h2d(buf.ctx.handle, buf.ptr, a.ctypes.data_as(c_void_p), c_size_t(a.nbytes)))
        return buf

    def to_array(self, dtype, count: int) -> np.ndarray:
        out = np.empty(int(count), dtype=dtype)
        check(lib.ipd_d2h(self.ctx.handle, out.ctypes.data_as(c_void_p), self.ptr,
                          c_size_t(out.nbytes)))
        return out

    def free(self):
        if self.ptr:
            lib.ipd_dfree(self.ctx.handle, self.ptr)
            self.ptr = c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
