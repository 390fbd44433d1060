"""ctypes binding of libfigh.so (include/figh.h).  No torch, no fallback.

``load()`` raises if the shared library is missing; every compute call raises
``FighError`` when the library reports a failure (no HIP device, bad
arguments ...).  Nothing in this package computes on the CPU instead.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FIGH_LIB_PATH: another build of the same ABI (same-box A/B of kernel variants); default is the in-tree library
LIB_PATH = os.environ.get("FIGH_LIB_PATH") or os.path.join(_HERE, "libfigh.so")

ABI_VERSION = 106  # include/figh.h FIGH_ABI_VERSION: load() refuses a library of another ABI

FIGH_OK = 0
ERR_INVALID, ERR_NO_DEVICE, ERR_ALLOC, ERR_UNSUPPORTED, ERR_COMM = -1, -2, -3, -4, -5

MODE_JOINT_TORQUE, MODE_EXT_WRENCH = 0, 1
FLAG_FRICTION, FLAG_ACT_INERTIA, FLAG_OFFSET, FLAG_TX40, FLAG_GENERIC, FLAG_BLOCKED_INPUTS = 1, 2, 4, 8, 256, 512
FLAG_COMPACT_BLOCKS = 2048  # figh_regressor_build_padded writes the block-compact W (figh.h)
FLAG_FORCE_COMPACT = 8192  # figh_regressor_build_padded, external wrench: force rows in their own region, a line per 4 links
FLAG_LINK_COMPACT = 4096  # figh_regressor_build_padded, external wrench on a free-flyer root: links without entries dropped
FLAG_ZEROS_PRESENT = 1024  # opt-in of figh_regressor_build_padded: structural zeros of W are already there (figh.h)

_c_double_p = C.POINTER(C.c_double)
_c_int32_p = C.POINTER(C.c_int32)

# symbol -> (restype, argtypes); exactly the declarations of include/figh.h
SIGNATURES = {
    "figh_version": (C.c_int, []),
    "figh_last_error": (C.c_char_p, []),
    "figh_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "figh_device_set": (C.c_int, [C.c_int]),
    "figh_host_wait_mode": (C.c_int, [C.c_int]),
    "figh_tsqr_null_pivot_tol": (C.c_int, [C.c_double]),
    "figh_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "figh_device_info": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    "figh_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "figh_free": (C.c_int, [C.c_void_p]),
    "figh_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "figh_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "figh_host_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "figh_host_free": (C.c_int, [C.c_void_p]),
    "figh_memcpy_d2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "figh_memset": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t]),
    "figh_synchronize": (C.c_int, []),
    "figh_profile_enable": (C.c_int, [C.c_int]),
    "figh_profile_reset": (C.c_int, []),
    "figh_profile_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), _c_double_p]),
    "figh_model_create": (C.c_int, [C.c_int, _c_int32_p, _c_int32_p, _c_double_p, _c_double_p, _c_int32_p,
                                    _c_int32_p, _c_double_p, _c_int32_p, C.POINTER(C.c_void_p)]),
    "figh_model_destroy": (C.c_int, [C.c_void_p]),
    "figh_regressor_shape": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "figh_regressor_build": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "figh_regressor_build_padded": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "figh_repack_samples": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "figh_coupling_tx40": (C.c_int, [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "figh_colsq": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p]),
    "figh_gather_cols": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int64]),
    "figh_place_block": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_void_p, C.c_int64]),
    "figh_matvec": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "figh_block_sqnorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "figh_tsqr": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, _c_double_p, C.c_int,
                            C.c_void_p]),
    "figh_tsqr_structured": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, _c_double_p,
                                       C.c_int, _c_int32_p, C.c_int, C.c_void_p]),
    "figh_tsqr_merge": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "figh_select_columns": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p]),
    "figh_tsqr_selected_wrench": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_double, C.c_int,
                                            C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int64]),
    "figh_regressor_force_layout": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "figh_model_set_active_rows": (C.c_int, [C.c_void_p, _c_int32_p, C.c_int]),
    "figh_regressor_link_layout": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _c_int32_p, C.POINTER(C.c_int)]),
    "figh_tsqr_selected_blocks": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_double, C.c_int,
                                            C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "figh_block_rows_residuals": (C.c_int, [C.c_void_p, C.c_int, _c_int32_p, C.c_int, C.c_void_p, C.c_void_p]),
    "figh_tsqr_selected": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]),
    "figh_regressor_tsqr_fused": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_double,
                                            C.c_void_p]),
    "figh_tsqr_merge_base": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]),
    "figh_compact_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_double, C.c_void_p,
                                    C.c_int64, C.c_void_p, C.POINTER(C.c_int64)]),
    "figh_base_permutation": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]),
    "figh_regressor_colsq": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int64, C.c_void_p]),
    "figh_regressor_tsqr": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, _c_double_p, C.c_int, C.c_int64,
                                      C.c_void_p]),
    "figh_regressor_tsqr_norms": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, _c_double_p, C.c_int, C.c_int64,
                                            C.c_void_p, C.c_void_p]),
    "figh_regressor_tsqr_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "figh_regressor_gram": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, _c_double_p, _c_double_p,
                                      _c_double_p]),
    "figh_filtfilt_cols": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int, _c_double_p, _c_double_p,
                                     C.c_int, C.c_int, _c_double_p, C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                     C.POINTER(C.c_int64)]),
    "figh_comm_available": (C.c_int, []),
    "figh_comm_unique_id": (C.c_int, [C.c_void_p]),
    "figh_comm_init": (C.c_int, [C.c_int, C.c_int, C.c_void_p]),
    "figh_comm_destroy": (C.c_int, []),
    "figh_comm_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "figh_comm_allreduce_sum": (C.c_int, [C.c_void_p, C.c_int64]),
}


class FighError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libfigh error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load():
    """Load libfigh.so (built by ``__graft_entry__.build()`` / ``make -C figaroh_plus_amd/csrc``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libfigh.so not found at %s -- build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        have = lib.figh_version()
        if have != ABI_VERSION:
            raise ImportError("%s implements ABI %d, this package is written against %d (include/figh.h FIGH_ABI_VERSION): "
                              "rebuild it (make -C figaroh_plus_amd/csrc)" % (LIB_PATH, have, ABI_VERSION))
        _lib = lib
    return _lib


def check(rc):
    if rc != FIGH_OK:
        msg = load().figh_last_error()
        raise FighError(rc, msg.decode() if msg else "")
    return rc


def device_count():
    n = C.c_int(0)
    check(load().figh_device_count(C.byref(n)))
    return n.value


def device_info():
    name = C.create_string_buffer(256)
    cus = C.c_int(0)
    mem = C.c_size_t(0)
    check(load().figh_device_info(name, 256, C.byref(cus), C.byref(mem)))
    return {"name": name.value.decode(), "cu_count": cus.value, "hbm_bytes": mem.value}


def synchronize():
    check(load().figh_synchronize())


class DeviceArray:
    """A float64 / int32 buffer in HBM owned by this object (hipMalloc through the C-ABI)."""

    def __init__(self, shape, dtype=np.float64):
        self.shape = (shape,) if np.isscalar(shape) else tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.size = int(np.prod(self.shape)) if len(self.shape) else 1
        self.nbytes = self.size * self.dtype.itemsize
        p = C.c_void_p()
        check(load().figh_malloc(C.byref(p), max(self.nbytes, 8)))
        self.ptr = p.value

    @classmethod
    def from_host(cls, arr, dtype=None):
        arr = np.ascontiguousarray(arr, dtype=dtype if dtype is not None else arr.dtype)
        d = cls(arr.shape, arr.dtype)
        if arr.nbytes:
            check(load().figh_memcpy_h2d(d.ptr, arr.ctypes.data, arr.nbytes))
        return d

    def to_host(self, out=None):
        if out is None:
            out = np.empty(self.shape, dtype=self.dtype)
        assert out.nbytes == self.nbytes and out.flags["C_CONTIGUOUS"]
        if self.nbytes:
            check(load().figh_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes))
        return out

    def zero_(self):
        check(load().figh_memset(self.ptr, 0, self.nbytes))
        return self

    def free(self):
        if getattr(self, "ptr", None):
            load().figh_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


_null_tol = 0.0


def tsqr_null_pivot_tol(tol):
    """Threshold of the null-pivot rule of the TSQR kernels (include/figh.h); 0 = exact zeros only.  Returns the previous one."""
    global _null_tol
    check(load().figh_tsqr_null_pivot_tol(float(tol)))
    prev, _null_tol = _null_tol, float(tol)
    return prev


class null_pivots:
    """``with null_pivots(tol_qr): ...`` -- the TSQR launches inside run with the null-pivot rule at tol_qr / 64: a column
    that is going to be classified as dependent (|R_kk| <= tol_qr, qrdecomposition.py:215-221) by a margin of 64 costs a
    norm per tile instead of a column step."""

    def __init__(self, tol_qr):  # None: leave the process-wide setting alone
        self.tol = None if tol_qr is None else max(float(tol_qr), 0.0) / 64.0

    def __enter__(self):
        self.prev = None if self.tol is None else tsqr_null_pivot_tol(self.tol)
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            tsqr_null_pivot_tol(self.prev)
        return False


def profile_enable(on=True, level=2):
    """level 1: dominant kernels only (cheap, usable inside a timed region); level 2: every launch."""
    check(load().figh_profile_enable(int(level) if on else 0))


def profile_reset():
    check(load().figh_profile_reset())


def profile_get(name):
    n = C.c_int(0)
    ms = C.c_double(0.0)
    check(load().figh_profile_get(name.encode(), C.byref(n), C.byref(ms)))
    return n.value, ms.value


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class ModelHandle:
    """figh_model_t for a flattened tree (``Model.to_flat()``)."""

    def __init__(self, flat):
        n = int(flat["njoints"])
        self.njoints, self.nq, self.nv = n, int(flat["nq"]), int(flat["nv"])
        parents, jtype = _i32(flat["parents"]), _i32(flat["jtype"])
        idx_q, idx_v = _i32(flat["idx_q"]), _i32(flat["idx_v"])
        axis, placement, gravity = _f64(flat["axis"]), _f64(flat["placement"]), _f64(flat["gravity"])
        body_mask = _i32(np.asarray(flat["mass"], dtype=float) != 0.0)
        h = C.c_void_p()
        check(load().figh_model_create(
            n, parents.ctypes.data_as(_c_int32_p), jtype.ctypes.data_as(_c_int32_p),
            axis.ctypes.data_as(_c_double_p), placement.ctypes.data_as(_c_double_p),
            idx_q.ctypes.data_as(_c_int32_p), idx_v.ctypes.data_as(_c_int32_p),
            gravity.ctypes.data_as(_c_double_p), body_mask.ctypes.data_as(_c_int32_p), C.byref(h)))
        self.handle = h.value
        # the library's own criterion (figh_model_create): fixed-base serial chain of <= 8 revolute joints
        self._chain = n - 1 <= 8 and all(int(jtype[i]) == 0 and int(parents[i]) == i - 1 for i in range(1, n))

    def is_chain(self):
        """True when the joint-torque regressor of this model is built by the chain kernel (dense row tiles)."""
        return self._chain

    def set_active_rows(self, rows=None):
        """figh_model_set_active_rows: the dof indices whose row blocks regressor_build_padded stores (None: all).  Changes
        what THIS handle writes: take a private handle (``ModelHandle(model.to_flat())``), not a shared one."""
        if rows is None or len(rows) == 0:
            check(load().figh_model_set_active_rows(self.handle, None, 0))
        else:
            arr = np.ascontiguousarray(rows, dtype=np.int32)
            check(load().figh_model_set_active_rows(self.handle, arr.ctypes.data_as(_c_int32_p), len(arr)))

    def shape(self, mode, flags):
        r, c = C.c_int(0), C.c_int(0)
        check(load().figh_regressor_shape(self.handle, mode, flags, C.byref(r), C.byref(c)))
        return r.value, c.value

    def destroy(self):
        if getattr(self, "handle", None):
            load().figh_model_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


# ------------------------------------------------------------------ thin typed wrappers (device pointers)
def regressor_build(model, mode, flags, ft_mask, N, d_q, d_v, d_a, d_W, ldw, d_colsq=None):
    check(load().figh_regressor_build(model.handle, mode, flags, ft_mask, N, d_q.ptr, d_v.ptr, d_a.ptr, d_W.ptr, ldw,
                                      d_colsq.ptr if d_colsq is not None else None))


def regressor_build_padded(model, mode, flags, ft_mask, N, d_q, d_v, d_a, d_W, ldw, d_colsq=None):
    """Link-padded W (16 columns per link) for device-resident use; ``d_W`` may be None (column norms only)."""
    check(load().figh_regressor_build_padded(model.handle, mode, flags, ft_mask, N, d_q.ptr, d_v.ptr, d_a.ptr,
                                             d_W.ptr if d_W is not None else None, ldw,
                                             d_colsq.ptr if d_colsq is not None else None))


def coupling_tx40(N, nv, d_v, d_a, d_out):
    check(load().figh_coupling_tx40(N, nv, d_v.ptr, d_a.ptr, d_out.ptr))


def colsq(d_W, rows, cols, ldw, d_out):
    check(load().figh_colsq(d_W.ptr, rows, cols, ldw, d_out.ptr))


def gather_cols(d_W, rows, ldw, d_idx, n, d_out, ldo):
    check(load().figh_gather_cols(d_W.ptr, rows, ldw, d_idx.ptr, n, d_out.ptr, ldo))


def place_block(src_ptr, ld_src, rows, cols, scale, dst_ptr, ld_dst):
    """Raw device addresses (``DeviceArray.ptr`` + a byte offset): the blocks are sub-matrices."""
    check(load().figh_place_block(src_ptr, ld_src, rows, cols, scale, dst_ptr, ld_dst))


def matvec(d_W, rows, ldw, d_idx, n, d_x, d_y):
    check(load().figh_matvec(d_W.ptr, rows, ldw, d_idx.ptr if d_idx is not None else None, n, d_x.ptr, d_y.ptr))


def block_sqnorm(d_a, d_b, rows, nblocks, d_out):
    check(load().figh_block_sqnorm(d_a.ptr, d_b.ptr if d_b is not None else None, rows, nblocks, d_out.ptr))


def tsqr(d_W, rows, ldw, d_idx, n, d_tau, block_weight, d_R, first_cols=None):
    """``first_cols``: structure hint (figh_tsqr_structured): per row block, the first gathered column that can be
    non-zero."""
    bw = None
    nb = 0
    if block_weight is not None:
        bwa = _f64(block_weight)
        bw = bwa.ctypes.data_as(_c_double_p)
        nb = len(bwa)
    if first_cols is not None:
        fc = _i32(first_cols)
        check(load().figh_tsqr_structured(d_W.ptr, rows, ldw, d_idx.ptr if d_idx is not None else None, n,
                                          d_tau.ptr if d_tau is not None else None, bw, nb,
                                          fc.ctypes.data_as(_c_int32_p), len(fc), d_R.ptr))
        return
    check(load().figh_tsqr(d_W.ptr, rows, ldw, d_idx.ptr if d_idx is not None else None, n,
                           d_tau.ptr if d_tau is not None else None, bw, nb, d_R.ptr))


def regressor_colsq(model, mode, flags, ft_mask, N, d_q, d_v, d_a, d_colsq, chunk_samples=0):
    check(load().figh_regressor_colsq(model.handle, mode, flags, ft_mask, N, d_q.ptr, d_v.ptr, d_a.ptr, chunk_samples,
                                      d_colsq.ptr))


def regressor_tsqr(model, mode, flags, ft_mask, N, d_q, d_v, d_a, d_idx, n, d_tau, block_weight, d_R, chunk_samples=0,
                   d_colsq=None):
    """``d_colsq`` (ncols doubles): also diag(W^T W) of all columns from the same pass (figh_regressor_tsqr_norms)."""
    bw, nb = None, 0
    if block_weight is not None:
        bwa = _f64(block_weight)
        bw, nb = bwa.ctypes.data_as(_c_double_p), len(bwa)
    args = (model.handle, mode, flags, ft_mask, N, d_q.ptr, d_v.ptr, d_a.ptr, d_idx.ptr if d_idx is not None else None, n,
            d_tau.ptr if d_tau is not None else None, bw, nb, chunk_samples, d_R.ptr)
    if d_colsq is not None:
        check(load().figh_regressor_tsqr_norms(*args, d_colsq.ptr))
    else:
        check(load().figh_regressor_tsqr(*args))


class PinnedArray:
    """float64 vector in page-locked host memory (figh_host_alloc); ``.array`` is a NumPy view, valid until ``free``."""

    def __init__(self, n):
        self._ptr = C.c_void_p()
        check(load().figh_host_alloc(C.byref(self._ptr), 8 * int(n)))
        self.size = int(n)
        self.array = np.ctypeslib.as_array(C.cast(self._ptr, C.POINTER(C.c_double)), shape=(self.size,))

    def free(self):
        if self._ptr:
            self.array = None
            check(load().figh_host_free(self._ptr))
            self._ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


def regressor_tsqr_batch(model, mode, flags, ft_mask, B, n_per, d_q, d_v, d_a, d_idx, n, d_R_stack, d_R):
    """B trajectories of n_per samples (back to back in d_q, d_v, d_a) -> B n x n triangles in d_R."""
    check(load().figh_regressor_tsqr_batch(model.handle, mode, flags, ft_mask, B, n_per, d_q.ptr, d_v.ptr, d_a.ptr,
                                           d_idx.ptr if d_idx is not None else None, n,
                                           d_R_stack.ptr if d_R_stack is not None else None, d_R.ptr))


def regressor_gram(model, mode, flags, ft_mask, N, d_q, d_v, d_a, d_idx, n, d_tau=None, chunk_samples=0):
    """(G, g, tau_sq): W_e^T W_e, W_e^T tau, tau^T tau (g, tau_sq None without tau)."""
    G = np.empty((n, n))
    g = np.empty(n) if d_tau is not None else None
    tt = C.c_double(0.0)
    check(load().figh_regressor_gram(model.handle, mode, flags, ft_mask, N, d_q.ptr, d_v.ptr, d_a.ptr,
                                     d_idx.ptr if d_idx is not None else None, n,
                                     d_tau.ptr if d_tau is not None else None, chunk_samples,
                                     G.ctypes.data_as(_c_double_p),
                                     g.ctypes.data_as(_c_double_p) if g is not None else None,
                                     C.byref(tt) if d_tau is not None else None))
    return G, g, (tt.value if d_tau is not None else None)


def filtfilt_cols(d_X, rows, cols, ldx, nblocks, form, b, a, zi, padlen, q, d_Y, ldy):
    """Zero-phase filter + keep every q-th sample of every (row block, column) sequence; returns the output rows."""
    b, a, zi = _f64(b), _f64(a), _f64(zi)
    nsec = b.shape[0] if form == 0 else 1
    order = 2 if form == 0 else b.size - 1
    out = C.c_int64(0)
    check(load().figh_filtfilt_cols(d_X.ptr, rows, cols, ldx, nblocks, form, b.ctypes.data_as(_c_double_p),
                                    a.ctypes.data_as(_c_double_p), nsec, order, zi.ctypes.data_as(_c_double_p), padlen, q,
                                    d_Y.ptr, ldy, C.byref(out)))
    return out.value


def compact_rows(W_ptr, rows, cols, ldw, tau_ptr, key_col, threshold, Wout_ptr, ld_out, tauout_ptr):
    """figh_compact_rows on raw device addresses (views into larger buffers); returns the number of kept rows."""
    out = C.c_int64(0)
    check(load().figh_compact_rows(W_ptr, rows, cols, ldw, tau_ptr, key_col, threshold, Wout_ptr, ld_out, tauout_ptr,
                                   C.byref(out)))
    return out.value


def base_permutation(d_R, nc, n, tol_qr, d_perm):
    check(load().figh_base_permutation(d_R.ptr, nc, n, tol_qr, d_perm.ptr))


def tsqr_merge(d_Rs, count, nc, d_R):
    check(load().figh_tsqr_merge(d_Rs.ptr, count, nc, d_R.ptr))


def select_columns(d_colsq, ncols, tol_e, link_stride, d_sel):
    check(load().figh_select_columns(d_colsq.ptr, ncols, tol_e, link_stride, d_sel.ptr))


def tsqr_selected(d_W, rows, ldw, d_colsq, ncols, tol_e, link_stride, nblocks, n_expected, d_tau, tol_qr, d_sel, d_R):
    """Elimination + TSQR of the kept columns back to back on the device (``n_expected`` <= 0: selection only);
    ``tol_qr`` >= 0 folds the rank decision and the regrouped factorisation into the last merge level (figh.h)."""
    check(load().figh_tsqr_selected(d_W.ptr, rows, ldw, d_colsq.ptr, ncols, tol_e, link_stride, nblocks, n_expected,
                                    d_tau.ptr if d_tau is not None else None, tol_qr, d_sel.ptr,
                                    d_R.ptr if d_R is not None else None))


def tsqr_selected_wrench(d_W, rows, ldw, d_colsq, ncols, tol_e, link_stride, n_expected, nf_expected, d_tau, tol_qr, d_sel,
                         d_R, d_link_pos=None, ld_force=0):
    """tsqr_selected for the external-wrench regressor of a free-flyer model: force rows over the ``nf_expected`` kept
    columns that can be non-zero there, torque rows chained onto their triangle (figh.h).  ``d_link_pos``: the device copy
    of :func:`regressor_link_layout`'s map when W is link-compact."""
    check(load().figh_tsqr_selected_wrench(d_W.ptr, rows, ldw, d_colsq.ptr, ncols, tol_e, link_stride, n_expected,
                                           nf_expected, d_tau.ptr if d_tau is not None else None, tol_qr, d_sel.ptr,
                                           d_R.ptr if d_R is not None else None,
                                           d_link_pos.ptr if d_link_pos is not None else None, int(ld_force)))


def regressor_force_layout(model, mode, flags, ft_mask):
    """Leading dimension of the force region of the force-compact layout (FLAG_FORCE_COMPACT, figh.h), 0 when it does not
    apply."""
    ldf = C.c_int64(0)
    rc = load().figh_regressor_force_layout(model.handle, mode, flags, ft_mask, C.byref(ldf))
    if rc == ERR_UNSUPPORTED:
        return 0
    check(rc)
    return int(ldf.value)


def regressor_link_layout(model, mode, flags, ft_mask):
    """(link_pos, nlive) of the link-compact layout (FLAG_LINK_COMPACT, figh.h) or None when it does not apply."""
    nlinks = model.njoints - 1
    pos = np.zeros(nlinks, dtype=np.int32)
    nlive = C.c_int(0)
    rc = load().figh_regressor_link_layout(model.handle, mode, flags, ft_mask, pos.ctypes.data_as(_c_int32_p), C.byref(nlive))
    if rc == ERR_UNSUPPORTED:
        return None
    check(rc)
    return pos, int(nlive.value)


def tsqr_selected_blocks(d_W, rows, ldw, d_colsq, ncols, tol_e, link_stride, n_expected, counts, d_cols, d_pos, d_tau,
                         tol_qr, d_sel, d_R, block_off=None, block_ld=None, d_block_tri=None):
    """tsqr_selected with one column list per row block (joint-torque regressor of a tree, figh.h); ``counts``: int32
    host array, one entry per row block; ``block_off`` / ``block_ld``: the block-compact W (element offsets, int64, and
    leading dimensions, int32, per row block); ``d_block_tri`` (optional) receives the compact stack of the embedded
    per-row-block triangles (figh.h)."""
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    off = ld = None
    if block_off is not None:
        off = np.ascontiguousarray(block_off, dtype=np.int64)
        ld = np.ascontiguousarray(block_ld, dtype=np.int32)
    check(load().figh_tsqr_selected_blocks(d_W.ptr, rows, ldw, d_colsq.ptr, ncols, tol_e, link_stride, n_expected,
                                           len(counts), counts.ctypes.data, d_cols.ptr, d_pos.ptr,
                                           off.ctypes.data if off is not None else None,
                                           ld.ctypes.data if ld is not None else None,
                                           d_tau.ptr if d_tau is not None else None, tol_qr, d_sel.ptr, d_R.ptr,
                                           d_block_tri.ptr if d_block_tri is not None else None))


def block_rows_residuals(d_rows, row_off, nc, d_v, d_r2):
    """d_r2[b] = sum over rows [row_off[b], row_off[b+1]) of (row . v)^2 (figh.h); ``row_off``: nblocks + 1 ints."""
    ro = _i32(row_off)
    check(load().figh_block_rows_residuals(d_rows.ptr, len(ro) - 1, ro.ctypes.data_as(_c_int32_p), nc, d_v.ptr, d_r2.ptr))


def regressor_tsqr_fused(model, flags, N, d_q, d_v, d_a, d_W, ldw, d_colsq, d_kept, n, d_tau, tol_qr, d_R):
    """K1 + level-0 TSQR over the caller's kept-column list in one launch (serial chains, figh.h).  Returns False -- nothing
    launched -- when the shape is not supported; raises on any other error."""
    rc = load().figh_regressor_tsqr_fused(model.handle, flags, N, d_q.ptr, d_v.ptr, d_a.ptr, d_W.ptr, ldw, d_colsq.ptr,
                                          d_kept.ptr, n, d_tau.ptr if d_tau is not None else None, tol_qr, d_R.ptr)
    if rc == ERR_UNSUPPORTED:
        return False
    check(rc)
    return True


def tsqr_merge_base(d_Rs, count, nc, n_free, tol_qr, d_Rk):
    check(load().figh_tsqr_merge_base(d_Rs.ptr, count, nc, n_free, tol_qr, d_Rk.ptr))


def repack_samples(d_src, N, width):
    """Tile-blocked copy of a sample-major N x width device array (figh_repack_samples): a new DeviceArray."""
    d_dst = DeviceArray((((N + 63) // 64) * 64 * width,), np.float64)
    check(load().figh_repack_samples(d_src.ptr, N, width, d_dst.ptr))
    return d_dst
