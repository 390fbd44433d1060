"""ORACLE (test infrastructure) -- ctypes binding of oracle/figh_oracle.c.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FIGH_ORACLE_ASAN=1 (with LD_PRELOAD=$(gcc -print-file-name=libasan.so)) runs the tests against the
# AddressSanitizer + UBSan build (`make -C oracle asan`): sanitizers are a CPU-build matter, the GPU pool has none.
_ASAN = os.environ.get("FIGH_ORACLE_ASAN") == "1"
_SO = os.path.join(_HERE, "libfigh_oracle_asan.so" if _ASAN else "libfigh_oracle.so")


class _Model(C.Structure):
    _fields_ = [("njoints", C.c_int), ("nq", C.c_int), ("nv", C.c_int),
                ("parents", C.POINTER(C.c_int)), ("jtype", C.POINTER(C.c_int)),
                ("idx_q", C.POINTER(C.c_int)), ("idx_v", C.POINTER(C.c_int)),
                ("axis", C.POINTER(C.c_double)), ("placement", C.POINTER(C.c_double)),
                ("gravity", C.POINTER(C.c_double))]


def build(force=False):
    src = os.path.join(_HERE, "figh_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "asan" if _ASAN else "libfigh_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.oracle_build_regressor_basic.restype = C.c_int
        _lib.oracle_householder_r.restype = C.c_int
    return _lib


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class OracleModel:
    def __init__(self, flat):
        self.flat = flat
        self._keep = {k: np.ascontiguousarray(flat[k], dtype=np.int32) for k in ("parents", "jtype", "idx_q", "idx_v")}
        self._keep.update({k: np.ascontiguousarray(flat[k], dtype=np.float64) for k in ("axis", "placement", "gravity")})
        k = self._keep
        self.c = _Model(int(flat["njoints"]), int(flat["nq"]), int(flat["nv"]), _ip(k["parents"]), _ip(k["jtype"]),
                        _ip(k["idx_q"]), _ip(k["idx_v"]), _dp(k["axis"]), _dp(k["placement"]), _dp(k["gravity"]))
        self.njoints, self.nq, self.nv = int(flat["njoints"]), int(flat["nq"]), int(flat["nv"])
        self.body_mask = np.ascontiguousarray(np.asarray(flat["mass"], dtype=float) != 0.0, dtype=np.int32)
        self._work = np.zeros(24 * self.njoints)

    def joint_torque_regressor(self, q, v, a, out=None):
        q, v, a = (np.ascontiguousarray(x, dtype=np.float64) for x in (q, v, a))
        Y = np.empty((self.nv, 10 * (self.njoints - 1))) if out is None else out
        lib().oracle_joint_torque_regressor(C.byref(self.c), _dp(q), _dp(v), _dp(a), _dp(Y), _dp(self._work))
        return Y

    def build_regressor_basic(self, q, v, a, mode, flags, ft_mask=63):
        q, v, a = (np.ascontiguousarray(x, dtype=np.float64) for x in (q, v, a))
        N = len(q)
        nl = self.njoints - 1
        rows = (self.nv if mode == 0 else 6) * N
        cols = 14 * nl + (3 if flags & 8 else 0)
        W = np.empty((rows, cols))
        rc = lib().oracle_build_regressor_basic(C.byref(self.c), mode, flags, ft_mask, _ip(self.body_mask),
                                                C.c_long(N), _dp(q), _dp(v), _dp(a), _dp(W), C.c_long(cols))
        if rc != 0:
            raise ValueError("oracle_build_regressor_basic failed (%d)" % rc)
        return W


def colsq(W):
    W = np.ascontiguousarray(W, dtype=np.float64)
    out = np.empty(W.shape[1])
    lib().oracle_colsq(_dp(W), C.c_long(W.shape[0]), W.shape[1], C.c_long(W.shape[1]), _dp(out))
    return out


def householder_r(W, col_idx=None, tau=None):
    W = np.ascontiguousarray(W, dtype=np.float64)
    col_idx = np.arange(W.shape[1], dtype=np.int32) if col_idx is None else np.ascontiguousarray(col_idx, dtype=np.int32)
    n = len(col_idx)
    R = np.empty((n, n))
    qtb = np.empty(n)
    t = None if tau is None else np.ascontiguousarray(tau, dtype=np.float64)
    rc = lib().oracle_householder_r(_dp(W), C.c_long(W.shape[0]), C.c_long(W.shape[1]), _ip(col_idx), n,
                                    _dp(t) if t is not None else None, _dp(R), _dp(qtb))
    if rc != 0:
        raise MemoryError("oracle_householder_r")
    return (R, qtb) if tau is not None else R


def param_flags(param, coupling=False):
    """param dict -> (mode, flags, ft_mask) of oracle_build_regressor_basic."""
    flags = (1 if param["has_friction"] else 0) | (2 if param["has_actuator_inertia"] else 0) | \
            (4 if param["has_joint_offset"] else 0) | (8 if coupling else 0)
    if param["is_joint_torques"]:
        return 0, flags, 63
    ft = 0
    for tok in param["force_torque"]:
        if tok == "All":
            ft |= 63
        else:
            ft |= 1 << {"Fx": 0, "Fy": 1, "Fz": 2, "Mx": 3, "My": 4, "Mz": 5}[tok]
    return 1, flags, ft
