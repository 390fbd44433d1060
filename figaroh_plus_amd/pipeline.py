"""Device-resident identification pass: (q, v, a[, tau]) in HBM -> idx_e, idx_base, beta, phi on the host.

This is the sequence every ``examples/*/identification.py`` of the reference runs
(SURVEY.md section 3.1-3.3)::

    W  = build_regressor_basic(...)            [+ add_coupling_TX40]      K1 (+ fused diag(W^T W))
    idx_e, params_r = get_index_eliminate(W, params_std, tol_e)           84..560 doubles to the host
    W_e = build_regressor_reduced(W, idx_e)                               never materialised: column gather
    _, params_base, idx_base = get_baseParams(W_e, params_r, params_std)  K3 TSQR over W[:, kept] (+ tau)
    phi_b = pinv(W_b) @ tau  /  double_QR                                 from the same triangle

kept in HBM between the steps, so one "step" of bench.py is exactly this function.  Multi-GPU:
each rank runs it on its own shard of samples; column norms are summed and the per-rank triangles
stacked (``exchange``), then reduced once more.
"""
import numpy as np

from . import _lib
from .device import GpuMatrix
from .tools import qrdecomposition as qrd
from .tools.regressor import build_regressor_device


class Exchange:
    """Single-process default: nothing to exchange."""

    world_size = 1
    rank = 0

    def sum_columns(self, d_colsq, ncols):
        return d_colsq.to_host()

    def stack_triangles(self, d_R, nc):
        return d_R, 1


class _View:
    """A window into a DeviceArray (no ownership): what the C-ABI wrappers need is ``.ptr``."""

    def __init__(self, base, byte_offset):
        self.ptr = base.ptr + int(byte_offset)


from ._host import single_threaded_blas as _single_threaded_blas


def _dtrtri(R1):
    from scipy.linalg.lapack import dtrtri
    return dtrtri(R1, lower=0)  # reads the upper triangle only


class IdentificationPipeline:
    def __init__(self, robot, param, params_std=None, coupling=False, tol_e=1e-6, tol_qr=qrd.TOL_QR, exchange=None,
                 chunk_samples=None):
        """``chunk_samples``: when the stacked regressor of all N samples does not fit HBM (human model at 1e7
        samples: 269 GB) the samples are processed in chunks of this size -- pass 1 accumulates diag(W^T W), pass 2
        rebuilds each chunk's W (recomputing is far cheaper than storing), factors it and stacks the triangles,
        which ``figh_tsqr_merge`` reduces.  Results are those of the one-shot pass (R is row-order independent)."""
        self.chunk_samples = chunk_samples
        self.robot, self.param, self.coupling = robot, param, coupling
        self.tol_e, self.tol_qr = tol_e, tol_qr
        self.params_std = params_std if params_std is not None else robot.get_standard_parameters(param)
        self.names = list(self.params_std.keys())
        self.exchange = exchange or Exchange()
        self.N = 0
        self.d_q = self.d_v = self.d_a = self.d_tau = None
        self.W = None

    # ------------------------------------------------------------------ inputs
    def set_samples(self, q, v, a, tau=None):
        """Upload this rank's samples (q: N x nq, v/a: N x nv) and optionally tau (rows of W,).

        Shapes are checked here, before anything reaches the device: the kernels index q, v, a and tau with N and the
        model's nq / nv, so a mismatched array would be read past its HBM allocation."""
        from .tools.regressor import _samples_to_device, regressor_flags
        mode, _, _ = regressor_flags(self.param, self.coupling)
        rps = self.robot.model.nv if mode == _lib.MODE_JOINT_TORQUE else 6  # figh_regressor_shape
        if tau is not None:
            tau = np.ascontiguousarray(tau, dtype=np.float64).reshape(-1)
            if tau.shape[0] != rps * len(q):
                raise ValueError("tau must have rows_per_sample * N = %d * %d = %d entries; got %d"
                                 % (rps, len(q), rps * len(q), tau.shape[0]))
        N, d_q, d_v, d_a = _samples_to_device(self.robot.model, q, v, a)  # raises ValueError on a shape mismatch
        d_tau = None if tau is None else _lib.DeviceArray.from_host(tau)
        self.N, self.d_q, self.d_v, self.d_a, self.d_tau = N, d_q, d_v, d_a, d_tau
        self.W = None

    def _chunks(self):
        c = self.chunk_samples
        return [(lo, min(lo + c, self.N)) for lo in range(0, self.N, c)]

    def _chunked(self):
        return bool(self.chunk_samples) and self.N > self.chunk_samples

    def _build_chunk(self, lo, hi, W, d_colsq):
        from .tools.regressor import regressor_flags
        mode, flags, ft_mask = regressor_flags(self.param, self.coupling)
        m = self.robot.model
        _lib.regressor_build(self.robot.device_model(), mode, flags, ft_mask, hi - lo, _View(self.d_q, lo * m.nq * 8),
                             _View(self.d_v, lo * m.nv * 8), _View(self.d_a, lo * m.nv * 8), W.buf, W.ld, d_colsq)

    def _tau_chunk(self, lo, hi, rps, d_out, scatter=False):
        """gather (or scatter back) the rows j*N + [lo, hi) of tau <-> the chunk's joint-major vector"""
        lib, nch = _lib.load(), hi - lo
        for j in range(rps):
            full, part = self.d_tau.ptr + (j * self.N + lo) * 8, d_out.ptr + j * nch * 8
            _lib.check(lib.figh_memcpy_d2d(full, part, nch * 8) if scatter else lib.figh_memcpy_d2d(part, full, nch * 8))

    def set_tau_from_parameters(self, phi, noise_std=0.0, seed=0):
        """Synthetic measurement tau = W phi + noise, built on the device (bench / tests)."""
        if self._chunked():
            from .tools.regressor import regressor_flags
            mode, flags, _ = regressor_flags(self.param, self.coupling)
            rps, ncols = self.robot.device_model().shape(mode, flags)
            d_phi = _lib.DeviceArray.from_host(np.ascontiguousarray(phi, dtype=np.float64))
            self.d_tau = _lib.DeviceArray((rps * self.N,), np.float64)
            Wc = GpuMatrix.empty(rps * self.chunk_samples, ncols)
            d_t = _lib.DeviceArray((rps * self.chunk_samples,), np.float64)
            for lo, hi in self._chunks():
                self._build_chunk(lo, hi, Wc, None)
                _lib.matvec(Wc.buf, rps * (hi - lo), Wc.ld, None, ncols, d_phi, d_t)
                self._tau_chunk(lo, hi, rps, d_t, scatter=True)
            if noise_std > 0.0:
                tau = self.d_tau.to_host()
                tau += np.random.default_rng(seed).standard_normal(tau.shape[0]) * noise_std
                self.d_tau = _lib.DeviceArray.from_host(tau)
            return self.d_tau
        W, _ = build_regressor_device(self.robot, self.d_q, self.d_v, self.d_a, self.N, self.param, self.coupling)
        d_phi = _lib.DeviceArray.from_host(np.ascontiguousarray(phi, dtype=np.float64))
        d_tau = _lib.DeviceArray((W.rows,), np.float64)
        _lib.matvec(W.buf, W.rows, W.ld, None, W.cols, d_phi, d_tau)
        if noise_std > 0.0:
            tau = d_tau.to_host()
            tau += np.random.default_rng(seed).standard_normal(tau.shape[0]) * noise_std
            d_tau = _lib.DeviceArray.from_host(tau)
        self.d_tau = d_tau
        W.buf.free()
        return d_tau

    # ------------------------------------------------------------------ one pass of the hot path
    def _run_chunked(self, strings):
        """Memory-bounded pass through the streamed C-ABI entry points (W exists one chunk at a time, in a library
        workspace): figh_regressor_colsq for the elimination, figh_regressor_tsqr for the triangle."""
        from .tools.regressor import regressor_flags
        ex, lib = self.exchange, _lib.load()
        mode, flags, ft_mask = regressor_flags(self.param, self.coupling)
        dm = self.robot.device_model()
        rps, ncols = dm.shape(mode, flags)
        if getattr(self, "_d_colsq", None) is None or self._d_colsq.size != ncols:
            cap = ncols + 1
            self._d_colsq = _lib.DeviceArray((ncols,), np.float64)
            self._d_idx = _lib.DeviceArray((cap,), np.int32)
            self._d_Rm = _lib.DeviceArray((cap * cap,), np.float64)
            self._d_Rp = _lib.DeviceArray((cap * cap,), np.float64)
            self._d_R2 = _lib.DeviceArray((cap * cap,), np.float64)
        # pass 1: column norms
        _lib.regressor_colsq(dm, mode, flags, ft_mask, self.N, self.d_q, self.d_v, self.d_a, self._d_colsq,
                             chunk_samples=self.chunk_samples)
        col_norm = ex.sum_columns(self._d_colsq, ncols)
        idx_e = [i for i in range(ncols) if col_norm[i] < self.tol_e]
        kept = [i for i in range(ncols) if not col_norm[i] < self.tol_e]
        params_r = [self.names[i] for i in kept]
        n = len(kept)
        with_tau = self.d_tau is not None
        nc = n + (1 if with_tau else 0)
        kept_i32 = np.asarray(kept, dtype=np.int32)
        _lib.check(lib.figh_memcpy_h2d(self._d_idx.ptr, kept_i32.ctypes.data, kept_i32.nbytes))
        # pass 2: every chunk rebuilt and factored, triangles merged
        d_R = self._d_Rm
        _lib.regressor_tsqr(dm, mode, flags, ft_mask, self.N, self.d_q, self.d_v, self.d_a, self._d_idx, n,
                            self.d_tau if with_tau else None, None, d_R, chunk_samples=self.chunk_samples)
        d_stack, count = ex.stack_triangles(d_R, nc)
        if count > 1:
            d_R = _lib.DeviceArray((nc * nc,), np.float64)
            _lib.tsqr_merge(d_stack, count, nc, d_R)
        return self._tail(d_R, n, nc, params_r, idx_e, col_norm, with_tau, rps * self.N * ex.world_size, strings)

    def _structure_hint(self, mode, kept_i32):
        """Joint-torque regressor of single-dof joints (regressor.py:45-87): the rows of joint j (row block j of N rows)
        only involve the links of j's subtree, which are numbered from j on -- the columns in front of 14 j are zeros
        written by K1.  Returns, per row block, how many kept columns lie in front (None when the structure is not
        guaranteed: external-wrench mode, multi-dof joints)."""
        m = self.robot.model
        if mode != _lib.MODE_JOINT_TORQUE or m.nv != m.njoints - 1 or self.N < 64:
            return None
        return np.searchsorted(kept_i32, 14 * np.arange(m.nv)).astype(np.int32)

    def run(self, strings=True):
        if self._chunked():
            return self._run_chunked(strings)
        ex = self.exchange
        # K1 (+ fused column norms)
        from .tools.regressor import regressor_flags
        mode, flags, ft_mask = regressor_flags(self.param, self.coupling)
        handle = self.robot.device_model()
        if self.W is None:  # HBM buffers are allocated once and reused by every step
            self._kept_cache = None
            rows_per_sample, ncols = handle.shape(mode, flags)
            # W stays in HBM.  Chains: the reference's dense layout (the chain kernel streams one contiguous run per tile).
            # Trees: the link-padded layout of figh_regressor_build_padded -- 16 columns per link, every (row, link)
            # segment one 128-byte line -- which K1' writes at about twice the rate; the TSQR takes a column list anyway.
            self._padded = not (handle.is_chain() and mode == _lib.MODE_JOINT_TORQUE) and not self.coupling
            if self._padded:
                self.W = GpuMatrix.empty(rows_per_sample * self.N, 16 * (self.robot.model.njoints - 1))
                self.W.ref_cols = ncols
            else:
                self.W = GpuMatrix.empty(rows_per_sample * self.N, ncols)
                self.W.ref_cols = ncols
            cap = ncols + 1
            self._d_colsq = _lib.DeviceArray((ncols,), np.float64)
            self._d_idx = _lib.DeviceArray((cap,), np.int32)
            self._d_R = _lib.DeviceArray((cap * cap,), np.float64)
            self._d_Rm = _lib.DeviceArray((cap * cap,), np.float64)
            self._d_Rp = _lib.DeviceArray((cap * cap,), np.float64)
            self._d_R2 = _lib.DeviceArray((cap * cap,), np.float64)
        W, d_colsq, lib = self.W, self._d_colsq, _lib.load()
        if self._padded:
            _lib.regressor_build_padded(handle, mode, flags, ft_mask, self.N, self.d_q, self.d_v, self.d_a, W.buf, W.ld,
                                        d_colsq)
        else:
            _lib.regressor_build(handle, mode, flags, ft_mask, self.N, self.d_q, self.d_v, self.d_a, W.buf, W.ld, d_colsq)
        col_norm = ex.sum_columns(d_colsq, W.ref_cols)
        small = col_norm < self.tol_e  # regressor.py:271-277 (NaN compares False: kept, as in the reference's loop)
        d_R, d_idx = self._d_R, self._d_idx
        cached = getattr(self, "_kept_cache", None)
        if cached is not None and np.array_equal(cached[0], small):
            # the same columns as in the previous pass: the lists derived from the mask and the column list that is
            # already in HBM are reused (the decision itself is taken afresh from this pass's norms every time)
            _, idx_e, params_r, n, hint = cached
            idx_e, params_r = list(idx_e), list(params_r)  # (the caller owns what run() returns)
        else:
            idx_e = np.flatnonzero(small).tolist()
            kept_i32 = np.flatnonzero(~small).astype(np.int32)
            params_r = [self.names[i] for i in kept_i32.tolist()]
            n = len(params_r)
            dev_cols = (kept_i32 // 14) * 16 + kept_i32 % 14 if self._padded else kept_i32  # W's own column numbering
            _lib.check(lib.figh_memcpy_h2d(d_idx.ptr, dev_cols.ctypes.data, dev_cols.nbytes))
            hint = self._structure_hint(mode, kept_i32)
            self._kept_cache = (small.copy(), list(idx_e), list(params_r), n, hint)
        # K3: TSQR over the kept columns (+ tau), then the cross-rank stack
        with_tau = self.d_tau is not None
        nc = n + (1 if with_tau else 0)
        _lib.tsqr(W.buf, W.rows, W.ld, d_idx, n, self.d_tau, None, d_R, first_cols=hint)
        d_stack, count = ex.stack_triangles(d_R, nc)
        if count > 1:
            _lib.tsqr_merge(d_stack, count, nc, self._d_Rm)
            d_R = self._d_Rm
        return self._tail(d_R, n, nc, params_r, idx_e, col_norm, with_tau, W.rows * ex.world_size, strings)

    def _tail(self, d_R, n, nc, params_r, idx_e, col_norm, with_tau, total_rows, strings):
        lib = _lib.load()
        # tail on the n x n triangle (qrdecomposition.py:215-266).  The rank decision |R_ii| > tol and the regrouped
        # order [base | rest | tau] are formed on the device (figh_base_permutation) and the regrouped factorisation
        # qr(R[:, perm]) goes through the TSQR kernel again (column gather in the kernel, one wavefront), so the whole
        # tail needs ONE host round trip: R, the regrouped triangle and the permutation come back in one copy.
        words = 2 * nc * nc + (nc + 1) // 2
        pack = getattr(self, "_d_pack", None)
        if pack is None or pack.size < words:
            pack = self._d_pack = _lib.DeviceArray((words,), np.float64)
        _lib.check(lib.figh_memcpy_d2d(pack.ptr, d_R.ptr, nc * nc * 8))
        d_perm = _View(pack, 2 * nc * nc * 8)
        _lib.base_permutation(d_R, nc, n, self.tol_qr, d_perm)
        _lib.tsqr(d_R, nc, nc, d_perm, nc, None, None, _View(pack, nc * nc * 8))
        host = np.empty(words)
        _lib.check(lib.figh_memcpy_d2h(host.ctypes.data, pack.ptr, host.nbytes))
        R = host[:nc * nc].reshape(nc, nc)          # upper triangular: the kernels write the zeros
        R_r = host[nc * nc:2 * nc * nc].reshape(nc, nc)
        perm = host[2 * nc * nc:].view(np.int32)[:nc]
        r = int(np.count_nonzero(np.abs(np.diag(R)[:n]) > self.tol_qr))
        idx_base, idx_regroup = perm[:r].tolist(), perm[r:n].tolist()
        assert len(params_r) == n, "params_r does not have same length with R"
        r = len(idx_base)
        R1, R2, z = R_r[:r, :r], R_r[:r, r:n], (R_r[:r, n] if with_tau else None)
        # inv(R1) of qrdecomposition.py:244 by LAPACK's triangular inverse (dtrtri): R1 is upper triangular, and
        # np.linalg.inv's general LU path pays ~30 us of BLAS thread start-up per call on a many-core host
        with _single_threaded_blas(n):
            R1_inv, info = _dtrtri(R1)
            if info != 0:
                raise np.linalg.LinAlgError("Singular matrix")
            beta = np.around(R1_inv @ R2, 6)
            phi_ls = R1_inv @ z if with_tau else None
        out = {
            "idx_e": idx_e, "params_r": params_r, "idx_base": idx_base, "beta": beta,
            "col_norm": col_norm, "absdiagR": np.abs(np.diag(R)[:n]), "rows": total_rows,
        }
        if strings:
            out["params_base"] = qrd._expressions([params_r[i] for i in idx_base],
                                                  [params_r[i] for i in idx_regroup], beta)
        if with_tau:
            out["phi_b"] = np.round(phi_ls, 6)
            out["phi_ls"] = phi_ls
            out["residual_norm"] = abs(R[n, n])
        return out
