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
    """Single-process default: nothing to exchange.  Also the BASE CLASS for user-supplied exchanges: a subclass overrides
    ``sum_columns`` and ``stack_triangles`` (and sets ``world_size`` / ``rank``); ``sum_columns_device`` and ``collective``
    have working defaults built on those."""

    world_size = 1
    rank = 0

    @property
    def collective(self):
        """False: run() folds the rank decision into the local merge tree and bypasses the exchanges below."""
        return self.world_size > 1

    def sum_columns(self, d_colsq, ncols):
        return d_colsq.to_host()

    def sum_columns_device(self, d_colsq, ncols):
        """In place: d_colsq holds the sum over the ranks afterwards.  Default: through ``sum_columns`` and one copy back
        (an exchange written against the two-method interface keeps working); nothing to do for one rank."""
        if self.world_size > 1:
            total = np.ascontiguousarray(self.sum_columns(d_colsq, ncols), dtype=np.float64)
            _lib.check(_lib.load().figh_memcpy_h2d(d_colsq.ptr, total.ctypes.data, total.nbytes))

    def stack_triangles(self, d_R, nc):
        return d_R, 1


class _View:
    """A window into a DeviceArray (no ownership): what the C-ABI wrappers need is ``.ptr``."""

    def __init__(self, base, byte_offset):
        self.ptr = base.ptr + int(byte_offset)


from ._host import relative_percent as _relative_percent
from ._host import single_threaded_blas as _single_threaded_blas


def _dtrtri(R1):
    from scipy.linalg.lapack import dtrtri
    return dtrtri(R1, lower=0)  # reads the upper triangle only


_dtrsm = None


def _solve_upper(R1, B):
    """R1^-1 B by one BLAS triangular solve (dtrsm reads the upper triangle only).  Raises like np.linalg.inv on an exactly
    zero pivot."""
    global _dtrsm
    if _dtrsm is None:
        from scipy.linalg.blas import dtrsm
        _dtrsm = dtrsm
    if R1.shape[0] and not np.diag(R1).all():
        raise np.linalg.LinAlgError("Singular matrix")
    return _dtrsm(1.0, R1, B, side=0, lower=0, trans_a=0, diag=0)


class IdentificationPipeline:
    def __init__(self, robot, param, params_std=None, coupling=False, tol_e=1e-6, tol_qr=qrd.TOL_QR, exchange=None,
                 chunk_samples=None, placement_trials=1, structural_zeros="every-pass", w_layout="dense", fuse=True, null_pivots=True,
                 row_blocks=None):
        """``chunk_samples``: when the stacked regressor of all N samples does not fit HBM (human model at 1e7
        samples: 269 GB) the samples are processed in chunks of this size -- pass 1 accumulates diag(W^T W), pass 2
        rebuilds each chunk's W (recomputing is far cheaper than storing), factors it and stacks the triangles,
        which ``figh_tsqr_merge`` reduces.  Results are those of the one-shot pass (R is row-order independent)."""
        self.chunk_samples = chunk_samples
        # row_blocks (ACTIVE JOINTS, examples/tiago/identification.py:148-187, :406-424): the dof indices (``act_idxv``) whose
        # row blocks carry measurements.  As in the script, the columns are eliminated on the norms of the FULL regressor
        # (every row block is walked for diag(W^T W)), but only the listed blocks are stored, and only they -- with their tau
        # -- are factored.  ``tau`` of set_samples then has len(row_blocks) * N entries, block i belonging to dof
        # row_blocks[i] (the script's tau[:, i]); out["rows"], sigma2_joint follow the list.  Joint-torque regressors of trees
        # of single-dof joints (the per-row-block TSQR), W resident.
        self.row_blocks = None if row_blocks is None else [int(b) for b in row_blocks]
        self._own_handle = None
        # fuse (serial chains, W in the reference layout): from the second pass on K1 and the level-0 TSQR run as ONE launch
        # (figh_regressor_tsqr_fused) over the kept-column list of the previous pass -- every 64-row tile of W is factored
        # while it is still in LDS, W is written but not read back.  The list is verified against the norms the pass produces;
        # a pass whose kept set changed falls back to the two launches below (and learns the new list).
        self.fuse = self._fuse_requested = bool(fuse)
        # null_pivots: the TSQR launches of a pass run under the null-pivot rule (include/figh.h) and the pass is certified
        # afterwards; a pass that cannot be certified is repeated without the rule, which then stays off until the next
        # set_samples (null_rule_fallbacks counts those)
        self.null_pivots = self._null_pivots_requested = bool(null_pivots)
        self.null_rule_fallbacks = 0
        self._cert_cache = None
        self._fused_kept = None
        self.fused_passes = 0
        # structural_zeros = "once" (opt-in, joint-torque regressor of a tree kept in HBM): W is zero-filled when it is
        # allocated and the regressor kernel is told that the structural zeros are present (FIGH_FLAG_ZEROS_PRESENT) -- it
        # rewrites every entry that depends on q, v, a in every pass and leaves the zeros alone.  The default re-creates
        # every byte of W in every pass.
        if structural_zeros not in ("every-pass", "once"):
            raise ValueError("structural_zeros must be 'every-pass' or 'once'")
        self.structural_zeros = structural_zeros
        # w_layout = "block-compact" (joint-torque regressor of a tree kept in HBM, FIGH_FLAG_COMPACT_BLOCKS): W is stored row
        # block by row block, block j as its own N x 16 |subtree_j| matrix -- the window of the row that can be non-zero.
        # Nothing else is written or read (the TSQR takes one column list per row block anyway); every stored byte is written
        # in every pass.  self.W is then that buffer (rows = nv N, `compact` = (offsets, leading dimensions)), not the
        # reference's matrix: build_regressor_basic still returns that.
        # "link-padded": trees in the plain link-padded layout even where the link-compact one applies (tests, A/B);
        # "link-compact": the external-wrench regressor link-compact but with the force rows as wide as the torque rows (no
        # force-compact region: what run(wls=True) switches to, its second pass reads W as one matrix).
        if w_layout not in ("dense", "block-compact", "link-padded", "link-compact"):
            raise ValueError("w_layout must be 'dense', 'block-compact', 'link-padded' or 'link-compact'")
        self.w_layout = w_layout
        # placement_trials > 1: when W is allocated, that many candidate buffers are allocated side by side, the regressor
        # kernel is timed on each and the fastest one is kept (set-up cost: a few passes of K1).  The time K1 needs for the
        # same 4 GB depends on the physical pages behind them -- 0.68 or 0.82 ms per allocation, hipMemset moves with it
        # (0.60 / 0.63 ms), measured with tools/k1_alloc_probe.py -- and the allocator offers no handle on that.
        self.placement_trials = max(1, int(placement_trials))
        self.placement_report = None
        self.robot, self.param, self.coupling = robot, param, coupling
        self.tol_e, self.tol_qr = tol_e, tol_qr
        self.params_std = params_std if params_std is not None else robot.get_standard_parameters(param)
        self.names = list(self.params_std.keys())
        self.exchange = exchange or Exchange()
        self.N = 0
        self.d_q = self.d_v = self.d_a = self.d_tau = None
        self.W = None

    def _handle(self):
        """The device model of this pipeline: the robot's shared handle, or -- with ``row_blocks`` -- a private one that
        stores the active row blocks only (figh_model_set_active_rows changes what a handle writes)."""
        if self.row_blocks is None:
            return self.robot.device_model()
        if self._own_handle is None:
            m = self.robot.model
            if sorted(set(self.row_blocks)) != sorted(self.row_blocks) or not all(0 <= b < m.nv for b in self.row_blocks):
                raise ValueError("row_blocks must be distinct dof indices in [0, %d)" % m.nv)
            self._own_handle = _lib.ModelHandle(m.to_flat())
            self._own_handle.set_active_rows(self.row_blocks)
        return self._own_handle

    # ------------------------------------------------------------------ inputs
    def set_samples(self, q, v, a, tau=None):
        """Upload this rank's samples (q: N x nq, v/a: N x nv) and optionally tau (rows of W,).

        Shapes are checked here, before anything reaches the device: the kernels index q, v, a and tau with N and the
        model's nq / nv, so a mismatched array would be read past its HBM allocation."""
        from .tools.regressor import _samples_to_device, regressor_flags
        mode, _, _ = regressor_flags(self.param, self.coupling)
        rps = self.robot.model.nv if mode == _lib.MODE_JOINT_TORQUE else 6  # figh_regressor_shape
        if self.row_blocks is not None:
            if mode != _lib.MODE_JOINT_TORQUE or self.coupling or self.chunk_samples:
                raise NotImplementedError("row_blocks: joint-torque regressor of a tree, W resident (no chunk_samples)")
            rps_in = len(self.row_blocks)
        else:
            rps_in = rps
        if tau is not None:
            tau = np.ascontiguousarray(tau, dtype=np.float64).reshape(-1)
            if tau.shape[0] != rps_in * len(q):
                raise ValueError("tau must have rows_per_sample * N = %d * %d = %d entries; got %d"
                                 % (rps_in, len(q), rps_in * len(q), tau.shape[0]))
            if self.row_blocks is not None:  # block i of the caller's tau belongs to dof row_blocks[i]
                full = np.zeros(rps * len(q))
                for i, b in enumerate(self.row_blocks):
                    full[b * len(q):(b + 1) * len(q)] = tau[i * len(q):(i + 1) * len(q)]
                tau = full
        N, d_q, d_v, d_a = _samples_to_device(self.robot.model, q, v, a)  # raises ValueError on a shape mismatch
        d_tau = None if tau is None else _lib.DeviceArray.from_host(tau)
        # Tree models: the generic regressor kernel takes one sample per lane, and in the reference's sample-major arrays a
        # lane's values lie nq * 8 bytes from its neighbour's.  The three arrays are re-laid once per tile of 64 samples,
        # value-major (figh_repack_samples); K1' then reads one 512-byte line per value.  Chains keep the original arrays
        # (their kernel reads 48-byte runs per lane).
        self._in_flags, self.repack_ms = 0, 0.0
        m = self.robot.model
        if N > 0 and not (self._handle().is_chain() and mode == _lib.MODE_JOINT_TORQUE) and not self.coupling:
            import time
            _lib.synchronize()
            t0 = time.perf_counter()
            blocked = [_lib.repack_samples(d, N, w) for d, w in ((d_q, m.nq), (d_v, m.nv), (d_a, m.nv))]
            _lib.synchronize()
            self.repack_ms = 1e3 * (time.perf_counter() - t0)
            for d in (d_q, d_v, d_a):
                d.free()
            d_q, d_v, d_a = blocked
            self._in_flags = _lib.FLAG_BLOCKED_INPUTS
            if self.chunk_samples:
                self.chunk_samples = max(64, (int(self.chunk_samples) // 64) * 64)  # chunks start on tile boundaries
        self.N, self.d_q, self.d_v, self.d_a, self.d_tau = N, d_q, d_v, d_a, d_tau
        self.W = None
        # Fused launch: eligibility is decided again for every sample set (a set that was too small to fuse says nothing
        # about the next one), and under a collective exchange from a value all ranks agree on -- the fused launch needs at
        # least 4096 local samples, and ranks whose shards straddle that size must not take different paths through the
        # pass: one rank that cannot fuse keeps every rank on the two-launch path.
        # The same holds for the force-compact layout of the external-wrench regressor: its consumer,
        # figh_tsqr_selected_wrench, refuses a shard with fewer than 16 nc / 3 samples (three row blocks of N rows each must
        # hold nc rows of 16-row tiles), and a rank-local fallback would re-run K1 and its collectives on that rank alone.
        # Whether ANY rank's shard is too small for it is agreed here, once per sample set.
        self.fuse = self._fuse_requested
        self.null_pivots = self._null_pivots_requested  # (a new sample set gets the rule -- and its certificate -- again)
        self._cert_cache = None
        ncols_ref = self._handle().shape(mode, self._flags()[1])[1] if N > 0 else 0
        self._force_compact_ok = N >= 64 and 3 * N >= 16 * (ncols_ref + 1)
        ex = self.exchange
        if getattr(ex, "collective", False) and ex.world_size > 1:
            flag = _lib.DeviceArray.from_host(np.array([0.0 if N >= 4096 else 1.0, 0.0 if self._force_compact_ok else 1.0]))
            agreed = np.asarray(ex.sum_columns(flag, 2)).reshape(-1)
            self.fuse = self.fuse and float(agreed[0]) == 0.0
            self._force_compact_ok = float(agreed[1]) == 0.0
            flag.free()

    def _flags(self):
        """(mode, flags, ft_mask) of this pipeline's regressor calls, including the layout of its input arrays."""
        from .tools.regressor import regressor_flags
        mode, flags, ft_mask = regressor_flags(self.param, self.coupling)
        return mode, flags | getattr(self, "_in_flags", 0), ft_mask

    def _chunks(self):
        c = self.chunk_samples
        return [(lo, min(lo + c, self.N)) for lo in range(0, self.N, c)]

    def _chunked(self):
        return bool(self.chunk_samples) and self.N > self.chunk_samples

    def _build_chunk(self, lo, hi, W, d_colsq):
        mode, flags, ft_mask = self._flags()
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
            mode, flags, _ = self._flags()
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
        W, _ = build_regressor_device(self.robot, self.d_q, self.d_v, self.d_a, self.N, self.param, self.coupling,
                                      extra_flags=self._in_flags)
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
        ex, lib = self.exchange, _lib.load()
        mode, flags, ft_mask = self._flags()
        dm = self.robot.device_model()
        rps, ncols = dm.shape(mode, flags)
        if getattr(self, "_dc_colsq", None) is None or self._dc_colsq.size != ncols:
            cap = ncols + 1
            self._dc_colsq = _lib.DeviceArray((ncols,), np.float64)
            self._dc_idx = _lib.DeviceArray((cap,), np.int32)
            self._dc_R = _lib.DeviceArray((cap * cap,), np.float64)
            self._chunk_kept = None  # (the cached column list lived in the old index buffer)
        with_tau = self.d_tau is not None
        d_R = self._dc_R
        # One pass over the samples when the kept set of the previous pass is known: the chunks are factored over THAT column
        # list while diag(W^T W) of all columns is accumulated by the same regressor launches (figh_regressor_tsqr_norms), and
        # the set is verified against the norms afterwards -- the same speculation as run() makes with the column count.
        # (The norms decide on every rank from the all-reduced sums, so a mismatch sends all ranks to the two-pass form.)
        cached = getattr(self, "_chunk_kept", None)
        if cached is not None and cached[0] == ncols:
            kept = cached[1]
            n = len(kept)
            _lib.regressor_tsqr(dm, mode, flags, ft_mask, self.N, self.d_q, self.d_v, self.d_a, self._dc_idx, n,
                                self.d_tau if with_tau else None, None, d_R, chunk_samples=self.chunk_samples,
                                d_colsq=self._dc_colsq)
            col_norm = ex.sum_columns(self._dc_colsq, ncols)
            if np.array_equal(np.flatnonzero(~(col_norm < self.tol_e)), cached[2]):
                nc = n + (1 if with_tau else 0)
                idx_e, params_r = list(cached[3]), list(cached[4])
                d_stack, count = ex.stack_triangles(d_R, nc)
                return self._tail(d_stack, count, n, nc, params_r, idx_e, col_norm, with_tau,
                                  rps * self.N * ex.world_size, strings)
            self._chunk_kept = None  # the kept set changed: two passes, below
        # pass 1: column norms
        _lib.regressor_colsq(dm, mode, flags, ft_mask, self.N, self.d_q, self.d_v, self.d_a, self._dc_colsq,
                             chunk_samples=self.chunk_samples)
        col_norm = ex.sum_columns(self._dc_colsq, ncols)
        idx_e = [i for i in range(ncols) if col_norm[i] < self.tol_e]
        kept = [i for i in range(ncols) if not col_norm[i] < self.tol_e]
        params_r = [self.names[i] for i in kept]
        n = len(kept)
        nc = n + (1 if with_tau else 0)
        kept_i32 = np.asarray(kept, dtype=np.int32)
        _lib.check(lib.figh_memcpy_h2d(self._dc_idx.ptr, kept_i32.ctypes.data, kept_i32.nbytes))
        self._chunk_kept = (ncols, kept, np.asarray(kept, dtype=np.int64), list(idx_e), list(params_r))
        # pass 2: every chunk rebuilt and factored, triangles merged
        _lib.regressor_tsqr(dm, mode, flags, ft_mask, self.N, self.d_q, self.d_v, self.d_a, self._dc_idx, n,
                            self.d_tau if with_tau else None, None, d_R, chunk_samples=self.chunk_samples)
        d_stack, count = ex.stack_triangles(d_R, nc)
        return self._tail(d_stack, count, n, nc, params_r, idx_e, col_norm, with_tau, rps * self.N * ex.world_size, strings)

    def run(self, strings=True, wls=False):
        """One pass.  ``wls=True`` appends the weighted least squares of the example scripts
        (examples/staubli_TX40/identification.py:305-346: per-joint variances from the OLS residuals, rows scaled by
        1 / sigma_j) on the data that is already in HBM: ``phi_wls``, ``std_wls`` (%), ``sigma2_joint``.

        One pass.  Nothing returns to the host before the end: K1 leaves diag(W^T W) in HBM, the kept-column list is
        formed there (figh_tsqr_selected), the TSQR is launched with the column COUNT of the previous pass, the merge tree
        ends in the rank-revealing level, and one copy brings back [column norms | selection | triangle rows].  The count
        is verified against the device's own afterwards; the first pass (count unknown) and a pass whose count changed
        repeat the solve with the right one."""
        # (null_pivots: the dependent columns of the regressor -- |R_kk| <= tol_qr by a margin of 64 -- cost a norm per tile
        # instead of a column step in every TSQR launch of the pass, include/figh.h: figh_tsqr_null_pivot_tol.  Round 6: a pass
        # that ran under the rule is CERTIFIED afterwards (_finish, _host.null_rule_certified): when its classification is not
        # provably plain Householder's -- a pivot close to tol_qr, or regrouping coefficients large enough for the folded
        # tol_qr / 64 to matter -- the pass is repeated without the rule and the rule stays off for this pipeline.)
        for _ in range(2):
            fused_before = self.fused_passes
            with _lib.null_pivots(self.tol_qr if self.null_pivots else None):
                out = self._run_once(strings, wls)
            if self.null_pivots and not out.get("null_rule_certified", True):
                self.null_pivots = False
                self.fused_passes = fused_before  # (counts the passes whose results were returned, not the discarded attempt)
                self.null_rule_fallbacks = getattr(self, "null_rule_fallbacks", 0) + 1
                # (kept for diagnostics, tools/cert_report.py: the bounds and the pivots of the pass that was not certified)
                self._cert_failed = (self._cert_cache, out["absdiagR"].copy(), list(out["idx_base"]))
                self._cert_cache = None
                continue
            return out
        return out

    def _run_once(self, strings, wls):
        if self._chunked():
            if wls:
                raise NotImplementedError("wls=True needs the regressor resident in HBM (no chunk_samples)")
            return self._run_chunked(strings)
        if wls and not getattr(self, "_no_force_compact", False):
            # the weighted solve's second pass reads W as ONE matrix: from now on without the force-compact region
            self._no_force_compact = True
            if getattr(self, "_force_ld", 0) and self.W is not None:
                self.W.buf.free()
                self.W = None
        # (with the WLS the expression strings of the base parameters -- host work, 0.2 - 0.4 ms for TIAGo -- are built while
        # the weighted factorisation runs on the device)
        out = self._run_resident(strings and not wls, wls)
        if wls and out.get("null_rule_certified", True):
            self._wls(out, strings)
        return out

    def _run_resident(self, strings, wls):
        ex = self.exchange
        # K1 (+ fused column norms)
        mode, flags, ft_mask = self._flags()
        handle = self._handle()
        if self.W is None:  # HBM buffers are allocated once and reused by every step
            self._kept_cache = None
            self._n_expected = -1
            self._mask_expected = None
            rows_per_sample, ncols = handle.shape(mode, flags)
            # W stays in HBM.  Chains: the reference's dense layout (the chain kernel streams one contiguous run per tile).
            # Trees: the link-padded layout of figh_regressor_build_padded -- 16 columns per link, every (row, link)
            # segment one 128-byte line -- which K1' writes at about twice the rate; the TSQR takes a column list anyway.
            self._padded = not (handle.is_chain() and mode == _lib.MODE_JOINT_TORQUE) and not self.coupling
            wcols = 16 * (self.robot.model.njoints - 1) if self._padded else ncols
            m = self.robot.model
            self._compact = None
            # external wrench on a free-flyer root: links that cannot have a non-zero entry in any of the six row blocks
            # (massless bodies, regressor.py:36-39; human model: 21 of 40) get no columns in the resident W -- their columns are
            # eliminated whatever the samples (figh.h, FIGH_FLAG_LINK_COMPACT: 304 instead of 640 columns, which is what lets
            # 1e7 human samples stay resident)
            self._link_pos = self._d_link_pos = None
            if self._padded and self.w_layout != "link-padded":
                layout = _lib.regressor_link_layout(handle, mode, flags & 7, ft_mask)
                if layout is not None and 0 < layout[1] < m.njoints - 1:
                    self._link_pos = layout[0].astype(np.int64)
                    self._d_link_pos = _lib.DeviceArray.from_host(layout[0])
                    wcols = 16 * layout[1]
            # ... and the three FORCE row blocks in a region of their own, one 128-byte line per four links (a force row only
            # has mx my mz m of every link: FIGH_FLAG_FORCE_COMPACT) -- 5/8 of the bytes of W, and the force rows' TSQR reads
            # lines that are all payload.  Not with friction / inertia / offset columns; not for the weighted solve, whose second
            # pass reads W as one matrix (run(wls=True) re-creates W without it).
            self._force_ld = 0
            if (self._padded and self.w_layout in ("dense", "block-compact") and not getattr(self, "_no_force_compact", False)
                    and getattr(self, "_force_compact_ok", self.N >= 64)):
                self._force_ld = _lib.regressor_force_layout(handle, mode, flags & 7, ft_mask)
            if (self.w_layout == "block-compact" and self._padded and mode == _lib.MODE_JOINT_TORQUE
                    and m.nv == m.njoints - 1 and self.N >= 64):
                sizes = self._subtree_sizes()
                for k in range(m.nv):  # depth-first numbering: the subtree of joint j + 1 is the links j .. j + size - 1
                    jid = k + 1
                    while jid > 0:
                        if not (jid - 1 <= k < jid - 1 + sizes[jid - 1]):
                            raise RuntimeError("block-compact W needs depth-first joint numbering")
                        jid = m.parents[jid]
                ld = 16 * sizes
                if self.row_blocks is not None:  # blocks without measurements are not stored at all
                    keep = np.zeros(m.nv, dtype=bool)
                    keep[self.row_blocks] = True
                    ld = np.where(keep, ld, 0)
                off = self.N * np.concatenate([[0], np.cumsum(ld)[:-1]])
                self._compact = (off.astype(np.int64), ld.astype(np.int32))
                self.W = GpuMatrix(_lib.DeviceArray((int(self.N * ld.sum()),), np.float64), rows_per_sample * self.N, wcols,
                                   wcols)
                self.W.compact = self._compact
            elif self._force_ld:
                half = (rows_per_sample // 2) * self.N  # rows of the force region = rows of the torque region
                self.W = GpuMatrix(_lib.DeviceArray((half * (self._force_ld + wcols),), np.float64), rows_per_sample * self.N,
                                   wcols, wcols)
                self.W.force_ld = self._force_ld  # (rows [0, half): force region, ld force_ld; torque rows behind it, ld wcols)
            else:
                self.W = self._place_W(rows_per_sample * self.N, wcols, handle, mode, flags, ft_mask)
            self.W.ref_cols = ncols
            self._zeros_once = (self.structural_zeros == "once" and self._padded and mode == _lib.MODE_JOINT_TORQUE
                                and self._compact is None)
            if self._zeros_once:
                _lib.check(_lib.load().figh_memset(self.W.buf.ptr, 0, self.W.rows * self.W.ld * 8))
            cap = ncols + 1
            # one buffer for everything that returns to the host: [colsq (ncols f64) | sel (2 + 2 ncols i32) | rows ((cap+1) cap f64)]
            self._sel_words = (2 + 2 * ncols + 1) // 2
            self._d_pack = _lib.DeviceArray((ncols + self._sel_words + (cap + 1) * cap,), np.float64)
            self._d_colsq = _View(self._d_pack, 0)
            self._d_sel = _View(self._d_pack, ncols * 8)
            self._d_rows = _View(self._d_pack, (ncols + self._sel_words) * 8)
            self._d_R = _lib.DeviceArray((cap * cap,), np.float64)
            # joint-torque regressor of single-dof joints (regressor.py:45-87): the rows of joint j (row block j of N rows)
            # only involve the links of j's subtree, which are numbered from j on -- the columns in front of 14 j are
            # zeros written by K1; the device derives the per-tile form of that hint from its own column list
            m = self.robot.model
            structured = mode == _lib.MODE_JOINT_TORQUE and m.nv == m.njoints - 1 and self.N >= 64
            self._hint_blocks = m.nv if structured else 0
            # external wrench on a free-flyer root: six row blocks, and in the three force blocks the rotational-inertia
            # columns of every link are exact zeros -- figh_tsqr_selected_wrench factors those rows over the other
            # columns only (the count of such kept columns is derived from the kept mask, like the column count)
            from .model import JT_FREEFLYER
            self._wrench_split = (mode == _lib.MODE_EXT_WRENCH and rows_per_sample == 6 and len(m.joints) > 1 and
                                  m.joints[1].jtype == JT_FREEFLYER and not self.coupling)
            self._nf_expected = -1
            # joint-torque regressor of a tree of single-dof joints with more than 80 kept columns (TIAGo): per row block
            # only the columns of the joint's subtree are non-zero -- figh_tsqr_selected_blocks (lists from the kept mask)
            self._tree_blocks = bool(structured and self._padded)
            if self._compact is not None and not self._tree_blocks:
                raise RuntimeError("block-compact W needs the per-row-block TSQR")
            if self.row_blocks is not None and not self._tree_blocks:
                raise NotImplementedError("row_blocks needs the per-row-block TSQR: joint-torque regressor of a tree of "
                                          "single-dof joints with at least 64 samples")
            self._block_cache = None
        W, d_colsq, lib = self.W, self._d_colsq, _lib.load()
        if (self.fuse and self._fused_kept is None and not self._padded and self.N >= 4096
                and handle.is_chain() and mode == _lib.MODE_JOINT_TORQUE and not self.coupling):
            self._learn_kept_set(handle, mode, flags, ft_mask)
        if self.fuse and self._fused_kept is not None and not self._padded:
            out = self._run_fused(handle, flags, strings)
            if out is not None:
                return out
        if self._padded:
            _lib.regressor_build_padded(handle, mode, flags | (_lib.FLAG_ZEROS_PRESENT if self._zeros_once else 0)
                                        | (_lib.FLAG_COMPACT_BLOCKS if self._compact is not None else 0)
                                        | (_lib.FLAG_LINK_COMPACT if self._link_pos is not None else 0)
                                        | (_lib.FLAG_FORCE_COMPACT if self._force_ld else 0), ft_mask,
                                        self.N, self.d_q, self.d_v, self.d_a, W.buf, W.ld, d_colsq)
        else:
            _lib.regressor_build(handle, mode, flags, ft_mask, self.N, self.d_q, self.d_v, self.d_a, W.buf, W.ld, d_colsq)
        ex.sum_columns_device(d_colsq, W.ref_cols)
        ncols, with_tau = W.ref_cols, self.d_tau is not None
        stride = 16 if self._padded else 14
        split = getattr(self, "_wrench_split", False)
        for attempt in range(4):
            n = self._n_expected
            nf = self._nf_expected if split else 0
            nc = n + (1 if with_tau else 0)
            local = not getattr(ex, "collective", True)
            blocks = None
            if getattr(self, "_tree_blocks", False) and (nc > 80 or self._compact is not None or self.row_blocks is not None):
                blocks = self._block_lists(ncols, stride)
                if blocks is None and self.row_blocks is not None and n > 0:
                    raise RuntimeError("row_blocks: no kept mask to build the column lists from")
            if self._compact is not None and blocks is None and n > 0:
                raise RuntimeError("block-compact W: no kept mask to build the column lists from")
            if split or self._link_pos is not None or self._force_ld:  # (nf = 0: the plain pass, through the entry that knows the layout)
                try:
                    _lib.tsqr_selected_wrench(W.buf, W.rows, W.ld, d_colsq, ncols, self.tol_e, stride, n, nf, self.d_tau,
                                              self.tol_qr if local else -1.0, self._d_sel,
                                              self._d_rows if local else self._d_R, d_link_pos=self._d_link_pos,
                                              ld_force=self._force_ld)
                except _lib.FighError as e:
                    if not (self._force_ld and e.code == _lib.ERR_UNSUPPORTED):
                        raise
                    # the kept columns turned out not to allow the force / torque split (at most 80 of them -- a property of
                    # the all-reduced norms, the same on every rank; shards too short for it were ruled out for all ranks in
                    # set_samples): this regressor is kept as one matrix from now on
                    self._no_force_compact = True
                    self.W.buf.free()
                    self.W = None
                    return self._run_resident(strings, wls)
            elif blocks is not None:
                coff, cld = self._compact if self._compact is not None else (None, None)
                _lib.tsqr_selected_blocks(W.buf, W.rows, W.ld, d_colsq, ncols, self.tol_e, stride, n, blocks[1],
                                          blocks[4] if coff is not None else blocks[2], blocks[3], self.d_tau,
                                          self.tol_qr if local else -1.0, self._d_sel, self._d_rows if local else self._d_R,
                                          block_off=coff, block_ld=cld, d_block_tri=self._block_tri(wls, len(blocks[1]), nc))
            else:
                _lib.tsqr_selected(W.buf, W.rows, W.ld, d_colsq, ncols, self.tol_e, stride, self._hint_blocks, n, self.d_tau,
                                   self.tol_qr if local else -1.0, self._d_sel, self._d_rows if local else self._d_R)
            self._have_block_tri = bool(wls and blocks is not None and not split)
            if not local:
                if n > 0:
                    d_stack, count = ex.stack_triangles(self._d_R, nc)
                    _lib.tsqr_merge_base(d_stack, count, nc, n, self.tol_qr, self._d_rows)
            words = ncols + self._sel_words + ((nc + 1) * nc if n > 0 else 0)
            # one page-locked landing buffer for the pass's results, reused by every step (everything run() returns is
            # copied or derived from it before the next step overwrites it)
            pin = getattr(self, "_host_pack", None)
            if pin is None or pin.size < self._d_pack.size:
                pin = self._host_pack = _lib.PinnedArray(self._d_pack.size)
            host = pin.array[:words]
            _lib.check(lib.figh_memcpy_d2h(host.ctypes.data, self._d_pack.ptr, host.nbytes))
            sel = host[ncols:ncols + self._sel_words].view(np.int32)
            nf_now = 0
            if split:  # kept columns that can be non-zero in force rows: slot >= 6 within the link
                nf_now = int(np.count_nonzero(sel[2 + ncols:2 + 2 * ncols][self._force_slots(ncols)]))
            mask_now = sel[2 + ncols:2 + 2 * ncols] != 0
            if int(sel[0]) == n and nf_now == nf and (blocks is None or np.array_equal(mask_now, blocks[0])):
                break
            self._mask_expected = mask_now.copy()  # (the per-block column lists are built from the mask)
            self._n_expected = int(sel[0])  # first pass, or the kept set changed size: solve again with the right shape
            self._nf_expected = nf_now
            if self._n_expected == 0:
                raise ValueError("every column of the regressor was eliminated")
        else:
            raise RuntimeError("the number of kept columns did not settle")
        col_norm = host[:ncols].copy()
        kept_mask = sel[2 + ncols:2 + 2 * ncols] != 0
        cached = self._kept_cache
        if cached is not None and np.array_equal(cached[0], kept_mask):
            _, idx_e, params_r = cached  # same columns as in the previous pass: the derived lists are reused
            idx_e, params_r = list(idx_e), list(params_r)  # (the caller owns what run() returns)
        else:
            idx_e = np.flatnonzero(~kept_mask).tolist()
            params_r = [self.names[i] for i in np.flatnonzero(kept_mask).tolist()]
            self._kept_cache = (kept_mask.copy(), list(idx_e), list(params_r))
        rows_k = host[ncols + self._sel_words:].reshape(nc + 1, nc)
        if self.fuse and not self._padded and self._compact is None and (
                self._fused_kept is None or not np.array_equal(self._fused_kept[0], kept_mask)):
            kept = np.flatnonzero(kept_mask).astype(np.int32)
            buf = getattr(self, "_d_kept_buf", None)
            if buf is None or buf.size < ncols:
                buf = self._d_kept_buf = _lib.DeviceArray((ncols,), np.int32)
            _lib.check(lib.figh_memcpy_h2d(buf.ptr, kept.ctypes.data, kept.nbytes))
            self._fused_kept = (kept_mask.copy(), buf, len(kept))
        rows_mine = W.rows if self.row_blocks is None else len(self.row_blocks) * self.N
        return self._finish(rows_k, n, nc, params_r, idx_e, col_norm, with_tau, rows_mine * ex.world_size, strings,
                            defer_beta=wls and with_tau)

    def forget(self):
        """Drop everything the pipeline has learnt from earlier passes (kept-column set, counts, per-block lists): the next
        ``run()`` is a FIRST pass again -- what a script that calls the reference's functions once pays
        (examples/ur10/identification.py:71-83) -- with the HBM buffers still allocated.  bench.py times it as
        ``ms_first_pass``."""
        self._fused_kept = None
        self._kept_cache = None
        self._n_expected = self._nf_expected = -1
        self._mask_expected = None
        self._block_cache = None
        self._chunk_kept = None
        self._cert_cache = None

    PREFIX_SAMPLES = 4096

    def _learn_kept_set(self, handle, mode, flags, ft_mask):
        """First pass of a serial chain: which columns get_index_eliminate (regressor.py:258-279) keeps is a property of the
        model -- the eliminated columns are structural zeros of the regressor (UR10: 35 of 84) -- so the set is learnt from the
        first 4096 samples (one K1 launch of a few microseconds into the head of W, which the pass overwrites; the norms are
        summed over the ranks like those of a full pass) and the full pass runs FUSED over it.  The fused pass verifies the
        set against the norms of all samples as always; should a column's norm only cross tol_e with more samples, the pass
        falls back to the two launches and learns the set from them."""
        ex, W = self.exchange, self.W
        n0 = min(self.N, self.PREFIX_SAMPLES)
        _lib.regressor_build(handle, mode, flags, ft_mask, n0, self.d_q, self.d_v, self.d_a, W.buf, W.ld, self._d_colsq)
        ex.sum_columns_device(self._d_colsq, W.ref_cols)
        _lib.select_columns(self._d_colsq, W.ref_cols, self.tol_e, 14, self._d_sel)
        sel = np.empty(2 + 2 * W.ref_cols, dtype=np.int32)
        _lib.check(_lib.load().figh_memcpy_d2h(sel.ctypes.data, self._d_sel.ptr, sel.nbytes))
        mask = sel[2 + W.ref_cols:] != 0
        if not mask.any():
            return
        kept = np.flatnonzero(mask).astype(np.int32)
        buf = getattr(self, "_d_kept_buf", None)
        if buf is None or buf.size < W.ref_cols:  # (allocated once: hipMalloc synchronises the device)
            buf = self._d_kept_buf = _lib.DeviceArray((W.ref_cols,), np.int32)
        _lib.check(_lib.load().figh_memcpy_h2d(buf.ptr, kept.ctypes.data, kept.nbytes))
        self._fused_kept = (mask.copy(), buf, len(kept))
        self.prefix_passes = getattr(self, "prefix_passes", 0) + 1

    def _run_fused(self, handle, flags, strings):
        """One pass with K1 and the level-0 TSQR in one launch over the kept-column list of the previous pass
        (figh_regressor_tsqr_fused).  Returns None -- the caller takes the two-launch path -- when the shape is not supported
        or the kept set turned out to be different."""
        ex, lib, W = self.exchange, _lib.load(), self.W
        self._have_block_tri = False
        mask, d_kept, n = self._fused_kept
        ncols, with_tau = W.ref_cols, self.d_tau is not None
        nc = n + (1 if with_tau else 0)
        local = not getattr(ex, "collective", True)
        if not _lib.regressor_tsqr_fused(handle, flags, self.N, self.d_q, self.d_v, self.d_a, W.buf, W.ld, self._d_colsq,
                                         d_kept, n, self.d_tau, self.tol_qr if local else -1.0,
                                         self._d_rows if local else self._d_R):
            self.fuse = False  # (this shape does not fuse: not asked again until the next set_samples)
            return None
        ex.sum_columns_device(self._d_colsq, ncols)
        _lib.select_columns(self._d_colsq, ncols, self.tol_e, 14, self._d_sel)
        if not local:
            d_stack, count = ex.stack_triangles(self._d_R, nc)
            _lib.tsqr_merge_base(d_stack, count, nc, n, self.tol_qr, self._d_rows)
        words = ncols + self._sel_words + (nc + 1) * nc
        pin = getattr(self, "_host_pack", None)
        if pin is None or pin.size < self._d_pack.size:
            pin = self._host_pack = _lib.PinnedArray(self._d_pack.size)
        host = pin.array[:words]
        _lib.check(lib.figh_memcpy_d2h(host.ctypes.data, self._d_pack.ptr, host.nbytes))
        sel = host[ncols:ncols + self._sel_words].view(np.int32)
        mask_now = sel[2 + ncols:2 + 2 * ncols] != 0
        if not np.array_equal(mask_now, mask):
            self._fused_kept = None  # the kept set changed: the two-launch path learns the new one
            return None
        self.fused_passes += 1
        self._n_expected = n
        cached = self._kept_cache
        if cached is not None and np.array_equal(cached[0], mask):
            idx_e, params_r = list(cached[1]), list(cached[2])
        else:
            idx_e = np.flatnonzero(~mask).tolist()
            params_r = [self.names[i] for i in np.flatnonzero(mask).tolist()]
            self._kept_cache = (mask.copy(), list(idx_e), list(params_r))
        rows_k = host[ncols + self._sel_words:].reshape(nc + 1, nc)
        return self._finish(rows_k, n, nc, params_r, idx_e, host[:ncols].copy(), with_tau, W.rows * ex.world_size, strings)

    def _block_tri(self, wls, nblocks, nc):
        """Device buffer for the per-row-block triangles of figh_tsqr_selected_blocks (kept for the weighted solve)."""
        if not wls:
            return None
        need = (nblocks + 2) * nc * nc  # (the compact stack is at most nblocks nc rows + nc + 1 rows of slack)
        buf = getattr(self, "_d_block_tri", None)
        if buf is None or buf.size < need:
            buf = self._d_block_tri = _lib.DeviceArray((need,), np.float64)
        return buf

    def _sb(self, name, count, dtype=np.float64):
        """A device scratch buffer of this pipeline, kept from pass to pass (figh_malloc / figh_free synchronise the stream:
        six of them per weighted solve were 0.3 ms of a 14 ms TIAGo step)."""
        pool = self.__dict__.setdefault("_scratch", {})
        buf = pool.get(name)
        if buf is None or buf.size != int(count) or buf.dtype != np.dtype(dtype):
            if buf is not None:
                buf.free()
            buf = pool[name] = _lib.DeviceArray((int(count),), dtype)
        return buf

    def _up(self, name, host):
        """``host`` (1-D array) in the scratch buffer ``name``."""
        host = np.ascontiguousarray(host)
        buf = self._sb(name, host.size, host.dtype)
        _lib.check(_lib.load().figh_memcpy_h2d(buf.ptr, host.ctypes.data, host.nbytes))
        return buf

    def _wls(self, out, strings=False):
        """Weighted least squares of examples/staubli_TX40/identification.py:305-346 (what
        identification_tools.weighted_least_squares_blocks does on a host W_b), device-resident:
        sigma_j^2 = ||tau_j - W_b,j phi_b||^2 / n_j per joint row block, phi = (W^T S^-1 W)^-1 W^T S^-1 tau, rounded to 6
        decimals, C_X = (W^T S^-1 W)^-1, std% = 100 sqrt(diag C_X) / |phi| (2 decimals).  phi_b is the rounded OLS solution
        of the pass, as in the script.

        Tree models whose pass went through per-row-block triangles (TIAGo): both the residual norms and the weighted
        triangle come from those nblocks small triangles -- W is not read again.  Otherwise: one pass over W for the
        residuals (figh_matvec + figh_block_sqnorm) and one weighted TSQR over the base columns (figh_tsqr with row-block
        weights), straight from the resident W in whatever layout it has."""
        if self.d_tau is None:
            raise ValueError("wls=True needs tau")
        ex, W = self.exchange, self.W
        n = len(out["params_r"])
        nc = n + 1
        base = np.asarray(out["idx_base"], dtype=np.int64)
        nb_par = len(base)
        phi_b = np.ascontiguousarray(out["phi_b"], dtype=np.float64)
        mode, _, _ = self._flags()
        nblocks = self.robot.model.nv if mode == _lib.MODE_JOINT_TORQUE else 6
        rows_blk = W.rows // nblocks
        # [residual norms of the nblocks row blocks | this rank's rows per block]: one sum over the ranks gives both -- the
        # shards of a run need not be equally long (dist.shard_range), so the divisor is the summed row count, not
        # rows_blk * world_size
        d_r2 = self._sb("wls_r2", nblocks + 1)
        _rows = np.array([float(rows_blk)])
        _lib.check(_lib.load().figh_memcpy_h2d(d_r2.ptr + 8 * nblocks, _rows.ctypes.data, 8))
        d_Rw = self._sb("wls_Rw", (nb_par + 1) * (nb_par + 1))
        kept = np.flatnonzero(self._kept_cache[0])
        if getattr(self, "_have_block_tri", False):
            # the compact stack of the pass: block j holds its n_j + 1 triangle rows over the kept columns + tau
            tri = self._d_block_tri
            counts = np.asarray(self._block_cache[1], dtype=np.int64) + 1
            row_off = np.concatenate([[0], np.cumsum(counts)])
            v = np.zeros(nc)
            v[base] = phi_b
            v[n] = -1.0
            _lib.block_rows_residuals(tri, row_off, nc, self._up("wls_v", v), d_r2)
            r2 = np.asarray(ex.sum_columns(d_r2, nblocks + 1))
            sig2 = r2[:nblocks] / r2[nblocks]
            if self.row_blocks is not None:  # (blocks without measurements: no rows in the stack, no variance)
                inactive = counts == 0
                sig2 = np.where(inactive, 1.0, sig2)
            d_cols = self._up("wls_cols", np.r_[base, n].astype(np.int32))
            _lib.tsqr(tri, int(row_off[-1]), nc, d_cols, nb_par + 1, None, np.repeat(1.0 / np.sqrt(sig2), counts), d_Rw)
            if self.row_blocks is not None:
                sig2 = sig2[self.row_blocks]
            source = "per-row-block triangles"
        else:
            if self._compact is not None or self.row_blocks is not None:
                raise RuntimeError("block-compact W / row_blocks without per-row-block triangles")
            d_cols = self._up("wls_cols", np.asarray(self.device_columns(kept[base]), dtype=np.int32))
            d_est = self._sb("wls_est", W.rows)
            _lib.matvec(W.buf, W.rows, W.ld, d_cols, nb_par, self._up("wls_phi", phi_b), d_est)
            _lib.block_sqnorm(self.d_tau, d_est, W.rows, nblocks, d_r2)
            r2 = np.asarray(ex.sum_columns(d_r2, nblocks + 1))
            sig2 = r2[:nblocks] / r2[nblocks]
            _lib.tsqr(W.buf, W.rows, W.ld, d_cols, nb_par, self.d_tau, 1.0 / np.sqrt(sig2), d_Rw)
            source = "second pass over W"
        if getattr(ex, "collective", True) and ex.world_size > 1:
            d_stack, count = ex.stack_triangles(d_Rw, nb_par + 1)
            d_one = self._sb("wls_one", (nb_par + 1) * (nb_par + 1))
            _lib.tsqr_merge(d_stack, count, nb_par + 1, d_one)
            d_Rw = d_one
        self._finish_beta(out)  # (the device is busy with the weighted factorisation meanwhile)
        if strings and "params_base" not in out:
            params_r = out["params_r"]
            regroup = np.setdiff1d(np.arange(n), base).tolist()
            out["params_base"] = qrd._expressions([params_r[i] for i in base.tolist()], [params_r[i] for i in regroup],
                                                  out["beta"])
        Rw = d_Rw.to_host().reshape(nb_par + 1, nb_par + 1)
        R, z = np.triu(Rw[:nb_par, :nb_par]), Rw[:nb_par, nb_par]
        with _single_threaded_blas(nb_par):
            phi = np.around(_solve_upper(np.asfortranarray(R), np.asfortranarray(z.reshape(-1, 1)))[:, 0], 6)
            R_inv, info = _dtrtri(np.ascontiguousarray(R))
            if info != 0:
                raise np.linalg.LinAlgError("Singular matrix")
            std = _relative_percent(np.sqrt(np.einsum("ij,ij->i", R_inv, R_inv)), phi)  # (phi_i == 0: inf, as in the script)
        out["phi_wls"], out["std_wls"], out["sigma2_joint"], out["wls_source"] = phi, std, sig2, source
        return out

    def _finish_beta(self, out):
        """The regrouping coefficients _finish left for later (``defer_beta``): beta = R1^-1 R2, rounded (qrdecomposition.py:244)."""
        later = getattr(self, "_beta_later", None)
        if later is None or out.get("beta") is not None:
            return
        R1, R2, _, _ = later
        self._beta_later = None
        with _single_threaded_blas(len(R1)):
            out["beta"] = np.around(_solve_upper(R1, np.asfortranarray(R2)), 6)

    def _block_lists(self, ncols, stride):
        """(mask, counts, d_cols, d_pos) for figh_tsqr_selected_blocks, from the kept mask this pass expects: row block j
        (joint j + 1) gets the kept inertial columns of the links in that joint's subtree and the kept Ia / fv / fs / off
        columns of its own link -- everything else in the block is a structural zero of the regressor
        (regressor.py:45-87).  None while no mask is known."""
        mask = getattr(self, "_mask_expected", None)
        if mask is None or len(mask) != ncols:
            return None
        cached = self._block_cache
        if cached is not None and np.array_equal(cached[0], mask):
            return cached
        m = self.robot.model
        nb = m.nv
        parents = list(m.parents)
        anc = np.zeros((nb, nb), dtype=bool)  # anc[j, k]: joint j + 1 is joint k + 1 or one of its ancestors
        for k in range(nb):
            jid = k + 1
            while jid > 0:
                anc[jid - 1, k] = True
                jid = parents[jid]
        kept = np.flatnonzero(mask)
        link, slot = kept // 14, kept % 14
        counts, cols, pos, ccols = [], [], [], []
        active = None if self.row_blocks is None else set(self.row_blocks)
        for j in range(nb):
            if active is not None and j not in active:  # no measurements on this joint: the block takes no part
                counts.append(-1)
                continue
            in_block = (anc[j][np.minimum(link, nb - 1)] & (link < nb) & (slot < 10)) | ((link == j) & (slot >= 10))
            p = np.flatnonzero(in_block)
            counts.append(len(p))
            pos.append(p)
            cols.append(link[p] * stride + slot[p])
            ccols.append((link[p] - j) * 16 + slot[p])  # block-compact W: the row block starts at its own joint's link
        pad = [np.zeros(1, dtype=np.int64)]
        d_cols = _lib.DeviceArray.from_host(np.concatenate(cols + pad).astype(np.int32))
        d_pos = _lib.DeviceArray.from_host(np.concatenate(pos + pad).astype(np.int32))
        d_ccols = _lib.DeviceArray.from_host(np.concatenate(ccols + pad).astype(np.int32))
        self._block_cache = (mask.copy(), np.asarray(counts, dtype=np.int32), d_cols, d_pos, d_ccols)
        return self._block_cache

    def _subtree_sizes(self):
        """Links in the subtree of every joint (depth-first numbering: the subtree of joint j + 1 is links j .. j + size - 1)."""
        m = self.robot.model
        nb = m.nv
        parents = list(m.parents)
        size = np.zeros(nb, dtype=np.int64)
        for k in range(nb):
            jid = k + 1
            while jid > 0:
                size[jid - 1] += 1
                jid = parents[jid]
        return size

    def _force_slots(self, ncols):
        """Mask over the reference's columns: slot >= 6 within a link (mx my mz m Ia fv fs off)."""
        m = getattr(self, "_force_slot_mask", None)
        if m is None or len(m) != ncols:
            m = self._force_slot_mask = (np.arange(ncols) % 14) >= 6
        return m

    def device_columns(self, ref_cols):
        """Column indices of the HBM-resident ``self.W`` that hold the reference's columns ``ref_cols`` (trees keep W
        link-padded: 16 columns per link, reference column c at 16 (c // 14) + c % 14)."""
        if getattr(self, "_compact", None) is not None:
            raise ValueError("block-compact W has no global column numbering: row block j is its own N x 16 |subtree_j| "
                             "matrix (pipe.W.compact = (element offsets, leading dimensions))")
        c = np.asarray(ref_cols, dtype=np.int64)
        if getattr(self, "_force_ld", 0):
            raise ValueError("force-compact W is two matrices (force rows: pipe.W.force_ld columns, torque rows behind them): "
                             "no single column numbering; IdentificationPipeline(w_layout='link-compact') keeps one")
        if getattr(self, "_link_pos", None) is not None:  # link-compact: only links with entries have a segment
            pos = self._link_pos[c // 14]
            if (pos < 0).any():
                raise ValueError("link-compact W holds no columns for links without entries (structural zeros)")
            return pos * 16 + c % 14
        return (c // 14) * 16 + c % 14 if getattr(self, "_padded", False) else c

    def _place_W(self, rows, cols, handle, mode, flags, ft_mask):
        """Allocate W; with ``placement_trials`` > 1 keep the candidate allocation on which K1 runs fastest."""
        import time
        nbytes = rows * cols * 8
        trials = self.placement_trials
        if trials > 1:
            trials = max(1, min(trials, int(0.5 * _lib.device_info()["hbm_bytes"] // max(nbytes, 1))))
        if trials <= 1:
            return GpuMatrix.empty(rows, cols)
        d_cs = _lib.DeviceArray((cols,), np.float64)
        build = _lib.regressor_build_padded if self._padded else _lib.regressor_build
        cands, times = [], []
        for _ in range(trials):  # all candidates are alive at once: a freed block would simply be handed out again
            W = GpuMatrix.empty(rows, cols)
            for rep in range(4):
                if rep == 1:
                    _lib.synchronize()
                    t0 = time.perf_counter()
                build(handle, mode, flags, ft_mask, self.N, self.d_q, self.d_v, self.d_a, W.buf, W.ld, d_cs)
            _lib.synchronize()
            cands.append(W)
            times.append((time.perf_counter() - t0) / 3)
        best = int(np.argmin(times))
        for i, W in enumerate(cands):
            if i != best:
                W.buf.free()
        self.placement_report = {"trials": trials, "k1_ms": [round(1e3 * t, 4) for t in times], "kept": best}
        return cands[best]

    def _tail(self, d_stack, count, n, nc, params_r, idx_e, col_norm, with_tau, total_rows, strings):
        """Stack of plain triangles in HBM (one per rank) -> results: reduction, rank decision and regrouped factorisation
        on the device (figh_tsqr_merge_base), one copy back."""
        lib = _lib.load()
        pack = getattr(self, "_d_tailpack", None)
        if pack is None or pack.size < (nc + 1) * nc:
            pack = self._d_tailpack = _lib.DeviceArray(((nc + 1) * nc,), np.float64)
        _lib.tsqr_merge_base(d_stack, count, nc, n, self.tol_qr, pack)
        pin = getattr(self, "_host_tailpack", None)
        if pin is None or pin.size < (nc + 1) * nc:
            pin = self._host_tailpack = _lib.PinnedArray((nc + 1) * nc)
        host = pin.array[:(nc + 1) * nc]
        _lib.check(lib.figh_memcpy_d2h(host.ctypes.data, pack.ptr, host.nbytes))
        return self._finish(host.reshape(nc + 1, nc), n, nc, params_r, idx_e, col_norm, with_tau, total_rows, strings)

    def _finish(self, rows_k, n, nc, params_r, idx_e, col_norm, with_tau, total_rows, strings, defer_beta=False):
        """Host tail on the n x n numbers the device returns (qrdecomposition.py:215-266): ``rows_k`` ((nc + 1) x nc) holds,
        in the original column order, the rows of the regrouped factorisation qr([W1 W2 tau]) at the base columns and, in
        its last row, the diagonal of the plain factorisation (include/figh.h, figh_tsqr_selected)."""
        diag = np.abs(rows_k[nc])
        rows_k = rows_k[:nc]
        base_mask = diag[:n] > self.tol_qr  # the device took the same decision on the same numbers
        base = np.flatnonzero(base_mask)
        rest = np.flatnonzero(~base_mask)
        idx_base, idx_regroup = base.tolist(), rest.tolist()
        assert len(params_r) == n, "params_r does not have same length with R"
        Rb = rows_k[base]
        R1, R2, z = Rb[:, base], Rb[:, rest], (Rb[:, n] if with_tau else None)
        if self.null_pivots:
            # the certificate of the null-pivot rule (_host.null_rule_certified): the coefficient sums come from the regrouped
            # triangle and are kept while the two index sets stay the same (they are properties of the model's geometry far more
            # than of the samples); the margins are this pass's own pivots
            from ._host import null_rule_bounds
            key = (n, base.tobytes())
            cache = getattr(self, "_cert_cache", None)
            if cache is None or cache[0] != key:
                with _single_threaded_blas(n):
                    cache = self._cert_cache = (key, null_rule_bounds(np.triu(R1), R2))
            self._cert_args = (diag[:n].copy(), idx_base, idx_regroup, cache[1])  # (phi joins below, once it is solved)
        # inv(R1) of qrdecomposition.py:244 by LAPACK's triangular inverse (dtrtri): R1 is upper triangular (the routine
        # reads the upper triangle only; below it the device leaves rounding residues), and np.linalg.inv's general LU
        # path pays ~30 us of BLAS thread start-up per call on a many-core host
        with _single_threaded_blas(n):
            if n <= 80:
                R1_inv, info = _dtrtri(np.ascontiguousarray(R1))
                if info != 0:
                    raise np.linalg.LinAlgError("Singular matrix")
                beta = np.around(R1_inv @ R2, 6)
                phi_ls = R1_inv @ z if with_tau else None
            elif defer_beta:
                # the weighted solve comes next and only needs phi: the regrouping coefficients (the larger solve: 55 right-hand
                # sides for TIAGo) wait until the weighted factorisation has been launched (_finish_beta, called by _wls)
                phi_ls = np.ascontiguousarray(_solve_upper(R1, np.asfortranarray(z.reshape(-1, 1)))[:, 0])
                beta = None
                self._beta_later = (np.ascontiguousarray(R1), R2, idx_base, idx_regroup)
            else:
                # wide problems (TALOS: 234 base columns, 96 regrouped): one triangular solve of [R2 z] is a third of the
                # flops of inverse + products (host tail 2.0 -> 1.2 ms)
                rhs = np.asfortranarray(np.c_[R2, z] if with_tau else R2)
                X = _solve_upper(R1, rhs)
                beta = np.around(X[:, :R2.shape[1]], 6)
                phi_ls = np.ascontiguousarray(X[:, -1]) if with_tau else None
        out = {
            "idx_e": idx_e, "params_r": params_r, "idx_base": idx_base, "beta": beta,
            "col_norm": col_norm, "absdiagR": diag[:n].copy(), "rows": total_rows,
        }
        if self.null_pivots:
            from ._host import null_rule_certified
            d_, ib_, ir_, bounds_ = self._cert_args
            out["null_rule_certified"] = null_rule_certified(d_, ib_, ir_, bounds_, self.tol_qr, phi=phi_ls if with_tau else None)
        if strings and beta is not None:
            out["params_base"] = qrd._expressions([params_r[i] for i in idx_base],
                                                  [params_r[i] for i in idx_regroup], beta)
        if with_tau:
            out["phi_b"] = np.round(phi_ls, 6)
            out["phi_ls"] = phi_ls
            out["residual_norm"] = abs(float(rows_k[n, n]))
        return out
