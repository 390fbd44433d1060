"""Device-resident matrices for the drop-in functions.

The reference's functions take and return NumPy arrays.  The
mirrors in ``tools/`` accept either a NumPy array (uploaded for the call) or a
:class:`GpuMatrix`, and return NumPy arrays unless the caller passed
``param["device_resident"] = True`` / a ``GpuMatrix``, so a script can keep the
stacked regressor in HBM between ``build_regressor_basic`` ->
``get_index_eliminate`` -> ``build_regressor_reduced`` -> ``get_baseParams``
instead of paying a PCIe round trip per call.
"""
import numpy as np

from . import _lib


_HUGE = 2 << 20


def host_empty(shape):
    """Uninitialised float64 host array for a result that comes back from HBM.  Large ones (>= 64 MB) are backed by an
    anonymous mapping advised to transparent huge pages: the first touch of a fresh 4 GB ``np.empty`` costs a million page
    faults (the D2H copy then runs at 12 - 24 GB/s), 2 MB pages make that two thousand (48 GB/s through the staged copy of
    ``figh_memcpy_d2h``; tools/microbench/d2h_probe2.hip).  An ordinary writable ndarray as far as the caller can tell: it owns
    the mapping, which goes away with it."""
    shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    count = int(np.prod(shape)) if shape else 1
    nbytes = 8 * count
    if nbytes < (64 << 20):
        return np.empty(shape)
    try:
        import mmap

        # (PRIVATE anonymous memory: Python's default for fileno -1 is a SHARED mapping -- shmem pages, no transparent huge pages
        # and a microsecond per fault: 3.9 GB/s measured)
        m = mmap.mmap(-1, nbytes + _HUGE, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
        if hasattr(m, "madvise") and hasattr(mmap, "MADV_HUGEPAGE"):
            try:
                m.madvise(mmap.MADV_HUGEPAGE)
            except OSError:  # (no THP on this kernel: plain pages, still correct)
                pass
        base = np.frombuffer(m, dtype=np.uint8)
        off = (-base.ctypes.data) % _HUGE  # start on a huge-page boundary
        return base[off:off + nbytes].view(np.float64).reshape(shape)
    except (ImportError, OSError, ValueError, BufferError):
        return np.empty(shape)


class GpuMatrix:
    """rows x cols float64, row-major with leading dimension ``ld``, in HBM."""

    def __init__(self, buf, rows, cols, ld=None):
        self.buf = buf
        self.rows, self.cols = int(rows), int(cols)
        self.ld = int(ld if ld is not None else cols)

    @property
    def shape(self):
        return (self.rows, self.cols)

    @property
    def ptr(self):
        return self.buf.ptr

    @classmethod
    def empty(cls, rows, cols, ld=None):
        """``ld``: leading dimension in elements (default ``cols``: the reference's dense layout).  Buffers that never
        leave the device may pad it to whole 128-byte lines (16 doubles): the tree regressor kernel then writes full
        cache lines."""
        ld = int(cols if ld is None else ld)
        return cls(_lib.DeviceArray((max(int(rows) * ld, 1),), np.float64), rows, cols, ld)

    @classmethod
    def from_host(cls, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        if arr.ndim != 2:
            raise ValueError("expected a 2-D array, got shape %r" % (arr.shape,))
        return cls(_lib.DeviceArray.from_host(arr.reshape(-1)), arr.shape[0], arr.shape[1])

    def numpy(self):
        if getattr(self, "compact", None) is not None:
            # (IdentificationPipeline(w_layout="block-compact"): rows / cols describe the regressor, the buffer holds
            # N sum_j ld_j doubles -- reading rows * ld doubles would run far past the allocation)
            raise ValueError("block-compact W is not a dense rows x ld matrix: read its row blocks through .compact "
                             "(element offsets, leading dimensions)")
        if getattr(self, "force_ld", 0):
            # (force-compact W of the external-wrench regressor: rows [0, rows / 2) live in a region of their own with
            # leading dimension force_ld, the torque rows behind it -- the buffer is smaller than rows x ld)
            raise ValueError("force-compact W is two matrices (force rows: .force_ld columns, torque rows behind them), not a "
                             "dense rows x ld matrix; IdentificationPipeline(w_layout='link-compact') keeps one matrix")
        if self.ld == self.cols:
            out = host_empty((self.rows, self.cols))
            if out.size:
                _lib.check(_lib.load().figh_memcpy_d2h(out.ctypes.data, self.buf.ptr, out.nbytes))
            return out
        full = host_empty((self.rows, self.ld))
        if full.size:
            _lib.check(_lib.load().figh_memcpy_d2h(full.ctypes.data, self.buf.ptr, full.nbytes))
        out = host_empty((self.rows, self.cols))
        out[...] = full[:, :self.cols]
        return out

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def __len__(self):
        return self.rows


def to_device(W):
    """(GpuMatrix, was_already_on_device)"""
    if isinstance(W, GpuMatrix):
        return W, True
    return GpuMatrix.from_host(W), False


def vector_to_device(x):
    if isinstance(x, _lib.DeviceArray) or hasattr(x, "ptr"):  # (device buffers and views of them pass through)
        return x
    return _lib.DeviceArray.from_host(np.ascontiguousarray(x, dtype=np.float64).reshape(-1))


def index_to_device(idx):
    return _lib.DeviceArray.from_host(np.ascontiguousarray(idx, dtype=np.int32))
