// K3, wide form, batched level 0: the BATCH instantiations of the blocked TSQR kernel (figh_tsqr_wide_kernel.h).  B
// independent tall matrices -- the base regressors of B candidate trajectories of one finite-difference gradient of the
// excitation objective (examples/tiago/optimal_trajectory.py:100-133, 296-313) -- are factored in ONE launch: they are
// the row segments of one joint-major regressor built by one K1 launch over all B * n_per samples.
#include "figh_tsqr_wide_kernel.h"

namespace figh {

// matrix b (b < B) = the rps row segments [j * seg_stride + b * n_per, + n_per) of (W, ldw); wgs workgroups per matrix,
// one triangle each: B * wgs compact nc x nc triangles in Rws_out, matrix-major
int launch_tsqr_wide_batch(const double *W, long ldw, const int *col_idx, int n, int nc, long B, long n_per, int rps,
                           long seg_stride, long wgs, double *Rws_out) {
    const long nwg = B * wgs;
    const int nch = (nc + 15) >> 4;
    const size_t blk_bytes = sizeof(double) * 256 * ((size_t)nch * (nch + 1) / 2) * (size_t)nwg;
    double *Rblk = static_cast<double *>(workspace(blk_bytes, 13));
    if (!Rblk) return FIGH_ERR_ALLOC;
    const bool ok = wy_dispatch(wy_config(nc), [&](auto NW, auto CPW, auto NRC, auto WPE, auto LDSC) {
        hipLaunchKernelGGL((tsqr_wy_kernel<decltype(NW)::value, decltype(CPW)::value, decltype(NRC)::value,
                                           decltype(WPE)::value, decltype(LDSC)::value, 2>),
                           dim3((unsigned)nwg), dim3(64 * decltype(NW)::value), 0, stream(), W, n_per, ldw, col_idx, n,
                           (const double *)nullptr, (const double *)nullptr, seg_stride, Rblk, Rws_out, nc,
                           (long long *)nullptr, wgs, rps, null_pivot_sq());
    });
    if (!ok) {
        set_error("figh_tsqr: no wide-kernel geometry for this column count");
        return FIGH_ERR_UNSUPPORTED;
    }
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}

// level 0 continued: like launch_tsqr_wide, but workgroup b starts from triangle b of Rws_out (what the previous launch of
// the same shape wrote there) instead of an empty one
int launch_tsqr_wide_chain(const double *W, long rows, long ldw, const int *col_idx, int n, const double *tau,
                           const double *d_blkw, long rows_per_blk, int nc, long nwg, double *Rws_out) {
    const int nch = (nc + 15) >> 4;
    const size_t blk_bytes = sizeof(double) * 256 * ((size_t)nch * (nch + 1) / 2) * (size_t)nwg;
    double *Rblk = static_cast<double *>(workspace(blk_bytes, 13));
    if (!Rblk) return FIGH_ERR_ALLOC;
    const bool ok = wy_dispatch(wy_config(nc), [&](auto NW, auto CPW, auto NRC, auto WPE, auto LDSC) {
        FIGH_LAUNCH_TIMED((tsqr_wy_kernel<decltype(NW)::value, decltype(CPW)::value, decltype(NRC)::value,
                                          decltype(WPE)::value, decltype(LDSC)::value, 3>),
                          dim3((unsigned)nwg), dim3(64 * decltype(NW)::value), 0, W, rows, ldw, col_idx, n, tau, d_blkw,
                          rows_per_blk, Rblk, Rws_out, nc, (long long *)nullptr, 0L, 0, null_pivot_sq());
    });
    if (!ok) {
        set_error("figh_tsqr: no wide-kernel geometry for this column count");
        return FIGH_ERR_UNSUPPORTED;
    }
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}

// rows of a tile of the geometry that serves nc columns (the batched launcher sizes its workgroup count with it)
int tsqr_wide_tile_rows(int nc) { return 16 * wy_config(nc).nrc; }

}  // namespace figh
