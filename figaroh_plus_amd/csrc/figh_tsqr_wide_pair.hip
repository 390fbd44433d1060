// K3, wide form, merge levels: the PAIR instantiations of the blocked TSQR kernel (figh_tsqr_wide_kernel.h).  One level
// halves a stack of compact nc x nc triangles: workgroup b starts FROM triangle 2b and absorbs triangle 2b + 1 (tiles of
// a triangle start at their first non-zero chunk).  The reference has no analogue: this is the reduction of
// np.linalg.qr(W_e) (src/figaroh/tools/qrdecomposition.py:205) over row blocks.
#include "figh_tsqr_wide_kernel.h"

namespace figh {

// stack of `count` triangles -> (count + 1) / 2 triangles in Rws_out (an odd last triangle passes through)
int launch_tsqr_wide_pairs(const double *stack, long count, int nc, double *Rws_out) {
    const long nwg = (count + 1) / 2;
    const int nch = (nc + 15) >> 4;
    const size_t blk_bytes = sizeof(double) * 256 * ((size_t)nch * (nch + 1) / 2) * (size_t)nwg;
    double *Rblk = static_cast<double *>(workspace(blk_bytes, 13));
    if (!Rblk) return FIGH_ERR_ALLOC;
    const bool ok = wy_dispatch(wy_config(nc), [&](auto NW, auto CPW, auto NRC, auto WPE, auto LDSC) {
        hipLaunchKernelGGL((tsqr_wy_kernel<decltype(NW)::value, decltype(CPW)::value, decltype(NRC)::value,
                                           decltype(WPE)::value, decltype(LDSC)::value, true>),
                           dim3((unsigned)nwg), dim3(64 * decltype(NW)::value), 0, stream(), stack, (long)nc, (long)nc,
                           (const int *)nullptr, nc, (const double *)nullptr, (const double *)nullptr, 1L, Rblk, Rws_out,
                           nc, (long long *)nullptr, count);
    });
    if (!ok) {
        set_error("figh_tsqr: no wide-kernel geometry for this column count");
        return FIGH_ERR_UNSUPPORTED;
    }
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}

}  // namespace figh
