// K3, wide form, merge levels: the PAIR instantiations of the blocked TSQR kernel (figh_tsqr_wide_kernel.h).  One level
// halves a stack of compact nc x nc triangles: workgroup b starts FROM triangle 2b and absorbs triangle 2b + 1 (tiles of
// a triangle start at their first non-zero chunk).  The reference has no analogue: this is the reduction of
// np.linalg.qr(W_e) (src/figaroh/tools/qrdecomposition.py:205) over row blocks.
#include "figh_tsqr_wide_kernel.h"

namespace figh {

namespace {

// Geometry of a pair merge.  One workgroup per CU at most and nothing else on the chip: the time of a level is the
// number of panel phases of one triangle times the length of a phase, so (i) EIGHT waves share the trailing chunks -- a
// sweep (36 MFMAs per chunk) is then shorter than the look-ahead chain and leaves the critical path, where with four
// waves the owner's deferred double sweep was the longest thing in a phase (25 k cycles per phase measured against the
// chain's 15-19 k) -- and (ii) the tile is as TALL as the registers of a wave with two or three chunk slots allow: fewer
// tiles per triangle, fewer phases (331 columns: 21 + 15 + 9 + 3 = 48 phases with 96-row tiles against 66 with 64 rows).
WyConfig wy_config_pair(const int nc) {
    const int nch = (nc + 15) >> 4;
    if (nch <= 16) return {8, 2, 8, 2, 0};
    if (nch <= 24) return {8, 3, 6, 2, 0};
    return {8, 4, 4, 2, 0};
}

template <class F>
bool wy_dispatch_pair(const WyConfig cfg, F &&f) {
#define FIGH_WY_CASE(NW_, CPW_, NRC_, WPE_)                                                          \
    if (cfg.nw == NW_ && cfg.cpw == CPW_ && cfg.nrc == NRC_ && cfg.wpe == WPE_) {                    \
        f(std::integral_constant<int, NW_>{}, std::integral_constant<int, CPW_>{},                   \
          std::integral_constant<int, NRC_>{}, std::integral_constant<int, WPE_>{},                  \
          std::integral_constant<bool, false>{});                                                    \
        return true;                                                                                 \
    }
    FIGH_WY_CASE(8, 2, 8, 2)
    FIGH_WY_CASE(8, 3, 6, 2)
    FIGH_WY_CASE(8, 4, 4, 2)
#undef FIGH_WY_CASE
    return false;
}

}  // namespace

// stack of `count` triangles -> (count + 1) / 2 triangles in Rws_out (an odd last triangle passes through)
int launch_tsqr_wide_pairs(const double *stack, long count, int nc, double *Rws_out) {
    const long nwg = (count + 1) / 2;
    const int nch = (nc + 15) >> 4;
    const size_t blk_bytes = sizeof(double) * 256 * ((size_t)nch * (nch + 1) / 2) * (size_t)nwg;
    double *Rblk = static_cast<double *>(workspace(blk_bytes, 13));
    if (!Rblk) return FIGH_ERR_ALLOC;
    auto launch = [&](auto NW, auto CPW, auto NRC, auto WPE, auto LDSC) {
        hipLaunchKernelGGL((tsqr_wy_kernel<decltype(NW)::value, decltype(CPW)::value, decltype(NRC)::value,
                                           decltype(WPE)::value, decltype(LDSC)::value, 1>),
                           dim3((unsigned)nwg), dim3(64 * decltype(NW)::value), 0, stream(), stack, (long)nc, (long)nc,
                           (const int *)nullptr, nc, (const double *)nullptr, (const double *)nullptr, 1L, Rblk, Rws_out,
                           nc, (long long *)nullptr, count, 0, null_pivot_sq());
    };
    // More pairs than CUs (the stacked triangles of a streamed run: 20 chunks x 512 for the human model): the level is
    // throughput-bound, and the level-0 geometry -- four waves, two workgroups = two chains per CU -- absorbs twice as
    // many triangles per CU at a time as the eight-wave one (5121 pairs of 191 columns: 4.7 ms with the latter).
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const bool ok = nwg > cus ? wy_dispatch(wy_config(nc), launch) : wy_dispatch_pair(wy_config_pair(nc), launch);
    if (!ok) {
        set_error("figh_tsqr: no wide-kernel geometry for this column count");
        return FIGH_ERR_UNSUPPORTED;
    }
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}

// One workgroup, one small matrix (rows of the order of nc: the regrouped factorisation qr(R[:, perm]) of the rank step)
// with the pair-merge geometry -- the same latency-bound situation, eight waves and the tallest tile
int launch_tsqr_wide_single(const double *W, long rows, long ldw, const int *col_idx, int n, int nc, double *R_out) {
    const int nch = (nc + 15) >> 4;
    double *Rblk = static_cast<double *>(workspace(sizeof(double) * 256 * ((size_t)nch * (nch + 1) / 2), 13));
    if (!Rblk) return FIGH_ERR_ALLOC;
    const bool ok = wy_dispatch_pair(wy_config_pair(nc), [&](auto NW, auto CPW, auto NRC, auto WPE, auto LDSC) {
        hipLaunchKernelGGL((tsqr_wy_kernel<decltype(NW)::value, decltype(CPW)::value, decltype(NRC)::value,
                                           decltype(WPE)::value, decltype(LDSC)::value, 0>),
                           dim3(1), dim3(64 * decltype(NW)::value), 0, stream(), W, rows, ldw, col_idx, n,
                           (const double *)nullptr, (const double *)nullptr, 1L, Rblk, R_out, nc, (long long *)nullptr, 0L, 0,
                           null_pivot_sq());
    });
    if (!ok) {
        set_error("figh_tsqr: no wide-kernel geometry for this column count");
        return FIGH_ERR_UNSUPPORTED;
    }
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}

}  // namespace figh
