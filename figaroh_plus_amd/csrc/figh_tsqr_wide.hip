// K3, wide form, level 0: launcher of the blocked (compact-WY) Householder TSQR for 80 < nc <= 512 columns.  The kernel
// and its geometry table live in figh_tsqr_wide_kernel.h; the pair-merge levels are figh_tsqr_wide_pair.hip.
#include "figh_tsqr_wide_kernel.h"

namespace figh {

// persistent workgroups the wide kernel wants for nc columns (one private triangle each)
long tsqr_wide_workgroups(const int nc, const int cus) {
    int occ = 1;
    wy_dispatch(wy_config(nc), [&](auto NW, auto CPW, auto NRC, auto WPE, auto LDSC) {
        occ = wy_occupancy<decltype(NW)::value, decltype(CPW)::value, decltype(NRC)::value, decltype(WPE)::value,
                           decltype(LDSC)::value>();
    });
    return (long)cus * occ;
}

// rows of (W, ldw) -> nwg triangles (nc x nc, row-major) in Rws_out; tiles are dealt round-robin to the workgroups.
// chain_flags: bit 0 = the workgroups continue from the triangles Rws_out holds (same nwg and nc as the launch before)
int launch_tsqr_wide(const double *W, long rows, long ldw, const int *col_idx, int n, const double *tau,
                     const double *d_blkw, long rows_per_blk, int nc, long nwg, double *Rws_out, int chain_flags) {
    if (chain_flags & 1)  // the CHAIN instantiations live in figh_tsqr_wide_batch.hip
        return launch_tsqr_wide_chain(W, rows, ldw, col_idx, n, tau, d_blkw, rows_per_blk, nc, nwg, Rws_out);
    const int nch = (nc + 15) >> 4;
    const size_t blk_bytes = sizeof(double) * 256 * ((size_t)nch * (nch + 1) / 2) * (size_t)nwg;
    double *Rblk = static_cast<double *>(workspace(blk_bytes, 13));
    if (!Rblk) return FIGH_ERR_ALLOC;
    long long *prof = nullptr;
#ifdef FIGH_ABLATION
    const WyConfig pcfg = wy_config(nc);
    static const bool want_prof = getenv("FIGH_WY_PROF") != nullptr;
    static bool alias_set = false;
    if (!alias_set) {
        const int v = getenv("FIGH_WY_RALIAS") != nullptr;
        hipMemcpyToSymbol(HIP_SYMBOL(g_wy_ralias), &v, sizeof(int));
        const int o = getenv("FIGH_WY_OFF") ? atoi(getenv("FIGH_WY_OFF")) : 0;
        hipMemcpyToSymbol(HIP_SYMBOL(g_wy_off), &o, sizeof(int));
        int dm = 0, dt = 0;
        if (const char *e = getenv("FIGH_WY_DELAY")) sscanf(e, "%d,%d", &dm, &dt);
        hipMemcpyToSymbol(HIP_SYMBOL(g_wy_delay_mode), &dm, sizeof(int));
        hipMemcpyToSymbol(HIP_SYMBOL(g_wy_delay_ticks), &dt, sizeof(int));
        alias_set = true;
    }
    if (want_prof && rows >= 65536) {
        prof = static_cast<long long *>(workspace(sizeof(long long) * 12 * nwg * pcfg.nw, 6));
        if (!prof) return FIGH_ERR_ALLOC;
    }
#endif
    const bool ok = wy_dispatch(wy_config(nc), [&](auto NW, auto CPW, auto NRC, auto WPE, auto LDSC) {
        FIGH_LAUNCH_TIMED((tsqr_wy_kernel<decltype(NW)::value, decltype(CPW)::value, decltype(NRC)::value,
                                          decltype(WPE)::value, decltype(LDSC)::value, 0>),
                          dim3((unsigned)nwg), dim3(64 * decltype(NW)::value), 0, W, rows, ldw, col_idx, n, tau, d_blkw,
                          rows_per_blk, Rblk, Rws_out, nc, prof, 0L, 0, null_pivot_sq());
    });
    if (!ok) {
        set_error("figh_tsqr: no wide-kernel geometry for this column count");
        return FIGH_ERR_UNSUPPORTED;
    }
    FIGH_HIP(hipGetLastError());
#ifdef FIGH_ABLATION
    if (prof) {
        const long nwv = nwg * pcfg.nw;
        std::vector<long long> h(12 * nwv);
        FIGH_HIP(hipMemcpyAsync(h.data(), prof, sizeof(long long) * 12 * nwv, hipMemcpyDeviceToHost, stream()));
        FIGH_HIP(hipStreamSynchronize(stream()));
        double acc[12] = {0};
        for (long w = 0; w < nwv; ++w)
            for (int k = 0; k < 12; ++k) acc[k] += (double)h[12 * w + k];
        const long ntiles = (rows + 16 * pcfg.nrc - 1) / (16 * pcfg.nrc);
        fprintf(stderr, "[wy prof] cfg %d,%d,%d,%d nc %d wgs %ld (occupancy %ld/CU) tiles/wg %.1f | ticks per wave: kernel %.0f = top %.0f + "
                        "first panel wait %.0f [owner: load %.0f panel %.0f retire %.0f] + la-update %.0f + la-X %.0f + la-panel %.0f + la-retire %.0f + updates %.0f + barrier %.0f\n",
                pcfg.nw, pcfg.cpw, pcfg.nrc, pcfg.wpe, nc, nwg, tsqr_wide_workgroups(nc, 1), (double)ntiles / nwg, acc[0] / nwv, acc[1] / nwv,
                acc[2] / nwv, acc[8] / nwv, acc[9] / nwv, acc[10] / nwv, acc[3] / nwv, acc[7] / nwv, acc[4] / nwv, acc[11] / nwv, acc[5] / nwv, acc[6] / nwv);
    }
#endif
    return FIGH_OK;
}

}  // namespace figh
