// K3, wide form, merge levels: the PAIR instantiations of the blocked TSQR kernel (figh_tsqr_wide_kernel.h).  One level
// halves a stack of compact nc x nc triangles: workgroup b starts FROM triangle 2b and absorbs triangle 2b + 1 (tiles of
// a triangle start at their first non-zero chunk).  The reference has no analogue: this is the reduction of
// np.linalg.qr(W_e) (src/figaroh/tools/qrdecomposition.py:205) over row blocks.
#include <cstring>

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

// Several stacks (the wide row blocks of a tree's joint-torque regressor: TIAGo has three of 81 .. 96 columns, 768 triangles
// each) reduced to one triangle each, level by level in lockstep: a level whose pairs of every stack fit the chip at once is
// latency-bound -- 65 us whatever the number of pairs -- and runs as ONE launch over all the stacks (MODE 4: the workgroup
// looks up its stack in a device table); a level with more pairs than CUs in some stack runs per stack as before.  The tables
// of all levels are built up front and uploaded when their content changed (the pipeline repeats the same structure).
int reduce_wide_stacks(std::vector<WyPairStack> &st) {
    const int nj = (int)st.size();
    if (nj == 0) return FIGH_OK;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    // buffers: two ping-pong areas and the packed blocks per stack
    std::vector<size_t> pp_off(nj), blk_off(nj);
    size_t pp_total = 0, blk_total = 0;
    for (int j = 0; j < nj; ++j) {
        const size_t tri = (size_t)st[j].nc * st[j].nc;
        const int nch = (st[j].nc + 15) >> 4;
        pp_off[j] = pp_total;
        pp_total += tri * (size_t)((st[j].count + 1) / 2);
        blk_off[j] = blk_total;
        blk_total += (size_t)256 * ((size_t)nch * (nch + 1) / 2) * (size_t)((st[j].count + 1) / 2);
    }
    double *pp[2] = {static_cast<double *>(workspace(sizeof(double) * pp_total, 32)),
                     static_cast<double *>(workspace(sizeof(double) * pp_total, 33))};
    double *blk = static_cast<double *>(workspace(sizeof(double) * blk_total, 34));
    if (!pp[0] || !pp[1] || !blk) return FIGH_ERR_ALLOC;
    struct Level {
        std::vector<int> grouped;  // stacks of this level's grouped launch
        std::vector<int> single;   // stacks that take their own launch
        size_t table_at = 0;       // first record of the grouped launch in the table
        int nwg = 0;
    };
    std::vector<Level> levels;
    std::vector<WyPairJob> table;
    std::vector<long> cnt(nj);
    std::vector<const double *> cur(nj);
    for (int j = 0; j < nj; ++j) {
        cnt[j] = st[j].count;
        cur[j] = st[j].tri;
        if (cnt[j] == 1)
            FIGH_HIP(hipMemcpyAsync(st[j].out, st[j].tri, sizeof(double) * (size_t)st[j].nc * st[j].nc, hipMemcpyDeviceToDevice,
                                    stream()));
    }
    struct Single {
        const double *src;
        long count;
        int nc;
        double *dst;
    };
    std::vector<std::vector<Single>> singles;
    for (int lvl = 0;; ++lvl) {
        Level L;
        std::vector<Single> sg;
        bool any = false;
        for (int j = 0; j < nj; ++j) {
            if (cnt[j] <= 1) continue;
            any = true;
            const long nb = (cnt[j] + 1) / 2;
            double *dst = nb == 1 ? st[j].out : pp[lvl & 1] + pp_off[j];
            const int nch = (st[j].nc + 15) >> 4;
            if (nb > cus || nch > 16) {
                sg.push_back({cur[j], cnt[j], st[j].nc, dst});
                L.single.push_back(j);
            } else {
                WyPairJob J;
                J.stack = cur[j];
                J.Rblk = blk + blk_off[j];
                J.Rout = dst;
                J.count = cnt[j];
                J.nc = st[j].nc;
                J.wg0 = L.nwg;
                if (L.grouped.empty()) L.table_at = table.size();
                table.push_back(J);
                L.grouped.push_back(j);
                L.nwg += (int)nb;
            }
            cur[j] = dst;
            cnt[j] = nb;
        }
        if (!any) break;
        levels.push_back(L);
        singles.push_back(sg);
    }
    const size_t tb = sizeof(WyPairJob) * table.size();
    const WyPairJob *d_table = nullptr;
    if (tb) {
        char *devp = static_cast<char *>(workspace(tb, 35));
        if (!devp) return FIGH_ERR_ALLOC;
        static std::vector<char> cached;
        static const char *cached_dev = nullptr;
        std::vector<char> blob(tb);
        std::memcpy(blob.data(), table.data(), tb);
        if (cached_dev != devp || cached != blob) {
            FIGH_HIP(hipMemcpyAsync(devp, blob.data(), tb, hipMemcpyHostToDevice, stream()));
            FIGH_HIP(hipStreamSynchronize(stream()));  // (blob is host memory of this call)
            cached = blob;
            cached_dev = devp;
        }
        d_table = reinterpret_cast<const WyPairJob *>(devp);
    }
    for (size_t l = 0; l < levels.size(); ++l) {
        ProfileScope scope("tsqr_reduce");
        for (const Single &s : singles[l])
            if (int rc = launch_tsqr_wide_pairs(s.src, s.count, s.nc, s.dst)) return rc;
        const Level &L = levels[l];
        if (!L.grouped.empty()) {
            hipLaunchKernelGGL((tsqr_wy_kernel<8, 2, 8, 2, false, 4>), dim3((unsigned)L.nwg), dim3(512), 0, stream(),
                               reinterpret_cast<const double *>(d_table + L.table_at), 0L, 0L, (const int *)nullptr, 0,
                               (const double *)nullptr, (const double *)nullptr, 1L, (double *)nullptr, (double *)nullptr, 0,
                               (long long *)nullptr, 0L, (int)L.grouped.size(), null_pivot_sq());
            FIGH_HIP(hipGetLastError());
        }
    }
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
