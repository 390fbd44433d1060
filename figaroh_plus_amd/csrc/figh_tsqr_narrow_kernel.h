// Level 0 of the register-tile TSQR (n <= 80): the body of tsqr2_kernel, shared by the one-matrix launch (figh_linalg.hip)
// and the grouped launch over the narrow row blocks of a tree's regressor (figh_tsqr_group.hip) -- `wave` / `tstep` are the
// wave's index and the number of waves working on THIS matrix (blockIdx.x / gridDim.x for the one-matrix launch).
#pragma once

#include <type_traits>

#include "figh_tsqr_narrow.h"

namespace figh {

template <int NCC, int NRC, bool LDSRED, bool RLAST = false>
__device__ __forceinline__ void tsqr2_level0_body(
    const double *__restrict__ W, const long rows, const long ldw, const int *__restrict__ col_idx, const int n,
    const double *__restrict__ tau, const double *__restrict__ blkw, const long rows_per_blk,
    double *__restrict__ Rws, const int nc, const int *__restrict__ tile_first, const long wave, const long tstep,
    const double null2) {
    // tile_first[t] (always a valid array; zeros without a structure hint, figh_tsqr_structured): the first kept column
    // that can hold a non-zero in tile t.  Lanes in front of it are not read at all (their registers are zeroed, the
    // loads run under a narrower EXEC mask): in the joint-major regressor of a chain, row block j only involves the
    // links >= j, so 41 % of the kept entries of UR10 -- and of this kernel's HBM reads -- are known zeros.  (The array is
    // unconditional on purpose: a `hint != nullptr` test inside the tile loop gets the loop unswitched and costs 70
    // spilled registers.)
    constexpr int RPL = 4 * NRC, M = 16 * NRC;
    extern __shared__ __attribute__((aligned(16))) double Rl[];  // packed triangle of the NCC panels
    const int lane = threadIdx.x;
    // Tiles are dealt round-robin (tile t -> wave t mod nwaves): in the joint-major row order the number of
    // non-zero leading columns, hence the work per tile, depends on the joint block, and contiguous ranges
    // would leave the waves of the last joints idle while those of joint 1 finish.
    const long ntiles = (rows + M - 1) / M;
    const long rend = rows;
    // The nc columns are RIGHT-aligned in the 16*NCC lane-columns (pad = 16 NCC - nc zero columns in front): a chunk
    // takes part in every step up to its last column, so the partially filled chunk must be the FIRST one -- for
    // nc = 50 the chunk holding 2 real columns is then live for 2 steps instead of 50 (-33 % chunk-steps on UR10).
    const int pad = 16 * NCC - nc;
    // LDS: [64 doubles of reduction scratch][packed triangle without the rows of the padding columns]
    // RLAST (figh_tsqr_narrow.h): the last 16 lane-columns of the triangle live in registers, 16 doubles of LDS pass a row of
    // them between the row groups, and the packed triangle has LCH = NCC - 1 chunks per row -- for 65 .. 80 columns 17 instead
    // of 28 KB per wave, i.e. two waves per SIMD instead of five per CU
    constexpr int LCH = RLAST ? NCC - 1 : NCC;
    constexpr int HEAD = RLAST ? 80 : 64;
    int skip = 0;
    for (int kp = 0; kp < pad; ++kp) skip += 16 * (LCH - (kp >> 4) > 0 ? LCH - (kp >> 4) : 0);
    Tsqr2State<NCC, NRC, RLAST> S;
    S.red = Rl;
    S.bc = Rl + 64;
    S.Rl = Rl + HEAD - skip;
    if constexpr (RLAST) {
#pragma unroll
        for (int sl = 0; sl < 4 * NCC; ++sl) S.Rq[sl] = 0.0;
    }
    S.lane_c = lane & 15;
    S.lane_g = lane >> 4;
    S.nc = nc;
    S.null2 = null2;
    // per-lane column sources: W[:, col_idx[col]] for col < n; tau is column n = nc - 1, i.e. lane-column 15 of the
    // last chunk; everything else (padding) is a dead lane-column whose registers stay exactly zero for the whole
    // kernel (zero data, zero R row entry => w_j = c_j = 0 in every step), so they are zeroed once and never loaded.
    bool wlive[NCC];
    int cidx[NCC], loff[NCC];
#pragma unroll
    for (int cc = 0; cc < NCC; ++cc) {
        const int col = 16 * cc + S.lane_c - pad;
        wlive[cc] = col >= 0 && col < n;
        cidx[cc] = wlive[cc] ? (col_idx ? col_idx[col] : col) : 0;
        loff[cc] = (int)(S.lane_g * ldw) + cidx[cc];  // the host side guarantees ldw < 2^24
    }
    const bool tau_lane = tau != nullptr && S.lane_c == 15;
#pragma unroll
    for (int cc = 0; cc < NCC; ++cc)
#pragma unroll
        for (int i = 0; i < RPL; ++i) S.T[cc][i] = 0.0;
    {
        constexpr int tot = 256 * (LCH * LCH - (LCH * (LCH - 1)) / 2);
        for (int e = lane; e < HEAD + tot - skip; e += 64) Rl[e] = 0.0;
    }
    __syncthreads();

    // Tile loads: lane (g, c) takes rows r0 + 16 rc + g + 4 reg of its column.  Full tiles use a wave-uniform row base
    // (SGPR pair) + a 32-bit per-lane element offset g*ldw + column: 16*NCC independent requests, all in flight at
    // once, under one EXEC mask per chunk.  The requests for the NEXT tile's chunk P are issued as soon as panel P of
    // the current tile is finished (its registers are dead from then on), so most of a tile's HBM latency is covered
    // by the wave's own remaining panels and only the last chunk (+ tau) is requested at the top of the iteration.
    auto load_chunk = [&](auto CC, const long r0, const int fpos) {
        constexpr int cc = decltype(CC)::value;
        // lanes whose column lies in front of the tile's first possible non-zero (hint) are not read: their registers
        // are zeroed and the loads run under the narrower EXEC mask
        if (fpos > 16 * cc) {
#pragma unroll
            for (int i = 0; i < RPL; ++i) S.T[cc][i] = 0.0;
        }
        if (wlive[cc] && 16 * cc + S.lane_c >= fpos) {
#pragma unroll
            for (int i = 0; i < RPL; ++i) S.T[cc][i] = (W + (r0 + 16 * (i >> 2) + 4 * (i & 3)) * ldw)[loff[cc]];
        }
    };
    auto load_head_chunks = [&](const long r0, const int fpos) {  // chunks 0 .. NCC-2
        if constexpr (NCC > 1) load_chunk(std::integral_constant<int, 0>{}, r0, fpos);
        if constexpr (NCC > 2) load_chunk(std::integral_constant<int, 1>{}, r0, fpos);
        if constexpr (NCC > 3) load_chunk(std::integral_constant<int, 2>{}, r0, fpos);
        if constexpr (NCC > 4) load_chunk(std::integral_constant<int, 3>{}, r0, fpos);
    };

    bool prefetched = false;
    int absorbed = 0;  // tiles this wave has factored
    // The SIMD arbiter favours the older of its two resident waves, which then finishes ~25 % earlier and leaves
    // the younger one running alone (at a single wave's issue efficiency) for the rest of the kernel.  The two
    // halves of the grid therefore alternate their issue priority per tile, in antiphase, so that both waves of a
    // SIMD progress at a more even rate and the SIMD stays doubly occupied for longer (measured 1.120 -> 1.091 ms;
    // in the paired phase the SIMD is issue-bound, so this only shortens the single-wave tail).
    // (no ties: the younger half of the grid stays at priority 1, the older half alternates 2 / 0 per tile)
    const bool younger = wave >= tstep / 2;
    int prio_phase = 0;
    if (younger) __builtin_amdgcn_s_setprio(1);
    // Tile order: the wave's k-th tile is not tile wave + k*nwaves itself but its image under an 8-way interleave of
    // the row range (position p -> tile (p mod 8) * ceil(ntiles/8) + p / 8).  In the joint-major row order whole
    // row blocks are either compute-bound (rows of joint 1: all columns non-zero) or HBM-bound (rows of the last
    // joints: a handful of column steps per 43 KB tile); with the plain order every wave walks through the blocks in
    // lockstep and the kernel is a compute-bound phase followed by a bandwidth-bound phase.  Interleaved, each SIMD
    // sees both kinds at any time and the two bounds overlap.
    constexpr int GI = 8;
    const long npg = (ntiles + GI - 1) / GI;
    auto tile_at = [&](const long p) { return (p % GI) * npg + p / GI; };  // may be >= ntiles
    long pos = wave;
    while (pos < GI * npg && tile_at(pos) >= ntiles) pos += tstep;
    while (pos < GI * npg) {
        long posn = pos + tstep;
        while (posn < GI * npg && tile_at(posn) >= ntiles) posn += tstep;
        const long t = tile_at(pos);
        const long tn = posn < GI * npg ? tile_at(posn) : ntiles;
        pos = posn;
        if (!younger) {
            if (prio_phase & 1) __builtin_amdgcn_s_setprio(0);
            else __builtin_amdgcn_s_setprio(2);
            ++prio_phase;
        }
        const long r0 = t * M;
        const long r0n = tn * M;
        const bool fast = r0 + M <= rend;
        const bool next_fast = r0n + M <= rend;
        const int fpos = fast ? __builtin_amdgcn_readfirstlane(pad + tile_first[t]) : 0;
        const int fposn = next_fast ? __builtin_amdgcn_readfirstlane(pad + tile_first[tn]) : 0;
        if (fast) {
            if (!prefetched) load_head_chunks(r0, fpos);
            // the last chunk (+ tau) is live until the end of the previous tile: requested here.  (Requesting it into
            // a separate 16-double buffer during the last panel was measured: the load phase shrinks, the panels
            // slow down by the same amount -- 256 VGPRs -- no net gain.)
            load_chunk(std::integral_constant<int, NCC - 1>{}, r0, fpos);
            if (tau_lane) {
#pragma unroll
                for (int i = 0; i < RPL; ++i) S.T[NCC - 1][i] = (tau + r0 + 16 * (i >> 2) + 4 * (i & 3))[S.lane_g];
            }
        } else {  // the last, partial tile: rows clamped to the last one, then masked
            const double *tb = W + r0 * ldw;
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const int rr = 16 * (i >> 2) + S.lane_g + 4 * (i & 3);
                const bool inb = r0 + rr < rend;
                const int rel = inb ? rr : (int)(rend - 1 - r0);
                const int ro = rel * (int)ldw;
#pragma unroll
                for (int cc = 0; cc < NCC; ++cc) {
                    double v = 0.0;
                    if (wlive[cc]) v = tb[ro + cidx[cc]];
                    if (cc == NCC - 1 && tau_lane) v = tau[r0 + rel];
                    S.T[cc][i] = inb ? v : 0.0;
                }
            }
        }
        if (blkw) {  // row-block weights (WLS): row r is scaled by blkw[r / rows_per_blk]
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const long row = r0 + 16 * (i >> 2) + S.lane_g + 4 * (i & 3);
                const double scale = blkw[(row < rend ? row : rend - 1) / rows_per_blk];
#pragma unroll
                for (int cc = 0; cc < NCC; ++cc) S.T[cc][i] *= scale;
            }
        }
        // zero-column map of the tile: bit = padded lane-column position with a non-zero entry
        unsigned long long nzlo = 0;
        unsigned nzhi = 0;
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc) {
            bool nz = false;
#pragma unroll
            for (int i = 0; i < RPL; ++i) nz |= (S.T[cc][i] != 0.0);
            const unsigned long long b = __ballot(nz);
            const unsigned m16 = (unsigned)((b | (b >> 16) | (b >> 32) | (b >> 48)) & 0xffffull);
            if (cc < 4) nzlo |= (unsigned long long)m16 << (16 * cc);
            else nzhi |= m16 << (16 * (cc - 4));
        }
        int first_nz = 16 * NCC;  // in padded lane-column positions
        if (nzlo) first_nz = __ffsll((long long)nzlo) - 1;
        else if (nzhi) first_nz = 64 + __ffs((int)nzhi) - 1;

        // null pivots only once the triangle is of full height: a tile that exhausts the rank of what has been absorbed so
        // far forms its last reflectors from small residuals, which leaves noise of 1e-11 (instead of 1e-13) in the columns
        // behind them -- harmless for a Householder step (the garbage reflectors of the dependent columns annihilate it),
        // but a null pivot would keep it as its |R_kk| (tools/null_pivot_noise.py)
        S.null2 = absorbed * M < nc + M / 8 ? 0.0 : null2;
        ++absorbed;
        tsqr2_panels<0, NCC, NRC, LDSRED, RLAST>(S, first_nz, [&](auto P) {
            if constexpr (decltype(P)::value < NCC - 1) {
                if (next_fast) load_chunk(P, r0n, fposn);
            }
        });
        prefetched = next_fast;
    }
    __syncthreads();
    double *Rg = Rws + wave * (long)nc * nc;
    const int nlds = 16 * LCH - pad;  // columns of the compact triangle that live in LDS
    for (int e = lane; e < nc * nc; e += 64) {
        const int k = e / nc, col = e - k * nc;
        if (RLAST && col >= nlds && col >= k) continue;  // (written from the registers below)
        const int kp = k + pad, colp = col + pad;  // padded positions
        const int pk = kp >> 4;
        Rg[e] = (k >= nc || col < k)  // below the diagonal the LDS rows hold rounding residues, not results
                    ? 0.0
                    : S.Rl[tsqr2_panel_off<LCH>(pk) + (kp & 15) * 16 * (LCH - pk) + (colp - 16 * pk)];
    }
    if constexpr (RLAST) {
        const int col = 16 * LCH + S.lane_c - pad;
#pragma unroll
        for (int sl = 0; sl < 4 * NCC; ++sl) {
            const int k = 4 * sl + S.lane_g - pad;
            if (k >= 0 && col >= k && col >= 0 && col < nc) Rg[(long)k * nc + col] = S.Rq[sl];
        }
    }
}

}  // namespace figh
