// Column steps of the register-tile TSQR (n <= 16*NCC <= 80 columns), shared by tsqr2_kernel (figh_linalg.hip) and the
// fused regressor + TSQR kernel (figh_fused.hip).
#pragma once

#include <type_traits>

#include "figh_wave.h"

#ifndef FIGH_NARROW_LDSTRAIL
#define FIGH_NARROW_LDSTRAIL 0
#endif

namespace figh {

// ------------------------------------------------------------------------------------------------------------
// tsqr2_kernel<NCC, NRC>: the n <= 16*NCC (<= 80) kernel.  The 16*NRC x 16*NCC tile sits in registers in the
// MFMA f64 C/D layout (figh_wave.h).  A column step needs
//   - the pivot column inside each row group: a DPP row_newbcast operand of v_fmac_f64 (no LDS crossbar),
//   - the dot products summed over the four row groups: 512 B of wave-private LDS,
// instead of 128 ds_bpermute per step (6.2 cycles each per CU, shared by the four SIMDs).  Finished chunks drop out
// of the update loops (the triangle's zero part costs nothing).
//
// RLAST (the fused kernel, figh_fused.hip: LDS is what limits the number of waves there): the last 16 lane-columns of the
// triangle live in REGISTERS -- row kp of that chunk in the lanes of row group kp % 4, register kp / 4 -- and only the other
// NCC - 1 chunks in LDS (7.4 KB instead of 13.8 KB for 50 columns).  A step passes its row of the register chunk to the
// other row groups through 128 B of LDS scratch (one masked ds_write + one ds_read, where the LDS form has a read and a
// masked write of the row itself).
template <int NCC, int NRC, bool RLAST = false>
struct Tsqr2State {
    static constexpr int RPL = 4 * NRC;  // rows per lane
    double T[NCC][RPL];
    double Rq[RLAST ? 4 * NCC : 1];  // RLAST: rows 4 s + lane_g of the last chunk's triangle columns
    double *Rl;   // LDS triangle (packed, biased so that the compile-time row offsets apply)
    double *red;  // LDS: 64 doubles of cross-row-group reduction scratch, private to the wave
    double *bc;   // RLAST: 16 doubles of LDS, the current row of the register chunk for all row groups
    int lane_c;   // lane & 15
    int lane_g;   // lane >> 4
    int nc;
    double null2;  // null-pivot rule (figh_tsqr_null_pivot_tol): threshold squared, 0 = only exact zeros
};

// doubles of LDS in front of row 0 of panel P in the packed triangle (LCH chunks per row of panel 0)
template <int LCH>
constexpr int tsqr2_panel_off(int P) { return 256 * (P * LCH - (P * (P - 1)) / 2); }

// The panel index P is a compile-time constant: the chunk registers T[P .. NCC-1] are addressed statically (no
// rotation copies), the number of live chunks is known, and a step is straight-line code -- after the pivot chunk's
// own dot product (the only input of the Householder scalars) the dot products of the trailing chunks and the LDS
// reads of row k are independent of the rsq/rcp chain and are interleaved with it by the scheduler.

template <bool LDSRED, int NCC, int NRC, bool RLAST>
__device__ __forceinline__ double tsqr2_reduce(Tsqr2State<NCC, NRC, RLAST> &S, const double x) {
    if constexpr (LDSRED) return allreduce_rowgroups_lds(S.red, 16 * S.lane_g + S.lane_c, x);
    else return allreduce_rowgroups(x);
}

template <int KK, int P, int NCC, int NRC, bool LDSRED, bool RLAST = false>
__device__ __forceinline__ void tsqr2_step(Tsqr2State<NCC, NRC, RLAST> &S) {
    constexpr int RPL = 4 * NRC;
    constexpr int LIVE = NCC - P;
    constexpr int NR = RPL;
    constexpr int LCH = RLAST ? NCC - 1 : NCC;  // chunks of the triangle kept in LDS
    constexpr int LLIVE = LCH - P;              // ... of which live in this panel (RLAST, last panel: none)
    // packed triangle: panel p keeps 16 rows of 16*(LCH-p) entries (columns 16p ..)
    constexpr int rowoff = tsqr2_panel_off<LCH>(P) + KK * 16 * LLIVE;
    constexpr int g0 = KK & 3, slot = 4 * P + (KK >> 2);  // RLAST: row 16 P + KK of the register chunk
    // The pivot column x = lane-column KK of chunk P is read in place through the DPP operand.
    double Rk[LIVE], d[LIVE];
    if constexpr (RLAST) {
        if (S.lane_g == g0) S.bc[S.lane_c] = S.Rq[slot];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        Rk[LIVE - 1] = S.bc[S.lane_c];
    }
#pragma unroll
    for (int cc = 0; cc < LLIVE; ++cc) Rk[cc] = S.Rl[rowoff + 16 * cc + S.lane_c];
    // the diagonal entry R_kk straight from LDS (one address for the whole wave: a broadcast read on the LDS port)
    // instead of a v_mov_b64_dpp of Rk[0] on the VALU (8 ticks)
    double alpha;
    if constexpr (RLAST && LLIVE == 0) alpha = S.bc[KK];
    else alpha = S.Rl[rowoff + KK];
    {
        double s0 = 0.0, s1 = 0.0;  // two chains (the second wave of the SIMD covers the FMA latency): 2 movs + 1 add
#pragma unroll
        for (int i = 0; i < NR; i += 2) {
            fmac_bcast_live<KK>(s0, S.T[P][i], S.T[P][i]);
            fmac_bcast_live<KK>(s1, S.T[P][i + 1], S.T[P][i + 1]);
        }
        d[0] = tsqr2_reduce<LDSRED>(S, s0 + s1);
    }
    // row k of the triangle is requested before the dot product and pinned here, so that the LDS latency is not
    // part of the dependent chain below (the compiler would otherwise sink the read below the sigma branch)
#pragma unroll
    for (int cc = 0; cc < LIVE; ++cc) asm volatile("" : "+v"(Rk[cc]));
    asm volatile("" : "+v"(alpha));
    const double sigma = row_bcast<KK>(d[0]);
    // s = sqrt(alpha^2 + sigma), beta = -sign(alpha) s, inv = 1/(alpha - beta) = sign(alpha)/(|alpha| + s),
    // tfac = (beta - alpha)/beta = (|alpha| + s)/s: v_rsq_f64 / v_rcp_f64 seeds (~2^-24) + ONE third-order step each
    // (y (1 + e/2 + 3e^2/8), e = 1 - q y^2: error e^3; r (1 + e + e^2), e = 1 - d r) instead of two Newton steps.
    // Round 5: (i) the reciprocal starts from the UNCORRECTED rsq seed and is corrected against the accurate |alpha| + s
    // (householder_scalars4, figh_wave.h): the v_rcp_f64 runs beside the correction of rs, 26 ticks off the dependent chain;
    // (ii) the chain starts BEFORE the zero-column test -- the ballot -> branch round trip (~60 ticks) runs in its shadow; a
    // zero column discards inf / NaN.  The asm pins the chain in front of the branch (the compiler sinks it otherwise).
    const double q2 = fma(alpha, alpha, sigma);
    // NULL PIVOT (dlarfg's H = I rule with a threshold, figh_tsqr_null_pivot_tol): the column is zero to working accuracy
    // at and below the diagonal -- a linearly dependent column of the regressor, whose residual is rounding noise in every
    // tile.  Its norm moves into R_kk (which therefore keeps the running residual norm of the column: the test is on
    // alpha^2 + sigma, so at most null2 of a column's energy is ever folded), the column leaves the tile, and no reflector
    // is formed: no trailing updates.  q2 is the same number in every lane.
    const double rs0 = __builtin_amdgcn_rsq(q2);
    double ri = __builtin_amdgcn_rcp(fma(q2, rs0, fabs(alpha)));
    double rs;
    {
        const double e = fma(-(q2 * rs0), rs0, 1.0);
        rs = fma(rs0, fma(e, 0.375, 0.5) * e, rs0);
    }
    const double dsum = fma(q2, rs, fabs(alpha));  // |alpha| + s
    {
        const double e = fma(-dsum, ri, 1.0);
        ri = fma(ri, fma(e, e, e), ri);
    }
    asm volatile("" : "+v"(ri), "+v"(rs));
    if (__builtin_amdgcn_ballot_w64(sigma != 0.0) == 0) return;  // column zero below the triangle: H = I (dlarfg)
    const double inv = copysign(ri, alpha);
    const double tfac = dsum * rs;
    const bool live = __builtin_amdgcn_ballot_w64(q2 > S.null2) != 0;
    // The trailing dot products stay OUTSIDE the test, where the scheduler interleaves them with the rsq / rcp chain as before
    // (inside it they cost the live steps 5 % and the null steps gain nothing net: same-box A/B, tools/null_ab.py); a null
    // pivot saves the trailing updates.  Every other arrangement tried -- the test in front of the chain, an else branch of
    // its own -- made the register allocator spill hundreds of registers.
#pragma unroll
    for (int cc = 1; cc < LIVE; ++cc) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int i = 0; i < NR; i += 2) {
            fmac_bcast<KK>(s0, S.T[P][i], S.T[P + cc][i]);
            fmac_bcast<KK>(s1, S.T[P][i + 1], S.T[P + cc][i + 1]);
        }
        // (FIGH_NARROW_LDSTRAIL, round-6 experiment on the fused launch: the sums of the TRAILING chunks -- off the step's dependent
        // chain -- through the wave's 512 B of LDS instead of four lane swaps: 3 VALU instructions instead of 10)
        d[cc] = tsqr2_reduce<LDSRED || (FIGH_NARROW_LDSTRAIL && RLAST)>(S, s0 + s1);
    }
    if (live) {
        // w_j = tau (R_kj + v^T B_j) for EVERY lane-column, no masks:
        //   - the pivot lane itself gets w = (alpha + sigma inv) tfac = alpha - beta, hence R_kk = alpha - w = beta and
        //     c = w inv = 1: its tile entries x - 1 x vanish (the finished column leaves the tile);
        //   - finished lane-columns (c < KK) and padding hold zeros (up to rounding residues that are never read as
        //     results), so their w is zero by itself.
        // Trailing chunks first, the pivot chunk last: its own update is the only write to the DPP source registers.
#pragma unroll
        for (int cc = LIVE - 1; cc >= 1; --cc) {
            const double wj = (Rk[cc] + d[cc] * inv) * tfac;
            const double ncj = -wj * inv;
#pragma unroll
            for (int i = 0; i < NR; ++i) fmac_bcast<KK>(S.T[P + cc][i], S.T[P][i], ncj);
            if (RLAST && cc == LIVE - 1) {
                if (S.lane_g == g0) S.Rq[slot] = Rk[cc] - wj;
            } else {
                if (S.lane_g == 0) S.Rl[rowoff + 16 * cc + S.lane_c] = Rk[cc] - wj;
            }
        }
    }
    {
        // the pivot chunk; null pivot: w = R_kk - sign(R_kk) s and c = 1 in the pivot lane, zero in the others
        const double wl = (Rk[0] + d[0] * inv) * tfac;
        const double wn = S.lane_c == KK ? alpha - copysign(q2 * rs, alpha) : 0.0;
        const double wj = live ? wl : wn;
        const double nn = S.lane_c == KK ? -1.0 : 0.0;
        const double ncj = live ? -wl * inv : nn;
#pragma unroll
        for (int i = 0; i < NR; ++i) fmac_bcast_live<KK>(S.T[P][i], S.T[P][i], ncj);
        if (RLAST && LIVE == 1) {
            if (S.lane_g == g0) S.Rq[slot] = Rk[0] - wj;
        } else {
            if (S.lane_g == 0) S.Rl[rowoff + S.lane_c] = Rk[0] - wj;
        }
    }
}

// all column steps of panel P, then the next panel (compile-time recursion over the panels).  after(P) runs when
// chunk P is retired (its registers are dead for the rest of the tile): the kernel requests the next tile's chunk P
// into them there.
template <int P, int NCC, int NRC, bool LDSRED, bool RLAST = false, class AfterPanel>
__device__ __forceinline__ void tsqr2_panels(Tsqr2State<NCC, NRC, RLAST> &S, const int first_nz, AfterPanel &&after) {
    if (16 * P + 15 >= first_nz) {
#define FIGH_STEP(KK) \
    if (16 * P + KK >= first_nz) tsqr2_step<KK, P, NCC, NRC, LDSRED, RLAST>(S);
        FIGH_STEP(0) FIGH_STEP(1) FIGH_STEP(2) FIGH_STEP(3) FIGH_STEP(4) FIGH_STEP(5) FIGH_STEP(6) FIGH_STEP(7)
        FIGH_STEP(8) FIGH_STEP(9) FIGH_STEP(10) FIGH_STEP(11) FIGH_STEP(12) FIGH_STEP(13) FIGH_STEP(14) FIGH_STEP(15)
#undef FIGH_STEP
    }
    after(std::integral_constant<int, P>{});
    if constexpr (P + 1 < NCC) tsqr2_panels<P + 1, NCC, NRC, LDSRED, RLAST>(S, first_nz, after);
}

// ------------------------------------------------------------------------------------------------------------
// tsqr_coop_kernel<NCC, NW>: the merge levels.  One workgroup of NW waves factors NW*64 stacked rows (about
// NW*64/nc triangles) in ONE sweep of column steps instead of NW sequential 64-row tiles: every wave keeps its
// own 64-row tile in registers, forms its part of x^T B, and the per-wave partial sums are combined through LDS
// (fixed order: bit-reproducible).  A merge level is latency-bound (few waves on the chip), so its time is the
// number of dependent column steps: nc per level here, against fan*nc for one wave walking `fan` tiles.
// The triangle being built starts empty (alpha = 0), so row k is final at step k and is written straight out.
template <int KK, int P, int NCC, int NW>
__device__ __forceinline__ void tsqr_coop_step(double (&T)[NCC][16], const int nc, const int pad, const int lane_c,
                                               const int lane_g, const int wave, double (*pw)[NW][16 * NCC],
                                               double *__restrict__ Rg) {
    constexpr int LIVE = NCC - P;
    constexpr int kpos = 16 * P + KK;  // padded position of the pivot column
    const int buf = kpos & 1;
    // per-wave partial dot products of the pivot column (read in place through the DPP operand) with the live chunks
#pragma unroll
    for (int cc = 0; cc < LIVE; ++cc) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            fmac_bcast<KK>(s0, T[P][i], T[P + cc][i]);
            fmac_bcast<KK>(s1, T[P][i + 1], T[P + cc][i + 1]);
            fmac_bcast<KK>(s2, T[P][i + 2], T[P + cc][i + 2]);
            fmac_bcast<KK>(s3, T[P][i + 3], T[P + cc][i + 3]);
        }
        const double dw = allreduce_rowgroups((s0 + s1) + (s2 + s3));
        if (lane_g == 0) pw[buf][wave][16 * (P + cc) + lane_c] = dw;
    }
    __syncthreads();
    // every wave sums the NW partials itself, in wave order (bit-reproducible): one barrier per column step; the
    // partial buffers ping-pong so that the next step's stores cannot overtake a slow reader of this one
    double d[LIVE];
#pragma unroll
    for (int cc = 0; cc < LIVE; ++cc) {
        double s = pw[buf][0][16 * (P + cc) + lane_c];
#pragma unroll
        for (int w = 1; w < NW; ++w) s += pw[buf][w][16 * (P + cc) + lane_c];
        d[cc] = s;
    }
    const double sigma = row_bcast<KK>(d[0]);
    if (__builtin_amdgcn_ballot_w64(sigma != 0.0) == 0) return;  // uniform over the workgroup: same totals in every wave
    // the output triangle starts empty, so alpha = 0: beta = -s, v = x / s, tau = 1
    const double hq = -0.5 * sigma;
    double rs = __builtin_amdgcn_rsq(sigma);
    rs = rs * fma(hq * rs, rs, 1.5);
    rs = rs * fma(hq * rs, rs, 1.5);
    // w_j = x^T B_j / s for every lane-column (no masks): the pivot lane gets w = s, c = 1 and cancels itself, its
    // R entry is -w = beta; finished and padding lane-columns hold (near) zeros.  Pivot chunk last: DPP source.
#pragma unroll
    for (int cc = LIVE - 1; cc >= 0; --cc) {
        const double wj = d[cc] * rs;
        const double ncj = -wj * rs;
#pragma unroll
        for (int i = 0; i < 16; ++i) fmac_bcast<KK>(T[P + cc][i], T[P][i], ncj);
        const int col = 16 * (P + cc) + lane_c - pad;
        if (wave == 0 && lane_g == 0 && col >= kpos - pad) Rg[(long)(kpos - pad) * nc + col] = -wj;
    }
}

template <int P, int NCC, int NW>
__device__ __forceinline__ void tsqr_coop_panels(double (&T)[NCC][16], const int nc, const int pad, const int lane_c,
                                                 const int lane_g, const int wave, double (*pw)[NW][16 * NCC],
                                                 double *__restrict__ Rg) {
#define FIGH_CSTEP(KK) \
    if (16 * P + KK >= pad) tsqr_coop_step<KK, P, NCC, NW>(T, nc, pad, lane_c, lane_g, wave, pw, Rg);
    FIGH_CSTEP(0) FIGH_CSTEP(1) FIGH_CSTEP(2) FIGH_CSTEP(3) FIGH_CSTEP(4) FIGH_CSTEP(5) FIGH_CSTEP(6) FIGH_CSTEP(7)
    FIGH_CSTEP(8) FIGH_CSTEP(9) FIGH_CSTEP(10) FIGH_CSTEP(11) FIGH_CSTEP(12) FIGH_CSTEP(13) FIGH_CSTEP(14)
    FIGH_CSTEP(15)
#undef FIGH_CSTEP
    if constexpr (P + 1 < NCC) tsqr_coop_panels<P + 1, NCC, NW>(T, nc, pad, lane_c, lane_g, wave, pw, Rg);
}


}  // namespace figh
