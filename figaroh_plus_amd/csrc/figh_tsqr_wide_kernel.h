// Device code of the blocked TSQR, shared by figh_tsqr_wide.hip (level 0) and figh_tsqr_wide_pair.hip (pair-merge
// levels): two translation units so that the two sets of instantiations compile in parallel.
#pragma once
// K3, wide form -- blocked (compact-WY) Householder TSQR for 80 < nc <= 512 columns on gfx950.
//
// Replaces np.linalg.qr of src/figaroh/tools/qrdecomposition.py:205,238 (and :105,286) for the regressors whose kept
// columns do not fit one wavefront's register tile: TIAGo (240 + tau), TALOS (330 + tau), human (190 + tau; 400 for the
// SIP program, identification_tools.py:528-531).  Like the narrow kernel it streams the rows of W once and keeps only
// the triangle ("triangle on top of a rectangle", LAPACK tpqrt), but the work is organised as 16-column panels:
//
//   workgroup = NW waves, tile = M = 16*NRC rows x nc columns in registers, COLUMN-split: wave w owns the 16-column
//   chunks w, w + NW, ... (CPW per wave) in the MFMA f64 C/D layout (figh_wave.h).
//
//   panel p (owner = wave p mod NW, VALU):  the 16 columns of chunk p are factored against the diagonal block R_pp
//       column by column -- pivot column through a DPP row_newbcast operand of v_fmac_f64, sums over the four row groups
//       through wave-private LDS, rsq/rcp + Newton for the Householder scalars -- and the T factor of the compact-WY
//       form H_0 ... H_15 = I - U T U^T, U = [I; V], is accumulated on the fly (LAPACK larft, forward/columnwise) from
//       the Gram entries v_m^T v_k that the panel's own dot products deliver.  V (M x 16) and T (16 x 16) are published
//       in LDS (ping-pong buffers).
//   trailing update (all waves, matrix pipe):  every chunk cc > p gets
//           G  = R_p,cc + V^T B_cc      4*NRC  v_mfma_f64_16x16x4   (A = V, B = tile chunk: register r of a row chunk
//                                                                    IS K-slice r in the C/D layout)
//           Wm = T^T G                  4      v_mfma_f64_16x16x4
//           R_p,cc -= Wm ;  B_cc -= V Wm   4*NRC  v_mfma_f64_16x16x4   (V read transposed from LDS)
//       i.e. 36 MFMAs = 73.7 kflop per 64 x 16 chunk against 65.5 kflop algorithmic -- no cross-row reductions, no
//       per-column barrier, no pivot broadcast for 89 % of the arithmetic.
//   look-ahead:  in phase p the owner of chunk p+1 updates that chunk FIRST and factors panel p+1 at once, while the
//       other waves are still applying panel p; one workgroup barrier per PANEL (not per column).  The panel is a
//       dependent chain of 16 column steps (latency-bound VALU work); a second workgroup on the same CU (two waves per
//       SIMD) fills the matrix pipe meanwhile.
//
// The triangle lives in global memory as packed 16 x 16 blocks (2 KB, block (p, cc) at index cc (cc+1)/2 + p).  Block
// (p, cc) is only ever touched by the owner of chunk cc, with one fixed lane -> element mapping (element lane + 64 r),
// so every access to R is thread-private: no fences, perfectly coalesced 512-byte requests.  Retired chunk registers
// are refilled with the next tile's chunk (loads in flight during the remaining phases).
//
// fp64 throughout (the rank decision |R_kk| > 1e-8 on dependent pivots needs Householder's eps*||col||, see
// figh_linalg.hip); roofline = the 78.6 TFLOP/s fp64 matrix peak, algorithmic flops 2 m nc^2 per m rows.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "figh_internal.h"
#include "figh_wave.h"

namespace figh {

namespace {

// in-kernel s_memtime accounting, ablation build only (FIGH_WY_PROF=1): per wave {kernel, tile top, first panel,
// look-ahead chunk update, look-ahead panel, trailing updates, barrier waits}
#ifdef FIGH_ABLATION
#define FIGH_PROF_DECL long long pc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long pt_ = __builtin_readcyclecounter(); const long long pbegin_ = pt_;
#define FIGH_PROF_ADD(k) do { const long long now_ = __builtin_readcyclecounter(); pc_[k] += now_ - pt_; pt_ = now_; } while (0)
#define FIGH_PROF_STORE(ptr, w, nw) do { if ((ptr) && lane == 0) { pc_[0] = __builtin_readcyclecounter() - pbegin_; for (int k_ = 0; k_ < 12; ++k_) (ptr)[((long)blockIdx.x * (nw) + (w)) * 12 + k_] = pc_[k_]; } } while (0)
#else
#define FIGH_PROF_DECL
#define FIGH_PROF_ADD(k) do {} while (0)
#define FIGH_PROF_STORE(ptr, w, nw) do {} while (0)
#endif

#ifdef FIGH_ABLATION
__device__ int g_wy_ralias = 0;
__device__ int g_wy_off = 0;  // FIGH_WY_OFF: wave relabelling per workgroup (which SIMD hosts the owner of panel p)
__device__ int g_wy_delay_mode = 0, g_wy_delay_ticks = 0;  // FIGH_WY_DELAY="mode,ticks": start-up delay of some workgroups
#endif

#ifndef FIGH_WY_TMODE
#define FIGH_WY_TMODE 0  // where T is formed: see wy_panel_step (0: inside the column steps; 2: once per panel -- measured in round 6, not faster)
#endif

constexpr int kLdv = 17;  // LDS row stride of V (doubles): the transposed reads of B -= V Wm hit 16 different banks

constexpr int kLdt = 17;  // LDS row stride of T (doubles): T[row][col] at row * kLdt + col

// Column KK of T, larft forward/columnwise: T[c][KK] = -tau_KK sum_{m < KK} T[c][m] (v_m^T v_KK) for c < KK.  Lane c
// reads row c of T from LDS (entries m < KK, written by earlier steps) and takes (v_m^T v_KK) / inv_KK = vg of
// lane-column m through the DPP operand of the FMA -- one instruction per term, no broadcast registers.  In two halves
// so that at most eight row entries are in registers at a time.
template <int M0, int M1>
struct TColumn {
    static __device__ __forceinline__ void load(double (&tr)[8], const double *__restrict__ Trow_lds) {
        tr[M0 & 7] = Trow_lds[M0];
        TColumn<M0 + 1, M1>::load(tr, Trow_lds);
    }
    static __device__ __forceinline__ void dot(double &acc0, double &acc1, const double (&tr)[8], const double vg) {
        if constexpr (M0 & 1) fmac_bcast<M0>(acc1, vg, tr[M0 & 7]);
        else fmac_bcast<M0>(acc0, vg, tr[M0 & 7]);
        TColumn<M0 + 1, M1>::dot(acc0, acc1, tr, vg);
    }
};
template <int M1>
struct TColumn<M1, M1> {
    static __device__ __forceinline__ void load(double (&)[8], const double *__restrict__) {}
    static __device__ __forceinline__ void dot(double &, double &, const double (&)[8], const double) {}
};

#ifdef FIGH_WY_LDSRED
#define FIGH_WY_REDUCE(red, lane, x) allreduce_rowgroups_lds(red, lane, x)
#else
#define FIGH_WY_REDUCE(red, lane, x) allreduce_rowgroups(x)
#endif

// One column step of a panel.  X = the panel's chunk (lane (g, c): rows 16 rc + 4 r + g of column c, i = 4 rc + r),
// Rl = the 16 x 16 diagonal block in wave-private LDS (row-major), Tl = the T factor being built (LDS, zero-filled),
// myinv = 1 / (alpha - beta) of reflector c (0 until column c has been factored).  Columns c < KK are finished
// reflectors and stay frozen (they are V, up to the scaling by myinv).
// TMODE (round 6): 0 = column KK of T is formed INSIDE the step (larft forward, as described above: 15 DPP FMAs + the LDS
// reads of T's row per step -- 106 of a step's 753 ticks, tools/microbench/step_bench.hip); 2 = the step only leaves the Gram
// entries v_m^T v_KK (m < KK) and tau_KK in column KK of the T buffer and wy_larft_rows turns the buffer into T once per panel;
// 1 = Gram entries only, no T at all (microbenchmark: the bound on what 2 can gain).
template <int KK, int RPL, int TMODE = 0>
__device__ __forceinline__ void wy_panel_step(double (&X)[RPL], double &myinv, double *__restrict__ Rl,
                                              double *__restrict__ Tl, double *__restrict__ red, const int lane,
                                              const int c_, const double null2) {
    // (the lane's column, opaque per step: the comparisons with KK below are then two v_cmp here -- hoisted out of the sixteen
    // steps they become 32 lane masks in SGPR pairs that the allocator parks in VGPR lanes and fetches back with four
    // v_readlane per step)
    int c = c_;
    asm volatile("" : "+v"(c));
    double rk = Rl[KK * 16 + c];  // row KK of the diagonal block: requested before the dot products
    constexpr int KH = KK < 8 ? KK : 8;
    double tr[8];
    if constexpr (TMODE == 0) TColumn<0, KH>::load(tr, Tl + c * kLdt);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int i = 0; i < RPL; i += 4) {
        fmac_bcast<KK>(s0, X[i], X[i]);
        fmac_bcast<KK>(s1, X[i + 1], X[i + 1]);
        fmac_bcast<KK>(s2, X[i + 2], X[i + 2]);
        fmac_bcast<KK>(s3, X[i + 3], X[i + 3]);
    }
    const double d = FIGH_WY_REDUCE(red, lane, (s0 + s1) + (s2 + s3));  // x^T X[:, c], identical in all row groups
    asm volatile("" : "+v"(rk));
    const double sigma = row_bcast<KK>(d);
    const double alpha = row_bcast<KK>(rk);
    // column zero below the triangle: H = I (dlarfg) -- or NULL PIVOT (figh_tsqr_null_pivot_tol, see tsqr2_step in
    // figh_tsqr_narrow.h): zero to working accuracy at and below the diagonal.  H = I as well: tau = 0 (column KK of T stays
    // zero), v = 0 (myinv stays zero: a zero column of V and of every later Gram entry), no update; the column's entries stay
    // in X untouched and wy_factor_panel folds their norm into R_kk after the last step.  A third of the dependent chain of a
    // step instead of all of it.
    if (__builtin_amdgcn_ballot_w64(sigma != 0.0 && fma(alpha, alpha, sigma) > null2) == 0) return;
    // (x_m^T x_KK) inv_m in lane-column m < KK, zero elsewhere (myinv is still zero for the columns not yet factored)
    double vg = d * myinv;
    asm volatile("s_nop 1" : "+v"(vg));  // VALU write -> DPP read of vg below: 2 wait states
    double acc0 = 0.0, acc1 = 0.0;
    if constexpr (TMODE == 0) {
        TColumn<0, KH>::dot(acc0, acc1, tr, vg);
        if constexpr (KK > 8) {
            TColumn<8, KK>::load(tr, Tl + c * kLdt);
            TColumn<8, KK>::dot(acc0, acc1, tr, vg);
        }
    }
    // (round 5: the reciprocal of the scalars runs beside the rsq correction, householder_scalars4; starting the whole chain
    // in front of the test -- as the register-tile step now does -- costs this kernel 12 .. 16 more bytes of scratch in the
    // panel chain and gained nothing: TALOS level 0 2 x 51.4 ms against 82.4 + 18.9)
    double inv, tfac;
    householder_scalars4(alpha, sigma, inv, tfac);
    // w_c = tau (R_kc + v^T X_c); the pivot lane gets w = alpha - beta, i.e. R_kk = alpha - w = beta
    // (frozen columns c < KK: w = 0; their row entry rk is a structural zero of the diagonal block and stays one)
    const double wj = (c >= KK) ? (rk + d * inv) * tfac : 0.0;
    const double ncj = (c == KK) ? 0.0 : -wj * inv;
#pragma unroll
    for (int i = 0; i < RPL; ++i) fmac_bcast<KK>(X[i], X[i], ncj);
    // row KK of the block and column KK of T are written by the whole first row group, no per-step lane masks: the
    // sum of T's row entries is zero by itself for c >= KK (T is upper triangular, vg is zero there)
    if (lane < 16) {
        Rl[KK * 16 + c] = rk - wj;
        if constexpr (TMODE == 0) Tl[c * kLdt + KK] = fma(-tfac * inv, acc0 + acc1, (c == KK) ? tfac : 0.0);
        else Tl[c * kLdt + KK] = (c == KK) ? tfac : vg * inv;  // v_c^T v_KK for c < KK (vg is zero from the diagonal on), tau_KK
    }
    myinv = (c == KK) ? inv : myinv;
    // the next step reads X through DPP operands of inline asm, which the hazard recognizer cannot see: nothing of it
    // may be scheduled in between this step's updates (a VALU write needs 2 wait states before a DPP read)
    __builtin_amdgcn_sched_barrier(0);
}

template <int M, int K>
struct LarftAxpy {  // Tr[k] += S[M][k] Tr[M] for k = K .. 15: independent FMAs, S[M][k] = lane-column M of the loaded column k
    static __device__ __forceinline__ void run(const double (&col)[16], double (&Tr)[16]) {
        if constexpr (K < 16) {
            fmac_bcast<M>(Tr[K], col[K], Tr[M]);
            LarftAxpy<M, K + 1>::run(col, Tr);
        }
    }
};
// right-looking: once T[r][M] is final it is added into every later column's sum at once -- the dependent chain is sixteen
// short links (finalise, one FMA), not the 120 FMAs of the row-by-row dot products (measured: 2000 against 1900 ticks saved)
template <int M>
__device__ __forceinline__ void larft_row(const double (&col)[16], double (&Tr)[16], const int c) {
    const double tau = row_bcast<M>(col[M]);
    Tr[M] = (c == M) ? tau : -tau * Tr[M];
    if constexpr (M < 15) {
        LarftAxpy<M, M + 1>::run(col, Tr);
        larft_row<M + 1>(col, Tr, c);
    }
}

// TMODE 2: the T buffer holds S = strict upper part of V^T V with tau on the diagonal (a null pivot left its column zero);
// larft forward, T[r][k] = -tau_k sum_{m = r .. k-1} T[r][m] S[m][k], T[k][k] = tau_k, is a recurrence along each ROW: lane r keeps
// its row of T in registers and takes S[m][k] from lane-column m of the loaded column k through the DPP operand of the FMA -- 120
// FMAs per lane, once per panel, in place of 120 + the row reloads spread over the sixteen column steps.  Every row group
// computes the same rows; the first one writes T back over S.
__device__ __forceinline__ void wy_larft_rows(double *__restrict__ Tl, const int lane, const int c) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double col[16], Tr[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) col[k] = Tl[c * kLdt + k];  // lane c: S[c][k] (c < k), tau_k (c == k), 0 (c > k)
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(col[k]));
    asm volatile("s_nop 1" ::: "memory");
#pragma unroll
    for (int k = 0; k < 16; ++k) Tr[k] = 0.0;
    larft_row<0>(col, Tr, c);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // (every lane has its column entries before the first one is overwritten)
    if (lane < 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) Tl[c * kLdt + k] = Tr[k];
    }
}

// Factor one panel: on return Rl holds the new diagonal block, Vl (M x kLdv) the reflectors V = X diag(inv), Tl (16 x
// kLdt, row-major) the T factor.
template <int RPL, int TMODE = FIGH_WY_TMODE>
__device__ __forceinline__ void wy_factor_panel(double (&X)[RPL], double *__restrict__ Rl, double *__restrict__ red,
                                                double *__restrict__ Vl, double *__restrict__ Tl, const int lane,
                                                const int c, const int g, const double null2) {
    double myinv = 0.0;
    {
        // (a zero made HERE: hoisted out of the tile loop the constant lived in a register pair for the whole kernel, and the
        // allocator spilled exactly that pair -- a scratch reload of 0.0 at the head of every panel chain)
        double zero = 0.0;
        asm volatile("" : "+v"(zero));
#pragma unroll
        for (int e = lane; e < 16 * kLdt; e += 64) Tl[e] = zero;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // X may still be in flight from the matrix pipe or from LDS, and its first readers are DPP operands of inline asm:
    // the required wait states (VALU write -> DPP read) are not inserted by the compiler for asm
#pragma unroll
    for (int i = 0; i < RPL; ++i) asm volatile("" : "+v"(X[i]));
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    wy_panel_step<0, RPL, TMODE>(X, myinv, Rl, Tl, red, lane, c, null2);
    wy_panel_step<1, RPL, TMODE>(X, myinv, Rl, Tl, red, lane, c, null2);
    wy_panel_step<2, RPL, TMODE>(X, myinv, Rl, Tl, red, lane, c, null2);
    wy_panel_step<3, RPL, TMODE>(X, myinv, Rl, Tl, red, lane, c, null2);
    wy_panel_step<4, RPL, TMODE>(X, myinv, Rl, Tl, red, lane, c, null2);
    wy_panel_step<5, RPL, TMODE>(X, myinv, Rl, Tl, red, lane, c, null2);
    wy_panel_step<6, RPL, TMODE>(X, myinv, Rl, Tl, red, lane, c, null2);
    wy_panel_step<7, RPL, TMODE>(X, myinv, Rl, Tl, red, lane, c, null2);
    wy_panel_step<8, RPL, TMODE>(X, myinv, Rl, Tl, red, lane, c, null2);
    wy_panel_step<9, RPL, TMODE>(X, myinv, Rl, Tl, red, lane, c, null2);
    wy_panel_step<10, RPL, TMODE>(X, myinv, Rl, Tl, red, lane, c, null2);
    wy_panel_step<11, RPL, TMODE>(X, myinv, Rl, Tl, red, lane, c, null2);
    wy_panel_step<12, RPL, TMODE>(X, myinv, Rl, Tl, red, lane, c, null2);
    wy_panel_step<13, RPL, TMODE>(X, myinv, Rl, Tl, red, lane, c, null2);
    wy_panel_step<14, RPL, TMODE>(X, myinv, Rl, Tl, red, lane, c, null2);
    wy_panel_step<15, RPL, TMODE>(X, myinv, Rl, Tl, red, lane, c, null2);
    // null pivots: the norm of what the column still holds below the triangle moves into R_kk (all sixteen lane-columns at
    // once; nothing to do for columns that formed a reflector or are exactly zero)
    if (null2 > 0.0) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int i = 0; i < RPL; i += 2) {
            s0 = fma(X[i], X[i], s0);
            s1 = fma(X[i + 1], X[i + 1], s1);
        }
        const double sc = FIGH_WY_REDUCE(red, lane, s0 + s1);
        if (lane < 16 && myinv == 0.0 && sc != 0.0) {
            const double r = Rl[c * 16 + c];
            const double q2 = fma(r, r, sc);
            double rs = __builtin_amdgcn_rsq(q2);
            const double e = fma(-(q2 * rs), rs, 1.0);
            rs = fma(rs, fma(e, 0.375, 0.5) * e, rs);
            Rl[c * 16 + c] = copysign(q2 * rs, r);
        }
    }
#pragma unroll
    for (int i = 0; i < RPL; ++i) Vl[(16 * (i >> 2) + 4 * (i & 3) + g) * kLdv + c] = X[i] * myinv;
    if constexpr (TMODE == 2) wy_larft_rows(Tl, lane, c);
}

// Apply the panel's block reflector to one trailing chunk B (NRC row chunks of 16 x 16, C/D layout) and to its block of
// the triangle (Rpt = the block's current content, element lane + 64 r = row g + 4 r, column c; the new content goes
// to Rblock).
template <int NRC>
__device__ __forceinline__ void wy_update_chunk(f64x4 (&B)[NRC], const double *__restrict__ Vl,
                                                const double *__restrict__ Tl, const f64x4 Rpt,
                                                double *__restrict__ Rblock, const int lane, const int c, const int g) {
    // OPERANDS ARE STREAMED, one row chunk ahead of the MFMAs that use them (four MFMAs = 256 cycles cover the LDS round
    // trip): the window is 8 + 2 NRC doubles instead of the 8 NRC doubles of "everything up front", and the registers
    // that frees go into taller tiles -- rows per look-ahead chain are what the kernel's throughput is proportional to.
    double tt[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) tt[r] = Tl[(g + 4 * r) * kLdt + c];
    double vc[NRC][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) vc[0][r] = Vl[(4 * r + g) * kLdv + c];
    f64x4 G0 = {0.0, 0.0, 0.0, 0.0}, G1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int rc = 0; rc < NRC; ++rc) {
        if (rc + 1 < NRC) {
#pragma unroll
            for (int r = 0; r < 4; ++r) vc[rc + 1][r] = Vl[(16 * (rc + 1) + 4 * r + g) * kLdv + c];
        }
        // A[i = c][k = g] = V[row 16 rc + 4 r + g][c], B[k = g][j = c] = the tile entry of the same row: K-slice r
        G0 = __builtin_amdgcn_mfma_f64_16x16x4f64(vc[rc][0], B[rc][0], G0, 0, 0, 0);
        G1 = __builtin_amdgcn_mfma_f64_16x16x4f64(vc[rc][1], B[rc][1], G1, 0, 0, 0);
        G0 = __builtin_amdgcn_mfma_f64_16x16x4f64(vc[rc][2], B[rc][2], G0, 0, 0, 0);
        G1 = __builtin_amdgcn_mfma_f64_16x16x4f64(vc[rc][3], B[rc][3], G1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);  // keeps the window: without it every operand is fetched before the first MFMA
    }
    // operands of the last stage (V transposed), K-slice 0: in flight while the first stage drains and the T stage runs
    double vr[4][NRC];
#pragma unroll
    for (int rc = 0; rc < NRC; ++rc) vr[0][rc] = Vl[(16 * rc + c) * kLdv + g];
    const f64x4 G = (G0 + G1) + Rpt;  // G[r] = row g + 4 r of R_p,cc + V^T B
    // Wm = T^T G: A[i = c][k = g + 4 r] = T[g + 4 r][c], B[k][j = c] = G[g + 4 r][c]; two accumulators
    const f64x4 zero = {0.0, 0.0, 0.0, 0.0};
    f64x4 W0 = __builtin_amdgcn_mfma_f64_16x16x4f64(tt[0], G[0], zero, 0, 0, 0);
    f64x4 W1 = __builtin_amdgcn_mfma_f64_16x16x4f64(tt[1], G[1], zero, 0, 0, 0);
    W0 = __builtin_amdgcn_mfma_f64_16x16x4f64(tt[2], G[2], W0, 0, 0, 0);
    W1 = __builtin_amdgcn_mfma_f64_16x16x4f64(tt[3], G[3], W1, 0, 0, 0);
    const f64x4 Wm = W0 + W1;
#pragma unroll
    for (int r = 0; r < 4; ++r) Rblock[lane + 64 * r] = Rpt[r] - Wm[r];
    const f64x4 Wn = -Wm;
    __builtin_amdgcn_sched_barrier(0);
    // B -= V Wm: A[i = c][k = g + 4 s] = V[row 16 rc + c][g + 4 s], B[k][j = c] = -Wm[g + 4 s][c]; the NRC row chunks
    // are independent accumulators, so consecutive MFMAs never wait for each other
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (s + 1 < 4) {
#pragma unroll
            for (int rc = 0; rc < NRC; ++rc) vr[s + 1][rc] = Vl[(16 * rc + c) * kLdv + g + 4 * (s + 1)];
        }
#pragma unroll
        for (int rc = 0; rc < NRC; ++rc)
            B[rc] = __builtin_amdgcn_mfma_f64_16x16x4f64(vr[s][rc], Wn[s], B[rc], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The same update for a chunk that lives in LDS (lch[64 i + lane] = tile entry i of this lane: rows 16 rc + 4 r + g, i =
// 4 rc + r, column c): one 16-row chunk at a time through eight registers, read twice (V^T B, then B -= V Wm).
template <int NRC>
__device__ __forceinline__ void wy_update_lds_chunk(double *__restrict__ lch, const double *__restrict__ Vl,
                                                    const double *__restrict__ Tl, double *__restrict__ Rblock,
                                                    const int lane, const int c, const int g) {
    f64x4 Rpt;
#pragma unroll
    for (int r = 0; r < 4; ++r) Rpt[r] = Rblock[lane + 64 * r];
    double tt[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) tt[r] = Tl[(g + 4 * r) * kLdt + c];
    f64x4 G0 = {0.0, 0.0, 0.0, 0.0}, G1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int rc = 0; rc < NRC; ++rc) {
        double vc[4], b[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            vc[r] = Vl[(16 * rc + 4 * r + g) * kLdv + c];
            b[r] = lch[64 * (4 * rc + r) + lane];
        }
        G0 = __builtin_amdgcn_mfma_f64_16x16x4f64(vc[0], b[0], G0, 0, 0, 0);
        G1 = __builtin_amdgcn_mfma_f64_16x16x4f64(vc[1], b[1], G1, 0, 0, 0);
        G0 = __builtin_amdgcn_mfma_f64_16x16x4f64(vc[2], b[2], G0, 0, 0, 0);
        G1 = __builtin_amdgcn_mfma_f64_16x16x4f64(vc[3], b[3], G1, 0, 0, 0);
    }
    const f64x4 G = (G0 + G1) + Rpt;
    const f64x4 zero = {0.0, 0.0, 0.0, 0.0};
    f64x4 W0 = __builtin_amdgcn_mfma_f64_16x16x4f64(tt[0], G[0], zero, 0, 0, 0);
    f64x4 W1 = __builtin_amdgcn_mfma_f64_16x16x4f64(tt[1], G[1], zero, 0, 0, 0);
    W0 = __builtin_amdgcn_mfma_f64_16x16x4f64(tt[2], G[2], W0, 0, 0, 0);
    W1 = __builtin_amdgcn_mfma_f64_16x16x4f64(tt[3], G[3], W1, 0, 0, 0);
    const f64x4 Wm = W0 + W1;
#pragma unroll
    for (int r = 0; r < 4; ++r) Rblock[lane + 64 * r] = Rpt[r] - Wm[r];
    const f64x4 Wn = -Wm;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rc = 0; rc < NRC; ++rc) {
        double vr[4];
        f64x4 b;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            vr[s] = Vl[(16 * rc + c) * kLdv + g + 4 * s];
            b[s] = lch[64 * (4 * rc + s) + lane];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) b = __builtin_amdgcn_mfma_f64_16x16x4f64(vr[s], Wn[s], b, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) lch[64 * (4 * rc + s) + lane] = b[s];
        __builtin_amdgcn_sched_barrier(0);
    }
}

// WPE = waves per SIMD the kernel is built for (register budget 512 / WPE).
//
// Register queue.  A wave's CPW chunks live in a queue of register slots: F holds the wave's next chunk to become a panel
// (chunk index `front`), Q[j] the chunk front + NW (j + 1).  When a chunk retires (its panel is factored, or it lies in
// front of the tile's first non-zero column) the queue is ROTATED by register moves and the freed slot at the back is
// refilled with the same chunk of the coming tile.  Every slot index in the code is therefore a compile-time constant;
// the only run-time quantity is `front`.  (The obvious alternative -- fixed slots and a run-time slot index through a
// switch or a chain of `if (s == sel)` -- makes the compiler route the selected slot through a shared register set
// (64 v_mov per use) or keep two copies of every slot alive: hundreds of spills, LDS operands fetched one by one.)
// LDSC: chunk NW * CPW -- one more than the register slots hold -- lives in LDS (16 M doubles) and is updated there by its
// owner until it becomes the last panel.  For column counts one chunk past a register geometry (TALOS: 21 chunks = 4 x 5
// + 1) this buys the taller tile of the smaller geometry (64 rows with five slots per wave instead of 48 with six).
template <int NW, int CPW, int NRC, int WPE, bool LDSC, int MODE>
__global__ __launch_bounds__(64 * NW, WPE) void tsqr_wy_kernel(const double *__restrict__ Wp_, const long rowsp_,
                                                               const long ldwp_, const int *__restrict__ col_idx,
                                                               const int np_, const double *__restrict__ tau,
                                                               const double *__restrict__ blkw, const long rows_per_blk,
                                                               double *__restrict__ Rblkp_, double *__restrict__ Routp_,
                                                               const int ncp_, long long *__restrict__ prof,
                                                               const long pair_countp_, const int aux, const double null2_) {
    // MODE 0: level 0 (rows of one tall matrix, tiles dealt round-robin).  MODE 1 = PAIR, MODE 2 = BATCH, below.
    // MODE 4 = PAIR for several stacks at once (the wide row blocks of a tree's regressor, figh_tsqr_wide_pair.hip): Wp_ is
    // a device table of `aux` WyPairJob records; the workgroup looks up its job and is then a PAIR workgroup of that job.
    constexpr bool PAIRG = MODE == 4;
    constexpr bool PAIR = MODE == 1 || PAIRG, BATCH = MODE == 2;
    const WyPairJob *job_ = nullptr;
    if constexpr (PAIRG) {
        const WyPairJob *jt = reinterpret_cast<const WyPairJob *>(Wp_);
        int j = 0;
        for (int k = 1; k < aux; ++k) j = (int)blockIdx.x >= jt[k].wg0 ? k : j;
        job_ = jt + j;
    }
    const unsigned bx = PAIRG ? blockIdx.x - (unsigned)job_->wg0 : blockIdx.x;
    const double *__restrict__ W_ = PAIRG ? job_->stack : Wp_;
    const int nc = PAIRG ? job_->nc : ncp_;
    const long rows_ = PAIRG ? (long)nc : rowsp_, ldw = PAIRG ? (long)nc : ldwp_;
    const int n = PAIRG ? nc : np_;
    double *__restrict__ Rblk = PAIRG ? job_->Rblk : Rblkp_;
    double *__restrict__ Rout = PAIRG ? job_->Rout : Routp_;
    const long pair_count = PAIRG ? job_->count : pair_countp_;
    // PAIR (pair-merge mode).  W_ is a stack of pair_count compact nc x nc triangles (ldw == nc); workgroup b
    // starts from triangle 2b (copied into its packed blocks: absorbing a triangle into an EMPTY one would cost a full
    // factorisation to reproduce it) and absorbs triangle 2b + 1, tile by tile, all tiles its own (tile i of a triangle
    // starts at column 16 NRC i).  A last workgroup without a partner passes its triangle through.
    // PAIR is a template parameter so that the level-0 instantiation is the round-2 code, register for register (as
    // run-time mode the extra scalar state moved the TALOS geometry from 68 to 88 bytes of scratch: level 0 +3.7 %).
    // BATCH: B independent matrices in ONE joint-major regressor of B * rows_ samples (one K1 launch): matrix b is the
    // `aux` row segments [j * rows_per_blk + b * rows_, + rows_), j < aux, of W_ (rows_per_blk = the joint stride = total
    // samples; no row-block weights in this mode).  pair_count = workgroups per matrix: workgroup (b, l) factors the
    // tiles l, l + pair_count, ... of matrix b -- tile t = segment t / tps, rows [M (t % tps), ...) of it, the ragged end
    // of every segment zero-filled by the range check of its own buffer descriptor -- into its own triangle.
    const long bidx = BATCH ? (long)bx / pair_count : 0L;
    const double *__restrict__ W = PAIR ? W_ + (2L * bx + 1) * nc * nc : (BATCH ? W_ + bidx * rows_ * ldw : W_);
    // (MODE 3 = CHAIN, chained level-0 launches: the workgroup's own triangle of the previous launch, read back from Rout;
    // a mode of its own for the same reason as PAIR: as a run-time flag of MODE 0 it cost the TALOS geometry 20 more
    // bytes of scratch)
    constexpr bool CHAIN = MODE == 3;
    const double *__restrict__ Rinit = PAIR ? W_ + (2L * bx) * nc * nc
                                            : (CHAIN ? Rout + (long)bx * nc * nc : nullptr);
    const long rows = PAIR ? ((2L * bx + 1 < pair_count) ? (long)nc : 0L) : rows_;
    const long tile0 = PAIR ? 0L : (BATCH ? (long)bx - bidx * pair_count : (long)bx);
    const long tstep = PAIR ? 1L : (BATCH ? pair_count : (long)gridDim.x);
    static_assert((NW & (NW - 1)) == 0 && NW >= 2, "NW must be a power of two >= 2");
    static_assert(CPW >= 2, "at least two chunk slots per wave");
    FIGH_PROF_DECL
    if constexpr (WPE == 1) asm volatile("" ::: "a255");  // the allocation covers the SIMD: never two waves on one
    constexpr int RPL = 4 * NRC, M = 16 * NRC, VBUF = M * kLdv + 16 * kLdt, NQ = CPW - 1;
    constexpr bool kLateRetire = CPW <= 4 || LDSC;  // empirical, per geometry (same-box A/B)  // see the look-ahead block
    __shared__ double vt[3][VBUF];      // V (M x kLdv) followed by T (16 x kLdt); three buffers: panel p is still read in
                                        // phase p + 1 (deferred sweep of the wave that factored panel p + 1)
    __shared__ double rpp[NW][256];     // the diagonal block of the panel a wave is factoring (wave-private)
    __shared__ double redbuf[NW][64];   // cross-row-group sums (wave-private)
    __shared__ int fnz[2][NW];
    __shared__ double lch[LDSC ? 16 * M : 8];
    const int lane = threadIdx.x & 63;
#ifdef FIGH_ABLATION
    // mode 1: workgroups of the second half of the grid rotated by two waves, 2: by one, 3: odd workgroups by two,
    // 4: by blockIdx / 8 (the XCD-local index)
    const int woff_ = g_wy_off == 1 ? (blockIdx.x >= gridDim.x / 2 ? 2 : 0)
                      : g_wy_off == 2 ? (blockIdx.x >= gridDim.x / 2 ? 1 : 0)
                      : g_wy_off == 3 ? ((blockIdx.x & 1) ? 2 : 0)
                      : g_wy_off == 4 ? (int)(blockIdx.x >> 3) : 0;
    const int wave = __builtin_amdgcn_readfirstlane(((threadIdx.x >> 6) + woff_) & (NW - 1));
#else
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#endif
    const int c = lane & 15, g = lane >> 4;
    const int nch = (nc + 15) >> 4;
    constexpr int LC = NW * CPW;                               // the chunk that lives in LDS (LDSC)
    const int nreg = LDSC ? (nch < LC ? nch : LC) : nch;       // chunks held in register slots
    const bool lhave = LDSC && nch > LC;                       // the matrix reaches into the LDS chunk
    const bool lowner = lhave && wave == (LC & (NW - 1));      // the wave that looks after it
#ifdef FIGH_ABLATION
    // FIGH_WY_RALIAS: the workgroups of an XCD share ONE triangle (garbage results, same instruction stream): what the
    // kernel would cost if the R blocks came from L2 instead of HBM / MALL
    double *Rb = Rblk + (long)(g_wy_ralias ? (blockIdx.x & 7) : blockIdx.x) * ((long)nch * (nch + 1) / 2) * 256;
#else
    double *Rb = Rblk + (long)bx * ((long)nch * (nch + 1) / 2) * 256;
#endif
    auto block = [&](const int p, const int cc) { return Rb + ((long)cc * (cc + 1) / 2 + p) * 256; };
    double *Rl = rpp[wave];
    double *red = redbuf[wave];

    // this wave's columns of the triangle start empty (pair-merge mode: as the workgroup's first triangle)
    auto init_block = [&](const int p, const int cc) {
        double *b = block(p, cc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double val = 0.0;
            if constexpr (PAIR || CHAIN) {
                const int row = 16 * p + g + 4 * r, col = 16 * cc + c;
                if (col < nc && row <= col) val = Rinit[(long)row * nc + col];
            }
            b[lane + 64 * r] = val;
        }
    };
    // CHAIN (chained launches of a streamed run): the workgroup starts from its own triangle of the previous
    // sample chunk (read back from Rout) -- one RUNNING triangle per workgroup instead of one per workgroup and chunk (the
    // human model's 20 chunks: 512 triangles to merge instead of 10 240).  Measured on one box (tools/chain_ab.sh): a launch
    // that starts from a filled triangle takes 7.6 ms where one that starts from zeros takes 7.2 ms, whether the blocks are
    // left in place or re-created, in or out of phase with the other workgroup of the CU (tools/wydelay_ab.sh) -- the
    // operands of the first half of a launch (force rows: the inertia columns are exact zeros) are then no longer mostly
    // zero; the merges saved (8.3 -> 1.8 ms) outweigh it.
    if (lowner)
        for (int p = 0; p <= LC; ++p) init_block(p, LC);
#pragma unroll
    for (int s = 0; s < CPW; ++s) {
        const int cc = wave + NW * s;
        if (cc < nreg)
            for (int p = 0; p <= cc; ++p) init_block(p, cc);
    }

    // One queue slot: the tile chunk (C/D layout) and its per-lane column source.  Full tiles are read with buffer loads:
    // row base in an SGPR resource descriptor, row offsets inside the tile in SGPR soffsets, ONE 32-bit byte offset per
    // lane (boff; tau is column n and uses its own descriptor).  Lane-columns beyond the matrix are dead: their registers
    // stay exactly zero for the whole kernel (zero data, zero triangle entries, and neither panel nor update changes a
    // zero column).
    struct Slot {
        f64x4 t[NRC];
        unsigned boff;
        bool wlive, tlive;
    };
    Slot F, Q[NQ];
    auto init_slot = [&](Slot &S, const int s) {
        const int col = 16 * (wave + NW * s) + c;
        S.wlive = col < n;
        S.tlive = col == n && tau != nullptr;
        const int wcol = S.wlive ? (col_idx ? col_idx[col] : col) : 0;  // (only boff travels with the slot)
        S.boff = 8u * ((unsigned)g * (unsigned)ldw + (unsigned)wcol);
#pragma unroll
        for (int rc = 0; rc < NRC; ++rc) S.t[rc] = f64x4{0.0, 0.0, 0.0, 0.0};
    };
    init_slot(F, 0);
#pragma unroll
    for (int j = 0; j < NQ; ++j) init_slot(Q[j], j + 1);

    const unsigned toff = 8u * (unsigned)g;
    const unsigned ldw8 = 8u * (unsigned)ldw;  // bytes per row (the host side guarantees 64 * ldw * 8 < 2^32)

    // requests for the chunk of slot S_ in the tile at row r0_: RPL independent loads per lane.  The descriptor ends
    // with the matrix, so the rows of a ragged last tile beyond `rows` come back as zeros from the buffer range check
    // -- no second, clamped-and-masked load path (whose 64-bit row arithmetic the compiler hoisted into scratch).
#define FIGH_WY_LOAD(S_, r0_, lf_)                                                                                  \
    do {                                                                                                          \
        const long left_ = BATCH ? (lf_) : rows - (r0_);                                                                        \
        if ((S_).wlive) {                                                                                         \
            const long bytes_ = left_ * ldw * 8;                                                                  \
            const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(                                 \
                const_cast<double *>(W + (r0_) * ldw), (short)0, (int)(bytes_ < 0x7fffffffL ? bytes_ : 0x7fffffffL), \
                0x00020000);                                                                                      \
            _Pragma("unroll") for (int i = 0; i < RPL; ++i) {                                                     \
                const u32x2 v_ = __builtin_amdgcn_raw_buffer_load_b64(                                            \
                    rs_, (S_).boff, (unsigned)(16 * (i >> 2) + 4 * (i & 3)) * ldw8, 0);                           \
                (S_).t[i >> 2][i & 3] = __hiloint2double((int)v_[1], (int)v_[0]);                                 \
            }                                                                                                     \
        }                                                                                                         \
        if ((S_).tlive) {                                                                                         \
            const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(                                 \
                const_cast<double *>(tau + (r0_)), (short)0, (int)(left_ * 8 < 0x7fffffffL ? left_ * 8 : 0x7fffffffL), \
                0x00020000);                                                                                      \
            _Pragma("unroll") for (int i = 0; i < RPL; ++i) {                                                     \
                const u32x2 v_ = __builtin_amdgcn_raw_buffer_load_b64(                                            \
                    rs_, toff, 8u * (unsigned)(16 * (i >> 2) + 4 * (i & 3)), 0);                                  \
                (S_).t[i >> 2][i & 3] = __hiloint2double((int)v_[1], (int)v_[0]);                                 \
            }                                                                                                     \
        }                                                                                                         \
    } while (0)
#define FIGH_WY_WAVE_SYNC()                                    \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)
    // The front chunk retires: rotate the queue (register moves) and refill the freed slot at the back with the same
    // chunk of the coming tile (full tiles only; `rn_` is its first row, opaque to the compiler so that the row
    // addresses are not hoisted out of the loops).
#define FIGH_WY_RETIRE(rn_)                                                            \
    do {                                                                               \
        Slot old_ = F;                                                                 \
        F = Q[0];                                                                      \
        _Pragma("unroll") for (int j = 0; j + 1 < NQ; ++j) Q[j] = Q[j + 1];            \
        Q[NQ - 1].boff = old_.boff;                                                    \
        Q[NQ - 1].wlive = old_.wlive;                                                  \
        Q[NQ - 1].tlive = old_.tlive;                                                  \
        _Pragma("unroll") for (int rc = 0; rc < NRC; ++rc) Q[NQ - 1].t[rc] = old_.t[rc]; \
        if (next_full) FIGH_WY_LOAD(Q[NQ - 1], rn_, leftn);                                \
        front += NW;                                                                   \
    } while (0)

    const long tps = (rows + M - 1) / M;  // tiles per row segment (BATCH)
    const long ntiles = BATCH ? tps * aux : tps;
    int parity = 0;
    int front = wave;       // chunk held by F; Q[j] holds front + NW (j + 1)
    bool prefetched = false;  // the queue already holds the coming tile (every chunk retires once per tile)
    // null pivots (figh_tsqr_null_pivot_tol) only once the workgroup's triangle is of full height (level 0: it starts
    // empty): see tsqr2_level0_body, figh_tsqr_narrow_kernel.h
    int young = (PAIR || CHAIN) ? 0 : (nc + M / 8 + M - 1) / M;
    for (long t = tile0; t < ntiles; t += tstep, parity ^= 1) {
        const double null2 = young > 0 ? 0.0 : null2_;
        young -= young > 0 ? 1 : 0;
        long r0, r0n, left = 0, leftn = 0;
        bool next_full;  // the workgroup has another tile (full or ragged: the descriptor zero-fills)
        if constexpr (BATCH) {
            const long sg = t / tps, tn = t + tstep, sgn = tn / tps;
            r0 = sg * rows_per_blk + (t - sg * tps) * M;
            left = rows - (t - sg * tps) * M;
            r0n = sgn * rows_per_blk + (tn - sgn * tps) * M;
            leftn = rows - (tn - sgn * tps) * M;
            next_full = tn < ntiles;
            asm volatile("" : "+s"(leftn));
        } else {
            r0 = t * M;
            r0n = (t + tstep) * M;
            next_full = r0n < rows;
        }
        if (!prefetched) {
            FIGH_WY_LOAD(F, r0, left);
#pragma unroll
            for (int j = 0; j < NQ; ++j) FIGH_WY_LOAD(Q[j], r0, left);
        }
        prefetched = next_full;
        // The LDS chunk is requested at the top of the tile (its LDS home is in use until the tile's last panel) into a
        // record that only lives until the data are in LDS, a few lines further down.
        // Its column source is re-derived here every tile (one index load) rather than held in registers across the tile.
        Slot L;
        {
            int lc_ = c;
            asm volatile("" : "+v"(lc_));  // opaque: not to be hoisted out of the tile loop
            const int lcol = 16 * LC + lc_;
            L.wlive = lowner && lcol < n;
            L.tlive = lowner && lcol == n && tau != nullptr;
            const int lwcol = L.wlive ? (col_idx ? col_idx[lcol] : lcol) : 0;
            L.boff = 8u * ((unsigned)g * (unsigned)ldw + (unsigned)lwcol);
        }
#pragma unroll
        for (int rc = 0; rc < NRC; ++rc) L.t[rc] = f64x4{0.0, 0.0, 0.0, 0.0};
        if (lowner) FIGH_WY_LOAD(L, r0, left);
        if (blkw) {  // row-block weights (WLS): row r is scaled by blkw[r / rows_per_blk]
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const long row = r0 + 16 * (i >> 2) + 4 * (i & 3) + g;
                const double scale = blkw[(row < rows ? row : rows - 1) / rows_per_blk];
                F.t[i >> 2][i & 3] *= scale;
                if (lowner) L.t[i >> 2][i & 3] *= scale;
#pragma unroll
                for (int j = 0; j < NQ; ++j) Q[j].t[i >> 2][i & 3] *= scale;
            }
        }
        // the first column with a non-zero in this tile: the column steps in front of it are identities (stacked
        // triangles in the merge levels, the joint-torque rows of a tree)
        int myfirst = 16 * nch;
        if (lowner) {
            bool nz = false;
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                nz |= (L.t[i >> 2][i & 3] != 0.0);
                lch[64 * i + lane] = L.t[i >> 2][i & 3];
            }
            const unsigned long long b = __ballot(nz);
            const unsigned m16 = (unsigned)((b | (b >> 16) | (b >> 32) | (b >> 48)) & 0xffffull);
            if (m16) myfirst = 16 * LC + __ffs((int)m16) - 1;
        }
#pragma unroll
        for (int s = CPW - 1; s >= 0; --s) {
            bool nz = false;
#pragma unroll
            for (int i = 0; i < RPL; ++i) nz |= ((s == 0 ? F.t[i >> 2][i & 3] : Q[s > 0 ? s - 1 : 0].t[i >> 2][i & 3]) != 0.0);
            const unsigned long long b = __ballot(nz);
            const unsigned m16 = (unsigned)((b | (b >> 16) | (b >> 32) | (b >> 48)) & 0xffffull);
            if (m16) myfirst = 16 * (wave + NW * s) + __ffs((int)m16) - 1;
        }
        if (lane == 0) fnz[parity][wave] = myfirst;
        __syncthreads();
        int first_nz = fnz[parity][0];
#pragma unroll
        for (int w = 1; w < NW; ++w) first_nz = min(first_nz, fnz[parity][w]);
        int p0 = __builtin_amdgcn_readfirstlane(first_nz) >> 4;
        if (p0 > nch) p0 = nch;
        FIGH_PROF_ADD(1);
        long rn = r0n;
        asm volatile("" : "+s"(rn));
        // chunks in front of the first non-zero column take no part in this tile
        while (front < p0 && front < wave + NW * CPW) FIGH_WY_RETIRE(rn);

        if (p0 < nch) {
            int vb = p0 % 3;        // V/T buffer of panel p (wave-uniform); vbp: the buffer of panel p - 1
            int vbp = vb;
            bool deferred = false;  // this wave factored panel p in phase p - 1 and owes its other chunks panel p - 1
            // ---- the first panel of the tile has nobody to overlap with
            if (wave == (p0 & (NW - 1))) {
                __builtin_amdgcn_s_setprio(3);
                double *Vn = vt[vb];
                double *bpp = block(p0, p0);
#pragma unroll
                for (int r = 0; r < 4; ++r) Rl[lane + 64 * r] = bpp[lane + 64 * r];
                double X[RPL];
                const bool from_lds = LDSC && p0 == LC;
#pragma unroll
                for (int i = 0; i < RPL; ++i) X[i] = F.t[i >> 2][i & 3];
                if constexpr (LDSC) {
                    if (from_lds) {
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int i = 0; i < RPL; ++i) X[i] = lch[64 * i + lane];
                    }
                }
                FIGH_WY_WAVE_SYNC();
                FIGH_PROF_ADD(8);
                wy_factor_panel<RPL>(X, Rl, red, Vn, Vn + M * kLdv, lane, c, g, null2);
                FIGH_WY_WAVE_SYNC();
                FIGH_PROF_ADD(9);
#pragma unroll
                for (int r = 0; r < 4; ++r) bpp[lane + 64 * r] = Rl[lane + 64 * r];
                if constexpr (!kLateRetire) {
                    if (!(LDSC && p0 == LC)) FIGH_WY_RETIRE(rn);  // (the LDS chunk has no register slot to rotate)
                }
                __builtin_amdgcn_s_setprio(0);
            }
            __syncthreads();
            FIGH_PROF_ADD(2);
            if constexpr (kLateRetire) {
                if (wave == (p0 & (NW - 1)) && !(LDSC && p0 == LC)) {
                    FIGH_WY_RETIRE(rn);
                    FIGH_PROF_ADD(10);
                }
            }

            // ---- phase p: apply panel p to the trailing chunks; the owner of chunk p + 1 updates that chunk first and
            // factors panel p + 1 meanwhile.  The block (p, p+1) of the triangle that the owner starts with was requested
            // one phase earlier (rp_next); the diagonal block (p+1, p+1) is requested at the start of the phase and has
            // the chunk update to arrive.
            f64x4 rp_next = {0.0, 0.0, 0.0, 0.0};
            bool have_next = false;
            for (int p = p0; p + 1 < nch; ++p) {
                const double *Vl = vt[vb];
                const double *Tl = Vl + M * kLdv;
                const int vbn = vb == 2 ? 0 : vb + 1;
                const int pn = p + 1;
                f64x4 rp = rp_next;
                const bool have = have_next;
                have_next = false;
                if (pn + 1 < nch && wave == ((pn + 1) & (NW - 1))) {  // owner of the phase after this one
                    const double *b1 = block(pn, pn + 1);
#pragma unroll
                    for (int r = 0; r < 4; ++r) rp_next[r] = b1[lane + 64 * r];
                    have_next = true;
                }
                const bool is_owner = wave == (pn & (NW - 1));
                if (is_owner) {  // front == pn
                    // this wave is the critical path of the workgroup until panel p + 1 is published: it wins the
                    // issue arbitration against the waves it shares the SIMD with
                    __builtin_amdgcn_s_setprio(3);
                    double *bpp = block(pn, pn);
                    f64x4 rq;
#pragma unroll
                    for (int r = 0; r < 4; ++r) rq[r] = bpp[lane + 64 * r];
                    if (!have) {
                        const double *b1 = block(p, pn);
#pragma unroll
                        for (int r = 0; r < 4; ++r) rp[r] = b1[lane + 64 * r];
                    }
                    const bool from_lds = LDSC && pn == LC;
                    if (from_lds) wy_update_lds_chunk<NRC>(lch, Vl, Tl, block(p, pn), lane, c, g);
                    else wy_update_chunk<NRC>(F.t, Vl, Tl, rp, block(p, pn), lane, c, g);
                    FIGH_PROF_ADD(3);
#pragma unroll
                    for (int r = 0; r < 4; ++r) Rl[lane + 64 * r] = rq[r];
                    double *Vn = vt[vbn];
                    double X[RPL];
#pragma unroll
                    for (int i = 0; i < RPL; ++i) X[i] = F.t[i >> 2][i & 3];
                    if constexpr (LDSC) {
                        if (from_lds) {  // (a real branch: as a select the sixteen LDS reads were issued for every panel)
                            asm volatile("" ::: "memory");
#pragma unroll
                            for (int i = 0; i < RPL; ++i) X[i] = lch[64 * i + lane];
                        }
                    }
                    FIGH_WY_WAVE_SYNC();
                    FIGH_PROF_ADD(7);
                    wy_factor_panel<RPL>(X, Rl, red, Vn, Vn + M * kLdv, lane, c, g, null2);
                    FIGH_WY_WAVE_SYNC();
                    FIGH_PROF_ADD(4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) bpp[lane + 64 * r] = Rl[lane + 64 * r];
                    // the queue rotation + the requests for the next tile's chunk (0.4-0.9 k cycles): with few slots per
                    // wave after the barrier, where nobody waits for them (same-box A/B: n = 191 +4 %); with six slots
                    // the rotation is long and last phase's owner, who has two sweeps to do, is the critical wave
                    // (TALOS -3 %), so there it stays in front of the barrier
                    if constexpr (!kLateRetire) {
                        if (!from_lds) FIGH_WY_RETIRE(rn);
                    }
                    __builtin_amdgcn_s_setprio(0);
                    // LOAD BALANCE: the look-ahead (chunk update + 16 dependent column steps) is about two trailing
                    // sweeps long, so this wave leaves panel p to its other chunks for the next phase, when it is not
                    // the owner (NW >= 2) -- otherwise every phase lasts look-ahead + sweep and the other waves wait at
                    // the barrier for half of it (in-kernel profile before the change: 44 % of the wave time).
                    deferred = true;
                }
                // Trailing sweeps of this wave in this phase -- ONE copy of the update code in a wave-uniform loop (a
                // second copy costs hundreds of spilled registers): none for the owner, panel p - 1 and then panel p for
                // last phase's owner, panel p for everybody else.
                const bool owe = deferred && !is_owner;
#pragma nounroll
                for (int rep = is_owner ? 2 : (owe ? 0 : 1); rep < 2; ++rep) {
                    const int pp = rep == 0 ? p - 1 : p;  // the panel applied in this sweep, to the chunks behind pp + 1
                    const double *Vx = rep == 0 ? vt[vbp] : Vl;
                    const double *Tx = Vx + M * kLdv;
                    if (front > pp + 1 && front < nreg) {
                        double *b = block(pp, front);
                        f64x4 rb;
#pragma unroll
                        for (int r = 0; r < 4; ++r) rb[r] = b[lane + 64 * r];
                        wy_update_chunk<NRC>(F.t, Vx, Tx, rb, b, lane, c, g);
                    }
                    __builtin_amdgcn_sched_barrier(0);  // one chunk at a time: hoisting the next chunk's operands costs registers
#pragma unroll
                    for (int j = 0; j < NQ; ++j) {
                        const int cc = front + NW * (j + 1);
                        if (cc > pp + 1 && cc < nreg) {
                            double *b = block(pp, cc);
                            f64x4 rb;
#pragma unroll
                            for (int r = 0; r < 4; ++r) rb[r] = b[lane + 64 * r];
                            wy_update_chunk<NRC>(Q[j].t, Vx, Tx, rb, b, lane, c, g);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (LDSC) {
                        // The LDS chunk is updated by a different wave every phase -- any wave can, it is in shared
                        // memory -- namely one that has a single sweep to do: wave pp + 2 is neither the owner of panel
                        // pp + 1 nor the wave that owes a deferred sweep (owner of panel pp), so this never runs in the
                        // deferred round either.  Consecutive panels are separated by the phase barrier.
                        if (lhave && wave == ((pp + 2) & (NW - 1)) && LC > pp + 1)
                            wy_update_lds_chunk<NRC>(lch, Vx, Tx, block(pp, LC), lane, c, g);
                    }
                }
                if (owe) deferred = false;
                FIGH_PROF_ADD(5);
                __syncthreads();
                FIGH_PROF_ADD(6);
                if constexpr (kLateRetire) {
                    if (is_owner && !(LDSC && pn == LC)) {
                        FIGH_WY_RETIRE(rn);
                        FIGH_PROF_ADD(11);
                    }
                }
                vbp = vb;
                vb = vbn;
            }
        }
        // slots beyond the last chunk of the matrix (and, for a zero tile, everything) leave the queue unused
        while (front < wave + NW * CPW) FIGH_WY_RETIRE(rn);
        front = wave;
    }
#undef FIGH_WY_LOAD
#undef FIGH_WY_WAVE_SYNC
#undef FIGH_WY_RETIRE
    FIGH_PROF_STORE(prof, wave, NW);

    // ---- write this wave's columns of the nc x nc row-major triangle (zeros below the diagonal)
    double *Ro = Rout + (long)bx * nc * nc;
#pragma unroll
    for (int s = 0; s < CPW + (LDSC ? 1 : 0); ++s) {
        const int cc = s < CPW ? wave + NW * s : LC;  // (the extra round: the LDS chunk's columns, by its owner)
        if (s == CPW && !lowner) continue;
        if (cc >= nch) continue;
        const int col = 16 * cc + c;
        for (int pb = 0; pb < nch; ++pb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * pb + g + 4 * r;
                double val = 0.0;
                if (pb <= cc) val = block(pb, cc)[lane + 64 * r];
                if (row < nc && col < nc) Ro[(long)row * nc + col] = (col >= row) ? val : 0.0;
            }
        }
    }
}

template <int NW, int CPW, int NRC, int WPE, bool LDSC>
int wy_occupancy() {  // resident workgroups per CU (registers and LDS decide)
    static int nb = 0;
    if (!nb) {
        int v = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, tsqr_wy_kernel<NW, CPW, NRC, WPE, LDSC, 0>, 64 * NW, 0) != hipSuccess ||
            v < 1)
            v = 1;
        nb = v;
    }
    return nb;
}

struct WyConfig {
    int nw, cpw, nrc, wpe, ldsc;
};

// Geometry by column count (one wave per SIMD throughout, see tsqr_wy_kernel): as few waves per workgroup -- as many
// independent panel chains per CU -- as the register file allows for the workgroup's tile.
[[maybe_unused]] WyConfig wy_config(const int nc) {
    const int nch = (nc + 15) >> 4;
#ifdef FIGH_ABLATION
    if (const char *e = getenv("FIGH_WY_CFG")) {  // "nw,cpw,nrc,wpe" (ablation build only)
        WyConfig cfg{0, 0, 0, 0, 0};  // a fifth number 1 selects the LDS-chunk form (one chunk beyond the register slots)
        const int got = sscanf(e, "%d,%d,%d,%d,%d", &cfg.nw, &cfg.cpw, &cfg.nrc, &cfg.wpe, &cfg.ldsc);
        if (got >= 4 && cfg.nw * cfg.cpw + (cfg.ldsc ? 1 : 0) >= nch) return cfg;
    }
#endif
    // the tallest tile the 256 registers of a wave hold: the rows one look-ahead chain covers are what the throughput is
    // proportional to (same-box A/B, n = 191: 64 -> 80 -> 96 rows +11 %, +5.5 %; n = 241: 64 -> 80 rows +6 %; the last
    // step of each costs 8 / 22 spilled registers and still wins)
    // up to six chunks (81 .. 96 columns: the widest row blocks of a tree's joint-torque regressor): two waves per
    // workgroup -- the tile time is the look-ahead chain whatever the wave count, so more, smaller workgroups per CU
    // (three by LDS instead of two) carry more chains at a time
    if (nch <= 6) return {2, 3, 6, 2, 0};
    if (nch <= 8) return {2, 4, 5, 2, 0};  // 97 .. 128 columns (TIAGo's torso block): two waves x four chunks, 80-row tiles
    if (nch <= 10) return {2, 5, 4, 2, 0};  // 129 .. 160 columns (TALOS' force rows): two waves x five chunks, 64-row tiles
    if (nch <= 12) return {4, 3, 6, 2, 0};  // (two waves x six chunks with 48-row tiles: human torque rows 71 -> 79 ms)
    if (nch <= 16) return {4, 4, 5, 2, 0};
    if (nch <= 20) return {4, 5, 4, 2, 0};
    if (nch == 21) return {4, 5, 4, 2, 1};  // TALOS (331 columns): 64-row tiles, chunk 20 in LDS
    if (nch <= 24) return {4, 6, 3, 2, 0};
    if (nch == 25) return {4, 6, 3, 2, 1};  // the human SIP program's 400 columns: two chains per CU instead of the one
                                            // of the 8-wave geometry (n = 400: 49.7 -> 45.2 ms)
    return {8, 4, 4, 2, 0};
}

template <class F>
bool wy_dispatch(const WyConfig cfg, F &&f) {
#define FIGH_WY_CASE_L(NW_, CPW_, NRC_, WPE_, L_)                                                  \
    if (cfg.nw == NW_ && cfg.cpw == CPW_ && cfg.nrc == NRC_ && cfg.wpe == WPE_ && (cfg.ldsc != 0) == L_) { \
        f(std::integral_constant<int, NW_>{}, std::integral_constant<int, CPW_>{},                 \
          std::integral_constant<int, NRC_>{}, std::integral_constant<int, WPE_>{},                \
          std::integral_constant<bool, L_>{});                                                     \
        return true;                                                                               \
    }
#define FIGH_WY_CASE(NW_, CPW_, NRC_, WPE_) FIGH_WY_CASE_L(NW_, CPW_, NRC_, WPE_, false)
    FIGH_WY_CASE(2, 3, 6, 2)
    FIGH_WY_CASE(2, 4, 5, 2)
    FIGH_WY_CASE(2, 5, 4, 2)
    FIGH_WY_CASE(4, 3, 6, 2)
    FIGH_WY_CASE(4, 4, 5, 2)
    FIGH_WY_CASE(4, 5, 4, 2)
    FIGH_WY_CASE_L(4, 5, 4, 2, true)
    FIGH_WY_CASE(4, 6, 3, 2)
    FIGH_WY_CASE_L(4, 6, 3, 2, true)
    FIGH_WY_CASE(8, 4, 4, 2)
#ifdef FIGH_ABLATION
    FIGH_WY_CASE(8, 3, 4, 2)
    FIGH_WY_CASE(4, 3, 4, 2)
    FIGH_WY_CASE(4, 3, 5, 2)
    FIGH_WY_CASE(4, 4, 4, 2)
    FIGH_WY_CASE(4, 3, 4, 3)
    FIGH_WY_CASE(4, 4, 3, 3)
    FIGH_WY_CASE(4, 3, 3, 3)
    FIGH_WY_CASE(4, 4, 3, 2)
    FIGH_WY_CASE(8, 2, 4, 3)
    FIGH_WY_CASE(8, 2, 4, 2)
    FIGH_WY_CASE(4, 3, 4, 1)
    FIGH_WY_CASE(4, 6, 4, 1)
    FIGH_WY_CASE(8, 3, 2, 4)
    FIGH_WY_CASE(8, 3, 2, 3)
    FIGH_WY_CASE(8, 4, 2, 3)
#endif
#undef FIGH_WY_CASE
#undef FIGH_WY_CASE_L
    return false;
}

}  // namespace

}  // namespace figh
