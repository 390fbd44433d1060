// Wave-level building blocks shared by the TSQR kernels (gfx950, wave64).
//
// Register tiles are kept in the v_mfma_f64_16x16x4 C/D layout: lane = 16*g + c (g = lane >> 4 is the "row group",
// c = lane & 15 the lane-column) holds rows 16*rc + g + 4*r (rc = 16-row chunk, r = 0..3) of column c.  In that layout
//   - a column of the tile inside one row group is a DPP row_newbcast (a VALU operand modifier, no LDS crossbar),
//   - a sum over the rows of a column is a per-lane sum followed by a sum over the four row groups,
//   - register r of a 16-row chunk IS the A/B operand of K-slice r of the f64 MFMA (A[i = c][k = g], B[k = g][j = c]).
#pragma once

#include <hip/hip_runtime.h>

#ifndef FIGH_NARROW_BANKMASK
#define FIGH_NARROW_BANKMASK 1
#endif

namespace figh {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int K>
__device__ __forceinline__ double row_bcast(double x) {  // value of lane-column K of my row group
    // v_mov_b64_dpp: gfx90a+ allows 64-bit DPP for row_newbcast, one instruction per double
    return __longlong_as_double(
        __builtin_amdgcn_update_dpp((long long)0, __double_as_longlong(x), 0x150 + K, 0xf, 0xf, true));
}

__device__ __forceinline__ double allreduce_rowgroups(double x) {  // sum over lanes c, c+16, c+32, c+48
    unsigned lo = __double2loint(x), hi = __double2hiint(x);
    u32x2 a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    u32x2 b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const double y = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    lo = __double2loint(y);
    hi = __double2hiint(y);
    a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}

__device__ __forceinline__ double uniform_of(double x) {  // SGPR copy of a value that is identical in all lanes
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)),
                            __builtin_amdgcn_readfirstlane(__double2loint(x)));
}

// Sum over the four row groups through the wave's own 512 B of LDS: one ds_write_b64 + three ds_read_b64 + three
// v_add_f64 instead of 4 v_mov + 4 v_permlane*_swap + 2 adds.  Every row group adds the same two pairs in the same
// order: the result is bit-identical in all lanes.  (Same-box A/B on the UR10 problem, round 1: 1.034 vs 1.080 ms.)
__device__ __forceinline__ double allreduce_rowgroups_lds(double *red, const int lane, const double x) {
    red[lane] = x;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double a = red[lane ^ 16], b = red[lane ^ 32], c = red[lane ^ 48];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return (x + a) + (b + c);
}

// acc += (value of pv in lane-column K of my row group) * b as ONE instruction: gfx90a+ allow a DPP row_newbcast
// operand on the DP ALU v_fmac_f64, so the pivot column is never materialised in registers (no v_mov_b64_dpp per
// row).  The compiler does not form this instruction by itself.  Hazard: a VGPR written by a VALU instruction needs
// 2 wait states before a DPP read -- callers order their updates so that a register is never read through DPP by the
// instruction right after the one that wrote it.
template <int K>
__device__ __forceinline__ void fmac_bcast(double &acc, const double pv, const double b) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(pv), "v"(b), "n"(K));
}

// The same with the banks of four lane-columns that lie entirely LEFT of column K switched off (DPP bank_mask: a disabled bank
// keeps its accumulator): in the pivot chunk of column step K the lane-columns c < K are finished -- their entries are zeros and
// stay zeros either way -- so up to twelve of a row's sixteen lanes need not execute the FMA.  Round 6, on by default
// (FIGH_NARROW_BANKMASK): the fused launch is clocked down by the power management (2.03 GHz, profiles/r06_pmc_clock.txt) and
// the same instruction stream with fewer active lanes runs 1 - 3 % faster (same-box A/B, four alternating pairs: fused kernel
// 1.344 - 1.369 ms against 1.361 - 1.411, profiles/r06_bankmask_ab.txt); results are bit-identical where they are read.
template <int K>
__device__ __forceinline__ void fmac_bcast_live(double &acc, const double pv, const double b) {
#if FIGH_NARROW_BANKMASK
    constexpr int BM = 0xf & ~((1 << (K / 4)) - 1);
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:%4" : "+v"(acc) : "v"(pv), "v"(b), "n"(K), "n"(BM));
#else
    fmac_bcast<K>(acc, pv, b);
#endif
}

// Householder scalars of the stacked column [alpha; x], sigma = x^T x != 0 (LAPACK dlarfg without the rescaling
// loop): beta = -sign(alpha) sqrt(alpha^2 + sigma), inv = 1 / (alpha - beta) = sign(alpha) / (|alpha| + s),
// tfac = tau = (beta - alpha) / beta = (|alpha| + s) / s.  v_rsq_f64 / v_rcp_f64 seeds (2^-24 accurate) + two Newton
// steps each instead of the IEEE sqrt and two divisions.
__device__ __forceinline__ void householder_scalars(const double alpha, const double sigma, double &inv, double &tfac) {
    const double q2 = fma(alpha, alpha, sigma);
    const double hq = -0.5 * q2;
    double rs = __builtin_amdgcn_rsq(q2);
    rs = rs * fma(hq * rs, rs, 1.5);
    rs = rs * fma(hq * rs, rs, 1.5);
    const double dsum = fma(q2, rs, fabs(alpha));  // |alpha| + s
    double ri = __builtin_amdgcn_rcp(dsum);
    ri = ri * fma(-dsum, ri, 2.0);
    ri = ri * fma(-dsum, ri, 2.0);
    inv = copysign(ri, alpha);
    tfac = dsum * rs;
}

// The same scalars with ONE third-order correction per seed (y (1 + e/2 + 3 e^2/8), e = 1 - q y^2; r (1 + e + e^2), e = 1 - d r:
// error e^3 from a 2^-24 seed) instead of two Newton steps each: 5 + 3 dependent operations instead of 7 + 4 -- what the
// register-tile kernel's step uses (figh_tsqr_narrow.h).
__device__ __forceinline__ void householder_scalars3(const double alpha, const double sigma, double &inv, double &tfac) {
    const double q2 = fma(alpha, alpha, sigma);
    double rs = __builtin_amdgcn_rsq(q2);
    {
        const double e = fma(-(q2 * rs), rs, 1.0);
        rs = fma(rs, fma(e, 0.375, 0.5) * e, rs);
    }
    const double dsum = fma(q2, rs, fabs(alpha));  // |alpha| + s
    double ri = __builtin_amdgcn_rcp(dsum);
    {
        const double e = fma(-dsum, ri, 1.0);
        ri = fma(ri, fma(e, e, e), ri);
    }
    inv = copysign(ri, alpha);
    tfac = dsum * rs;
}

// The same scalars with the reciprocal started from the UNCORRECTED rsq seed: dsum0 = q2 rs0 + |alpha| is 2^-23 accurate, and
// so is rcp(dsum0) as an approximation of 1 / dsum -- its third-order correction is taken against the accurate dsum (e = 1 -
// dsum ri0 ~ 1e-7, error e^3), so the v_rcp_f64 (26 ticks) runs beside the correction of rs instead of behind it: 77 instead
// of 103 ticks from the rsq to inv on the dependent chain of a column step (tools/microbench/latency.hip for the latencies).
__device__ __forceinline__ void householder_scalars4(const double alpha, const double sigma, double &inv, double &tfac) {
    const double q2 = fma(alpha, alpha, sigma);
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
    inv = copysign(ri, alpha);
    tfac = dsum * rs;
}

}  // namespace figh
