// K2 / K3 -- column norms, gathers, residual norms and the tall-skinny Householder QR on gfx950.
//
// figh_tsqr replaces the np.linalg.qr calls of src/figaroh/tools/qrdecomposition.py:105,205,238,286.  The
// reference only consumes R (|diag R| > tol rank test, R1 / R2 regrouping) and Q1^T tau, never Q itself
// (W_b = Q1 R1 is re-derived as the gathered base columns, qrdecomposition.py:268-269), so the kernel streams
// the rows of W once and keeps only the n x n triangle.
//
// Kernel mapping (tsqr_kernel<CPL, M>): one wavefront owns a contiguous range of rows and a private R.
// Lane l owns columns l, l+64, ... (CPL per lane): a tile of M rows sits in registers, B[c][r].  For each
// column k the pivot column is broadcast across the wave (ds_bpermute, no LDS traffic), every lane forms
// x^T B[:, col] for its own columns in M FMAs, and the Householder update of the stacked [R; tile]
// ("triangle on top of a rectangle", LAPACK tpqrt structure) costs another M FMAs per column -- 2*m*n^2
// flops for m appended rows, no wasted work on the triangle.  Tiles whose leading columns are structurally
// zero (rows of joint j have zeros for links < j in the joint-torque layout) start at their first non-zero
// column.  The per-wave triangles are then reduced by the same kernel over the stacked R factors (fan-in 4
// per level).  Householder throughout: the rank decision |R_kk| > 1e-8 needs ~eps*||col|| accuracy on
// dependent pivots, which a Gram/Cholesky route cannot give.
//
// fp64 on gfx950: v_mfma_f64_16x16x4 and the fp64 VALU FMA have the SAME peak (78.6 TFLOP/s), so for n ~ 50
// (a 16-wide Householder panel would leave the matrix pipe waiting on the panel's reductions) the wave-level
// VALU formulation is used; the roofline this kernel is priced against is that fp64 peak.
#include <cstdlib>
#include <algorithm>
#include <cstdio>
#include <type_traits>
#include <vector>

#include "figh_internal.h"

namespace figh {

__device__ __forceinline__ double bcast_v(double x, int src_lane) {  // value of lane src_lane, in a VGPR
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(x));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(x));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bcast_s(double x, int src_lane) {  // wave-uniform (SGPR) copy
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src_lane);
    return __hiloint2double(hi, lo);
}

// ------------------------------------------------------------------------------------------------------------
// tsqr2_kernel<NCC, NRC>: the n <= 16*NCC (<= 80) kernel.  The 16*NRC x 16*NCC tile sits in registers in the
// MFMA f64 C/D layout: lane = 16*g + c holds rows 16*rc + g + 4*reg, column 16*cc + c.  A column step then needs
//   - the pivot column inside each row group: DPP row_newbcast (a VALU mov, no LDS crossbar),
//   - the dot products summed over the four row groups: v_permlane32_swap / v_permlane16_swap (gfx950) + add,
// instead of 128 ds_bpermute per step (6.2 cycles each per CU, shared by the four SIMDs).  Column chunks are
// ROTATED after each 16-column panel so the pivot panel is always register slot 0 (keeps the unrolled step code
// at 16 variants); finished chunks drop out of the update loops (the triangle's zero part costs nothing).
template <int K>
__device__ __forceinline__ double row_bcast(double x) {  // value of lane-column K of my row group
    // v_mov_b64_dpp: gfx90a+ allows 64-bit DPP for row_newbcast, one instruction per double
    return __longlong_as_double(
        __builtin_amdgcn_update_dpp((long long)0, __double_as_longlong(x), 0x150 + K, 0xf, 0xf, true));
}
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
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
// The same sum on the matrix pipe: with A = all ones, v_mfma_f64_16x16x4 computes D[m][n] = sum_k B[k][n], and the
// K index of the B operand IS the row group (lane = 16 k + n), so one MFMA replaces 4 v_mov + 4 v_permlane*_swap
// (12 ticks each) + 2 v_add_f64.  Measured: NOT faster (1.129 vs 1.107 ms on the UR10 problem) -- the f64 MFMA
// occupies the FP64 datapath for its 64 ticks (tools/microbench/latency.hip: MFMA + independent v_fma_f64 do not
// overlap), so it only trades VALU issue slots for FP64-pipe time.  Kept for reference, not used.
typedef double f64x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double allreduce_rowgroups_mfma(double x) {
    const f64x4_t zero = {0.0, 0.0, 0.0, 0.0};
    const f64x4_t r = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0, x, zero, 0, 0, 0);
    return r[0];
}
__device__ __forceinline__ double uniform_of(double x) {  // SGPR copy of a value that is identical in all lanes
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)),
                            __builtin_amdgcn_readfirstlane(__double2loint(x)));
}

template <int NCC, int NRC>
struct Tsqr2State {
    static constexpr int RPL = 4 * NRC;  // rows per lane
    double T[NCC][RPL];
    double *Rl;   // LDS triangle (packed, biased so that the compile-time row offsets apply)
    double *red;  // LDS: 64 doubles of cross-row-group reduction scratch, private to the wave
    int lane_c;   // lane & 15
    int lane_g;   // lane >> 4
    int nc;
};

// TRI: the tile is one upper-triangular R factor (merge levels): rows 16 rc .. are zero in the columns of panels
// p < rc, so row chunks rc > p take no part in panel p (neither in the pivot column nor in the update).
// The panel index P is a compile-time constant: the chunk registers T[P .. NCC-1] are addressed statically (no
// rotation copies), the number of live chunks is known, and a step is straight-line code -- after the pivot chunk's
// own dot product (the only input of the Householder scalars) the dot products of the trailing chunks and the LDS
// reads of row k are independent of the rsq/rcp chain and are interleaved with it by the scheduler.
// Sum over the four row groups through the wave's own 512 B of LDS: one ds_write_b64 + three ds_read_b64 + three
// v_add_f64 instead of 4 v_mov + 4 v_permlane*_swap + 2 adds.  Every row group adds the same two pairs in the same
// order: the result is bit-identical in all lanes.  Same-box A/B on the UR10 problem (3 repetitions each,
// FIGH_TSQR_DBG=64 selects the swap version): 1.034 vs 1.080 ms -- the default for the 4-chunk level-0 kernel.
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
// row, 32 VGPRs less).  The compiler does not form this instruction by itself.  Hazard: a VGPR written by a VALU
// instruction needs 2 wait states before a DPP read -- every use below reads pivot-chunk registers that were last
// written in the previous column step, i.e. before that step's closing scalar compare + branch.
template <int K>
__device__ __forceinline__ void fmac_bcast(double &acc, const double pv, const double b) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(pv), "v"(b), "n"(K));
}

template <bool LDSRED, int NCC, int NRC>
__device__ __forceinline__ double tsqr2_reduce(Tsqr2State<NCC, NRC> &S, const double x) {
    if constexpr (LDSRED) return allreduce_rowgroups_lds(S.red, 16 * S.lane_g + S.lane_c, x);
    else return allreduce_rowgroups(x);
}

template <int KK, int P, int NCC, int NRC, bool TRI, bool LDSRED>
__device__ __forceinline__ void tsqr2_step(Tsqr2State<NCC, NRC> &S) {
    constexpr int RPL = 4 * NRC;
    constexpr int LIVE = NCC - P;
    constexpr int NR = TRI ? (4 * (P + 1) < RPL ? 4 * (P + 1) : RPL) : RPL;  // rows per lane that take part
    // packed triangle: panel p keeps 16 rows of 16*(NCC-p) entries (columns 16p ..)
    constexpr int rowoff = 256 * (P * NCC - (P * (P - 1)) / 2) + KK * 16 * LIVE;
    // The pivot column x = lane-column KK of chunk P is read in place through the DPP operand.
    double Rk[LIVE], d[LIVE];
#pragma unroll
    for (int cc = 0; cc < LIVE; ++cc) Rk[cc] = S.Rl[rowoff + 16 * cc + S.lane_c];
    {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;  // four chains: no back-to-back dependent DPP instructions
#pragma unroll
        for (int i = 0; i < NR; i += 4) {
            fmac_bcast<KK>(s0, S.T[P][i], S.T[P][i]);
            fmac_bcast<KK>(s1, S.T[P][i + 1], S.T[P][i + 1]);
            fmac_bcast<KK>(s2, S.T[P][i + 2], S.T[P][i + 2]);
            fmac_bcast<KK>(s3, S.T[P][i + 3], S.T[P][i + 3]);
        }
        d[0] = tsqr2_reduce<LDSRED>(S, (s0 + s1) + (s2 + s3));
    }
    // row k of the triangle is requested before the dot product and pinned here, so that the LDS latency is not
    // part of the dependent chain below (the compiler would otherwise sink the read below the sigma branch)
#pragma unroll
    for (int cc = 0; cc < LIVE; ++cc) asm volatile("" : "+v"(Rk[cc]));
    const double alpha = row_bcast<KK>(Rk[0]);   // identical in all lanes (kept in VGPRs: no SGPR round trip)
    const double sigma = row_bcast<KK>(d[0]);
    if (__builtin_amdgcn_ballot_w64(sigma != 0.0) == 0) return;  // column zero below the triangle: H = I (dlarfg)
    // s = sqrt(alpha^2 + sigma), beta = -sign(alpha) s, inv = 1/(alpha - beta) = sign(alpha)/(|alpha| + s),
    // tfac = (beta - alpha)/beta = (|alpha| + s)/s: v_rsq_f64 / v_rcp_f64 seeds + Newton steps instead of the
    // IEEE sqrt and two divisions
    const double q2 = fma(alpha, alpha, sigma);
    const double hq = -0.5 * q2;
    double rs = __builtin_amdgcn_rsq(q2);
    rs = rs * fma(hq * rs, rs, 1.5);
    rs = rs * fma(hq * rs, rs, 1.5);
    const double dsum = fma(q2, rs, fabs(alpha));  // |alpha| + s
    double ri = __builtin_amdgcn_rcp(dsum);
    ri = ri * fma(-dsum, ri, 2.0);
    ri = ri * fma(-dsum, ri, 2.0);
    const double inv = copysign(ri, alpha);
    const double tfac = dsum * rs;
#pragma unroll
    for (int cc = 1; cc < LIVE; ++cc) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int i = 0; i < NR; i += 2) {
            fmac_bcast<KK>(s0, S.T[P][i], S.T[P + cc][i]);
            fmac_bcast<KK>(s1, S.T[P][i + 1], S.T[P + cc][i + 1]);
        }
        d[cc] = tsqr2_reduce<LDSRED>(S, s0 + s1);
    }
    // w_j = tau (R_kj + v^T B_j) for EVERY lane-column, no masks:
    //   - the pivot lane itself gets w = (alpha + sigma inv) tfac = alpha - beta, hence R_kk = alpha - w = beta and
    //     c = w inv = 1: its tile entries x - 1 x vanish (the finished column leaves the tile);
    //   - finished lane-columns (c < KK) and padding hold zeros (up to rounding residues that are never read as
    //     results), so their w is zero by itself.
    // Trailing chunks first, the pivot chunk last: its own update is the only write to the DPP source registers.
#pragma unroll
    for (int cc = LIVE - 1; cc >= 0; --cc) {
        const double wj = (Rk[cc] + d[cc] * inv) * tfac;
        const double ncj = -wj * inv;
#pragma unroll
        for (int i = 0; i < NR; ++i) fmac_bcast<KK>(S.T[P + cc][i], S.T[P][i], ncj);
        if (S.lane_g == 0) S.Rl[rowoff + 16 * cc + S.lane_c] = Rk[cc] - wj;
    }
}

// all column steps of panel P, then the next panel (compile-time recursion over the panels).  after(P) runs when
// chunk P is retired (its registers are dead for the rest of the tile): the kernel requests the next tile's chunk P
// into them there.
template <int P, int NCC, int NRC, bool TRI, bool LDSRED, class AfterPanel>
__device__ __forceinline__ void tsqr2_panels(Tsqr2State<NCC, NRC> &S, const int first_nz, AfterPanel &&after) {
    if (16 * P + 15 >= first_nz) {
#define FIGH_STEP(KK) \
    if (16 * P + KK >= first_nz) tsqr2_step<KK, P, NCC, NRC, TRI, LDSRED>(S);
        FIGH_STEP(0) FIGH_STEP(1) FIGH_STEP(2) FIGH_STEP(3) FIGH_STEP(4) FIGH_STEP(5) FIGH_STEP(6) FIGH_STEP(7)
        FIGH_STEP(8) FIGH_STEP(9) FIGH_STEP(10) FIGH_STEP(11) FIGH_STEP(12) FIGH_STEP(13) FIGH_STEP(14) FIGH_STEP(15)
#undef FIGH_STEP
    }
    after(std::integral_constant<int, P>{});
    if constexpr (P + 1 < NCC) tsqr2_panels<P + 1, NCC, NRC, TRI, LDSRED>(S, first_nz, after);
}

template <int NCC, int NRC, bool TRI, bool PROF = false, bool LDSRED = false>
__global__ __launch_bounds__(64, NCC <= 4 ? 2 : 1) void tsqr2_kernel(
    const double *__restrict__ W, const long rows, const long ldw, const int *__restrict__ col_idx, const int n,
    const double *__restrict__ tau, const double *__restrict__ blkw, const long rows_per_blk,
    double *__restrict__ Rws, const int nc, const int dbg, const int out_rows, long long *__restrict__ prof = nullptr,
    const int *__restrict__ tile_first = nullptr) {
    // tile_first[t] (always a valid array; zeros without a structure hint, figh_tsqr_structured): the first kept column
    // that can hold a non-zero in tile t.  Lanes in front of it are not read at all (their registers are zeroed, the
    // loads run under a narrower EXEC mask): in the joint-major regressor of a chain, row block j only involves the
    // links >= j, so 41 % of the kept entries of UR10 -- and of this kernel's HBM reads -- are known zeros.  (The array is
    // unconditional on purpose: a `hint != nullptr` test inside the tile loop gets the loop unswitched and costs 70
    // spilled registers.)
    // PROF (FIGH_TSQR_DBG & 4): per-wave s_memtime totals {kernel, load + delivery, factorisation, column steps}
    long long pc_load = 0, pc_fact = 0, pc_steps = 0;
    const long long pc_begin = PROF ? (long long)__builtin_readcyclecounter() : 0;
    // out_rows: row stride of the triangles written to Rws (nc = compact; 64 = one zero-padded R per 64-row tile,
    // the input format of the TRI merge levels).  dbg & 1: loads only, no factorisation (ablation).
    constexpr int RPL = 4 * NRC, M = 16 * NRC;
    extern __shared__ __attribute__((aligned(16))) double Rl[];  // packed triangle of the NCC panels
    const int lane = threadIdx.x;
    const long wave = blockIdx.x;
    // Tiles are dealt round-robin (tile t -> wave t mod nwaves): in the joint-major row order the number of
    // non-zero leading columns, hence the work per tile, depends on the joint block, and contiguous ranges
    // would leave the waves of the last joints idle while those of joint 1 finish.
    const long ntiles = (rows + M - 1) / M;
    const long tstep = gridDim.x;
    const long rend = rows;
    // The nc columns are RIGHT-aligned in the 16*NCC lane-columns (pad = 16 NCC - nc zero columns in front): a chunk
    // takes part in every step up to its last column, so the partially filled chunk must be the FIRST one -- for
    // nc = 50 the chunk holding 2 real columns is then live for 2 steps instead of 50 (-33 % chunk-steps on UR10).
    const int pad = 16 * NCC - nc;
    // LDS: [64 doubles of reduction scratch][packed triangle without the rows of the padding columns]
    int skip = 0;
    for (int kp = 0; kp < pad; ++kp) skip += 16 * (NCC - (kp >> 4));
    Tsqr2State<NCC, NRC> S;
    S.red = Rl;
    S.Rl = Rl + 64 - skip;
    S.lane_c = lane & 15;
    S.lane_g = lane >> 4;
    S.nc = nc;
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
        constexpr int tot = 256 * (NCC * NCC - (NCC * (NCC - 1)) / 2);
        for (int e = lane; e < 64 + tot - skip; e += 64) Rl[e] = 0.0;
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
    // The SIMD arbiter favours the older of its two resident waves, which then finishes ~25 % earlier and leaves
    // the younger one running alone (at a single wave's issue efficiency) for the rest of the kernel.  The two
    // halves of the grid therefore alternate their issue priority per tile, in antiphase, so that both waves of a
    // SIMD progress at a more even rate and the SIMD stays doubly occupied for longer (measured 1.120 -> 1.091 ms;
    // in the paired phase the SIMD is issue-bound, so this only shortens the single-wave tail).
    // (no ties: the younger half of the grid stays at priority 1, the older half alternates 2 / 0 per tile)
    const bool younger = wave >= tstep / 2;
    int prio_phase = 0;
    if (!(dbg & 8) && younger) __builtin_amdgcn_s_setprio(1);
    // Tile order: the wave's k-th tile is not tile wave + k*nwaves itself but its image under an 8-way interleave of
    // the row range (position p -> tile (p mod 8) * ceil(ntiles/8) + p / 8).  In the joint-major row order whole
    // row blocks are either compute-bound (rows of joint 1: all columns non-zero) or HBM-bound (rows of the last
    // joints: a handful of column steps per 43 KB tile); with the plain order every wave walks through the blocks in
    // lockstep and the kernel is a compute-bound phase followed by a bandwidth-bound phase.  Interleaved, each SIMD
    // sees both kinds at any time and the two bounds overlap.
    constexpr int GI = 8;
    const long npg = (ntiles + GI - 1) / GI;
    auto tile_at = [&](const long p) { return (dbg & 32) ? p : (p % GI) * npg + p / GI; };  // may be >= ntiles
    long pos = wave;
    while (pos < GI * npg && tile_at(pos) >= ntiles) pos += tstep;
    while (pos < GI * npg) {
        long posn = pos + tstep;
        while (posn < GI * npg && tile_at(posn) >= ntiles) posn += tstep;
        const long t = tile_at(pos);
        const long tn = posn < GI * npg ? tile_at(posn) : ntiles;
        pos = posn;
        if (!(dbg & 8) && !younger) {
            if (prio_phase & 1) __builtin_amdgcn_s_setprio(0);
            else __builtin_amdgcn_s_setprio(2);
            ++prio_phase;
        }
        long r0 = t * M;
        long r0n = tn * M;
        const bool fast = r0 + M <= rend;
        const bool next_fast = r0n + M <= rend;
        if (dbg & 16) {  // ablation: every tile re-reads one of the first 8 tiles of its sixth of the rows (L2-resident)
            const long blk = rows / 6;
            r0 = (r0 / blk) * blk + (r0 % blk) % (8 * M);
            r0n = (r0n / blk) * blk + (r0n % blk) % (8 * M);
        }
        const long long pc_a = PROF ? (long long)__builtin_readcyclecounter() : 0;
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
        if (dbg & 1) first_nz = 16 * NCC + 16;
        const long long pc_b = PROF ? (long long)__builtin_readcyclecounter() : 0;

        tsqr2_panels<0, NCC, NRC, TRI, LDSRED>(S, first_nz, [&](auto P) {
            if constexpr (decltype(P)::value < NCC - 1) {
                if (next_fast) load_chunk(P, r0n, fposn);
            }
        });
        prefetched = next_fast;
        if constexpr (PROF) {
            const long long pc_c = (long long)__builtin_readcyclecounter();
            pc_load += pc_b - pc_a;
            pc_fact += pc_c - pc_b;
            pc_steps += first_nz < 16 * NCC ? 16 * NCC - first_nz : 0;
        }
    }
    __syncthreads();
    if constexpr (PROF) {
        if (lane == 0 && prof) {
            prof[4 * wave + 0] = (long long)__builtin_readcyclecounter() - pc_begin;
            prof[4 * wave + 1] = pc_load;
            prof[4 * wave + 2] = pc_fact;
            prof[4 * wave + 3] = pc_steps;
        }
    }
    double *Rg = Rws + wave * (long)out_rows * nc;
    for (int e = lane; e < out_rows * nc; e += 64) {
        const int k = e / nc, col = e - k * nc;
        const int kp = k + pad, colp = col + pad;  // padded positions
        const int pk = kp >> 4;
        Rg[e] = (k >= nc || col < k)  // below the diagonal the LDS rows hold rounding residues, not results
                    ? 0.0
                    : S.Rl[256 * (pk * NCC - (pk * (pk - 1)) / 2) + (kp & 15) * 16 * (NCC - pk) + (colp - 16 * pk)];
    }
}

// ------------------------------------------------------------------------------------------------------------
// tsqr_wide_kernel<CPW>: 80 < n <= 512 columns (TIAGo 241, TALOS 331, human 191; 400 for the human SIP program).  The tile is too wide for one
// wave, so a workgroup of 8 waves splits the COLUMNS of the same 64 rows: wave w owns the 16-column chunks
// w, w+8, w+16 (CPW per wave) in the C-layout registers of tsqr2.  Per column step the owner of the pivot chunk
// broadcasts the pivot column (64 doubles) and alpha through LDS (ping-pong buffers: one barrier per step), every
// wave forms sigma = x^T x and its own dot products, and updates its own chunks and its part of row k of the
// triangle, which lives in global memory (one private nc x nc triangle per workgroup; row k+1 is prefetched while
// step k runs).  Tiles are dealt round-robin to the workgroups; the same kernel reduces the stacked triangles.
template <int KK, int CPW>
__device__ __forceinline__ void tsqr_wide_step(double (&T)[CPW][16], const int p, const int nchunks, const int nc,
                                               const int lane_c, const int lane_g, const int wave,
                                               double (*xl)[80], double *__restrict__ red,
                                               double *__restrict__ Rg, double (&Rk)[CPW], double (&Rn)[CPW]) {
    constexpr int NW = 8;
    const int k = 16 * p + KK;
    const int buf = k & 1;
    const int wo = p & (NW - 1), so = p >> 3;  // owner wave and its slot of the pivot chunk
    // The owner of the pivot chunk publishes, through LDS, the pivot column (row group g's 16 values contiguous:
    // xl[buf][16 g + 4 rc + r]) AND the Householder scalars inv = 1/(alpha - beta), tfac = tau -- it has x and alpha
    // in registers anyway, and the other seven waves then skip sigma = x^T x (16 FMAs + a cross-row reduction) and
    // the 18-operation rsq/rcp chain.  The kernel is issue-bound at two waves per SIMD, so the instructions saved in
    // the non-owners are time saved; the critical path (chain before the barrier instead of after it) is unchanged.
    if (wave == wo) {
        double xo[16];
        double a0 = 0.0;
#pragma unroll
        for (int s = 0; s < CPW; ++s)
            if (s == so) {
#pragma unroll
                for (int i = 0; i < 16; ++i) xo[i] = row_bcast<KK>(T[s][i]);
                a0 = row_bcast<KK>(Rk[s]);
            }
        double ss0 = 0.0, ss1 = 0.0;
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            ss0 = fma(xo[i], xo[i], ss0);
            ss1 = fma(xo[i + 1], xo[i + 1], ss1);
        }
        const double sigma = allreduce_rowgroups(ss0 + ss1);
        double inv = 0.0, tfac = 0.0;  // sigma == 0: H = I
        if (uniform_of(sigma) != 0.0) {
            const double q2 = fma(a0, a0, sigma);
            const double hq = -0.5 * q2;
            double rs = __builtin_amdgcn_rsq(q2);
            rs = rs * fma(hq * rs, rs, 1.5);
            rs = rs * fma(hq * rs, rs, 1.5);
            const double dsum = fma(q2, rs, fabs(a0));
            double ri = __builtin_amdgcn_rcp(dsum);
            ri = ri * fma(-dsum, ri, 2.0);
            ri = ri * fma(-dsum, ri, 2.0);
            inv = copysign(ri, a0);
            tfac = dsum * rs;
        }
        if (lane_c == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) xl[buf][16 * lane_g + i] = xo[i];
            if (lane_g == 0) {
                xl[buf][64] = inv;
                xl[buf][65] = tfac;
            }
        }
    }
    __syncthreads();
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = xl[buf][16 * lane_g + i];
    const double inv = xl[buf][64], tfac = xl[buf][65];
    // prefetch row k+2 of the triangle (rows k+1, k+2 are not touched by this step; two steps of distance cover the
    // L2 round trip, one does not)
    double Rn3[CPW];
#pragma unroll
    for (int s = 0; s < CPW; ++s) {
        const int col = 16 * (wave + NW * s) + lane_c;
        Rn3[s] = (k + 2 < nc && col < nc && wave + NW * s < nchunks) ? Rg[(long)(k + 2) * nc + col] : 0.0;
    }
    if (uniform_of(tfac) != 0.0) {
        // w_j = tau (R_kj + v^T B_j) in every lane-column, no masks: the pivot lane gets w = alpha - beta (so
        // R_kk = alpha - w = beta) and c = 1 (its tile entries cancel); finished columns hold (near) zeros
#pragma unroll
        for (int s = 0; s < CPW; ++s) {
            const int chunk = wave + NW * s;
            if (chunk >= p && chunk < nchunks) {
                const int col = 16 * chunk + lane_c;
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    s0 += x[i] * T[s][i];
                    s1 += x[i + 1] * T[s][i + 1];
                }
                const double d = allreduce_rowgroups_lds(red, 16 * lane_g + lane_c, s0 + s1);
                const double wj = (Rk[s] + d * inv) * tfac;
                const double cj = wj * inv;
#pragma unroll
                for (int i = 0; i < 16; ++i) T[s][i] -= cj * x[i];
                if (lane_g == 0 && col >= k && col < nc) Rg[(long)k * nc + col] = Rk[s] - wj;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < CPW; ++s) {
        Rk[s] = Rn[s];
        Rn[s] = Rn3[s];
    }
}

template <int CPW>
__global__ __launch_bounds__(512) void tsqr_wide_kernel(const double *__restrict__ W, const long rows, const long ldw,
                                                        const int *__restrict__ col_idx, const int n,
                                                        const double *__restrict__ tau, const double *__restrict__ blkw,
                                                        const long rows_per_blk, double *__restrict__ Rws, const int nc) {
    constexpr int NW = 8;
    __shared__ double xl[2][80];
    __shared__ double redbuf[NW][64];  // per-wave scratch of the cross-row-group sums
    __shared__ int fnz[NW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lane_c = lane & 15, lane_g = lane >> 4;
    const int nchunks = (nc + 15) >> 4;
    double *Rg = Rws + (long)blockIdx.x * nc * nc;
    for (long e = threadIdx.x; e < (long)nc * nc; e += 512) Rg[e] = 0.0;
    __syncthreads();

    const double *src[CPW];
    long stride[CPW];
    bool live[CPW];
#pragma unroll
    for (int s = 0; s < CPW; ++s) {
        const int col = 16 * (wave + NW * s) + lane_c;
        if (col < n) {
            src[s] = W + (col_idx ? col_idx[col] : col);
            stride[s] = ldw;
            live[s] = true;
        } else if (col == n && tau != nullptr) {
            src[s] = tau;
            stride[s] = 1;
            live[s] = true;
        } else {
            src[s] = W;
            stride[s] = 0;
            live[s] = false;
        }
    }
    const long ntiles = (rows + 63) / 64;
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long r0 = t * 64;
        double T[CPW][16];
#pragma unroll
        for (int s = 0; s < CPW; ++s)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const long row = r0 + 16 * (i >> 2) + lane_g + 4 * (i & 3);
                const bool inb = row < rows;
                const long rowc = inb ? row : rows - 1;
                double scale = 1.0;
                if (blkw) scale = blkw[rowc / rows_per_blk];
                const double v = src[s][rowc * stride[s]];
                T[s][i] = (inb && live[s]) ? v * scale : 0.0;
            }
        // first column with a non-zero in this tile (all 8 waves): the steps before it are identities.  Stacked
        // triangles (merge levels) and the joint-torque rows of a tree (row block j only touches the links of its
        // subtree) start far to the right.
        int myfirst = nc;
#pragma unroll
        for (int s = CPW - 1; s >= 0; --s) {
            bool nz = false;
#pragma unroll
            for (int i = 0; i < 16; ++i) nz |= (T[s][i] != 0.0);
            const unsigned long long b = __ballot(nz);
            const unsigned m16 = (unsigned)((b | (b >> 16) | (b >> 32) | (b >> 48)) & 0xffffull);
            if (m16) myfirst = 16 * (wave + NW * s) + __ffs((int)m16) - 1;
        }
        if (lane == 0) fnz[wave] = myfirst;
        __syncthreads();
        int first_nz = fnz[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) first_nz = min(first_nz, fnz[w]);
        first_nz = __builtin_amdgcn_readfirstlane(first_nz);
        double Rk[CPW], Rn[CPW];  // rows first_nz and first_nz + 1 of the triangle; step k requests row k + 2
#pragma unroll
        for (int s = 0; s < CPW; ++s) {
            const int col = 16 * (wave + NW * s) + lane_c;
            const bool ok = col < nc && wave + NW * s < nchunks;
            Rk[s] = (first_nz < nc && ok) ? Rg[(long)first_nz * nc + col] : 0.0;
            Rn[s] = (first_nz + 1 < nc && ok) ? Rg[(long)(first_nz + 1) * nc + col] : 0.0;
        }
        for (int p = first_nz >> 4; p < nchunks; ++p) {
#define FIGH_WSTEP(KK) \
    if (16 * p + KK >= first_nz && 16 * p + KK < nc) \
        tsqr_wide_step<KK, CPW>(T, p, nchunks, nc, lane_c, lane_g, wave, xl, redbuf[wave], Rg, Rk, Rn);
            FIGH_WSTEP(0) FIGH_WSTEP(1) FIGH_WSTEP(2) FIGH_WSTEP(3) FIGH_WSTEP(4) FIGH_WSTEP(5) FIGH_WSTEP(6)
            FIGH_WSTEP(7) FIGH_WSTEP(8) FIGH_WSTEP(9) FIGH_WSTEP(10) FIGH_WSTEP(11) FIGH_WSTEP(12) FIGH_WSTEP(13)
            FIGH_WSTEP(14) FIGH_WSTEP(15)
#undef FIGH_WSTEP
        }
        __syncthreads();  // every wave's row stores of this tile precede the next tile's row loads
    }
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

template <int NCC, int NW>
__global__ __launch_bounds__(64 * NW) void tsqr_coop_kernel(const double *__restrict__ Rs, const long rows, const int nc,
                                                            double *__restrict__ Rout) {
    __shared__ double pw[2][NW][16 * NCC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lane_c = lane & 15, lane_g = lane >> 4;
    const long r0 = ((long)blockIdx.x * NW + wave) * 64;
    const int pad = 16 * NCC - nc;  // columns right-aligned, as in tsqr2_kernel
    double *Rg = Rout + (long)blockIdx.x * nc * nc;
    for (int e = threadIdx.x; e < nc * nc; e += 64 * NW) Rg[e] = 0.0;
    double T[NCC][16];
#pragma unroll
    for (int cc = 0; cc < NCC; ++cc)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const long row = r0 + 16 * (i >> 2) + lane_g + 4 * (i & 3);
            const int col = 16 * cc + lane_c - pad;
            const bool ok = row < rows && col >= 0;
            const double v = Rs[(ok ? row : 0) * nc + (ok ? col : 0)];
            T[cc][i] = ok ? v : 0.0;
        }
    __syncthreads();  // the zero fill of Rg is ordered before the row stores of wave 0 (same workgroup)
    tsqr_coop_panels<0, NCC, NW>(T, nc, pad, lane_c, lane_g, wave, pw, Rg);
}

// ------------------------------------------------------------------------------------------------------------
// tsqr3_kernel<NCC>: blocked Householder (compact WY) on the same register tile.  A 16-column panel is factored
// with the DPP / permlane step above restricted to the panel itself; its 16 reflectors are then applied to every
// trailing 16-column chunk as three v_mfma_f64_16x16x4_f64 contractions
//        G  = R_pt + V^T B      (16 MFMA, A = V and B = the tile chunk, both straight from the tile registers:
//                                the f64 C/D layout row = g + 4 reg IS the A/B operand layout of K-slice `reg`)
//        Wm = T^T G             ( 4 MFMA)
//        R_pt -= Wm,  B -= V Wm (16 MFMA, V transposed once per panel through 8.7 KB of LDS)
// T is built column by column during the panel from the Gram entries v_c^T v_k, which the panel's own dot
// products already deliver for the finished columns c < k (T^-1 = striu(V^T V) + diag(1/tau), LAPACK larft).
typedef double f64x4 __attribute__((ext_vector_type(4)));


template <int NCC>
struct Tsqr3State {
    f64x4 T[NCC][4];   // [col chunk][row chunk]: lane (g, c) holds rows 16 rc + g + 4 r, column 16 cc + c
    double Trow[16];   // row c of the current panel's T factor
    double myinv;      // 1 / (alpha - beta) of reflector c
    int lane_c, lane_g;
};

// sum over the finished reflectors m < KK of T[c][m] * vg(lane-column m): DPP lane selects must be immediates
template <int M0, int KK>
struct TColumn {
    static __device__ __forceinline__ double dot(const double *Trow, const double vg) {
        return fma(Trow[M0], row_bcast<M0>(vg), TColumn<M0 + 1, KK>::dot(Trow, vg));
    }
};
template <int KK>
struct TColumn<KK, KK> {
    static __device__ __forceinline__ double dot(const double *, const double) { return 0.0; }
};

// one column step of panel P (pivot = lane-column KK of chunk P); Rrow = LDS row 16P+KK from column 16P on
template <int P, int KK, int NCC>
__device__ __forceinline__ void tsqr3_step(Tsqr3State<NCC> &S, double *__restrict__ Rrow) {
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = row_bcast<KK>(S.T[P][i >> 2][i & 3]);
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        s0 += x[i] * S.T[P][i >> 2][i & 3];
        s1 += x[i + 1] * S.T[P][(i + 1) >> 2][(i + 1) & 3];
    }
    const double d = allreduce_rowgroups(s0 + s1);
    const double sigma = uniform_of(row_bcast<KK>(d));
    if (sigma == 0.0) return;  // H = I: T row/column KK and V column KK stay zero
    const double rk = Rrow[S.lane_c];
    const double alpha = uniform_of(row_bcast<KK>(rk));
    const double q2 = fma(alpha, alpha, sigma);
    double rs = __builtin_amdgcn_rsq(q2);
    rs = rs * fma(-0.5 * q2 * rs, rs, 1.5);
    rs = rs * fma(-0.5 * q2 * rs, rs, 1.5);
    double sq = q2 * rs;
    sq = fma(fma(-sq, sq, q2), 0.5 * rs, sq);
    const double dsum = fabs(alpha) + sq;
    double ri = __builtin_amdgcn_rcp(dsum);
    ri = ri * fma(-dsum, ri, 2.0);
    ri = ri * fma(-dsum, ri, 2.0);
    const double beta = -copysign(sq, alpha);
    const double inv = copysign(ri, alpha);
    const double tfac = dsum * rs;
    const bool trail = S.lane_c > KK;
    const double wj = trail ? (rk + d * inv) * tfac : 0.0;
    const double cj = wj * inv;
    if (S.lane_g == 0) Rrow[S.lane_c] = (S.lane_c == KK) ? beta : rk - wj;
#pragma unroll
    for (int i = 0; i < 16; ++i) S.T[P][i >> 2][i & 3] -= cj * x[i];
    // Gram entry v_c^T v_KK of the finished columns c < KK, then column KK of T
    const double vg = (S.lane_c < KK) ? d * S.myinv * inv : 0.0;
    if (S.lane_c == KK) S.myinv = inv;
    const double acc = TColumn<0, KK>::dot(S.Trow, vg);  // sum_{m<KK} T[c][m] * (v_m^T v_KK)
    S.Trow[KK] = (S.lane_c < KK) ? -tfac * acc : ((S.lane_c == KK) ? tfac : 0.0);
}

// panel P of the current tile: factor chunk P, then apply its reflectors to chunks P+1 .. np-1 with MFMA
template <int P, int NCC>
__device__ __forceinline__ void tsqr3_panel(Tsqr3State<NCC> &S, double *__restrict__ Rl, double *__restrict__ Vl,
                                            double *__restrict__ Tl, const int first_nz, const int nc, const int np) {
    constexpr int LDR = 16 * NCC, LDV = 17;
    if (16 * P + 15 < first_nz || 16 * P >= nc) return;  // all 16 columns zero in this tile, or padding: H = I
    const int c = S.lane_c, g = S.lane_g;
#pragma unroll
    for (int i = 0; i < 16; ++i) S.Trow[i] = 0.0;
    S.myinv = 0.0;
    double *Rdiag = Rl + (16 * P) * LDR + 16 * P;
#define FIGH_STEP3(KK) \
    if (16 * P + KK >= first_nz && 16 * P + KK < nc) tsqr3_step<P, KK, NCC>(S, Rdiag + KK * LDR);
    FIGH_STEP3(0) FIGH_STEP3(1) FIGH_STEP3(2) FIGH_STEP3(3) FIGH_STEP3(4) FIGH_STEP3(5) FIGH_STEP3(6) FIGH_STEP3(7)
    FIGH_STEP3(8) FIGH_STEP3(9) FIGH_STEP3(10) FIGH_STEP3(11) FIGH_STEP3(12) FIGH_STEP3(13) FIGH_STEP3(14)
    FIGH_STEP3(15)
#undef FIGH_STEP3
    if constexpr (P + 1 < NCC) {
        if (P + 1 >= np) return;  // no trailing chunk holds real columns
        // ---- V = X diag(inv), in place (chunk P is finished): A operand of V^T B as is; -V goes transposed through
        // LDS for B -= V Wm; T goes through LDS to become the A operand of T^T G
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const double vv = S.T[P][i >> 2][i & 3] * S.myinv;
            S.T[P][i >> 2][i & 3] = vv;
            Vl[(16 * (i >> 2) + g + 4 * (i & 3)) * LDV + c] = -vv;
        }
        if (g == 0) {
#pragma unroll
            for (int m = 0; m < 16; ++m) Tl[c * 16 + m] = S.Trow[m];
        }
        __syncthreads();
#pragma unroll
        for (int cc = P + 1; cc < NCC; ++cc) {
            if (cc < np) {
                double Tt[4], Vt[4][4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    Tt[s] = Tl[(4 * s + g) * 16 + c];
#pragma unroll
                    for (int rc = 0; rc < 4; ++rc) Vt[rc][s] = Vl[(16 * rc + c) * LDV + 4 * s + g];
                }
                f64x4 Rpt, G0, G1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 4; ++r) Rpt[r] = Rl[(16 * P + g + 4 * r) * LDR + 16 * cc + c];
                G0 = Rpt;
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    G0 = __builtin_amdgcn_mfma_f64_16x16x4f64(S.T[P][i >> 2][i & 3], S.T[cc][i >> 2][i & 3], G0, 0, 0, 0);
                    G1 = __builtin_amdgcn_mfma_f64_16x16x4f64(S.T[P][(i + 1) >> 2][(i + 1) & 3],
                                                              S.T[cc][(i + 1) >> 2][(i + 1) & 3], G1, 0, 0, 0);
                }
                const f64x4 G = G0 + G1;
                f64x4 Wm = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 4; ++s) Wm = __builtin_amdgcn_mfma_f64_16x16x4f64(Tt[s], G[s], Wm, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) Rl[(16 * P + g + 4 * r) * LDR + 16 * cc + c] = Rpt[r] - Wm[r];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int rc = 0; rc < 4; ++rc)
                        S.T[cc][rc] = __builtin_amdgcn_mfma_f64_16x16x4f64(Vt[rc][s], Wm[s], S.T[cc][rc], 0, 0, 0);
            }
        }
        __syncthreads();
    }
}

template <int NCC, bool PF>
__global__ __launch_bounds__(64) void tsqr3_kernel(const double *__restrict__ W, const long rows, const long ldw,
                                                   const int *__restrict__ col_idx, const int n,
                                                   const double *__restrict__ tau, const double *__restrict__ blkw,
                                                   const long rows_per_blk, double *__restrict__ Rws, const int nc,
                                                   const int dbg) {
    constexpr int M = 64, LDR = 16 * NCC, LDV = 17;
    extern __shared__ __attribute__((aligned(16))) double lds3[];
    double *Rl = lds3;                 // nc x LDR
    double *Vl = lds3 + nc * LDR;      // 64 x LDV: -V, row-major
    double *Tl = Vl + 64 * LDV;        // 16 x 16: T, row-major
    const int lane = threadIdx.x;
    const long wave = blockIdx.x;
    const long ntiles = (rows + M - 1) / M;
    const long tstep = gridDim.x;
    const long rend = rows;
    Tsqr3State<NCC> S;
    S.lane_c = lane & 15;
    S.lane_g = lane >> 4;
    const int c = S.lane_c, g = S.lane_g;
    const int np = (dbg & 1) ? 0 : (nc + 15) >> 4;

    const double *src[NCC];
    long stride[NCC];
    bool livecol[NCC];
#pragma unroll
    for (int cc = 0; cc < NCC; ++cc) {
        const int col = 16 * cc + c;
        if (col < n) {
            src[cc] = W + (col_idx ? col_idx[col] : col);
            stride[cc] = ldw;
            livecol[cc] = true;
        } else if (col == n && tau != nullptr) {
            src[cc] = tau;
            stride[cc] = 1;
            livecol[cc] = true;
        } else {
            src[cc] = W;
            stride[cc] = 0;
            livecol[cc] = false;
        }
    }
    for (int e = lane; e < nc * LDR; e += 64) Rl[e] = 0.0;
    __syncthreads();

    // raw loads of a tile (no dependent instruction: all 16*NCC requests in flight together)
    double Tn[NCC][16];
    auto request_tile = [&](const long r0) {
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const long row = r0 + 16 * (i >> 2) + g + 4 * (i & 3);
                const long rowc = row < rend ? row : rend - 1;
                Tn[cc][i] = (dbg & 2) ? 1.0 + (double)(cc * 16 + i + lane) : src[cc][rowc * stride[cc]];
            }
    };
    if constexpr (PF) {
        if (wave < ntiles) request_tile(wave * M);
    }

    for (long t = wave; t < ntiles; t += tstep) {
        const long r0 = t * M;
        if constexpr (!PF) request_tile(r0);
        unsigned long long nzlo = 0;
        unsigned nzhi = 0;
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc) {
            bool nz = false;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const long row = r0 + 16 * (i >> 2) + g + 4 * (i & 3);
                const bool inb = row < rend;
                double scale = 1.0;
                if (blkw) scale = blkw[(inb ? row : rend - 1) / rows_per_blk];
                const double val = (inb && livecol[cc]) ? Tn[cc][i] * scale : 0.0;
                S.T[cc][i >> 2][i & 3] = val;
                nz |= (val != 0.0);
            }
            const unsigned long long b = __ballot(nz);
            const unsigned m16 = (unsigned)((b | (b >> 16) | (b >> 32) | (b >> 48)) & 0xffffull);
            if (cc < 4) nzlo |= (unsigned long long)m16 << (16 * cc);
            else nzhi |= m16 << (16 * (cc - 4));
        }
        int first_nz = nc;
        if (nzlo) first_nz = __ffsll((long long)nzlo) - 1;
        else if (nzhi) first_nz = 64 + __ffs((int)nzhi) - 1;

        // Panels are unrolled at compile time, so the register allocator sees that only chunk NCC-1 is alive
        // during the last panel: the next tile is requested there (software prefetch into the freed registers;
        // one wave per SIMD cannot rely on other waves to hide HBM latency).
        tsqr3_panel<0, NCC>(S, Rl, Vl, Tl, first_nz, nc, np);
        if constexpr (NCC > 2) tsqr3_panel<1, NCC>(S, Rl, Vl, Tl, first_nz, nc, np);
        if constexpr (NCC > 3) tsqr3_panel<2, NCC>(S, Rl, Vl, Tl, first_nz, nc, np);
        if constexpr (NCC > 4) tsqr3_panel<3, NCC>(S, Rl, Vl, Tl, first_nz, nc, np);
        if constexpr (PF) {
            if (t + tstep < ntiles) request_tile((t + tstep) * M);
        }
        tsqr3_panel<NCC - 1, NCC>(S, Rl, Vl, Tl, first_nz, nc, np);
    }
    __syncthreads();
    double *Rg = Rws + wave * (long)nc * nc;
    for (int e = lane; e < nc * nc; e += 64) {
        const int k = e / nc, col = e - k * nc;
        Rg[e] = Rl[k * LDR + col];
    }
}

template <int CPL, int M, bool RLDS>
__global__ __launch_bounds__(64) void tsqr_kernel(const double *__restrict__ W, const long rows, const long ldw,
                                                  const int *__restrict__ col_idx, const int n,
                                                  const double *__restrict__ tau, const double *__restrict__ blkw,
                                                  const long rows_per_blk, const long rows_per_wave,
                                                  double *__restrict__ Rws, const int nc) {
    const int lane = threadIdx.x;
    const long wave = blockIdx.x;
    const long rbeg = wave * rows_per_wave;
    const long rend = (rbeg + rows_per_wave < rows) ? rbeg + rows_per_wave : rows;

    __shared__ double Rl[RLDS ? 64 * 64 : 1];
    double *Rg = Rws + wave * (long)nc * nc;  // private triangle; also the working copy when !RLDS

    // per-lane column sources: W[:, col_idx[col]] (stride ldw), tau (stride 1) or nothing.  Loads are
    // unconditional (clamped row, select afterwards) so a tile's M loads are all in flight together.
    const double *src[CPL];
    long stride[CPL];
    bool live[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int col = lane + 64 * c;
        if (col < n) {
            src[c] = W + (col_idx ? col_idx[col] : col);
            stride[c] = ldw;
            live[c] = true;
        } else if (col == n && tau != nullptr) {
            src[c] = tau;
            stride[c] = 1;
            live[c] = true;
        } else {
            src[c] = W;
            stride[c] = 0;
            live[c] = false;
        }
    }
    if constexpr (RLDS) {
        for (int k = 0; k < 64; ++k) Rl[k * 64 + lane] = 0.0;
    } else {
        for (int k = 0; k < nc; ++k)
#pragma unroll
            for (int c = 0; c < CPL; ++c)
                if (lane + 64 * c < nc) Rg[(long)k * nc + lane + 64 * c] = 0.0;
    }
    __syncthreads();

    auto load_tile = [&](double (&T)[CPL][M], const long r0) {
        long blk = blkw ? r0 / rows_per_blk : 0;
        long next_blk = (blk + 1) * rows_per_blk;
#pragma unroll
        for (int r = 0; r < M; ++r) {
            const long row = r0 + r;
            const bool inb = row < rend;
            const long rowc = inb ? row : rend - 1;
            double scale = 1.0;
            if (blkw) {
                if (rowc >= next_blk) {
                    ++blk;
                    next_blk += rows_per_blk;
                }
                scale = blkw[blk];
            }
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const double x = src[c][rowc * stride[c]];
                T[c][r] = (inb && live[c]) ? x * scale : 0.0;
            }
        }
    };

    for (long r0 = rbeg; r0 < rend; r0 += M) {
        double B[CPL][M];
        load_tile(B, r0);
        int kstart = nc;
#pragma unroll
        for (int c = CPL - 1; c >= 0; --c) {
            bool nz = false;
#pragma unroll
            for (int r = 0; r < M; ++r) nz |= (B[c][r] != 0.0);
            const unsigned long long mask = __ballot(nz);
            if (mask) kstart = 64 * c + (__ffsll((long long)mask) - 1);
        }

        for (int k = kstart; k < nc; ++k) {
            const int kc = k >> 6, kl = k & 63;
            double xs[M];
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                if (c == kc) {
#pragma unroll
                    for (int r = 0; r < M; ++r) xs[r] = bcast_v(B[c][r], kl);
                }
            }
            double d[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
                for (int r = 0; r < M; r += 4) {
                    s0 += xs[r] * B[c][r];
                    s1 += xs[r + 1] * B[c][r + 1];
                    s2 += xs[r + 2] * B[c][r + 2];
                    s3 += xs[r + 3] * B[c][r + 3];
                }
                d[c] = (s0 + s1) + (s2 + s3);
            }
            double dk = 0.0;
#pragma unroll
            for (int c = 0; c < CPL; ++c)
                if (c == kc) dk = d[c];
            const double sigma = bcast_s(dk, kl);
            if (sigma == 0.0) continue;  // column already zero below the triangle: H = I (LAPACK dlarfg)

            double Rk[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int col = lane + 64 * c;
                if constexpr (RLDS) {
                    Rk[c] = Rl[k * 64 + lane];
                } else {
                    Rk[c] = col < nc ? Rg[(long)k * nc + col] : 0.0;
                }
            }
            double rkk = 0.0;
#pragma unroll
            for (int c = 0; c < CPL; ++c)
                if (c == kc) rkk = Rk[c];
            const double alpha = bcast_s(rkk, kl);
            const double beta = -copysign(sqrt(alpha * alpha + sigma), alpha);
            const double inv = 1.0 / (alpha - beta);
            const double tfac = (beta - alpha) / beta;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int col = lane + 64 * c;
                const bool trail = col > k && col < nc;
                const double wj = trail ? (Rk[c] + d[c] * inv) * tfac : 0.0;
                const double cj = wj * inv;
                Rk[c] = (col == k) ? beta : Rk[c] - wj;
#pragma unroll
                for (int r = 0; r < M; ++r) B[c][r] -= cj * xs[r];
                if constexpr (RLDS) {
                    Rl[k * 64 + lane] = Rk[c];
                } else {
                    if (col < nc) Rg[(long)k * nc + col] = Rk[c];
                }
            }
        }
    }
    if constexpr (RLDS) {
        __syncthreads();
        for (int k = 0; k < nc; ++k)
            if (lane < nc) Rg[(long)k * nc + lane] = Rl[k * 64 + lane];
    }
}

// diag(W^T W): block b owns a slab of rows; thread t owns columns t, t+256, ...; partial[b][c] then a
// fixed-order reduction (deterministic).
__global__ __launch_bounds__(256) void colsq_kernel(const double *__restrict__ W, long rows, int cols, long ldw,
                                                    long rows_per_block, double *__restrict__ part) {
    const long rb = (long)blockIdx.x * rows_per_block;
    const long re = rb + rows_per_block < rows ? rb + rows_per_block : rows;
    for (int c = threadIdx.x; c < cols; c += blockDim.x) {
        double s0 = 0.0, s1 = 0.0;
        long r = rb;
        for (; r + 1 < re; r += 2) {
            const double x0 = W[r * ldw + c], x1 = W[(r + 1) * ldw + c];
            s0 += x0 * x0;
            s1 += x1 * x1;
        }
        if (r < re) {
            const double x0 = W[r * ldw + c];
            s0 += x0 * x0;
        }
        part[(long)blockIdx.x * cols + c] = s0 + s1;
    }
}

__global__ __launch_bounds__(256) void reduce_cols_kernel(const double *__restrict__ part, int nblocks, int ncols,
                                                          double *__restrict__ out) {
    __shared__ double sm[256];
    const int c = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += part[(long)b * ncols + c];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = sm[0];
}

__global__ __launch_bounds__(256) void gather_cols_kernel(const double *__restrict__ W, long rows, long ldw,
                                                          const int *__restrict__ col_idx, int n,
                                                          double *__restrict__ out, long ldo) {
    const long total = rows * n;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long r = e / n;
        const int c = (int)(e - r * n);
        out[r * ldo + c] = W[r * ldw + col_idx[c]];
    }
}

// qrdecomposition.py:215-236: idx_base = {i : |R_ii| > tol}, then the regrouped column order [base | rest | tau].
// One wave, stable partition by ballot prefix counts; n <= 512.
__global__ __launch_bounds__(64) void base_permutation_kernel(const double *__restrict__ R, const int nc, const int n,
                                                              const double tol, int *__restrict__ perm) {
    const int lane = threadIdx.x;
    int nbase = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {  // count the base columns
        const int i = i0 + lane;
        const bool big = i < n && fabs(R[(long)i * nc + i]) > tol;
        nbase += __popcll(__ballot(big));
    }
    int pb = 0, pr = nbase;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const bool in = i < n;
        const bool big = in && fabs(R[(long)i * nc + i]) > tol;
        const unsigned long long mb = __ballot(big), mr = __ballot(in && !big);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (big) perm[pb + __popcll(mb & below)] = i;
        else if (in) perm[pr + __popcll(mr & below)] = i;
        pb += __popcll(mb);
        pr += __popcll(mr);
    }
    for (int i = n + lane; i < nc; i += 64) perm[i] = i;  // the tau column stays last
}

// y[r] = sum_c W[r, idx[c]] x[c]; one wave per row-group, lanes across columns, wave reduction
__global__ __launch_bounds__(256) void matvec_kernel(const double *__restrict__ W, long rows, long ldw,
                                                     const int *__restrict__ col_idx, int n,
                                                     const double *__restrict__ x, double *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    for (long r = wave; r < rows; r += nwaves) {
        double s = 0.0;
        for (int c = lane; c < n; c += 64) s += W[r * ldw + (col_idx ? col_idx[c] : c)] * x[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) y[r] = s;
    }
}

// out[b] = sum over block b of (a - b)^2, one workgroup per block, fixed reduction order
__global__ __launch_bounds__(256) void block_sqnorm_kernel(const double *__restrict__ a, const double *__restrict__ b,
                                                           long rows_per_block, double *__restrict__ out) {
    __shared__ double sm[256];
    const long base = (long)blockIdx.x * rows_per_block;
    double s = 0.0;
    for (long r = threadIdx.x; r < rows_per_block; r += 256) {
        const double d = a[base + r] - (b ? b[base + r] : 0.0);
        s += d * d;
    }
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sm[0];
}

static int cu_count() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    return cus;
}

// A/B switches (within-run comparisons).  Default: tsqr2 (unblocked DPP/permlane kernel, 2 waves per SIMD, no
// register prefetch) -- measured fastest: 2.29 ms vs 2.98 ms with register prefetch at 1 wave/SIMD, 3.44 ms for the
// blocked MFMA kernel tsqr3 (1 wave/SIMD: its column steps are a ~1000-cycle dependent chain that only a second
// wave can fill), 10.0 ms for the round-1 ds_bpermute kernel.  FIGH_TSQR_V1 / FIGH_TSQR_V3 select the others,
// FIGH_TSQR_PF the register-prefetch variants, FIGH_TSQR_DBG ablates (1 = no factorisation, 2 = no loads).
static const bool g_force_v1 = getenv("FIGH_TSQR_V1") != nullptr;
static const bool g_force_v2 = getenv("FIGH_TSQR_V3") == nullptr;
static const bool g_pf = getenv("FIGH_TSQR_PF") != nullptr;
static const int g_dbg = getenv("FIGH_TSQR_DBG") ? atoi(getenv("FIGH_TSQR_DBG")) : 0;

// LDS of one tsqr2 wave: 64 doubles of reduction scratch + the packed triangle minus the rows of the padding columns
static size_t tsqr2_lds_bytes(int ncc, int nc) {
    const int pad = 16 * ncc - nc;
    size_t skip = 0;
    for (int kp = 0; kp < pad; ++kp) skip += 16 * (ncc - (kp >> 4));
    return sizeof(double) * (64 + 256 * (size_t)(ncc * ncc - (ncc * (ncc - 1)) / 2) - skip);
}

// per-tile structure hint of the register-tile kernel: g_tile_hint (set by figh_tsqr_structured for the next level-0
// launch) or an all-zero array
static const int *g_tile_hint = nullptr;

__global__ __launch_bounds__(256) void tile_hint_kernel(const int *__restrict__ first, const long hint_rows,
                                                        const long rows, const long ntiles, int *__restrict__ out) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= ntiles) return;
    const long r0 = t * 64, r1 = (r0 + 63 < rows ? r0 + 63 : rows - 1);
    int f = first[r0 / hint_rows];
    for (long b = r0 / hint_rows + 1; b <= r1 / hint_rows; ++b) f = min(f, first[b]);
    out[t] = f;
}

static const int *tile_hint_or_zeros(long ntiles) {
    if (g_tile_hint) return g_tile_hint;
    static size_t zeroed = 0;
    const size_t need = sizeof(int) * (size_t)(ntiles + 1);
    int *z = static_cast<int *>(workspace(need, 15));
    if (z && zeroed < need) {  // (re)allocated: workspace() hands out at least `need` bytes, growing by 25 %
        if (hipMemsetAsync(z, 0, need, stream()) != hipSuccess) return nullptr;
        zeroed = need;
    }
    return z;
}

// one TSQR level: rows of (W, ldw) -> nw triangles in Rws.  Returns nw (>0) or a negative status.
static int tsqr_level(const double *W, long rows, long ldw, const int *col_idx, int n, const double *tau,
                       const double *d_blkw, long rows_per_blk, int nc, long target_waves, long align,
                       double *Rws_out, long *nw_out, int out_rows = 0, bool tri = false) {
    if (out_rows == 0) out_rows = nc;
    int M;
    if (nc <= 64) M = 64;
    else if (nc <= 128) M = 32;
    else M = 16;
    if (align < M) align = M;
    long rpw = (rows + target_waves - 1) / target_waves;
    rpw = ((rpw + align - 1) / align) * align;
    const long nw = (rows + rpw - 1) / rpw;
    *nw_out = nw;
    dim3 grid((unsigned)nw), block(64);
    const int *th = nullptr;  // per-tile structure hint of the register-tile kernel (64-row tiles)
    if (nc <= 80 && !g_force_v1) {
        th = tile_hint_or_zeros((rows + 63) / 64);
        if (!th) return FIGH_ERR_ALLOC;
    }
#define FIGH_TSQR_LAUNCH(CPL, MM, RL)                                                                          \
    hipLaunchKernelGGL((tsqr_kernel<CPL, MM, RL>), grid, block, 0, stream(), W, rows, ldw, col_idx, n, tau,     \
                       d_blkw, rows_per_blk, rpw, Rws_out, nc)
    const size_t lds3_extra = sizeof(double) * (64 * 17 + 16 * 16);  // -V^T staging + T of the blocked kernel
    if (nc <= 64 && !g_force_v1 && !g_force_v2) {
        if (g_pf)
            hipLaunchKernelGGL((tsqr3_kernel<4, true>), grid, block, sizeof(double) * nc * 64 + lds3_extra, stream(), W,
                               rows, ldw, col_idx, n, tau, d_blkw, rows_per_blk, Rws_out, nc, g_dbg);
        else
            hipLaunchKernelGGL((tsqr3_kernel<4, false>), grid, block, sizeof(double) * nc * 64 + lds3_extra, stream(), W,
                               rows, ldw, col_idx, n, tau, d_blkw, rows_per_blk, Rws_out, nc, g_dbg);
    } else if (nc <= 80 && !g_force_v1 && !g_force_v2) {
        hipLaunchKernelGGL((tsqr3_kernel<5, false>), grid, block, sizeof(double) * nc * 80 + lds3_extra, stream(), W, rows,
                           ldw, col_idx, n, tau, d_blkw, rows_per_blk, Rws_out, nc, g_dbg);
    } else if (nc <= 64 && !g_force_v1) {
        const size_t lds2 = tsqr2_lds_bytes(4, nc);
        if (tri)
            hipLaunchKernelGGL((tsqr2_kernel<4, 4, true>), grid, block, lds2, stream(), W, rows, ldw, col_idx, n, tau,
                               d_blkw, rows_per_blk, Rws_out, nc, g_dbg, out_rows, nullptr, th);
        else if (g_dbg & 4) {
            long long *prof = static_cast<long long *>(workspace(sizeof(long long) * 4 * nw, 6));
            if (!prof) return FIGH_ERR_ALLOC;
            hipLaunchKernelGGL((tsqr2_kernel<4, 4, false, true>), grid, block, lds2, stream(), W, rows, ldw, col_idx, n,
                               tau, d_blkw, rows_per_blk, Rws_out, nc, g_dbg, out_rows, prof, th);
            std::vector<long long> h(4 * nw);
            FIGH_HIP(hipMemcpyAsync(h.data(), prof, sizeof(long long) * 4 * nw, hipMemcpyDeviceToHost, stream()));
            FIGH_HIP(hipStreamSynchronize(stream()));
            double tot = 0, ld = 0, fa = 0, st = 0, mx = 0;
            for (long w = 0; w < nw; ++w) {
                tot += h[4 * w]; ld += h[4 * w + 1]; fa += h[4 * w + 2]; st += h[4 * w + 3];
                if (h[4 * w] > mx) mx = h[4 * w];
            }
            fprintf(stderr, "[tsqr2 prof] waves %ld rows %ld: ticks/wave avg %.0f max %.0f; load+delivery %.0f; "
                            "factorisation %.0f; steps/wave %.0f -> %.0f ticks per column step\n",
                    nw, rows, tot / nw, mx, ld / nw, fa / nw, st / nw, st > 0 ? fa / st : 0.0);
            if (nw >= 64) {  // by XCD (workgroups are dealt round-robin to the 8 XCDs) and by position in the grid
                double xs[8] = {0}, xn[8] = {0}, ss[8] = {0};
                for (long w = 0; w < nw; ++w) { xs[w & 7] += h[4 * w]; ss[w & 7] += h[4 * w + 3]; xn[w & 7] += 1; }
                fprintf(stderr, "[tsqr2 prof] ticks/wave by XCD:");
                for (int x = 0; x < 8; ++x) fprintf(stderr, " %.0f(%.0f steps)", xs[x] / xn[x], ss[x] / xn[x]);
                fprintf(stderr, "\n[tsqr2 prof] ticks/wave by grid octile:");
                for (int o = 0; o < 8; ++o) {
                    double a = 0; long c = 0;
                    for (long w = o * nw / 8; w < (o + 1) * nw / 8; ++w) { a += h[4 * w]; ++c; }
                    fprintf(stderr, " %.0f", a / c);
                }
                std::vector<long long> d(nw);
                for (long w = 0; w < nw; ++w) d[w] = h[4 * w];
                std::sort(d.begin(), d.end());
                fprintf(stderr, "\n[tsqr2 prof] ticks/wave quantiles: min %lld 10%% %lld 50%% %lld 90%% %lld 99%% %lld max %lld\n",
                        d[0], d[nw / 10], d[nw / 2], d[nw * 9 / 10], d[nw * 99 / 100], d[nw - 1]);
            }
        } else if (g_dbg & 64)  // A/B: permlane-swap reduction instead of the LDS one
            hipLaunchKernelGGL((tsqr2_kernel<4, 4, false, false, false>), grid, block, lds2, stream(), W, rows, ldw,
                               col_idx, n, tau, d_blkw, rows_per_blk, Rws_out, nc, g_dbg, out_rows, nullptr, th);
        else
            hipLaunchKernelGGL((tsqr2_kernel<4, 4, false, false, true>), grid, block, lds2, stream(), W, rows, ldw,
                               col_idx, n, tau, d_blkw, rows_per_blk, Rws_out, nc, g_dbg, out_rows, nullptr, th);
    } else if (nc <= 80 && !g_force_v1) {
        const size_t lds2 = tsqr2_lds_bytes(5, nc);
        hipLaunchKernelGGL((tsqr2_kernel<5, 4, false>), grid, block, lds2, stream(), W, rows, ldw, col_idx, n, tau,
                           d_blkw, rows_per_blk, Rws_out, nc, g_dbg, out_rows, nullptr, th);
    } else if (nc <= 64) FIGH_TSQR_LAUNCH(1, 64, true);
    else if (nc <= 512 && !g_force_v1) {
        // column-split workgroups: nw here counts workgroups (one private triangle each)
        if (nc <= 256)
            hipLaunchKernelGGL((tsqr_wide_kernel<2>), grid, dim3(512), 0, stream(), W, rows, ldw, col_idx, n, tau, d_blkw,
                               rows_per_blk, Rws_out, nc);
        else if (nc <= 384)
            hipLaunchKernelGGL((tsqr_wide_kernel<3>), grid, dim3(512), 0, stream(), W, rows, ldw, col_idx, n, tau, d_blkw,
                               rows_per_blk, Rws_out, nc);
        else  // the human model's 400 inertial columns (SIP quadratic program)
            hipLaunchKernelGGL((tsqr_wide_kernel<4>), grid, dim3(512), 0, stream(), W, rows, ldw, col_idx, n, tau, d_blkw,
                               rows_per_blk, Rws_out, nc);
    } else if (nc <= 128) FIGH_TSQR_LAUNCH(2, 32, false);
    else if (nc <= 256) FIGH_TSQR_LAUNCH(4, 16, false);
    else if (nc <= 384) FIGH_TSQR_LAUNCH(6, 16, false);
    else {
        set_error("figh_tsqr: more than 512 columns not supported yet");
        return FIGH_ERR_UNSUPPORTED;
    }
#undef FIGH_TSQR_LAUNCH
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}

// reduce `count` stacked nc x nc triangles (in Rs, contiguous) down to one, result in d_R_out
// padded: every input triangle occupies 64 rows (nc real + zero rows) so that one 64-row tile is exactly one upper
// triangular factor and the merge levels can skip the row chunks below the current panel (TRI kernel)
static int tsqr_reduce_tree(const double *Rs, long count, int nc, double *d_R_out, bool padded);

// reduce `count` stacked compact nc x nc triangles (contiguous in Rs) to one in d_R_out
static int tsqr_reduce(const double *Rs, long count, int nc, double *d_R_out, bool padded = false) {
    static const bool g_tree = getenv("FIGH_TSQR_TREE") != nullptr;  // A/B: single-wave 4:1 tree levels
    if (padded || nc > 80 || g_force_v1 || g_tree) return tsqr_reduce_tree(Rs, count, nc, d_R_out, padded);
    const size_t tri = sizeof(double) * (size_t)nc * nc;
    const double *cur = Rs;
    long cnt = count;
    int slot = 2;
    while (cnt > 1) {
        const long rows = cnt * nc;
        const int nwv = (rows > 256 && nc <= 64) ? 8 : 4;  // 5 column chunks per lane need > 256 registers: 4 waves
        const long nb = (rows + 64L * nwv - 1) / (64L * nwv);
        double *dst = nb == 1 ? d_R_out : static_cast<double *>(workspace(tri * nb, slot));
        if (!dst) return FIGH_ERR_ALLOC;
        ProfileScope scope("tsqr_reduce");
        if (nwv == 8)
            hipLaunchKernelGGL((tsqr_coop_kernel<4, 8>), dim3((unsigned)nb), dim3(512), 0, stream(), cur, rows, nc, dst);
        else if (nc > 64)
            hipLaunchKernelGGL((tsqr_coop_kernel<5, 4>), dim3((unsigned)nb), dim3(256), 0, stream(), cur, rows, nc, dst);
        else
            hipLaunchKernelGGL((tsqr_coop_kernel<4, 4>), dim3((unsigned)nb), dim3(256), 0, stream(), cur, rows, nc, dst);
        FIGH_HIP(hipGetLastError());
        cur = dst;
        cnt = nb;
        slot = slot == 2 ? 3 : 2;
    }
    if (cur != d_R_out) FIGH_HIP(hipMemcpyAsync(d_R_out, cur, tri, hipMemcpyDeviceToDevice, stream()));
    return FIGH_OK;
}

static int tsqr_reduce_tree(const double *Rs, long count, int nc, double *d_R_out, bool padded) {
    const size_t tri = sizeof(double) * (size_t)nc * nc;
    const bool use_tri = padded && nc <= 64 && !g_force_v1 && g_force_v2;
    const long in_rows = padded ? 64 : nc;
    const double *cur = Rs;
    long cnt = count;
    int slot = 2;
    while (cnt > 1) {
        const long fan = 4;
        const long nw_next = (cnt + fan - 1) / fan;
        const bool last = nw_next == 1;
        const int out_rows = (use_tri && !last) ? 64 : nc;
        double *dst = last ? d_R_out
                           : static_cast<double *>(workspace(sizeof(double) * (size_t)out_rows * nc * nw_next, slot));
        if (!dst) return FIGH_ERR_ALLOC;
        long nw = 0;
        ProfileScope scope("tsqr_reduce");
        const long in_r = (cur == Rs) ? in_rows : (use_tri ? 64 : nc);
        if (int rc = tsqr_level(cur, cnt * in_r, nc, nullptr, nc, nullptr, nullptr, 1, nc, nw_next, fan * in_r, dst, &nw,
                                out_rows, use_tri && in_r == 64))
            return rc;
        cur = dst;
        cnt = nw;
        slot = slot == 2 ? 3 : 2;
    }
    if (cur != d_R_out) FIGH_HIP(hipMemcpyAsync(d_R_out, cur, tri, hipMemcpyDeviceToDevice, stream()));
    return FIGH_OK;
}

}  // namespace figh

using namespace figh;

extern "C" {

int figh_colsq(const double *d_W, int64_t rows, int cols, int64_t ldw, double *d_out) {
    FIGH_REQUIRE(d_W && d_out, "NULL device pointer");
    FIGH_REQUIRE(rows >= 0 && cols > 0 && ldw >= cols, "bad shape");
    if (int rc = ensure_device()) return rc;
    if (rows == 0) {
        FIGH_HIP(hipMemsetAsync(d_out, 0, sizeof(double) * cols, stream()));
        return FIGH_OK;
    }
    long nblocks = cu_count() * 8L;
    long rpb = (rows + nblocks - 1) / nblocks;
    if (rpb < 16) rpb = 16;
    nblocks = (rows + rpb - 1) / rpb;
    double *part = static_cast<double *>(workspace(sizeof(double) * nblocks * cols, 1));
    if (!part) return FIGH_ERR_ALLOC;
    ProfileScope scope("colsq");
    hipLaunchKernelGGL(colsq_kernel, dim3((unsigned)nblocks), dim3(256), 0, stream(), d_W, (long)rows, cols, (long)ldw,
                       rpb, part);
    hipLaunchKernelGGL(reduce_cols_kernel, dim3(cols), dim3(256), 0, stream(), part, (int)nblocks, cols,
                       d_out);
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}

int figh_gather_cols(const double *d_W, int64_t rows, int64_t ldw, const int32_t *d_col_idx, int n, double *d_out,
                     int64_t ldo) {
    FIGH_REQUIRE(d_W && d_out && d_col_idx, "NULL device pointer");
    FIGH_REQUIRE(rows >= 0 && n >= 0 && ldo >= n, "bad shape");
    if (int rc = ensure_device()) return rc;
    if (rows == 0 || n == 0) return FIGH_OK;
    long blocks = (rows * n + 255) / 256;
    if (blocks > cu_count() * 16L) blocks = cu_count() * 16L;
    ProfileScope scope("gather_cols");
    hipLaunchKernelGGL(gather_cols_kernel, dim3((unsigned)blocks), dim3(256), 0, stream(), d_W, (long)rows, (long)ldw,
                       d_col_idx, n, d_out, (long)ldo);
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}

int figh_base_permutation(const double *d_R, int nc, int n, double tol_qr, int32_t *d_perm) {
    FIGH_REQUIRE(d_R && d_perm, "NULL device pointer");
    FIGH_REQUIRE(n >= 1 && nc >= n && nc <= 512, "bad shape");
    if (int rc = ensure_device()) return rc;
    ProfileScope scope("base_permutation");
    hipLaunchKernelGGL(base_permutation_kernel, dim3(1), dim3(64), 0, stream(), d_R, nc, n, tol_qr, d_perm);
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}

int figh_matvec(const double *d_W, int64_t rows, int64_t ldw, const int32_t *d_col_idx, int n, const double *d_x,
                double *d_y) {
    FIGH_REQUIRE(d_W && d_x && d_y, "NULL device pointer");
    FIGH_REQUIRE(rows >= 0 && n > 0, "bad shape");
    if (int rc = ensure_device()) return rc;
    if (rows == 0) return FIGH_OK;
    long blocks = (rows + 3) / 4;
    if (blocks > cu_count() * 8L) blocks = cu_count() * 8L;
    ProfileScope scope("matvec");
    hipLaunchKernelGGL(matvec_kernel, dim3((unsigned)blocks), dim3(256), 0, stream(), d_W, (long)rows, (long)ldw,
                       d_col_idx, n, d_x, d_y);
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}

int figh_block_sqnorm(const double *d_a, const double *d_b, int64_t rows, int nblocks, double *d_out) {
    FIGH_REQUIRE(d_a && d_out, "NULL device pointer");
    FIGH_REQUIRE(nblocks > 0 && rows >= 0 && rows % nblocks == 0, "rows must be a multiple of nblocks");
    if (int rc = ensure_device()) return rc;
    ProfileScope scope("block_sqnorm");
    hipLaunchKernelGGL(block_sqnorm_kernel, dim3((unsigned)nblocks), dim3(256), 0, stream(), d_a, d_b,
                       (long)(rows / nblocks), d_out);
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}

// Level 0 only (internal, figh_internal.h): the per-wave / per-workgroup triangles of W go to d_tri_out (compact
// nc x nc each, at most `capacity` of them, *count_out written) -- or, with d_tri_out == nullptr, to the library
// workspace whose address is returned in *ws_out.  The streamed entry points stack the triangles of all their sample
// chunks this way and run the merge tree once.
int figh_tsqr_level0(const double *d_W, int64_t rows, int64_t ldw, const int32_t *d_col_idx, int n, const double *d_tau,
                     const double *h_block_weight, int nblocks, double *d_tri_out, int64_t capacity, int64_t *count_out,
                     double **ws_out, int *padded_out) {
    FIGH_REQUIRE(d_W && count_out, "NULL device pointer");
    FIGH_REQUIRE(rows > 0 && n > 0 && ldw > 0, "bad shape");
    const int nc = n + (d_tau ? 1 : 0);
    FIGH_REQUIRE(nc <= 512, "figh_tsqr: more than 512 columns not supported yet");
    FIGH_REQUIRE(ldw < (1L << 24), "figh_tsqr: leading dimension must be below 2^24 elements");
    if (int rc = ensure_device()) return rc;
    const double *d_blkw = nullptr;
    long rows_per_blk = 1;
    if (h_block_weight) {
        FIGH_REQUIRE(nblocks > 0 && rows % nblocks == 0, "rows must be a multiple of nblocks");
        double *wbuf = static_cast<double *>(workspace(sizeof(double) * nblocks, 4));
        if (!wbuf) return FIGH_ERR_ALLOC;
        FIGH_HIP(hipMemcpyAsync(wbuf, h_block_weight, sizeof(double) * nblocks, hipMemcpyHostToDevice, stream()));
        FIGH_HIP(hipStreamSynchronize(stream()));
        d_blkw = wbuf;
        rows_per_blk = rows / nblocks;
    }
    // level 0: one wave per SIMD for the register-resident n <= 64 kernel, fewer for the wide ones
    long target = cu_count() * 2L;
    if (nc > 80 && !g_force_v1) target = cu_count();  // column-split workgroups of 8 waves: one per CU
    if (nc <= 80 && !g_force_v1) {  // register-tile kernel: as many waves per CU as its LDS triangle admits
        long per_cu = 4;  // tile + prefetched tile in registers: one wave per SIMD
        if (g_force_v2) {  // 256 registers: two waves per SIMD when the LDS triangles allow it
            per_cu = (long)((160 * 1024) / tsqr2_lds_bytes(nc <= 64 ? 4 : 5, nc));
            if (per_cu > (nc <= 64 ? 8 : 4)) per_cu = nc <= 64 ? 8 : 4;
            if (per_cu < 1) per_cu = 1;
        }
        static const int g_wpc = getenv("FIGH_TSQR_WPC") ? atoi(getenv("FIGH_TSQR_WPC")) : 0;  // A/B: waves per CU
        if (g_wpc > 0) per_cu = g_wpc;
        target = cu_count() * per_cu;
    } else if (nc <= 64) {
        target = cu_count() * 4L;
    }
    {   // a leaf must be much taller than wide, or the "reduction" produces more triangle rows than it consumed
        long cap = rows / (8L * nc);
        if (cap < 1) cap = 1;
        if (target > cap) target = cap;
    }
    const size_t tri = sizeof(double) * (size_t)nc * nc;
    // level-0 triangles in the zero-padded TRI merge format: measured SLOWER (reduce level 0.208 vs 0.167 ms: 28 %
    // more tiles outweigh the skipped row chunks), so it stays an A/B option (FIGH_TSQR_TRI)
    static const bool g_tri = getenv("FIGH_TSQR_TRI") != nullptr;
    const bool padded = g_tri && nc <= 64 && !g_force_v1 && g_force_v2 && d_tri_out == nullptr;
    const int out_rows = padded ? 64 : nc;
    long nw_est = target + 1;
    double *Rws = d_tri_out;
    if (Rws) {
        FIGH_REQUIRE(capacity >= nw_est, "figh_tsqr_level0: triangle buffer too small");
    } else {
        Rws = static_cast<double *>(workspace(sizeof(double) * (size_t)out_rows * nc * nw_est, 5));
        if (!Rws) return FIGH_ERR_ALLOC;
    }
    long nw = 0;
    {
        ProfileScope scope(rows >= 65536 ? "tsqr" : "tsqr_small");
        if (int rc = tsqr_level(d_W, rows, ldw, d_col_idx, n, d_tau, d_blkw, rows_per_blk, nc, target, 64, Rws, &nw,
                                out_rows))
            return rc;
    }
    (void)tri;
    *count_out = nw;
    if (ws_out) *ws_out = Rws;
    if (padded_out) *padded_out = padded ? 1 : 0;
    return FIGH_OK;
}

// upper bound of the triangles one figh_tsqr_level0 call can produce (for sizing the stack of a streamed run)
int64_t figh_tsqr_level0_capacity(int nc) {
    const long per_cu = (nc > 80 && !g_force_v1) ? 1 : 8;  // column-split workgroups: one triangle per CU
    return (int64_t)cu_count() * per_cu + 1;
}

int figh_tsqr(const double *d_W, int64_t rows, int64_t ldw, const int32_t *d_col_idx, int n, const double *d_tau,
              const double *h_block_weight, int nblocks, double *d_R_out) {
    FIGH_REQUIRE(d_W && d_R_out, "NULL device pointer");
    int64_t nw = 0;
    double *Rws = nullptr;
    int padded = 0;
    if (int rc = figh_tsqr_level0(d_W, rows, ldw, d_col_idx, n, d_tau, h_block_weight, nblocks, nullptr, 0, &nw, &Rws,
                                  &padded))
        return rc;
    const int nc = n + (d_tau ? 1 : 0);
    if (nw == 1) {
        FIGH_HIP(hipMemcpyAsync(d_R_out, Rws, sizeof(double) * (size_t)nc * nc, hipMemcpyDeviceToDevice, stream()));
        return FIGH_OK;
    }
    return tsqr_reduce(Rws, nw, nc, d_R_out, padded != 0);
}


// internal (figh_internal.h): install / remove the structure hint for the next level-0 launches on `rows` rows
int figh_tsqr_hint_begin(const int32_t *h_first_col, int nfirst, int64_t rows, int n, int nc) {
    FIGH_REQUIRE(h_first_col && nfirst > 0 && rows > 0 && rows % nfirst == 0, "rows must be a multiple of the hint blocks");
    for (int b = 0; b < nfirst; ++b) FIGH_REQUIRE(h_first_col[b] >= 0 && h_first_col[b] <= n, "first column out of range");
    if (int rc = ensure_device()) return rc;
    g_tile_hint = nullptr;
    if (nc > 80 || g_force_v1) return FIGH_OK;  // only the register-tile kernel uses the hint
    // the per-tile form of the hint is cached: the pipeline passes the same structure every step
    static std::vector<int32_t> cached_first;
    static int64_t cached_rows = -1;
    static const int *cached_ptr = nullptr;
    const long ntiles = (rows + 63) / 64;
    int *d_tile = static_cast<int *>(workspace(sizeof(int) * (size_t)(ntiles + 1), 14));
    if (!d_tile) return FIGH_ERR_ALLOC;
    if (cached_ptr != d_tile || cached_rows != rows || cached_first.size() != (size_t)nfirst ||
        !std::equal(cached_first.begin(), cached_first.end(), h_first_col)) {
        int *d_first = static_cast<int *>(workspace(sizeof(int) * nfirst, 7));
        if (!d_first) return FIGH_ERR_ALLOC;
        FIGH_HIP(hipMemcpyAsync(d_first, h_first_col, sizeof(int) * nfirst, hipMemcpyHostToDevice, stream()));
        hipLaunchKernelGGL(tile_hint_kernel, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, stream(), d_first,
                           (long)(rows / nfirst), (long)rows, ntiles, d_tile);
        FIGH_HIP(hipGetLastError());
        FIGH_HIP(hipStreamSynchronize(stream()));  // h_first_col is the caller's memory
        cached_first.assign(h_first_col, h_first_col + nfirst);
        cached_rows = rows;
        cached_ptr = d_tile;
    }
    g_tile_hint = d_tile;
    return FIGH_OK;
}
void figh_tsqr_hint_end(void) { g_tile_hint = nullptr; }

int figh_tsqr_structured(const double *d_W, int64_t rows, int64_t ldw, const int32_t *d_col_idx, int n,
                         const double *d_tau, const double *h_block_weight, int nblocks, const int32_t *h_first_col,
                         int nfirst, double *d_R_out) {
    if (int rc = figh_tsqr_hint_begin(h_first_col, nfirst, rows, n, n + (d_tau ? 1 : 0))) return rc;
    const int rc = figh_tsqr(d_W, rows, ldw, d_col_idx, n, d_tau, h_block_weight, nblocks, d_R_out);
    figh_tsqr_hint_end();
    return rc;
}

int figh_tsqr_merge(const double *d_Rs, int count, int nc, double *d_R_out) {
    FIGH_REQUIRE(d_Rs && d_R_out, "NULL device pointer");
    FIGH_REQUIRE(count >= 1 && nc >= 1 && nc <= 512, "bad shape");
    if (int rc = ensure_device()) return rc;
    if (count == 1) {
        FIGH_HIP(hipMemcpyAsync(d_R_out, d_Rs, sizeof(double) * (size_t)nc * nc, hipMemcpyDeviceToDevice, stream()));
        return FIGH_OK;
    }
    return tsqr_reduce(d_Rs, count, nc, d_R_out);
}

}  // extern "C"
