// K1 + K3 level 0 in ONE launch for serial chains (UR10: BASELINE configs[1]) -- W is written to HBM exactly as
// figh_regressor_build writes it, and every 64-row tile is factored while it is still in LDS: the TSQR no longer reads W
// back (the two-launch form moves 4.0 GB out and 3.4 GB back in for 10^6 UR10 samples).
//
// Replaces, in one pass over the samples, build_regressor_basic (src/figaroh/tools/regressor.py:20-194, the
// pin.computeJointTorqueRegressor loop :45-87), the column norms of get_index_eliminate (:258-279) and the np.linalg.qr
// of get_baseParams (src/figaroh/tools/qrdecomposition.py:205) on the kept columns [+ tau].
//
// One workgroup per CU, 2 + C waves with two roles (the roles never meet at a barrier inside the tile loop):
//
//   producers (waves 0, 1) regressor_chain_kernel's tile body, one sample per lane, every other sample tile of the workgroup
//                         each: forward recursion once per 64 samples, then for every joint row j the lane's 14 NJ row entries
//                         (+ the row's tau entry) go to the producer's OWN tile buffer in LDS [64][NC + 2]; the tile is
//                         announced (LDS counter), streamed to W as contiguous 1 KB runs of 16-byte stores, and its
//                         diag(W^T W) contribution is accumulated (LDS accumulators, fixed order).  256 registers (two waves
//                         per SIMD): the per-link rotations are re-formed from (cos, sin) where the two-launch kernel keeps
//                         them (431 registers there), q', q'' wait in LDS.
//   consumers (waves 2..)  tsqr2_kernel's wave: a private triangle, the 64 x 64 register tile in the MFMA C/D layout.  A free
//                         consumer claims the next announced tile of either buffer (LDS compare-and-swap: tiles go to
//                         whoever is free -- row blocks of joint 1 cost 10x those of joint 6), gathers the kept columns
//                         [+ tau] of the LDS tile into its registers, hands the buffer back and runs the column steps
//                         (figh_tsqr_narrow.h -- the code of the two-launch kernel, RLAST form: the last 16 columns of the
//                         triangle in registers, 7.4 instead of 13.8 KB of LDS per wave).
//
// Measured on the first form (one producer, seven consumers with 13.8 KB triangles, tools/fused_prof.py): the producer needs
// 10 000 cycles per tile (row emission 3 900, stream-out 2 000, column norms 1 600, forward recursion 1 100 -- a single wave
// is latency-bound) where seven consumers finish a tile every 5 000, and everything that touches THE buffer is serial.
// Hence two producers with a buffer each: 2 x 44 KB tiles + 2 x 9 KB staging + C x 7.4 KB triangles = 152 KB (C = 6).
// All arithmetic is fp64.  The kept-column list is the caller's (the previous pass's, verified afterwards against the
// norms this launch produces -- the same speculation figh_tsqr_selected makes with the column count).
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "figh_internal.h"
#include "figh_spatial.h"
#include "figh_chain.h"
#include "figh_wave.h"
#include "figh_tsqr_narrow.h"

namespace figh {

typedef double f64x2 __attribute__((ext_vector_type(2)));

#ifdef FIGH_ABLATION
// in-kernel time buckets (s_memtime ticks summed over all workgroups): producer [0] waiting for the buffer, [1] inputs +
// forward recursion, [2] row emission, [3] stream-out, [4] column norms, [5] total; consumers [8] waiting for a tile,
// [9] gather, [10] column steps, [11] total, [12] tiles
__device__ unsigned long long g_fused_prof[16];
__device__ int g_fused_opts = 0;  // 1: no column norms, 2: no stream-out, 4: no row emission arithmetic
#define FUSED_TICK(var) const long long var = __builtin_readcyclecounter()
#define FUSED_ADD(slot, dt) prof_##slot += (dt)
#else
#define FUSED_TICK(var)
#define FUSED_ADD(slot, dt)
#endif

struct FusedCtrl {  // one per producer / tile buffer
    int produced;  // tiles announced so far (written by the producer only)
    int taken;     // tiles whose gather is finished (written by the consumer that claimed the tile)
    int claimed;   // tiles claimed so far (compare-and-swap by the consumers: only announced tiles are claimed)
    int pad[5];
};

__device__ __forceinline__ int lds_peek(const int *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_post(int *p, int v) {
    // LDS executes one wave's instructions in order: everything this wave read or wrote in LDS before this store is done
    // when the store is.  Only the compiler has to be kept from moving accesses across it.
    asm volatile("" ::: "memory");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}

// a wave-uniform double the optimiser may not look through: uniform products of the tree constants (a_i a_j of the
// Rodrigues formula) would otherwise be hoisted out of the sample loop into VECTOR registers, 2 per value
__device__ __forceinline__ double opaque_s(double x) {
    asm volatile("" : "+s"(x));
    return x;
}

// the same for a value that is to live in a vector register from here on: whatever is computed from it stays inside the loop
// iteration that computes it (hoisted, such values are spilled to scratch, and a scratch reload waits -- in-order vmcnt --
// for every global store the wave has in flight)
__device__ __forceinline__ double opaque_v(double x) {
    asm volatile("" : "+v"(x));
    return x;
}

template <int NJ>
struct FusedGeom {
    static constexpr int NC = 14 * NJ;
    static constexpr int LDT = NC + 2;  // LDS row stride (doubles); column NC carries the row's tau entry
    static constexpr int CH = NC / 2;   // 16-byte store chunks per row
    static constexpr int NPROD = 2;
    // LDS map (doubles), per producer: tile | tau of the sample tile's NJ rows | q', q'' | column-norm accumulators | control
    static constexpr int TILE = 64 * LDT;
    static constexpr int TAU0 = TILE;
    static constexpr int QD0 = TAU0 + 64 * NJ;
    static constexpr int CS0 = QD0 + 2 * 64 * NJ;
    static constexpr int CTRL0 = CS0 + ((NC + 7) / 8) * 8;
    static constexpr int PSIZE = CTRL0 + (int)(sizeof(FusedCtrl) / sizeof(double));
    static constexpr int CONS0 = NPROD * PSIZE;
};

template <int NJ>
__global__ __launch_bounds__(512) void fused_chain_tsqr_kernel(
    const ChainParams<NJ> P, const int flags, const long N, const double *__restrict__ q, const double *__restrict__ v,
    const double *__restrict__ a, double *__restrict__ W, const double *__restrict__ tau, const int *__restrict__ col_idx,
    const int n, const int nc, double *__restrict__ colsq_part, double *__restrict__ Rws, const int ncons,
    const int tri_doubles, const double null2) {
    using G = FusedGeom<NJ>;
    constexpr int NC = G::NC, LDT = G::LDT, CH = G::CH;
    constexpr int NCC = 4, NRC = 4, RPL = 16;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long ntiles_s = (N + 63) / 64;  // sample tiles
    const long b = blockIdx.x, Gd = gridDim.x;
    // this workgroup's sample tiles: b, b + Gd, ...; producer p takes every other one, starting with its p-th
    const long count_s = b < ntiles_s ? (ntiles_s - b + Gd - 1) / Gd : 0;
    const int T0 = (int)(((count_s + 1) / 2) * NJ), T1 = (int)((count_s / 2) * NJ);  // row tiles of producer 0 / 1

    // zero: accumulators, control words, the consumers' triangles and scratch
    for (int pz = 0; pz < G::NPROD; ++pz)
        for (int e = pz * G::PSIZE + G::CS0 + threadIdx.x; e < (pz + 1) * G::PSIZE; e += blockDim.x) lds[e] = 0.0;
    for (int e = G::CONS0 + threadIdx.x; e < G::CONS0 + ncons * tri_doubles; e += blockDim.x) lds[e] = 0.0;
    __syncthreads();

#ifndef FUSED_NO_PROD
    if (wave < G::NPROD) {
        // ================================================================================================== producers
        __builtin_amdgcn_s_setprio(3);
        const bool fric = flags & FIGH_FLAG_FRICTION, actin = flags & FIGH_FLAG_ACT_INERTIA, offs = flags & FIGH_FLAG_OFFSET;
        double *tile = lds + wave * G::PSIZE;
        FusedCtrl *ctrl = reinterpret_cast<FusedCtrl *>(tile + G::CTRL0);
        double *my = tile + lane * LDT;
        double *stau = tile + G::TAU0, *sqd = tile + G::QD0, *cacc = tile + G::CS0;
        int t = 0;
#ifdef FIGH_ABLATION
        long long prof_0 = 0, prof_1 = 0, prof_2 = 0, prof_3 = 0, prof_4 = 0;
        const int opts = g_fused_opts;
        const long long prof_start = __builtin_readcyclecounter();
#endif
        auto sample_tile = [&](auto FAST_T, const long s) {
            FUSED_TICK(c0);
            constexpr bool FAST = decltype(FAST_T)::value;
            const long i0 = s * 64;
            const int nvalid = FAST ? 64 : (int)((N - i0) < 64 ? (N - i0) : 64);
            const bool live = FAST || lane < nvalid;
            const long is = i0 + (live ? lane : nvalid - 1);
            double cs[NJ], sn[NJ];
            double acc[NJ][3], dw[NJ][3], w[NJ][3];
            {
                double pq[NJ], pqd[NJ], pqdd[NJ];
#pragma unroll
                for (int k = 0; k < NJ; ++k) {
                    pq[k] = q[is * NJ + k];
                    pqd[k] = v[is * NJ + k];
                    pqdd[k] = a[is * NJ + k];
                }
                if (tau) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const double tv = tau[(long)j * N + is];
                        stau[64 * j + lane] = live ? tv : 0.0;
                    }
                }
#pragma unroll
                for (int k = 0; k < NJ; ++k) {
                    sqd[64 * (2 * k) + lane] = pqd[k];
                    sqd[64 * (2 * k + 1) + lane] = pqdd[k];
                }
                // forward recursion (regressor_chain_kernel, statement for statement)
                double vl[3] = {0, 0, 0}, om[3] = {0, 0, 0}, da[3] = {0, 0, 0};
                double al[3] = {opaque_v(-P.g[0]), opaque_v(-P.g[1]), opaque_v(-P.g[2])};
#pragma unroll
                for (int k = 0; k < NJ; ++k) {
                    double s_, c_;
                    sincos(pq[k], &s_, &c_);
                    cs[k] = c_;
                    sn[k] = s_;
                    double Rj[9], R[9];
                    const double ax[3] = {opaque_s(P.axis[k][0]), opaque_s(P.axis[k][1]), opaque_s(P.axis[k][2])};
                    rodrigues(ax, c_, s_, Rj);
                    matmul3(P.Rp[k], Rj, R);
                    double t1[3], t2[3], nvl[3], nom[3], nal[3], nda[3];
                    cross3(P.pp[k], om, t1);
#pragma unroll
                    for (int d = 0; d < 3; ++d) t2[d] = vl[d] - t1[d];
                    rotT(R, t2, nvl);
                    rotT(R, om, nom);
                    cross3(P.pp[k], da, t1);
#pragma unroll
                    for (int d = 0; d < 3; ++d) t2[d] = al[d] - t1[d];
                    rotT(R, t2, nal);
                    rotT(R, da, nda);
                    double vj[3] = {P.axis[k][0] * pqd[k], P.axis[k][1] * pqd[k], P.axis[k][2] * pqd[k]};
#pragma unroll
                    for (int d = 0; d < 3; ++d) nom[d] += vj[d];
                    cross3(nvl, vj, t1);
                    cross3(nom, vj, t2);
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        nal[d] += t1[d];
                        nda[d] += P.axis[k][d] * pqdd[k] + t2[d];
                        vl[d] = nvl[d];
                        om[d] = nom[d];
                        al[d] = nal[d];
                        da[d] = nda[d];
                    }
                    cross3(om, vl, t1);
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        acc[k][d] = al[d] + t1[d];
                        dw[k][d] = da[d];
                        w[k][d] = om[d];
                    }
                }
            }
            FUSED_TICK(c1);
            FUSED_ADD(1, c1 - c0);
#pragma unroll
            for (int jo = 0; jo < NJ; ++jo) {
                // Row order 0, NJ-1, 1, NJ-2, ...: a row block of joint 1 costs a consumer ten times what one of joint 6 costs
                // (104 against 11 chunk-steps for UR10), and with only one tile of slack per producer the consumers would
                // all be busy during the heavy half of a sample tile and all be waiting during the light half.
                const int j = (jo & 1) ? NJ - 1 - (jo >> 1) : (jo >> 1);
                const int jprev = jo == 0 ? (((NJ - 1) & 1) ? NJ - 1 - ((NJ - 1) >> 1) : ((NJ - 1) >> 1))
                                          : (((jo - 1) & 1) ? NJ - 1 - ((jo - 1) >> 1) : ((jo - 1) >> 1));
                FUSED_TICK(c2);
                // the buffer is free once the previous tile has been gathered by its consumer (this wave's own reads of it
                // -- stream-out, column norms -- are behind it in program order)
                while (lds_peek(&ctrl->taken) < t) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
                FUSED_TICK(c3);
                FUSED_ADD(0, c3 - c2);
                // ---- this lane's row (j, i): 14 columns per link; links < j are structurally zero.  The buffer holds the row
                // of joint jprev (zeros below 14 jprev, values from there on; the kernel's very first tile is row 0, which
                // writes every column): only [14 jprev, 14 j) has to be cleared
                if (j > jprev) {
#pragma unroll
                    for (int c = 14 * jprev; c < 14 * j; c += 2) *reinterpret_cast<f64x2 *>(my + c) = f64x2{0.0, 0.0};
                }
                double Jl[3] = {0, 0, 0}, Ja[3] = {opaque_v(P.axis[j][0]), opaque_v(P.axis[j][1]), opaque_v(P.axis[j][2])};
#pragma unroll
                for (int k = j; k < NJ; ++k) {
                    if (k > j) {
                        // (opaque to the optimiser: the rotation is re-formed for every row instead of being kept for all
                        // rows -- 54 doubles this wave has no registers for)
                        asm volatile("" : "+v"(cs[k]), "+v"(sn[k]));
                        double Rj[9], R[9];
                        const double ax[3] = {opaque_s(P.axis[k][0]), opaque_s(P.axis[k][1]), opaque_s(P.axis[k][2])};
                        rodrigues(ax, cs[k], sn[k], Rj);
                        matmul3(P.Rp[k], Rj, R);
                        double t1[3], t2[3], nJl[3], nJa[3];
                        cross3(P.pp[k], Ja, t1);
#pragma unroll
                        for (int d = 0; d < 3; ++d) t2[d] = Jl[d] - t1[d];
                        rotT(R, t2, nJl);
                        rotT(R, Ja, nJa);
#pragma unroll
                        for (int d = 0; d < 3; ++d) {
                            Jl[d] = nJl[d];
                            Ja[d] = nJa[d];
                        }
                    }
                    double o[14];
                    o[9] = Jl[0] * acc[k][0] + Jl[1] * acc[k][1] + Jl[2] * acc[k][2];
                    double h1[3], h2[3], h3[3], h4[3];
                    cross3(Jl, dw[k], h1);
                    cross3(w[k], Jl, h2);
                    cross3(w[k], h2, h3);
                    cross3(acc[k], Ja, h4);
                    o[6] = h1[0] + h3[0] + h4[0];
                    o[7] = h1[1] + h3[1] + h4[1];
                    o[8] = h1[2] + h3[2] + h4[2];
                    double u[3];
                    cross3(w[k], Ja, u);
                    const double *x = dw[k], *z = w[k];
                    o[0] = x[0] * Ja[0] - z[0] * u[0];
                    o[1] = x[1] * Ja[0] + x[0] * Ja[1] - (z[1] * u[0] + z[0] * u[1]);
                    o[3] = x[1] * Ja[1] - z[1] * u[1];
                    o[2] = x[2] * Ja[0] + x[0] * Ja[2] - (z[2] * u[0] + z[0] * u[2]);
                    o[4] = x[2] * Ja[1] + x[1] * Ja[2] - (z[2] * u[1] + z[1] * u[2]);
                    o[5] = x[2] * Ja[2] - z[2] * u[2];
                    if (k == j) {  // Ia fv fs off: only on the link's own row (regressor.py:55-70,84-87)
                        const double qdk = sqd[64 * (2 * k) + lane], qddk = sqd[64 * (2 * k + 1) + lane];
                        o[10] = actin ? qddk : 0.0;
                        o[11] = fric ? qdk : 0.0;
                        o[12] = fric ? sgn(qdk) : 0.0;
                        o[13] = offs ? 1.0 : 0.0;
                    } else {
                        o[10] = o[11] = o[12] = o[13] = 0.0;
                    }
                    double *dst = my + 14 * k;
#pragma unroll
                    for (int c = 0; c < 14; ++c) dst[c] = live ? o[c] : 0.0;
                }
                my[NC] = tau ? stau[64 * j + lane] : 0.0;
                lds_post(&ctrl->produced, t + 1);
                ++t;
                FUSED_TICK(c4);
                FUSED_ADD(2, c4 - c3);

                // ---- stream the 64 x NC tile to W rows j*N+i0 .. : one contiguous run (the launcher guarantees ldw == NC and
                // 16-byte alignment).  Chunk id (16 B) sits at byte 16 id in W and at 16 (id + r) in the padded tile,
                // r = id / CH by a magic multiply
                double *dstW = W + ((long)j * N + i0) * NC;
                const char *tbase = reinterpret_cast<const char *>(tile);
#ifdef FIGH_ABLATION
                if (opts & 2) {
                } else
#endif
                if constexpr (FAST) {
                    // R rows per two passes of the wave, R = 128 / CH (UR10: three rows = 126 chunks of 16 B; seven links: two rows
                    // = 98): pass A takes slots 0..63 of the R rows, pass B slots 64..127 -- the slots behind R CH are the first
                    // chunks of the NEXT row (written again, with the same bytes, by the next group's pass A; masked in the last
                    // group when no row is left over).  A slot's place in the padded tile is then one per-lane constant per pass
                    // + a compile-time offset per row group: two address registers for the whole stream-out and no branches,
                    // where chunk id -> (id + id / CH) needs one register per chunk (42 the producer does not have) or 5 VALU
                    // operations per chunk.  Every pass is one contiguous 1 KB run in W.  64 % R rows are left over: lanes < CH.
                    constexpr int RG = 128 / CH;           // rows per group
                    constexpr int NG = 64 / RG;            // groups
                    constexpr int LEFT = 64 - NG * RG;     // rows behind the last group
                    static_assert(CH <= 64 && RG * CH > 64 && 128 - RG * CH <= CH && LEFT <= 1, "R rows per two passes");
                    constexpr int ROWB = LDT * 8;
                    const unsigned sa = lane, sb = lane + 64;
                    const unsigned ra = sa / CH, rb = sb / CH;
                    const unsigned la = ra * ROWB + 16u * (sa - ra * CH);
                    const unsigned lb = rb * ROWB + 16u * (sb - rb * CH);
                    char *gbase = reinterpret_cast<char *>(dstW) + 16 * lane;
                    constexpr int GRP = RG * CH * 16;  // bytes of a row group in W
                    constexpr int BATCH = NG % 3 == 0 ? 3 : 4;  // 2 BATCH reads in flight, then 2 BATCH stores
                    static_assert(NG % BATCH == 0, "row groups per batch");
#pragma unroll
                    for (int g0 = 0; g0 < NG; g0 += BATCH) {
                        f64x2 va[BATCH], vb[BATCH];
#pragma unroll
                        for (int u = 0; u < BATCH; ++u) {
                            va[u] = *reinterpret_cast<const f64x2 *>(tbase + la + (g0 + u) * RG * ROWB);
                            vb[u] = *reinterpret_cast<const f64x2 *>(tbase + lb + (g0 + u) * RG * ROWB);
                        }
#pragma unroll
                        for (int u = 0; u < BATCH; ++u) {
                            *reinterpret_cast<f64x2 *>(gbase + (g0 + u) * GRP) = va[u];
                            // (the last group has no next row to run into unless one is left over)
                            if (LEFT > 0 || g0 + u + 1 < NG || sb < (unsigned)(RG * CH))
                                *reinterpret_cast<f64x2 *>(gbase + (g0 + u) * GRP + 1024) = vb[u];
                        }
                    }
                    if constexpr (LEFT > 0) {
                        if (lane < CH) {
                            const f64x2 vc = *reinterpret_cast<const f64x2 *>(tbase + 63 * ROWB + 16 * lane);
                            *reinterpret_cast<f64x2 *>(gbase + NG * GRP) = vc;
                        }
                    }
                } else {
                    constexpr unsigned MAGIC = ((1u << 20) + CH - 1) / CH;
                    static_assert(64 * CH <= 4096, "magic division range");
                    const int total = nvalid * CH;
                    for (int id = lane; id < total; id += 64) {
                        const unsigned r = ((unsigned)id * MAGIC) >> 20;
                        const double2 val = *reinterpret_cast<const double2 *>(tbase + 16u * (id + r));
                        *reinterpret_cast<double2 *>(reinterpret_cast<char *>(dstW) + 16 * id) = val;
                    }
                }
                FUSED_TICK(c5);
                FUSED_ADD(3, c5 - c4);
                // ---- diag(W^T W): on joint row j the live columns are 14 j ..; lane owns 14 j + lane (and + 64)
#ifdef FIGH_ABLATION
                if (!(opts & 1))
#endif
                {
                    const int lo = 14 * j, width = NC - lo;
                    // RB rows requested at a time (the LDS latency is paid 64 / RB times per pass, not 64 times); sixteen where
                    // the registers allow it -- with six links the producer is at the 256-register limit exactly here and
                    // sixteen cost two spilled registers whose reloads wait behind the stores in flight: eight
                    constexpr int RB = NJ >= 6 ? 8 : 16;
                    auto column_sum = [&](const double *col) {
                        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll
                        for (int r0 = 0; r0 < 64; r0 += RB) {
                            double x[RB];
#pragma unroll
                            for (int r = 0; r < RB; ++r) x[r] = col[(r0 + r) * LDT];
#pragma unroll
                            for (int r = 0; r < RB; ++r) asm volatile("" : "+v"(x[r]));
#pragma unroll
                            for (int r = 0; r < RB; r += 4) {
                                t0 += x[r] * x[r];
                                t1 += x[r + 1] * x[r + 1];
                                t2 += x[r + 2] * x[r + 2];
                                t3 += x[r + 3] * x[r + 3];
                            }
                        }
                        return (t0 + t1) + (t2 + t3);
                    };
                    if (lane < width) cacc[lo + lane] += column_sum(tile + lo + lane);
                    if (width > 64 && lane + 64 < width) cacc[lo + 64 + lane] += column_sum(tile + lo + 64 + lane);
                }
                FUSED_TICK(c6);
                FUSED_ADD(4, c6 - c5);
            }
        };
        for (long s = b + wave * Gd; s < ntiles_s; s += G::NPROD * Gd) {
            if ((s + 1) * 64 <= N) sample_tile(std::true_type{}, s);
            else sample_tile(std::false_type{}, s);
        }
        asm volatile("" ::: "memory");
        for (int c = lane; c < NC; c += 64) colsq_part[(b * G::NPROD + wave) * NC + c] = cacc[c];
#ifdef FIGH_ABLATION
        if (lane == 0) {
            atomicAdd(&g_fused_prof[0], (unsigned long long)prof_0);
            atomicAdd(&g_fused_prof[1], (unsigned long long)prof_1);
            atomicAdd(&g_fused_prof[2], (unsigned long long)prof_2);
            atomicAdd(&g_fused_prof[3], (unsigned long long)prof_3);
            atomicAdd(&g_fused_prof[4], (unsigned long long)prof_4);
            atomicAdd(&g_fused_prof[5], (unsigned long long)(__builtin_readcyclecounter() - prof_start));
        }
#endif
        return;
    }
#endif
#ifndef FUSED_NO_CONS
    // ======================================================================================================= consumers
    constexpr int LCH = NCC - 1;  // chunks of the triangle kept in LDS (RLAST form)
    const int c_id = wave - G::NPROD;
    const int pad = 16 * NCC - nc;
    int skip = 0;
    for (int kp = 0; kp < pad; ++kp) skip += 16 * (LCH - (kp >> 4) > 0 ? LCH - (kp >> 4) : 0);
    double *mine = lds + G::CONS0 + c_id * tri_doubles;
    Tsqr2State<NCC, NRC, true> S;
    S.red = mine;
    S.bc = mine + 64;
    S.Rl = mine + 80 - skip;
    S.lane_c = lane & 15;
    S.lane_g = lane >> 4;
    S.nc = nc;
    S.null2 = null2;
#pragma unroll
    for (int sl = 0; sl < 4 * NCC; ++sl) S.Rq[sl] = 0.0;
    // per-lane column sources inside an LDS tile: kept column col_idx[col] for col < n, the tau slot (column NC) for
    // col == n == nc - 1; padding lane-columns stay exactly zero for the whole kernel
    bool wlive[NCC];
    int loff[NCC];
#pragma unroll
    for (int cc = 0; cc < NCC; ++cc) {
        const int col = 16 * cc + S.lane_c - pad;
        wlive[cc] = col >= 0 && col < nc;
        const int src = (col >= 0 && col < n) ? col_idx[col] : NC;
        loff[cc] = S.lane_g * LDT + src;
    }
#pragma unroll
    for (int cc = 0; cc < NCC; ++cc)
#pragma unroll
        for (int i = 0; i < RPL; ++i) S.T[cc][i] = 0.0;
    // structure of the joint-torque regressor of a chain (regressor.py:45-87): the row block of joint j has exact zeros in the
    // columns of the links in front of j -- the producer writes them as such -- so a tile's first kept column that can be
    // non-zero is known from its row block: fpos[j] = pad + #{kept columns < 14 j}.  Chunks in front of it are not gathered.
    int fpos[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) fpos[j] = pad;
    for (int e = 0; e < n; ++e) {
        const int c = col_idx[e];
#pragma unroll
        for (int j = 1; j < NJ; ++j) fpos[j] += (c < 14 * j) ? 1 : 0;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) fpos[j] = __builtin_amdgcn_readfirstlane(fpos[j]);  // (uniform: scalar registers)
    FusedCtrl *ctrl0 = reinterpret_cast<FusedCtrl *>(lds + G::CTRL0);
    FusedCtrl *ctrl1 = reinterpret_cast<FusedCtrl *>(lds + G::PSIZE + G::CTRL0);

#ifdef FIGH_ABLATION
    long long prof_8 = 0, prof_9 = 0, prof_10 = 0, prof_12 = 0;
    const long long prof_start = __builtin_readcyclecounter();
#endif
    int pref = c_id & 1;  // the buffer this wave looks at first
    int absorbed = 0;     // tiles this wave has factored
    for (;;) {
        FUSED_TICK(c0);
        // claim the next announced tile of either buffer (lane 0 decides, the wave follows): -1 = everything is claimed
        int pick = -2, tpick = 0;
        while (pick == -2) {
            int res = -2, tres = 0;
            if (lane == 0) {
                bool done = true;
#pragma unroll
                for (int tr = 0; tr < 2; ++tr) {
                    const int pb = pref ^ tr;
                    FusedCtrl *cp = pb ? ctrl1 : ctrl0;
                    const int Tp = pb ? T1 : T0;
                    int cl = lds_peek(&cp->claimed);
                    if (cl < Tp) {
                        done = false;
                        if (res == -2 && lds_peek(&cp->produced) > cl &&
                            __hip_atomic_compare_exchange_strong(&cp->claimed, &cl, cl + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_WORKGROUP)) {
                            res = pb;
                            tres = cl;
                        }
                    }
                }
                if (done) res = -1;
            }
            pick = __builtin_amdgcn_readfirstlane(res);
            tpick = __builtin_amdgcn_readfirstlane(tres);
            if (pick == -2) __builtin_amdgcn_s_sleep(1);
        }
        if (pick < 0) break;
        pref = pick ^ 1;  // alternate between the buffers
        asm volatile("" ::: "memory");
        FUSED_TICK(c1);
        FUSED_ADD(8, c1 - c0);
        const double *tile = lds + pick * G::PSIZE;
        FusedCtrl *ctrl = pick ? ctrl1 : ctrl0;
        // row block of the tile: the producer's row order 0, NJ-1, 1, NJ-2, ...
        const int jo = tpick % NJ;
        const int jrow = (jo & 1) ? NJ - 1 - (jo >> 1) : (jo >> 1);
        int first_nz = fpos[0];
#pragma unroll
        for (int j = 1; j < NJ; ++j) first_nz = jrow == j ? fpos[j] : first_nz;
        first_nz = __builtin_amdgcn_readfirstlane(first_nz);
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc) {
            // (cleared first: what the previous tile left in the registers is dead, no merge with it)
#pragma unroll
            for (int i = 0; i < RPL; ++i) S.T[cc][i] = 0.0;
            // a chunk whose lane-columns all lie in front of first_nz holds structural zeros only (wave-uniform test)
            if (wlive[cc] && 16 * cc + 15 >= first_nz) {
#pragma unroll
                for (int i = 0; i < RPL; ++i) S.T[cc][i] = tile[(16 * (i >> 2) + 4 * (i & 3)) * LDT + loff[cc]];
            }
        }
        // the values are in registers before the buffer is handed back
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc)
#pragma unroll
            for (int i = 0; i < RPL; ++i) asm volatile("" : "+v"(S.T[cc][i]));
        lds_post(&ctrl->taken, tpick + 1);
        FUSED_TICK(c2);
        FUSED_ADD(9, c2 - c1);
        S.null2 = absorbed * 64 < nc + 8 ? 0.0 : null2;  // (null pivots once the triangle is of full height: see tsqr2_level0_body)
        ++absorbed;
        tsqr2_panels<0, NCC, NRC, false, true>(S, first_nz, [](auto) {});
        FUSED_TICK(c3);
        FUSED_ADD(10, c3 - c2);
        FUSED_ADD(12, 1);
    }
#ifdef FIGH_ABLATION
    if (lane == 0) {
        atomicAdd(&g_fused_prof[8], (unsigned long long)prof_8);
        atomicAdd(&g_fused_prof[9], (unsigned long long)prof_9);
        atomicAdd(&g_fused_prof[10], (unsigned long long)prof_10);
        atomicAdd(&g_fused_prof[11], (unsigned long long)(__builtin_readcyclecounter() - prof_start));
        atomicAdd(&g_fused_prof[12], (unsigned long long)prof_12);
    }
#endif
    // this wave's triangle (compact nc x nc, row-major): the LDS chunks, then the register chunk
    double *Rg = Rws + ((long)b * ncons + c_id) * (long)nc * nc;
    const int nlds = 16 * LCH - pad;  // columns of the compact triangle that live in LDS
    for (int e = lane; e < nc * nc; e += 64) {
        const int k = e / nc, col = e - k * nc;
        if (col >= nlds && col >= k) continue;  // (written from the registers below)
        const int kp = k + pad, colp = col + pad;
        const int pk = kp >> 4;
        Rg[e] = (col < k) ? 0.0 : S.Rl[tsqr2_panel_off<LCH>(pk) + (kp & 15) * 16 * (LCH - pk) + (colp - 16 * pk)];
    }
    {
        const int col = 16 * LCH + S.lane_c - pad;
#pragma unroll
        for (int sl = 0; sl < 4 * NCC; ++sl) {
            const int k = 4 * sl + S.lane_g - pad;
            if (k >= 0 && col >= k && col >= 0) Rg[(long)k * nc + col] = S.Rq[sl];
        }
    }
#endif
}

// partial[b][c] -> out[c]: one workgroup per column, strided partial sums + LDS tree (fixed order: deterministic)
__global__ __launch_bounds__(256) void fused_reduce_partials_kernel(const double *__restrict__ part, int nblocks, int ncols,
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

static size_t fused_tri_doubles(int nc) {  // LDS per consumer: 64 + 16 doubles of scratch + the three LDS chunks of the triangle
    const int pad = 64 - nc;
    size_t skip = 0;
    for (int kp = 0; kp < pad; ++kp) skip += 16 * (3 - (kp >> 4) > 0 ? 3 - (kp >> 4) : 0);
    return 80 + 256 * (size_t)(3 + 2 + 1) - skip;
}

template <int NJ>
static int launch_fused(const figh_model_s *m, int flags, long N, const double *q, const double *v, const double *a,
                        double *W, const double *tau, const int *d_kept, int n, double *d_colsq, double **Rws_out,
                        long *count_out) {
    using G = FusedGeom<NJ>;
    const int nc = n + (tau ? 1 : 0);
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const size_t tri = fused_tri_doubles(nc);
    const size_t fixed = (size_t)G::CONS0;
    const size_t budget = (160 * 1024) / sizeof(double);
    int ncons = (int)((budget - fixed) / tri);
    if (ncons > 6) ncons = 6;
    if (ncons < 3) {
        set_error("fused regressor + TSQR: the triangles of three consumer waves do not fit next to the tile");
        return FIGH_ERR_UNSUPPORTED;
    }
    const long ntiles_s = (N + 63) / 64;
    long grid = cus < ntiles_s / 2 ? cus : ntiles_s / 2;  // (at least two sample tiles = 12 row tiles per workgroup)
    const size_t lds = sizeof(double) * (fixed + (size_t)ncons * tri);
    static bool attr_set[16] = {};
    if (!attr_set[NJ]) {
        FIGH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_chain_tsqr_kernel<NJ>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[NJ] = true;
    }
    double *part = static_cast<double *>(workspace(sizeof(double) * grid * G::NPROD * G::NC, 0));
    double *Rws = static_cast<double *>(workspace(sizeof(double) * (size_t)nc * nc * (grid * ncons + 1), 5));
    if (!part || !Rws) return FIGH_ERR_ALLOC;
    const ChainParams<NJ> P = chain_params<NJ>(m);
#ifdef FIGH_ABLATION
    {
        const char *e = getenv("FIGH_FUSED_OPTS");
        const int o = e ? atoi(e) : 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fused_opts), &o, sizeof(int));
    }
#endif
    {
        ProfileScope scope("fused_chain_tsqr", true);
        FIGH_LAUNCH_TIMED((fused_chain_tsqr_kernel<NJ>), dim3((unsigned)grid), dim3(64 * (G::NPROD + ncons)), lds, P, flags, N, q,
                          v, a, W, tau, d_kept, n, nc, part, Rws, ncons, (int)tri, null_pivot_sq());
    }
    hipLaunchKernelGGL(fused_reduce_partials_kernel, dim3(G::NC), dim3(256), 0, stream(), part, (int)(grid * G::NPROD),
                       G::NC, d_colsq);
    FIGH_HIP(hipGetLastError());
    *Rws_out = Rws;
    *count_out = grid * ncons;
    return FIGH_OK;
}

}  // namespace figh

using namespace figh;

#ifdef FIGH_ABLATION
extern "C" int figh_fused_prof(double *out16, int reset) {
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fused_prof), sizeof(h)) != hipSuccess) return FIGH_ERR_NO_DEVICE;
    for (int i = 0; i < 16; ++i) out16[i] = (double)h[i];
    if (reset) {
        unsigned long long z[16] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fused_prof), z, sizeof(z));
    }
    return FIGH_OK;
}
#endif

extern "C" int figh_regressor_tsqr_fused(figh_model_t model, int flags, int64_t N, const double *d_q, const double *d_v,
                                         const double *d_a, double *d_W, int64_t ldw, double *d_colsq,
                                         const int32_t *d_kept, int n, const double *d_tau, double tol_qr,
                                         double *d_R_out) {
    FIGH_REQUIRE(model && d_q && d_v && d_a && d_W && d_colsq && d_kept && d_R_out, "NULL pointer");
    FIGH_REQUIRE(N >= 0 && n >= 1, "bad shape");
    const DevModel &h = model->host;
    const int nc = n + (d_tau ? 1 : 0);
    const int ncols = 14 * h.nlinks;
    if (!model->is_chain || (flags & (FIGH_FLAG_TX40 | FIGH_FLAG_GENERIC | FIGH_FLAG_BLOCKED_INPUTS)) || ldw != ncols ||
        (reinterpret_cast<uintptr_t>(d_W) % 16) != 0 || nc > 64 || n > ncols || N < 4096) {
        set_error("fused regressor + TSQR: serial chain, joint-torque mode, packed 16-byte aligned W, at most 64 columns "
                  "(tau included) and at least 4096 samples");
        return FIGH_ERR_UNSUPPORTED;
    }
    if (int rc = ensure_device()) return rc;
    double *Rws = nullptr;
    long count = 0;
    int rc;
    const int f = flags & 7;
    switch (h.nlinks) {
#define FIGH_FUSED_CASE(NJ)                                                                                       \
    case NJ:                                                                                                      \
        rc = launch_fused<NJ>(model, f, (long)N, d_q, d_v, d_a, d_W, d_tau, d_kept, n, d_colsq, &Rws, &count);    \
        break;
        FIGH_FUSED_CASE(5)
        FIGH_FUSED_CASE(6)
        FIGH_FUSED_CASE(7)
#undef FIGH_FUSED_CASE
        default:
            // (eight links: two 58 KB tile buffers leave 17 KB of the 160 KB of LDS, less than three consumer triangles)
            set_error("fused regressor + TSQR: serial chains of 5 to 7 joints (LDS: two tile buffers + three consumer triangles)");
            return FIGH_ERR_UNSUPPORTED;
    }
    if (rc) return rc;
    return tsqr_reduce_stack(Rws, count, nc, n, tol_qr, d_R_out);
}
