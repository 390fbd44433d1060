// K1 -- regressor assembly on gfx950.
//
// Replaces the per-sample Python loop around pin.computeJointTorqueRegressor plus the scatter / friction /
// permutation statements of build_regressor_basic (src/figaroh/tools/regressor.py:20-194) and
// add_coupling_TX40 (:198-227).  Output is written directly in the reference's layout: row r = j*N + i
// (joint-major), 14 columns per link in FIGAROH order.
//
// Two kernels (the second one in figh_regressor_tree.hip):
//
//  regressor_chain_kernel<NJ>   fixed-base serial chains of revolute joints (TX40, UR10: BASELINE configs 1-2).
//      One wavefront = a tile of 64 consecutive samples, one sample per lane.  The tree constants arrive as a
//      by-value kernel argument (SGPRs).  Each lane runs the forward recursion once, keeps per link the
//      rotation liMi and (acc, dw, w) in registers, then walks the output ROW-major: for joint row j the motion
//      axis S_j is pushed down the chain (J_{j,k} = liMi_k^{-1} J_{j,k-1}) and the 10 entries of link k are
//      J^T B_k evaluated in closed form -- 5x fewer flops than propagating the 6x10 body regressor up the
//      chain, and no per-thread arrays.  The lane's 14*NJ row entries go to an LDS tile [64][LDT]; because
//      rows j*N+i0 .. j*N+i0+63 are CONTIGUOUS in W, the wave then streams the tile out as one fully
//      coalesced run of 16-byte stores (43 KB for UR10).  HBM-write bound: 8*(3*NJ) B read + 8*NJ*ncols B
//      written per sample.
//
//  regressor_tape_kernel (figh_regressor_tree.hip)  any tree (TIAGo, TALOS, human: free-flyer, prismatic, continuous
//      joints, external-wrench mode).
//
// All arithmetic is fp64.  6-vectors are (linear, angular).
#include <cmath>
#include <type_traits>
#include <cstdlib>

#include "figh_internal.h"
#include "figh_spatial.h"
#include "figh_chain.h"

namespace figh {

// ---------------------------------------------------------------------------------------------- chain kernel
#ifdef FIGH_ABLATION
__device__ int g_chain_hotin = 0;
__device__ int g_chain_blocked = 0;
#endif

// FIGAROH slot of Pinocchio inertia entry: [xx xy yy xz yz zz] -> Ixx Ixy Ixz Iyy Iyz Izz = slots 0 1 3 2 4 5
template <int NJ, bool TX40, bool COLSQ>
__global__ __launch_bounds__(64) void regressor_chain_kernel(const ChainParams<NJ> P, const int flags, const long N,
                                                             const double *__restrict__ q,
                                                             const double *__restrict__ v,
                                                             const double *__restrict__ a, double *__restrict__ W,
                                                             const long ldw, const int vec_ok,
                                                             double *__restrict__ colsq_part) {
    using G = ChainGeom<NJ, TX40>;
    constexpr int NC = G::NC, LDT = G::LDT;
    extern __shared__ __attribute__((aligned(16))) double tile[];
    const int lane = threadIdx.x;
    double *my = tile + lane * LDT;
    const long ntiles = (N + 63) / 64;
    const bool fric = flags & FIGH_FLAG_FRICTION, actin = flags & FIGH_FLAG_ACT_INERTIA, offs = flags & FIGH_FLAG_OFFSET;

    // COLSQ: on joint row j the columns below 14 j are structurally zero, so the 64 lanes own columns 14 j + lane
    // (csA[j]) and, where the row is wider than 64 live columns, 14 j + 64 + lane (csB[j]): 8 passes over the tile per
    // 64 samples instead of 12.  The per-row owners are combined through LDS once per wave, rows in ascending order.
    double csA[NJ], csB[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) csA[j] = csB[j] = 0.0;

    // The tile body, in two instantiations.  FAST: full tile, packed rows, 16-byte stores -- straight-line code with exactly
    // NJ * CH stores, used by the main loop below.  Otherwise: any tile (ragged last tile, padded leading dimension).
    double pq[NJ], pqd[NJ], pqdd[NJ];  // this tile's inputs (requested one tile ahead in the main loop)
    auto fetch_inputs = [&](const long tile) {
        const long j0 = tile * 64;
        const int nv = (int)((N - j0) < 64 ? (N - j0) : 64);
#ifdef FIGH_ABLATION
        const long is = g_chain_hotin ? (lane < nv ? lane : nv - 1) : j0 + (lane < nv ? lane : nv - 1);
#else
        const long is = j0 + (lane < nv ? lane : nv - 1);
#endif
#pragma unroll
        for (int k = 0; k < NJ; ++k) {
            pq[k] = q[is * NJ + k];
            pqd[k] = v[is * NJ + k];
            pqdd[k] = a[is * NJ + k];
        }
    };
    auto tile_body = [&](auto FAST_T, const long t, const long tnext) {
        constexpr bool FAST = decltype(FAST_T)::value;
        const long i0 = t * 64;
        const int nvalid = FAST ? 64 : (int)((N - i0) < 64 ? (N - i0) : 64);
        double qd[NJ], qdd[NJ];
        double R[NJ][9];
        double acc[NJ][3], dw[NJ][3], w[NJ][3];
        {
            double vl[3] = {0, 0, 0}, om[3] = {0, 0, 0}, al[3] = {-P.g[0], -P.g[1], -P.g[2]}, da[3] = {0, 0, 0};
#pragma unroll
            for (int k = 0; k < NJ; ++k) {
                const double qk = pq[k];
                qd[k] = pqd[k];
                qdd[k] = pqdd[k];
                double s, c;
                sincos(qk, &s, &c);
                double Rj[9];
                rodrigues(P.axis[k], c, s, Rj);
                matmul3(P.Rp[k], Rj, R[k]);
                double t1[3], t2[3], nvl[3], nom[3], nal[3], nda[3];
                cross3(P.pp[k], om, t1);
#pragma unroll
                for (int d = 0; d < 3; ++d) t2[d] = vl[d] - t1[d];
                rotT(R[k], t2, nvl);
                rotT(R[k], om, nom);
                cross3(P.pp[k], da, t1);
#pragma unroll
                for (int d = 0; d < 3; ++d) t2[d] = al[d] - t1[d];
                rotT(R[k], t2, nal);
                rotT(R[k], da, nda);
                double vj[3] = {P.axis[k][0] * qd[k], P.axis[k][1] * qd[k], P.axis[k][2] * qd[k]};
#pragma unroll
                for (int d = 0; d < 3; ++d) nom[d] += vj[d];
                cross3(nvl, vj, t1);
                cross3(nom, vj, t2);
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    nal[d] += t1[d];
                    nda[d] += P.axis[k][d] * qdd[k] + t2[d];
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

        if constexpr (FAST) fetch_inputs(tnext);  // (pq / pqd / pqdd have been consumed by the recursion above)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            // ---- this lane's row (j, i): 14 columns per link
            // links < j are structurally zero; columns below 14 (j - 1) are still zero from the previous row of this tile
            if (j > 0) {
#pragma unroll
                for (int c = 14 * (j - 1); c < 14 * j; ++c) my[c] = 0.0;
            }
            double Jl[3] = {0, 0, 0}, Ja[3] = {P.axis[j][0], P.axis[j][1], P.axis[j][2]};
#pragma unroll
            for (int k = j; k < NJ; ++k) {
                if (k > j) {
                    double t1[3], t2[3], nJl[3], nJa[3];
                    cross3(P.pp[k], Ja, t1);
#pragma unroll
                    for (int d = 0; d < 3; ++d) t2[d] = Jl[d] - t1[d];
                    rotT(R[k], t2, nJl);
                    rotT(R[k], Ja, nJa);
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        Jl[d] = nJl[d];
                        Ja[d] = nJa[d];
                    }
                }
                double *o = my + 14 * k;
                // m
                o[9] = Jl[0] * acc[k][0] + Jl[1] * acc[k][1] + Jl[2] * acc[k][2];
                // mx my mz : Jl x dw + w x (w x Jl) + acc x Ja
                double h1[3], h2[3], h3[3], h4[3];
                cross3(Jl, dw[k], h1);
                cross3(w[k], Jl, h2);
                cross3(w[k], h2, h3);
                cross3(acc[k], Ja, h4);
                o[6] = h1[0] + h3[0] + h4[0];
                o[7] = h1[1] + h3[1] + h4[1];
                o[8] = h1[2] + h3[2] + h4[2];
                // inertia: L(dw)^T Ja - L(w)^T (w x Ja),  L(x)^T y = [x0y0, x1y0+x0y1, x1y1, x2y0+x0y2, x2y1+x1y2, x2y2]
                double u[3];
                cross3(w[k], Ja, u);
                const double *x = dw[k], *z = w[k];
                o[0] = x[0] * Ja[0] - z[0] * u[0];                                        // Ixx
                o[1] = x[1] * Ja[0] + x[0] * Ja[1] - (z[1] * u[0] + z[0] * u[1]);          // Ixy
                o[3] = x[1] * Ja[1] - z[1] * u[1];                                        // Iyy
                o[2] = x[2] * Ja[0] + x[0] * Ja[2] - (z[2] * u[0] + z[0] * u[2]);          // Ixz
                o[4] = x[2] * Ja[1] + x[1] * Ja[2] - (z[2] * u[1] + z[1] * u[2]);          // Iyz
                o[5] = x[2] * Ja[2] - z[2] * u[2];                                        // Izz
                // Ia fv fs off: only on the link's own row (regressor.py:55-70,84-87)
                const bool own = (k == j);
                o[10] = (own && actin) ? qdd[k] : 0.0;
                o[11] = (own && fric) ? qd[k] : 0.0;
                o[12] = (own && fric) ? sgn(qd[k]) : 0.0;
                o[13] = (own && offs) ? 1.0 : 0.0;
            }
            if constexpr (TX40) {
                // regressor.py:216-225: rows of joints 5 and 6 (0-based 4, 5)
                const double sc = sgn(qd[4] + qd[5]);
                my[14 * NJ + 0] = j == 4 ? qdd[5] : (j == 5 ? qdd[4] : 0.0);
                my[14 * NJ + 1] = j == 4 ? qd[5] : (j == 5 ? qd[4] : 0.0);
                my[14 * NJ + 2] = (j == 4 || j == 5) ? sc : 0.0;
            }
            __syncthreads();

            // ---- stream the 64 x NC tile to W rows j*N+i0 .. (contiguous when ldw == NC)
            double *dst = W + ((long)j * N + i0) * ldw;
            bool streamed = false;
            if constexpr (G::VEC == 2) {
                if (FAST || (vec_ok && nvalid == 64 && ldw == NC)) {
                    // full tile, packed rows: the 64 rows are one contiguous run of 64 NC doubles in W.  Chunk id (16 B)
                    // sits at byte 16 id in W and at 16 (id + r) in the padded tile, r = id / CH by a magic multiply
                    // (exact for id < 64 CH <= 4096): 5 instructions per chunk instead of a division and two 64-bit
                    // address computations.
                    constexpr int CH = G::CH;
                    constexpr unsigned MAGIC = ((1u << 20) + CH - 1) / CH;  // exact for id < 64 CH <= 3584 (checked, NJ <= 8)
                    static_assert(64 * CH <= 4096, "magic division range");
                    char *gbase = reinterpret_cast<char *>(dst) + 16 * lane;
                    const char *tbase = reinterpret_cast<const char *>(tile);
#pragma unroll
                    for (int it = 0; it < CH; ++it) {
                        const unsigned id = lane + 64u * it;
                        const unsigned r = (id * MAGIC) >> 20;
                        const double2 val = *reinterpret_cast<const double2 *>(tbase + 16u * (id + r));
                        *reinterpret_cast<double2 *>(gbase + 1024 * it) = val;
                    }
                    streamed = true;
                }
            }
            if (FAST || streamed) {
            } else if (vec_ok) {
                constexpr int CH = G::CH, VEC = G::VEC;
                const int total = nvalid * CH;
                for (int id = lane; id < total; id += 64) {
                    const int r = id / CH, c2 = id - r * CH;
                    if constexpr (VEC == 2) {
                        const double2 val = *reinterpret_cast<const double2 *>(tile + r * LDT + 2 * c2);
                        *reinterpret_cast<double2 *>(dst + (long)r * ldw + 2 * c2) = val;
                    } else {
                        dst[(long)r * ldw + c2] = tile[r * LDT + c2];
                    }
                }
            } else {
                const int total = nvalid * NC;
                for (int id = lane; id < total; id += 64) {
                    const int r = id / NC, c = id - r * NC;
                    dst[(long)r * ldw + c] = tile[r * LDT + c];
                }
            }
            if constexpr (COLSQ) {
                const int lo = 14 * j, width = NC - lo;  // live columns of this row
                if (nvalid == 64) {
                    if (lane < width) {
                        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll
                        for (int r = 0; r < 64; r += 4) {
                            const double x0 = tile[r * LDT + lo + lane], x1 = tile[(r + 1) * LDT + lo + lane];
                            const double x2 = tile[(r + 2) * LDT + lo + lane], x3 = tile[(r + 3) * LDT + lo + lane];
                            t0 += x0 * x0;
                            t1 += x1 * x1;
                            t2 += x2 * x2;
                            t3 += x3 * x3;
                        }
                        csA[j] += (t0 + t1) + (t2 + t3);
                    }
                    if (width > 64 && lane + 64 < width) {
                        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll
                        for (int r = 0; r < 64; r += 4) {
                            const double x0 = tile[r * LDT + lo + 64 + lane], x1 = tile[(r + 1) * LDT + lo + 64 + lane];
                            const double x2 = tile[(r + 2) * LDT + lo + 64 + lane], x3 = tile[(r + 3) * LDT + lo + 64 + lane];
                            t0 += x0 * x0;
                            t1 += x1 * x1;
                            t2 += x2 * x2;
                            t3 += x3 * x3;
                        }
                        csB[j] += (t0 + t1) + (t2 + t3);
                    }
                } else {
                    if (lane < width)
                        for (int r = 0; r < nvalid; ++r) csA[j] += tile[r * LDT + lo + lane] * tile[r * LDT + lo + lane];
                    if (lane + 64 < width)
                        for (int r = 0; r < nvalid; ++r)
                            csB[j] += tile[r * LDT + lo + 64 + lane] * tile[r * LDT + lo + 64 + lane];
                }
            }
            __syncthreads();
        }

    };
    // Main loop over the full tiles when W is packed and aligned.  The inputs of the NEXT tile are requested before this
    // tile's NJ * CH stores.  Loads and stores retire through one in-order counter (vmcnt): a load issued BEHIND the stores
    // can only be waited for together with all of them, and the wave used to sit at the top of every tile until HBM had
    // acknowledged its last store and then delivered q, v, a (with the inputs served from cache the kernel is 0.065 ms
    // faster, tools/k1_alloc_probe.py with FIGH_CHAIN_HOTIN).  Requested up front they are older than the stores, and
    // because this loop's body has a fixed number of stores on every path the compiler's wait for them is vmcnt(63):
    // all but the youngest 63 operations -- i.e. the stores of the last row block stay in flight.
    long t = blockIdx.x;
    if constexpr (G::VEC == 2) {
        const long nfast = (vec_ok && ldw == NC) ? N / 64 : 0;
        long tstep = gridDim.x, tend = nfast;
#ifdef FIGH_ABLATION
        if (g_chain_blocked) {  // FIGH_K1_BLOCKED: every wave walks its own contiguous range of tiles (tools/k1_alloc_probe.py)
            const long tpw = (nfast + gridDim.x - 1) / gridDim.x;
            t = blockIdx.x * tpw;
            tstep = 1;
            tend = t + tpw < nfast ? t + tpw : nfast;
        }
#endif
        if (t < tend) {
            fetch_inputs(t);
            // waited for here, outside the loop: a request still pending at the loop header would put a full wait at the
            // top of every iteration
#pragma unroll
            for (int k = 0; k < NJ; ++k) asm volatile("" : "+v"(pq[k]), "+v"(pqd[k]), "+v"(pqdd[k]));
        }
        for (; t < tend; t += tstep) {
            const long tn = t + tstep;
            tile_body(std::true_type{}, t, tn < tend ? tn : t);
        }
#ifdef FIGH_ABLATION
        if (g_chain_blocked) t = nfast + blockIdx.x;
#endif
    }
    for (; t < ntiles; t += gridDim.x) {
        fetch_inputs(t);
        tile_body(std::false_type{}, t, -1);
    }
    if constexpr (COLSQ) {
        static_assert(NC <= 128, "colsq ownership covers two columns per lane");
        __syncthreads();
        for (int c = lane; c < NC; c += 64) tile[c] = 0.0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int lo = 14 * j, width = NC - lo;
            if (lane < width) tile[lo + lane] += csA[j];
            if (lane + 64 < width) tile[lo + 64 + lane] += csB[j];
            __syncthreads();
        }
        for (int c = lane; c < NC; c += 64) colsq_part[(long)blockIdx.x * NC + c] = tile[c];
    }
}

// partial[b][c] -> out[c]: one workgroup per column, strided partial sums + LDS tree (fixed order: deterministic)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double *__restrict__ part, int nblocks, int ncols,
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

__global__ __launch_bounds__(256) void coupling_tx40_kernel(const long N, const int nv, const double *__restrict__ v,
                                                            const double *__restrict__ a, double *__restrict__ out) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < 6 * N; e += (long)gridDim.x * blockDim.x) {
        const long j = e / N, i = e - j * N;
        double o0 = 0.0, o1 = 0.0, o2 = 0.0;
        if (j >= 4) {
            const double v4 = v[i * nv + 4], v5 = v[i * nv + 5];
            o0 = j == 4 ? a[i * nv + 5] : a[i * nv + 4];
            o1 = j == 4 ? v5 : v4;
            o2 = sgn(v4 + v5);
        }
        out[3 * e] = o0;
        out[3 * e + 1] = o1;
        out[3 * e + 2] = o2;
    }
}

// ---------------------------------------------------------------------------------------------- launchers
template <int NJ, bool TX40>
static int launch_chain(const figh_model_s *m, int flags, long N, const double *q, const double *v, const double *a,
                        double *W, long ldw, double *d_colsq) {
    using G = ChainGeom<NJ, TX40>;
    ChainParams<NJ> P;
    const DevModel &h = m->host;
    for (int k = 0; k < NJ; ++k) {
        for (int d = 0; d < 3; ++d) P.axis[k][d] = h.axis[k + 1][d];
        for (int d = 0; d < 9; ++d) P.Rp[k][d] = h.placement[k + 1][d];
        for (int d = 0; d < 3; ++d) P.pp[k][d] = h.placement[k + 1][9 + d];
    }
    for (int d = 0; d < 3; ++d) P.g[d] = h.gravity[d];
    const long ntiles = (N + 63) / 64;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const size_t lds = sizeof(double) * 64 * G::LDT;
    const int per_cu = (int)((160 * 1024) / lds);
    long grid = (long)cus * (per_cu > 0 ? per_cu : 1) * 2;
    if (grid > ntiles) grid = ntiles;
    if (grid < 1) grid = 1;
    const int vec_ok = (ldw % G::VEC == 0) && ((reinterpret_cast<uintptr_t>(W) % (8 * G::VEC)) == 0);
#ifdef FIGH_ABLATION
    {
        const int hot = getenv("FIGH_CHAIN_HOTIN") != nullptr;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_chain_hotin), &hot, sizeof(int));
        const int blk = getenv("FIGH_K1_BLOCKED") != nullptr;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_chain_blocked), &blk, sizeof(int));
    }
#endif
    ProfileScope scope("regressor_chain", true);
    if (d_colsq) {
        double *part = static_cast<double *>(workspace(sizeof(double) * grid * G::NC, 0));
        if (!part) return FIGH_ERR_ALLOC;
        FIGH_LAUNCH_TIMED((regressor_chain_kernel<NJ, TX40, true>), dim3((unsigned)grid), dim3(64), lds, P, flags, N, q, v, a,
                          W, ldw, vec_ok, part);
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(G::NC), dim3(256), 0, stream(), part, (int)grid,
                           G::NC, d_colsq);
    } else {
        FIGH_LAUNCH_TIMED((regressor_chain_kernel<NJ, TX40, false>), dim3((unsigned)grid), dim3(64), lds, P, flags, N, q, v,
                          a, W, ldw, vec_ok, (double *)nullptr);
    }
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}

}  // namespace figh

using namespace figh;

extern "C" int figh_colsq(const double *d_W, int64_t rows, int cols, int64_t ldw, double *d_out);

extern "C" int figh_coupling_tx40(int64_t N, int nv, const double *d_v, const double *d_a, double *d_out) {
    FIGH_REQUIRE(N >= 0 && nv >= 6, "TX40 coupling needs nv >= 6");
    FIGH_REQUIRE(d_v && d_a && d_out, "NULL device pointer");
    if (int rc = ensure_device()) return rc;
    if (N == 0) return FIGH_OK;
    long blocks = (6 * N + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    ProfileScope scope("coupling_tx40");
    hipLaunchKernelGGL(coupling_tx40_kernel, dim3((unsigned)blocks), dim3(256), 0, stream(), (long)N, nv, d_v, d_a, d_out);
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}

// 64 x width block of a sample-major array -> [value][lane]: read as one contiguous run, transposed through LDS
__global__ __launch_bounds__(256) void repack_samples_kernel(const double *__restrict__ src, const long N, const int width,
                                                             double *__restrict__ dst) {
    extern __shared__ double blk[];  // 64 x (width + 1)
    const long t = blockIdx.x;
    const long i0 = t * 64;
    const int nvalid = (int)((N - i0) < 64 ? (N - i0) : 64);
    const int ld = width | 1;  // odd stride: the transposed reads spread over the banks
    for (int e = threadIdx.x; e < 64 * width; e += 256) {
        const int l = e / width, k = e - l * width;
        blk[l * ld + k] = src[(i0 + (l < nvalid ? l : nvalid - 1)) * width + k];
    }
    __syncthreads();
    double *out = dst + t * 64 * width;
    for (int e = threadIdx.x; e < 64 * width; e += 256) {
        const int k = e >> 6, l = e & 63;
        out[e] = blk[l * ld + k];
    }
}

extern "C" int figh_repack_samples(const double *d_src, int64_t N, int width, double *d_dst) {
    FIGH_REQUIRE(d_src && d_dst, "NULL device pointer");
    FIGH_REQUIRE(N >= 0 && width >= 1 && width <= 256, "figh_repack_samples: 1 .. 256 values per sample");
    if (int rc = ensure_device()) return rc;
    if (N == 0) return FIGH_OK;
    ProfileScope scope("repack_samples");
    const long ntiles = (N + 63) / 64;
    hipLaunchKernelGGL(repack_samples_kernel, dim3((unsigned)ntiles), dim3(256), sizeof(double) * 64 * (width | 1), stream(),
                       d_src, (long)N, width, d_dst);
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}

extern "C" int figh_regressor_build(figh_model_t model, int mode, int flags, int ft_mask, int64_t N,
                                    const double *d_q, const double *d_v, const double *d_a, double *d_W, int64_t ldw,
                                    double *d_colsq) {
    int rps = 0, ncols = 0;
    if (int rc = figh_regressor_shape(model, mode, flags, &rps, &ncols)) return rc;
    FIGH_REQUIRE(N >= 0, "N < 0");
    FIGH_REQUIRE(d_q && d_v && d_a && d_W, "NULL device pointer");
    FIGH_REQUIRE(ldw >= ncols, "ldw smaller than the number of columns");
    const DevModel &h = model->host;
    if (mode == FIGH_MODE_EXT_WRENCH) {
        FIGH_REQUIRE((ft_mask & ~63) == 0, "Please enter valid parameters");  // regressor.py:140
        if (flags & (FIGH_FLAG_FRICTION | FIGH_FLAG_ACT_INERTIA)) {
            // the reference indexes v[i, link] / a[i, link] (regressor.py:146-156): IndexError beyond nv
            FIGH_REQUIRE(h.nlinks <= h.nv, "external-wrench friction/inertia columns need njoints-1 <= nv");
        }
    }
    if (int rc = ensure_device()) return rc;
    if (N == 0) {
        if (d_colsq) FIGH_HIP(hipMemsetAsync(d_colsq, 0, sizeof(double) * ncols, stream()));
        return FIGH_OK;
    }
    const bool tx40 = flags & FIGH_FLAG_TX40;
    int rc;
    if (model->is_chain && mode == FIGH_MODE_JOINT_TORQUE && !(flags & FIGH_FLAG_GENERIC)) {
        FIGH_REQUIRE(!(flags & FIGH_FLAG_BLOCKED_INPUTS), "tile-blocked inputs are for the generic-tree kernel (the chain "
                                                          "kernel reads 48-byte runs per lane from the original arrays)");
        const int f = flags & 7;
        switch (h.nlinks) {
#define FIGH_CHAIN_CASE(NJ)                                                                             \
    case NJ:                                                                                            \
        rc = launch_chain<NJ, false>(model, f, N, d_q, d_v, d_a, d_W, ldw, d_colsq);                    \
        break;
            FIGH_CHAIN_CASE(1)
            FIGH_CHAIN_CASE(2)
            FIGH_CHAIN_CASE(3)
            FIGH_CHAIN_CASE(4)
            FIGH_CHAIN_CASE(5)
            case 6:
                rc = tx40 ? launch_chain<6, true>(model, f, N, d_q, d_v, d_a, d_W, ldw, d_colsq)
                          : launch_chain<6, false>(model, f, N, d_q, d_v, d_a, d_W, ldw, d_colsq);
                break;
            FIGH_CHAIN_CASE(7)
            FIGH_CHAIN_CASE(8)
#undef FIGH_CHAIN_CASE
            default:
                set_error("chain kernel: unsupported link count");
                return FIGH_ERR_UNSUPPORTED;
        }
        return rc;
    }
    int colsq_done = 0;
    rc = launch_regressor_tree(model, mode, flags, ft_mask, N, d_q, d_v, d_a, d_W, ldw, ncols, 14, d_colsq, &colsq_done);
    if (rc) return rc;
    if (colsq_done) return FIGH_OK;
    if (d_colsq) return figh_colsq(d_W, (int64_t)rps * N, ncols, ldw, d_colsq);
    return FIGH_OK;
}

extern "C" int figh_regressor_link_layout(figh_model_t model, int mode, int flags, int ft_mask, int32_t *h_link_pos,
                                          int *nlive) {
    FIGH_REQUIRE(model && h_link_pos && nlive, "NULL pointer");
    int pos[kMaxJoints];
    const int live = tree_link_positions(model, mode, flags, ft_mask, pos);
    if (live < 0) {
        set_error("link-compact W: external-wrench regressor of a model with a free-flyer root only");
        return FIGH_ERR_UNSUPPORTED;
    }
    for (int l = 0; l < model->host.nlinks; ++l) h_link_pos[l] = pos[l];
    *nlive = live;
    return FIGH_OK;
}

extern "C" int figh_regressor_force_layout(figh_model_t model, int mode, int flags, int ft_mask, int64_t *ld_force) {
    FIGH_REQUIRE(model && ld_force, "NULL pointer");
    const long ldf = tree_force_ld(model, mode, flags, ft_mask);
    if (ldf <= 0) {
        set_error("force-compact W: external-wrench regressor of a model with a free-flyer root, no friction / inertia / offset columns");
        return FIGH_ERR_UNSUPPORTED;
    }
    *ld_force = ldf;
    return FIGH_OK;
}

// Link-padded form of figh_regressor_build for W that stays on the device (see figh.h): 16 columns per link.
extern "C" int figh_regressor_build_padded(figh_model_t model, int mode, int flags, int ft_mask, int64_t N,
                                           const double *d_q, const double *d_v, const double *d_a, double *d_W,
                                           int64_t ldw, double *d_colsq) {
    int rps = 0, ncols = 0;
    if (int rc = figh_regressor_shape(model, mode, flags, &rps, &ncols)) return rc;
    FIGH_REQUIRE(N >= 0, "N < 0");
    FIGH_REQUIRE(d_q && d_v && d_a && (d_W || d_colsq), "NULL device pointer");
    FIGH_REQUIRE(!(model->is_chain && mode == FIGH_MODE_JOINT_TORQUE) && !(flags & FIGH_FLAG_TX40),
                 "the link-padded layout is for tree models (chains are written as dense row tiles)");
    const DevModel &h = model->host;
    if (mode == FIGH_MODE_EXT_WRENCH) {
        FIGH_REQUIRE((ft_mask & ~63) == 0, "Please enter valid parameters");  // regressor.py:140
        if (flags & (FIGH_FLAG_FRICTION | FIGH_FLAG_ACT_INERTIA))
            FIGH_REQUIRE(h.nlinks <= h.nv, "external-wrench friction/inertia columns need njoints-1 <= nv");
    }
    if (int rc = ensure_device()) return rc;
    if (N == 0) {
        if (d_colsq) FIGH_HIP(hipMemsetAsync(d_colsq, 0, sizeof(double) * ncols, stream()));
        return FIGH_OK;
    }
    int done = 0;
    return launch_regressor_tree(model, mode, flags & (7 | FIGH_FLAG_BLOCKED_INPUTS | FIGH_FLAG_ZEROS_PRESENT | FIGH_FLAG_COMPACT_BLOCKS |
                                                       FIGH_FLAG_LINK_COMPACT | FIGH_FLAG_FORCE_COMPACT),
                                 ft_mask, N,
                                 d_q, d_v, d_a, d_W, ldw,
                                 ncols, 16, d_colsq, &done);
}
