// K3, merge tree -- all merge levels of the tall-skinny QR (nc <= 80) in ONE launch, the last level rank-revealing.
//
// After level 0 (figh_linalg.hip) the R factor of W is spread over ~2000 per-wave triangles.  Reducing them is pure
// latency: a level is one sweep of nc dependent column steps (~0.8 us each), however few rows it covers.  Round 1/2 ran
// one launch per level (fan-in 10: 2039 -> 204 -> 21 -> 3 -> 1), a device-side rank decision (figh_base_permutation),
// the regrouped factorisation qr(R[:, perm]) as a one-wave launch, and three copies.  This kernel does the same work as
//
//   * a persistent grid of workgroups that walk the levels themselves: workgroup b of level l factors the stacked rows
//     [b, b+1) * NW*16*NRC of the level's input and publishes its triangle; the next level starts when a device-scope
//     counter says every workgroup of the level has published (all workgroups of the first level are resident at once --
//     the host side checks the grid against the CU count -- so the wait cannot deadlock);
//   * taller register tiles (16*NRC = 96 rows per wave, 768 per workgroup = 15 triangles of UR10): three levels
//     instead of four;
//   * the rank decision and the regrouped factorisation (qrdecomposition.py:215-244) appended to the last level: the
//     workgroup that produced the final triangle R forms perm = [k : |R_kk| > tol | the others | tau] (the reference
//     decides on the diagonal of the PLAIN factorisation, and so does this kernel -- skipping the reflectors of the
//     dependent columns on the fly would give other pivots for every column behind a dependent one: in the plain
//     factorisation the row of a dependent column keeps a direction out of reach of the later reflectors) and one of
//     its waves factors R[:, perm], a matrix of nc rows that fits its register tile: r dependent steps without any
//     cross-wave exchange.  The rows it produces are the rows of qr([W1 W2 tau]), the reference's second
//     factorisation; they are stored in the ORIGINAL column order under the base column they belong to.
//
// Same column-step formulation as tsqr_coop_kernel (DPP pivot operand, one barrier per step, per-wave partial sums
// combined in wave order: bit-reproducible).
#include "figh_internal.h"
#include "figh_wave.h"

namespace figh {

struct TreePlan {
    int nlevels;
    int nb[8];     // workgroups of level l
    long rows[8];  // stacked rows entering level l
};

template <int KK, int P, int NCC, int NRC, int NW>
__device__ __forceinline__ void tree_step(double (&T)[NCC][4 * NRC], const int nc, const int pad, const int lane_c,
                                          const int lane_g, const int wave, const int cstore, double (*pw)[NW][16 * NCC],
                                          double *Rg) {
    constexpr int RPL = 4 * NRC;
    constexpr int LIVE = NCC - P;
    constexpr int kpos = 16 * P + KK;  // padded position of the pivot column
    const int buf = kpos & 1;
#pragma unroll
    for (int cc = 0; cc < LIVE; ++cc) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int i = 0; i < RPL; i += 4) {
            fmac_bcast<KK>(s0, T[P][i], T[P + cc][i]);
            fmac_bcast<KK>(s1, T[P][i + 1], T[P + cc][i + 1]);
            fmac_bcast<KK>(s2, T[P][i + 2], T[P + cc][i + 2]);
            fmac_bcast<KK>(s3, T[P][i + 3], T[P + cc][i + 3]);
        }
        const double dw = allreduce_rowgroups((s0 + s1) + (s2 + s3));
        if (lane_g == 0) pw[buf][wave][16 * (P + cc) + lane_c] = dw;
    }
    __syncthreads();
    double d[LIVE];
#pragma unroll
    for (int cc = 0; cc < LIVE; ++cc) {
        double s = pw[buf][0][16 * (P + cc) + lane_c];
#pragma unroll
        for (int w = 1; w < NW; ++w) s += pw[buf][w][16 * (P + cc) + lane_c];
        d[cc] = s;
    }
    const double sigma = row_bcast<KK>(d[0]);
    if (__builtin_amdgcn_ballot_w64(sigma != 0.0) == 0) return;  // zero column: H = I, the row stays zero
    const int k = kpos - pad;
    const double hq = -0.5 * sigma;
    double rs = __builtin_amdgcn_rsq(sigma);
    rs = rs * fma(hq * rs, rs, 1.5);
    rs = rs * fma(hq * rs, rs, 1.5);
#pragma unroll
    for (int cc = LIVE - 1; cc >= 0; --cc) {
        const double wj = d[cc] * rs;
        const double ncj = -wj * rs;
#pragma unroll
        for (int i = 0; i < RPL; ++i) fmac_bcast<KK>(T[P + cc][i], T[P][i], ncj);
        const int col = 16 * (P + cc) + cstore;  // cstore = lane_c - pad in the lanes that store, hugely negative elsewhere
        if (col >= k) Rg[(unsigned)(k * nc + col)] = -wj;
    }
}

template <int P, int NCC, int NRC, int NW>
__device__ __forceinline__ void tree_panels(double (&T)[NCC][4 * NRC], const int nc, const int pad, const int lane_c,
                                            const int lane_g, const int wave, const int cstore,
                                            double (*pw)[NW][16 * NCC], double *Rg) {
#define FIGH_TSTEP(KK) \
    if (16 * P + KK >= pad) tree_step<KK, P, NCC, NRC, NW>(T, nc, pad, lane_c, lane_g, wave, cstore, pw, Rg);
    FIGH_TSTEP(0) FIGH_TSTEP(1) FIGH_TSTEP(2) FIGH_TSTEP(3) FIGH_TSTEP(4) FIGH_TSTEP(5) FIGH_TSTEP(6) FIGH_TSTEP(7)
    FIGH_TSTEP(8) FIGH_TSTEP(9) FIGH_TSTEP(10) FIGH_TSTEP(11) FIGH_TSTEP(12) FIGH_TSTEP(13) FIGH_TSTEP(14)
    FIGH_TSTEP(15)
#undef FIGH_TSTEP
    if constexpr (P + 1 < NCC) tree_panels<P + 1, NCC, NRC, NW>(T, nc, pad, lane_c, lane_g, wave, cstore, pw, Rg);
}

// One wave, no cross-wave exchange: column step of the regrouped factorisation (the whole nc-row matrix sits in the
// wave's tile).  Row i of the result goes to rows_out[perm[i]][perm[col]] -- the original column order.
template <int KK, int P, int NCC, int NRC>
__device__ __forceinline__ void solo_step(double (&T)[NCC][4 * NRC], const int nc, const int pad, const int lane_c,
                                          const int lane_g, const int (&pcol)[NCC], const int *sperm, double *rows_out) {
    constexpr int RPL = 4 * NRC;
    constexpr int LIVE = NCC - P;
    constexpr int kpos = 16 * P + KK;
    double d[LIVE];
#pragma unroll
    for (int cc = 0; cc < LIVE; ++cc) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int i = 0; i < RPL; i += 4) {
            fmac_bcast<KK>(s0, T[P][i], T[P + cc][i]);
            fmac_bcast<KK>(s1, T[P][i + 1], T[P + cc][i + 1]);
            fmac_bcast<KK>(s2, T[P][i + 2], T[P + cc][i + 2]);
            fmac_bcast<KK>(s3, T[P][i + 3], T[P + cc][i + 3]);
        }
        d[cc] = allreduce_rowgroups((s0 + s1) + (s2 + s3));
    }
    const double sigma = row_bcast<KK>(d[0]);
    if (__builtin_amdgcn_ballot_w64(sigma != 0.0) == 0) return;
    const int i = kpos - pad;
    double *Rrow = rows_out + (long)sperm[i] * nc;
    const double hq = -0.5 * sigma;
    double rs = __builtin_amdgcn_rsq(sigma);
    rs = rs * fma(hq * rs, rs, 1.5);
    rs = rs * fma(hq * rs, rs, 1.5);
#pragma unroll
    for (int cc = LIVE - 1; cc >= 0; --cc) {
        const double wj = d[cc] * rs;
        const double ncj = -wj * rs;
#pragma unroll
        for (int e = 0; e < RPL; ++e) fmac_bcast<KK>(T[P + cc][e], T[P][e], ncj);
        const int col = 16 * (P + cc) + lane_c - pad;
        if (lane_g == 0 && col >= i) Rrow[pcol[P + cc]] = -wj;
    }
}

template <int P, int NCC, int NRC>
__device__ __forceinline__ void solo_panels(double (&T)[NCC][4 * NRC], const int nc, const int pad, const int r,
                                            const int lane_c, const int lane_g, const int (&pcol)[NCC], const int *sperm,
                                            double *rows_out) {
#define FIGH_SSTEP(KK) \
    if (16 * P + KK >= pad && 16 * P + KK < pad + r) solo_step<KK, P, NCC, NRC>(T, nc, pad, lane_c, lane_g, pcol, sperm, rows_out);
    FIGH_SSTEP(0) FIGH_SSTEP(1) FIGH_SSTEP(2) FIGH_SSTEP(3) FIGH_SSTEP(4) FIGH_SSTEP(5) FIGH_SSTEP(6) FIGH_SSTEP(7)
    FIGH_SSTEP(8) FIGH_SSTEP(9) FIGH_SSTEP(10) FIGH_SSTEP(11) FIGH_SSTEP(12) FIGH_SSTEP(13) FIGH_SSTEP(14)
    FIGH_SSTEP(15)
#undef FIGH_SSTEP
    if constexpr (P + 1 < NCC) solo_panels<P + 1, NCC, NRC>(T, nc, pad, r, lane_c, lane_g, pcol, sperm, rows_out);
}

// in: plan.rows[0] x nc stacked triangles; bufA / bufB: ping-pong triangles of the inner levels; out: the plain nc x nc
// triangle (plan.nlevels == 0: `in` already is that triangle).  rows_out != nullptr: the rank decision over the columns
// k < n_free with threshold tol and the regrouped factorisation follow (layout: figh.h, figh_tsqr_selected): nc x nc rows
// + one more row with the diagonal of the plain triangle.
// counters: one per level, zero on entry, zero again on exit.
template <int NCC, int NRC, int NW>
__global__ __launch_bounds__(64 * NW) void tsqr_tree_kernel(const double *in, double *bufA, double *bufB, double *out,
                                                            const TreePlan plan, const int nc0, const int n_free,
                                                            const double tol, double *rows_out, unsigned *counters) {
    __shared__ double pw[2][NW][16 * NCC];
    __shared__ int sperm[16 * NCC];
    constexpr int RPL = 4 * NRC, M = 16 * NRC;
    static_assert(M >= 16 * NCC, "the regrouped factorisation needs the whole triangle in one wave's tile");
    const int lane = threadIdx.x & 63, wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane_c0 = lane & 15, lane_g0 = lane >> 4;
    const int pad0 = 16 * NCC - nc0;  // columns right-aligned, as in tsqr2_kernel
    const int b = blockIdx.x;
    for (int l = 0; l < plan.nlevels; ++l) {
        if (b >= plan.nb[l]) return;  // uniform over the workgroup
        const bool last = l == plan.nlevels - 1;
        const double *src = l == 0 ? in : ((l & 1) ? bufA : bufB);
        double *dst = last ? out : ((l & 1) ? bufB : bufA);
        if (l > 0) {
            if (threadIdx.x == 0) {
                while (__hip_atomic_load(&counters[l - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) <
                       (unsigned)plan.nb[l - 1])
                    __builtin_amdgcn_s_sleep(1);
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // every wave: no stale line of the level's input
        }
        // per-level copies that the optimiser cannot see through: without them every LDS / row address of the 64 unrolled
        // column steps is loop-invariant, gets hoisted out of the level loop and is kept live across it (measured at
        // compile time: 7500 spilled registers)
        int lane_c = lane_c0, lane_g = lane_g0, wave = wave0, nc = nc0, pad = pad0;
        asm volatile("" : "+v"(lane_c), "+v"(lane_g), "+s"(wave), "+s"(nc), "+s"(pad));
        const long rows = plan.rows[l];
        const long r0 = ((long)b * NW + wave) * M;
        double *Rg = dst + (long)b * nc * nc;
        for (int e = threadIdx.x; e < nc * nc; e += 64 * NW) Rg[e] = 0.0;
        double T[NCC][RPL];
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc)
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                // 32-bit element offsets from the uniform base (the host side keeps rows * nc below 2^31)
                const unsigned row = (unsigned)r0 + 16u * (i >> 2) + lane_g + 4u * (i & 3);
                const int col = 16 * cc + lane_c - pad;
                const bool ok = row < (unsigned)rows && col >= 0;
                const double v = src[ok ? row * (unsigned)nc + (unsigned)col : 0u];
                T[cc][i] = ok ? v : 0.0;
            }
        __syncthreads();  // the zero fill of Rg is ordered before the row stores of wave 0 (same workgroup)
        const int cstore = (wave == 0 && lane_g == 0) ? lane_c - pad : -(1 << 24);
        tree_panels<0, NCC, NRC, NW>(T, nc, pad, lane_c, lane_g, wave, cstore, pw, Rg);
        if (!last) {
            __threadfence();  // this thread's part of the triangle is visible device-wide ...
            __syncthreads();  // ... for every thread of the workgroup, before the workgroup is counted
            if (threadIdx.x == 0) __hip_atomic_fetch_add(&counters[l], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (b != 0) return;  // (plan.nlevels == 0: a grid of one)
    // only the workgroup of the last level gets here: every wait on the counters has been passed (a level cannot end
    // before all workgroups of the level before it were counted), so they can be cleared for the next call
    if ((int)threadIdx.x < plan.nlevels - 1) counters[threadIdx.x] = 0u;
    if (!rows_out) return;

    // ---- rank decision + regrouped factorisation (qrdecomposition.py:215-244)
    int lane_c = lane_c0, lane_g = lane_g0, nc = nc0, pad = pad0;
    asm volatile("" : "+v"(lane_c), "+v"(lane_g), "+s"(nc), "+s"(pad));
    const double *R = plan.nlevels == 0 ? in : out;
    __threadfence();
    __syncthreads();  // wave 0's rows of the plain triangle are visible to every wave of this workgroup
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (int e = threadIdx.x; e < nc * nc; e += 64 * NW) rows_out[e] = 0.0;
    for (int e = threadIdx.x; e < nc; e += 64 * NW) rows_out[(long)nc * nc + e] = R[(long)e * nc + e];
    if (wave0 != 0) return;
    // stable partition [base | rest | tau ...] by ballot prefix counts (nc <= 16 NCC <= 128: two rounds of 64)
    int nbase = 0;
    for (int i0 = 0; i0 < n_free; i0 += 64) {
        const int i = i0 + lane;
        nbase += __popcll(__ballot(i < n_free && fabs(R[(long)i * nc + i]) > tol));
    }
    {
        int pb = 0, pr = nbase;
        for (int i0 = 0; i0 < n_free; i0 += 64) {
            const int i = i0 + lane;
            const bool in_range = i < n_free;
            const bool big = in_range && fabs(R[(long)i * nc + i]) > tol;
            const unsigned long long mb = __ballot(big), mr = __ballot(in_range && !big);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (big) sperm[pb + __popcll(mb & below)] = i;
            else if (in_range) sperm[pr + __popcll(mr & below)] = i;
            pb += __popcll(mb);
            pr += __popcll(mr);
        }
        for (int i = n_free + lane; i < nc; i += 64) sperm[i] = i;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    int pcol[NCC];
    double T[NCC][RPL];
#pragma unroll
    for (int cc = 0; cc < NCC; ++cc) {
        const int col = 16 * cc + lane_c - pad;
        pcol[cc] = col >= 0 ? sperm[col] : 0;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int row = 16 * (i >> 2) + lane_g + 4 * (i & 3);
            const bool ok = row < nc && col >= 0;
            const double v = R[ok ? (unsigned)row * (unsigned)nc + (unsigned)pcol[cc] : 0u];
            T[cc][i] = ok ? v : 0.0;
        }
    }
    // the zero fill of rows_out by the other waves must land before this wave's row stores: they have exited or are
    // past their stores only after a workgroup barrier -- but they have left, so order through the memory system instead
    __threadfence();
    solo_panels<0, NCC, NRC>(T, nc, pad, nbase, lane_c, lane_g, pcol, sperm, rows_out);
    if (n_free == nc - 1) {
        // one right-hand side (tau): what is left of it after the base reflectors is the least-squares residual
        // || tau - W1 phi || (the (n, n) entry of the PLAIN triangle also lost what the reflectors of the dependent
        // columns took along their noise directions)
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int i = 0; i < RPL; i += 2) {
            s0 = fma(T[NCC - 1][i], T[NCC - 1][i], s0);
            s1 = fma(T[NCC - 1][i + 1], T[NCC - 1][i + 1], s1);
        }
        const double ss = allreduce_rowgroups(s0 + s1);
        if (lane_g == 0 && lane_c == 15) rows_out[(long)(nc - 1) * nc + nc - 1] = sqrt(ss);
    }
}

// Host side.  Returns FIGH_ERR_UNSUPPORTED when the first level would need more workgroups than the device has CUs (the
// caller then reduces with one launch per level first).  count == 0: Rs already is the plain triangle (regrouping only).
template <int NCC, int NRC, int NW>
static int launch_tree(const double *Rs, long count, int nc, int n_free, double tol, double *d_out, double *d_rows_out,
                       int cus) {
    constexpr long ROWS = 16L * NRC * NW;
    TreePlan plan{};
    long rows = count * nc;
    int l = 0;
    while (count > 0) {
        if (l >= 8) return FIGH_ERR_UNSUPPORTED;
        const long nb = (rows + ROWS - 1) / ROWS;
        plan.rows[l] = rows;
        plan.nb[l] = (int)nb;
        ++l;
        if (nb == 1) break;
        rows = nb * nc;
    }
    plan.nlevels = l;
    if (l > 0 && (plan.nb[0] > cus || plan.rows[0] * nc >= (1L << 31))) return FIGH_ERR_UNSUPPORTED;
    const size_t tri = sizeof(double) * (size_t)nc * nc;
    double *bufA = nullptr, *bufB = nullptr;
    if (plan.nlevels > 1) {
        bufA = static_cast<double *>(workspace(tri * plan.nb[0], 2));
        if (!bufA) return FIGH_ERR_ALLOC;
    }
    if (plan.nlevels > 2) {
        bufB = static_cast<double *>(workspace(tri * plan.nb[1], 3));
        if (!bufB) return FIGH_ERR_ALLOC;
    }
    static unsigned *counters = nullptr;
    if (!counters) FIGH_HIP(hipMalloc(&counters, 8 * sizeof(unsigned)));
    // zeroed before every launch (not left to the previous launch's epilogue: an aborted launch would leave them set)
    FIGH_HIP(hipMemsetAsync(counters, 0, 8 * sizeof(unsigned), stream()));
    const unsigned grid = plan.nlevels > 0 ? (unsigned)plan.nb[0] : 1u;
    // cooperative launch: co-residency of the whole grid is the runtime's guarantee, or the launch is refused (the level
    // counters are spun on by the workgroups of the next level)
    const double *a_Rs = Rs;
    double *a_A = bufA, *a_B = bufB, *a_out = d_out, *a_rows = d_rows_out;
    TreePlan a_plan = plan;
    int a_nc = nc, a_nf = n_free;
    double a_tol = tol;
    unsigned *a_cnt = counters;
    void *args[] = {&a_Rs, &a_A, &a_B, &a_out, &a_plan, &a_nc, &a_nf, &a_tol, &a_rows, &a_cnt};
    const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(&tsqr_tree_kernel<NCC, NRC, NW>), dim3(grid),
                                                    dim3(64 * NW), args, 0, stream());
    if (e == hipErrorCooperativeLaunchTooLarge || e == hipErrorNotSupported || e == hipErrorInvalidConfiguration) {
        (void)hipGetLastError();
        return FIGH_ERR_UNSUPPORTED;
    }
    FIGH_HIP(e);
    return FIGH_OK;
}

int launch_tsqr_tree(const double *Rs, long count, int nc, int n_free, double tol, double *d_out, double *d_rows_out) {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    ProfileScope scope("tsqr_tree");
    if (nc <= 64) return launch_tree<4, 6, 8>(Rs, count, nc, n_free, tol, d_out, d_rows_out, cus);
    if (nc <= 80) return launch_tree<5, 5, 4>(Rs, count, nc, n_free, tol, d_out, d_rows_out, cus);
    return FIGH_ERR_UNSUPPORTED;
}

}  // namespace figh
