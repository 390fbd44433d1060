// K3, streamed merge tree -- every merge level of the tall-skinny QR (nc <= 80), the rank decision and the regrouped
// factorisation as ONE software pipeline: about nc dependent column steps in total instead of nc per level.
//
// Why it can be pipelined.  A merge workgroup factors the stacked triangles of its children with the "output starts
// empty" Householder step (figh_linalg.hip, tsqr_coop_kernel): row k of its own triangle is FINAL after column step k.
// A parent's column step k only involves the rows 0 .. k of its children (rows further down are zero in column k, so the
// reflector leaves them alone).  Hence a parent may run one step behind its children, its parent one step behind it, and
// so on: levels overlap, and the whole tree costs ~nc steps + a few steps of lag per level.  Measured before (one level
// after the other, figh_tsqr_tree.hip): ~60 us per level, 2039 UR10 triangles -> 1 in 250 us + 40 us of regrouping.
//
// Data flow.  Every workgroup owns a triangle buffer in HBM that it fills row by row with device-coherent (sc1) stores.
// The buffers are kept POISONED (a NaN payload no computation produces): a consumer loads the row it needs next one step
// ahead with coherent loads and simply retries while it still sees poison -- each 8-byte entry is written once, atomically,
// so "no poison" == "final".  No flags, no fences, no ordering assumptions between different addresses.  The consumer
// re-poisons what it has read, so the buffers are clean again when the launch ends.
//
// Tile layout of a merge workgroup (8 waves): 16 NCC padded column positions x FAN children.  The FAN rows "position p of
// every child" form one row chunk of the MFMA C/D layout (lane = 16 g + c holds child g + 4 r in register r); position p
// lives in wave p mod 8, slot p / 8.  Rows of position p are zero in the column chunks below p / 16, which are not
// allocated at all: 80 doubles per lane for nc <= 64 and a fan-in of 16 (a dense 16-child tile would need 200), and every
// wave has the same work in every step.  The step code is unrolled over the (compile-time) pivot position, so the
// registers a row arrives in are static.
//
// Last stage (one wave of its own workgroup): the rows of the root's triangle R stream into a 64-row tile; at step k the
// diagonal entry R_kk has arrived, the reference's rank decision |R_kk| > tol (qrdecomposition.py:215-221) is taken on it,
// and only base columns get a reflector -- on R this is the regrouped factorisation qr([W1 W2 tau]) of
// qrdecomposition.py:223-244 (W2's entries under later base rows vanish: a dependent column lies in the span of the base
// columns before it).  Output layout: include/figh.h, figh_tsqr_selected.
#include "figh_internal.h"
#include "figh_wave.h"

namespace figh {

#ifdef FIGH_ABLATION
// in-kernel step profile of the ablation build (tools/stream_prof.py): s_memtime at five points of every column step of
// wave 0 of workgroup 0, accumulated per point
__device__ long long g_stream_prof[8];
#define FIGH_PROF_MARK(i) \
    if (blockIdx.x == 0 && threadIdx.x == 0) { const long long now_ = (long long)__builtin_readcyclecounter(); g_stream_prof[i] += now_ - prof_t; prof_t = now_; }
#define FIGH_PROF_BEGIN long long prof_t = (long long)__builtin_readcyclecounter();
#else
#define FIGH_PROF_MARK(i)
#define FIGH_PROF_BEGIN
#endif

constexpr unsigned long long kPoison = 0xFFFBADC0FFEE5EEDull;
// how many column steps ahead a row is requested.  Measured (tools/merge_tree_bench.py, 2039 triangles of 50 columns):
// depth 1 / 2 / 3 / 4 -> 107 / 117 / 124 / 129 us: a deeper request only adds lag per level, the step itself does not wait
// for its loads
#ifndef FIGH_STREAM_DEPTH
#define FIGH_STREAM_DEPTH 1
#endif
constexpr int kDepth = FIGH_STREAM_DEPTH;

struct StreamPlan {
    int nlevels;   // merge levels; 0: `in` already is the plain triangle (regrouping only)
    int nb[6];     // workgroups of level l
    int first[6];  // first blockIdx of level l (first[nlevels] = the regrouping workgroup)
    int nin[6];    // triangles entering level l
};

__device__ __forceinline__ double ld_coherent(const double *p) {
    return __longlong_as_double(__hip_atomic_load(reinterpret_cast<const long long *>(p), __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_coherent(double *p, const double v) {
    __hip_atomic_store(reinterpret_cast<long long *>(p), __double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool is_poison(const double v) { return (unsigned long long)__double_as_longlong(v) == kPoison; }

// ---- tile of a merge workgroup: slot j (position w + 8 j) keeps the chunks j/2 .. NCC-1, FANR registers each
constexpr int merge_slot_base(int ncc, int fanr, int j) {  // fanr * sum_{q<j} (ncc - q/2), in closed form (m = j/2)
    return fanr * (2 * (j / 2) * ncc - (j / 2) * (j / 2 - 1) + (j & 1) * (ncc - j / 2));
}
constexpr int merge_tile_n(int ncc, int fanr) { return merge_slot_base(ncc, fanr, 2 * ncc); }
template <int NCC, int FANR>
struct MergeTile {
    static constexpr int ix(int j, int cc, int r) { return merge_slot_base(NCC, FANR, j) + (cc - j / 2) * FANR + r; }
};

template <int NCC, int FANR>
struct MergeCtx {
    const double *child;    // triangle 0 of this workgroup's children (child c at child + c nc^2)
    double *repoison;       // same address when the children's rows are to be re-poisoned after reading, else nullptr
    double *stream;         // this workgroup's own streamed triangle (nullptr: not streamed)
    double *plain;          // the root also writes the plain triangle here (nullptr otherwise)
    int coff[FANR];         // per lane: child (g + 4 r) * nc^2 + lane_c - pad, in elements (>= 0 wherever a lane is active)
    bool have[FANR];        // child g + 4 r exists
    int nc, pad, wave, lane_c, lane_g, cstore;
};

// loads of the rows at padded position Q (all children) into their slot; lanes left of the diagonal are exact zeros
template <int Q, int NCC, int FANR>
__device__ __forceinline__ void arrive_issue(double (&T)[merge_tile_n(NCC, FANR)], const MergeCtx<NCC, FANR> &c) {
    using MT = MergeTile<NCC, FANR>;
    constexpr int J = Q / 8, PQ = Q / 16, KQ = Q % 16;
    const double *row = c.child + (long)(Q - c.pad) * c.nc;  // uniform
#pragma unroll
    for (int cc = PQ; cc < NCC; ++cc)
#pragma unroll
        for (int r = 0; r < FANR; ++r) {
            const bool act = c.have[r] && (cc > PQ || c.lane_c >= KQ);
            double v = 0.0;
            if (act) v = ld_coherent(row + 16 * cc + c.coff[r]);
            T[MT::ix(J, cc, r)] = v;
        }
}

template <int Q, int NCC, int FANR>
__device__ __forceinline__ void arrive_validate(double (&T)[merge_tile_n(NCC, FANR)], const MergeCtx<NCC, FANR> &c) {
    using MT = MergeTile<NCC, FANR>;
    constexpr int J = Q / 8, PQ = Q / 16, KQ = Q % 16;
    for (;;) {
        bool bad = false;
#pragma unroll
        for (int cc = PQ; cc < NCC; ++cc)
#pragma unroll
            for (int r = 0; r < FANR; ++r) bad |= is_poison(T[MT::ix(J, cc, r)]);
        if (__builtin_amdgcn_ballot_w64(bad) == 0) break;
        __builtin_amdgcn_s_sleep(2);
        arrive_issue<Q, NCC, FANR>(T, c);
    }
    if (c.repoison) {
        double *row = c.repoison + (long)(Q - c.pad) * c.nc;
#pragma unroll
        for (int cc = PQ; cc < NCC; ++cc)
#pragma unroll
            for (int r = 0; r < FANR; ++r)
                if (c.have[r] && (cc > PQ || c.lane_c >= KQ))
                    st_coherent(row + 16 * cc + c.coff[r], __longlong_as_double((long long)kPoison));
    }
}

template <int KK, int P, int NCC, int FANR, int NW>
__device__ __forceinline__ void stream_step(double (&T)[merge_tile_n(NCC, FANR)], const MergeCtx<NCC, FANR> &c,
                                            double (*pw)[NW][16 * NCC]) {
    using MT = MergeTile<NCC, FANR>;
    constexpr int LIVE = NCC - P;
    constexpr int kpos = 16 * P + KK;
    constexpr int J = kpos / 8, OW = kpos % 8;  // slot and owner wave of the arriving position
    const int buf = kpos & 1;
    FIGH_PROF_BEGIN
    // the rows of position kpos + kDepth are requested now by the wave that owns them (its slot is not read before then)
    if constexpr (kpos + kDepth < 16 * NCC) {
        if (c.wave == (kpos + kDepth) % 8) arrive_issue<kpos + kDepth, NCC, FANR>(T, c);
    }
    if (c.wave == OW) {
        if (kpos - c.pad < kDepth) arrive_issue<kpos, NCC, FANR>(T, c);  // the first steps: nobody asked before
        arrive_validate<kpos, NCC, FANR>(T, c);
    }
    FIGH_PROF_MARK(0)
    const bool own = c.wave <= OW;  // this wave's slot J has arrived (position wave + 8 J <= kpos)
    double a0[LIVE], a1[LIVE];
#pragma unroll
    for (int cc = 0; cc < LIVE; ++cc) {
        a0[cc] = 0.0;
        a1[cc] = 0.0;
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int r = 0; r < FANR; ++r) {
                if ((j * FANR + r) & 1) fmac_bcast<KK>(a1[cc], T[MT::ix(j, P, r)], T[MT::ix(j, P + cc, r)]);
                else fmac_bcast<KK>(a0[cc], T[MT::ix(j, P, r)], T[MT::ix(j, P + cc, r)]);
            }
    }
    if (own) {
#pragma unroll
        for (int cc = 0; cc < LIVE; ++cc)
#pragma unroll
            for (int r = 0; r < FANR; ++r) {
                if (r & 1) fmac_bcast<KK>(a1[cc], T[MT::ix(J, P, r)], T[MT::ix(J, P + cc, r)]);
                else fmac_bcast<KK>(a0[cc], T[MT::ix(J, P, r)], T[MT::ix(J, P + cc, r)]);
            }
    }
#pragma unroll
    for (int cc = 0; cc < LIVE; ++cc) {
        const double dw = allreduce_rowgroups(a0[cc] + a1[cc]);
        if (c.lane_g == 0) pw[buf][c.wave][16 * (P + cc) + c.lane_c] = dw;
    }
    FIGH_PROF_MARK(1)
    __syncthreads();
    FIGH_PROF_MARK(2)
    // every wave sums the partials itself, in wave order: same bits everywhere.  Pivot chunk first: it feeds the
    // rsq chain, behind which the reads of the trailing chunks hide.
    double d[LIVE];
#pragma unroll
    for (int cc = 0; cc < LIVE; ++cc) {
        double pp[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) pp[w] = pw[buf][w][16 * (P + cc) + c.lane_c];
        // fixed pairwise order (three dependent additions instead of seven)
        d[cc] = ((pp[0] + pp[1]) + (pp[2] + pp[3])) + ((pp[4] + pp[5]) + (pp[6] + pp[7]));
        if (cc == 0) __builtin_amdgcn_sched_barrier(0);
    }
    const double sigma = row_bcast<KK>(d[0]);
    double rs = __builtin_amdgcn_rsq(sigma);
    {   // one third-order step on the 2^-24 seed: y (1 + e/2 + 3 e^2 / 8), e = 1 - sigma y^2
        const double e = fma(-(sigma * rs), rs, 1.0);
        rs = fma(rs, fma(e, 0.375, 0.5) * e, rs);
    }
    rs = sigma != 0.0 ? rs : 0.0;  // zero column: H = I and a row of zeros (which must still be published)
    const int k = kpos - c.pad;
#ifdef FIGH_ABLATION
    asm volatile("" : "+v"(rs));
#endif
    FIGH_PROF_MARK(3)
#pragma unroll
    for (int cc = LIVE - 1; cc >= 0; --cc) {
        const double wj = d[cc] * rs;
        const double ncj = -wj * rs;
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int r = 0; r < FANR; ++r) fmac_bcast<KK>(T[MT::ix(j, P + cc, r)], T[MT::ix(j, P, r)], ncj);
        if (own) {
#pragma unroll
            for (int r = 0; r < FANR; ++r) fmac_bcast<KK>(T[MT::ix(J, P + cc, r)], T[MT::ix(J, P, r)], ncj);
        }
        const int col = 16 * (P + cc) + c.cstore;  // lane_c - pad in the storing lanes, hugely negative elsewhere
        if (col >= k) {
            const double val = 0.0 - wj;
            if (c.stream) st_coherent(c.stream + (unsigned)(k * c.nc + col), val);
            if (c.plain) c.plain[(unsigned)(k * c.nc + col)] = val;
        }
    }
    FIGH_PROF_MARK(4)
}

template <int P, int NCC, int FANR, int NW>
__device__ __forceinline__ void stream_panels(double (&T)[merge_tile_n(NCC, FANR)], const MergeCtx<NCC, FANR> &c,
                                              double (*pw)[NW][16 * NCC]) {
#define FIGH_QSTEP(KK) \
    if (16 * P + KK >= c.pad) stream_step<KK, P, NCC, FANR, NW>(T, c, pw);
    FIGH_QSTEP(0) FIGH_QSTEP(1) FIGH_QSTEP(2) FIGH_QSTEP(3) FIGH_QSTEP(4) FIGH_QSTEP(5) FIGH_QSTEP(6) FIGH_QSTEP(7)
    FIGH_QSTEP(8) FIGH_QSTEP(9) FIGH_QSTEP(10) FIGH_QSTEP(11) FIGH_QSTEP(12) FIGH_QSTEP(13) FIGH_QSTEP(14)
    FIGH_QSTEP(15)
#undef FIGH_QSTEP
    if constexpr (P + 1 < NCC) stream_panels<P + 1, NCC, FANR, NW>(T, c, pw);
}

// ---- last stage: one wave, rows of R stream into a 16 NCC-row tile (row chunk rc keeps the chunks rc .. NCC-1)
constexpr int solo_chunk_base(int ncc, int rc) { return 4 * (rc * ncc - rc * (rc - 1) / 2); }  // 4 sum_{q<rc} (ncc - q)
constexpr int solo_tile_n(int ncc) { return solo_chunk_base(ncc, ncc); }
template <int NCC>
struct SoloTile {
    static constexpr int ix(int rc, int cc, int r) { return solo_chunk_base(NCC, rc) + (cc - rc) * 4 + r; }
};

struct SoloCtx {
    const double *R;    // the root's streamed triangle (or the complete input)
    double *repoison;   // R when it is a stream buffer
    double *rows_out;   // (nc + 1) x nc
    int nc, pad, lane_c, lane_g, n_free;
    double tol;
};

template <int Q, int NCC>
__device__ __forceinline__ void solo_issue(double (&S)[NCC], const SoloCtx &c) {
    constexpr int PQ = Q / 16, KQ = Q % 16;
    const double *row = c.R + (long)(Q - c.pad) * c.nc - c.pad;
#pragma unroll
    for (int cc = PQ; cc < NCC; ++cc) {
        const bool act = c.lane_g == KQ % 4 && (cc > PQ || c.lane_c >= KQ);
        double v = 0.0;
        if (act) v = ld_coherent(row + 16 * cc + c.lane_c);
        S[cc] = v;
    }
}

template <int KK, int P, int NCC>
__device__ __forceinline__ void solo_stream_step(double (&T)[solo_tile_n(NCC)], double (&SS)[kDepth][NCC], const SoloCtx &c) {
    using ST = SoloTile<NCC>;
    constexpr int LIVE = NCC - P;
    constexpr int kpos = 16 * P + KK;
    constexpr int RR = KK / 4, RG = KK % 4;  // register and row group of the arriving row inside row chunk P
    // arrival of row kpos (requested kDepth steps ago into staging set kpos mod kDepth)
    double (&S)[NCC] = SS[kpos % kDepth];
    if (kpos - c.pad < kDepth) solo_issue<kpos, NCC>(S, c);  // the first steps: nobody asked before
    for (;;) {
        bool bad = false;
#pragma unroll
        for (int cc = P; cc < NCC; ++cc) bad |= is_poison(S[cc]);
        if (__builtin_amdgcn_ballot_w64(bad) == 0) break;
        __builtin_amdgcn_s_sleep(2);
        solo_issue<kpos, NCC>(S, c);
    }
    const int k = kpos - c.pad;
    if (c.repoison) {
        double *row = c.repoison + (long)k * c.nc - c.pad;
#pragma unroll
        for (int cc = P; cc < NCC; ++cc)
            if (c.lane_g == RG && (cc > P || c.lane_c >= KK))
                st_coherent(row + 16 * cc + c.lane_c, __longlong_as_double((long long)kPoison));
    }
#pragma unroll
    for (int cc = P; cc < NCC; ++cc)
        if (c.lane_g == RG) T[ST::ix(P, cc, RR)] = S[cc];
    if constexpr (kpos + kDepth < 16 * NCC) solo_issue<kpos + kDepth, NCC>(S, c);  // same staging set: it is free now
    // the diagonal entry of the plain triangle: lane (RG, KK) of the pivot chunk
    const double mine = T[ST::ix(P, P, RR)];
    const double dkk = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(mine), 16 * RG + KK),
                                        __builtin_amdgcn_readlane(__double2loint(mine), 16 * RG + KK));
    if (c.lane_g == 0 && c.lane_c == KK) c.rows_out[(long)c.nc * c.nc + k] = dkk;
    // qrdecomposition.py:215-221: base column iff |R_kk| > tol (NaN: not base); columns k >= n_free (tau) always are
    if (k < c.n_free && !(fabs(dkk) > c.tol)) return;
    double d[LIVE];
#pragma unroll
    for (int cc = 0; cc < LIVE; ++cc) {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int rc = 0; rc <= P; ++rc)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                fmac_bcast<KK>(a0, T[ST::ix(rc, P, r)], T[ST::ix(rc, P + cc, r)]);
                fmac_bcast<KK>(a1, T[ST::ix(rc, P, r + 1)], T[ST::ix(rc, P + cc, r + 1)]);
            }
        d[cc] = allreduce_rowgroups(a0 + a1);
    }
    const double sigma = row_bcast<KK>(d[0]);
    double rs = __builtin_amdgcn_rsq(sigma);
    {
        const double e = fma(-(sigma * rs), rs, 1.0);
        rs = fma(rs, fma(e, 0.375, 0.5) * e, rs);
    }
    rs = sigma != 0.0 ? rs : 0.0;
    double *Rrow = c.rows_out + (long)k * c.nc - c.pad;
#pragma unroll
    for (int cc = LIVE - 1; cc >= 0; --cc) {
        const double wj = d[cc] * rs;
        const double ncj = -wj * rs;
#pragma unroll
        for (int rc = 0; rc <= P; ++rc)
#pragma unroll
            for (int r = 0; r < 4; ++r) fmac_bcast<KK>(T[ST::ix(rc, P + cc, r)], T[ST::ix(rc, P, r)], ncj);
        // the whole live part of the row: columns left of the pivot inside its chunk are finished base columns
        // (residues ~ 0) or dependent columns (their entry of R2)
        if (c.lane_g == 0 && 16 * (P + cc) + c.lane_c >= c.pad) Rrow[16 * (P + cc) + c.lane_c] = 0.0 - wj;
    }
}

template <int P, int NCC>
__device__ __forceinline__ void solo_stream_panels(double (&T)[solo_tile_n(NCC)], double (&S)[kDepth][NCC], const SoloCtx &c) {
#define FIGH_RSTEP(KK) \
    if (16 * P + KK >= c.pad) solo_stream_step<KK, P, NCC>(T, S, c);
    FIGH_RSTEP(0) FIGH_RSTEP(1) FIGH_RSTEP(2) FIGH_RSTEP(3) FIGH_RSTEP(4) FIGH_RSTEP(5) FIGH_RSTEP(6) FIGH_RSTEP(7)
    FIGH_RSTEP(8) FIGH_RSTEP(9) FIGH_RSTEP(10) FIGH_RSTEP(11) FIGH_RSTEP(12) FIGH_RSTEP(13) FIGH_RSTEP(14)
    FIGH_RSTEP(15)
#undef FIGH_RSTEP
    if constexpr (P + 1 < NCC) solo_stream_panels<P + 1, NCC>(T, S, c);
}

// in: plan.nin[0] stacked nc x nc triangles.  sbuf: one poisoned nc x nc triangle per merge workgroup (by blockIdx).
// out (nullable): the plain triangle.  rows_out (nullable): (nc + 1) x nc, rank decision over the columns k < n_free.
template <int NCC, int FANR, int NW>
__global__ __launch_bounds__(64 * NW) void tsqr_stream_kernel(const double *in, double *sbuf, double *out, double *rows_out,
                                                              const StreamPlan plan, const int nc, const int n_free,
                                                              const double tol) {
    static_assert(NW == 8, "positions are dealt to eight waves");
    constexpr int FAN = 4 * FANR;
    __shared__ double pw[2][NW][16 * NCC];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane_c = lane & 15, lane_g = lane >> 4;
    const int pad = 16 * NCC - nc;
    const int b = blockIdx.x;
    const long tri = (long)nc * nc;
    if (b < plan.first[plan.nlevels]) {
        // ---- a merge workgroup
        int l = 0;
        while (l + 1 < plan.nlevels && b >= plan.first[l + 1]) ++l;
        const int bl = b - plan.first[l];
        const bool root = l == plan.nlevels - 1;
        MergeCtx<NCC, FANR> c;
        c.nc = nc;
        c.pad = pad;
        c.wave = wave;
        c.lane_c = lane_c;
        c.lane_g = lane_g;
        c.cstore = (wave == 0 && lane_g == 0) ? lane_c - pad : -(1 << 24);
        const int nchild = min(FAN, plan.nin[l] - FAN * bl);
        if (l == 0) {
            c.child = in + (long)FAN * bl * tri;
            c.repoison = nullptr;
        } else {
            double *cb = sbuf + (long)(plan.first[l - 1] + FAN * bl) * tri;
            c.child = cb;
            c.repoison = cb;
        }
#pragma unroll
        for (int r = 0; r < FANR; ++r) {
            const int ch = lane_g + 4 * r;
            c.have[r] = ch < nchild;
            c.coff[r] = (c.have[r] ? ch : 0) * (int)tri + lane_c - pad;
        }
        c.stream = (!root || rows_out) ? sbuf + (long)b * tri : nullptr;
        c.plain = (root && out) ? out : nullptr;
        if (c.plain) {
            for (int e = threadIdx.x; e < nc * nc; e += 64 * NW) c.plain[e] = 0.0;
        }
        double T[merge_tile_n(NCC, FANR)];
#pragma unroll
        for (int e = 0; e < merge_tile_n(NCC, FANR); ++e) T[e] = 0.0;
        __syncthreads();  // the zero fill of the plain triangle is ordered before wave 0's row stores
        stream_panels<0, NCC, FANR, NW>(T, c, pw);
        return;
    }
    // ---- the regrouping workgroup (present iff rows_out)
    for (int e = threadIdx.x; e < nc * nc; e += 64 * NW) rows_out[e] = 0.0;
    __syncthreads();
    if (wave != 0) return;
    SoloCtx c;
    c.nc = nc;
    c.pad = pad;
    c.lane_c = lane_c;
    c.lane_g = lane_g;
    c.n_free = n_free;
    c.tol = tol;
    c.rows_out = rows_out;
    if (plan.nlevels == 0) {
        c.R = in;
        c.repoison = nullptr;
    } else {
        double *rb = sbuf + (long)(plan.first[plan.nlevels] - 1) * tri;  // the root is the last merge workgroup
        c.R = rb;
        c.repoison = rb;
    }
    double T[solo_tile_n(NCC)], S[kDepth][NCC];
#pragma unroll
    for (int e = 0; e < solo_tile_n(NCC); ++e) T[e] = 0.0;
#pragma unroll
    for (int dd = 0; dd < kDepth; ++dd)
#pragma unroll
        for (int e = 0; e < NCC; ++e) S[dd][e] = 0.0;
    solo_stream_panels<0, NCC>(T, S, c);
}

__global__ __launch_bounds__(256) void poison_fill_kernel(double *p, const long n) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256)
        p[e] = __longlong_as_double((long long)kPoison);
}

template <int NCC, int FANR>
static int launch_stream(const double *Rs, long count, int nc, int n_free, double tol, double *d_out, double *d_rows_out,
                         int cus) {
    constexpr int FAN = 4 * FANR;
    StreamPlan plan{};
    int l = 0, total = 0;
    long cnt = count;
    while (cnt > 0) {  // count == 0: the input already is the plain triangle
        if (l >= 5) return FIGH_ERR_UNSUPPORTED;
        const long nb = (cnt + FAN - 1) / FAN;
        plan.nin[l] = (int)cnt;
        plan.nb[l] = (int)nb;
        plan.first[l] = total;
        total += (int)nb;
        ++l;
        if (nb == 1) break;
        cnt = nb;
    }
    plan.nlevels = l;
    plan.first[l] = total;
    const int grid = total + (d_rows_out ? 1 : 0);
    // every workgroup must be resident at once (consumers spin on their producers): one 512-thread workgroup per CU
    if (grid > cus || grid < 1 || count * nc * nc >= (1L << 31)) return FIGH_ERR_UNSUPPORTED;
    // the streamed triangles: poisoned when (re)allocated, left poisoned by every launch
    static double *sbuf = nullptr;
    static long sbuf_elems = 0;
    const long need = (long)(total > 0 ? total : 1) * nc * nc;
    if (need > sbuf_elems) {
        if (sbuf) {
            FIGH_HIP(hipStreamSynchronize(stream()));
            FIGH_HIP(hipFree(sbuf));
            sbuf = nullptr;
            sbuf_elems = 0;
        }
        const long want = need + need / 4 + 4096;
        FIGH_HIP(hipMalloc(&sbuf, sizeof(double) * want));
        sbuf_elems = want;
        hipLaunchKernelGGL(poison_fill_kernel, dim3(256), dim3(256), 0, stream(), sbuf, want);
        FIGH_HIP(hipGetLastError());
    }
    // Residency: consumers spin on their producers, so the whole grid must be co-resident.  The bound above (one workgroup
    // per CU) is checked against what the runtime says this instantiation can hold (registers, LDS); a refusal sends the
    // caller to the level-by-level reduction.  (A cooperative launch would make the runtime the guarantor -- it also covers
    // CU masks -- but costs this 0.11 ms launch 20 us, measured: 0.129 against 0.110 ms; the level-by-level fallback,
    // figh_tsqr_tree.hip, is launched that way.)
    static int occ = -1;
    if (occ < 0) {
        int v = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, tsqr_stream_kernel<NCC, FANR, 8>, 512, 0) != hipSuccess) v = 0;
        occ = v;
    }
    if (occ < 1) return FIGH_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((tsqr_stream_kernel<NCC, FANR, 8>), dim3((unsigned)grid), dim3(512), 0, stream(), Rs, sbuf, d_out,
                       d_rows_out, plan, nc, n_free, tol);
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}

int launch_tsqr_stream(const double *Rs, long count, int nc, int n_free, double tol, double *d_out, double *d_rows_out) {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    ProfileScope scope("tsqr_tree");
    if (nc <= 64) return launch_stream<4, 4>(Rs, count, nc, n_free, tol, d_out, d_rows_out, cus);
    if (nc <= 80) return launch_stream<5, 2>(Rs, count, nc, n_free, tol, d_out, d_rows_out, cus);
    return FIGH_ERR_UNSUPPORTED;
}

}  // namespace figh

#ifdef FIGH_ABLATION
extern "C" int figh_ab_stream_prof(long long *h_out, int reset) {
    long long zero[8] = {0};
    if (hipMemcpyFromSymbol(h_out, HIP_SYMBOL(figh::g_stream_prof), sizeof(zero)) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(figh::g_stream_prof), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
#endif
