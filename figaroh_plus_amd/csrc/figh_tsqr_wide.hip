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

#include "figh_internal.h"
#include "figh_wave.h"

namespace figh {

namespace {

constexpr int kLdv = 17;  // LDS row stride of V (doubles): the transposed reads of B -= V Wm hit 16 different banks

// sum over the finished reflectors m < KK of Trow[m] * vg(lane-column m): DPP lane selects must be immediates
template <int M0, int KK>
struct TColumn {
    static __device__ __forceinline__ double dot(const double (&Trow)[16], const double vg) {
        return fma(Trow[M0], row_bcast<M0>(vg), TColumn<M0 + 1, KK>::dot(Trow, vg));
    }
};
template <int KK>
struct TColumn<KK, KK> {
    static __device__ __forceinline__ double dot(const double (&)[16], const double) { return 0.0; }
};

// One column step of a panel.  X = the panel's chunk (lane (g, c): rows 16 rc + 4 r + g of column c, i = 4 rc + r),
// Rl = the 16 x 16 diagonal block in wave-private LDS (row-major), Trow = row c of T, myinv = 1 / (alpha - beta) of
// reflector c.  Columns c < KK are finished reflectors and stay frozen (they are V, up to the scaling by myinv).
template <int KK, int RPL>
__device__ __forceinline__ void wy_panel_step(double (&X)[RPL], double (&Trow)[16], double &myinv,
                                              double *__restrict__ Rl, double *__restrict__ red, const int lane,
                                              const int c) {
    double rk = Rl[KK * 16 + c];  // row KK of the diagonal block: requested before the dot products
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int i = 0; i < RPL; i += 4) {
        fmac_bcast<KK>(s0, X[i], X[i]);
        fmac_bcast<KK>(s1, X[i + 1], X[i + 1]);
        fmac_bcast<KK>(s2, X[i + 2], X[i + 2]);
        fmac_bcast<KK>(s3, X[i + 3], X[i + 3]);
    }
    const double d = allreduce_rowgroups_lds(red, lane, (s0 + s1) + (s2 + s3));  // x^T X[:, c], all row groups
    asm volatile("" : "+v"(rk));
    const double sigma = row_bcast<KK>(d);
    const double alpha = row_bcast<KK>(rk);
    if (__builtin_amdgcn_ballot_w64(sigma != 0.0) == 0) return;  // column zero below the triangle: H = I (dlarfg)
    // column KK of T, the part that does not depend on this step's scalars: sum_m T[c][m] (x_m^T x_KK) inv_m
    const double vg = (c < KK) ? d * myinv : 0.0;
    const double acc = TColumn<0, KK>::dot(Trow, vg);
    double inv, tfac;
    householder_scalars(alpha, sigma, inv, tfac);
    // w_c = tau (R_kc + v^T X_c) for c >= KK; the pivot lane gets w = alpha - beta, i.e. R_kk = alpha - w = beta
    const double wj = (c >= KK) ? (rk + d * inv) * tfac : 0.0;
    const double ncj = (c > KK) ? -wj * inv : 0.0;
#pragma unroll
    for (int i = 0; i < RPL; ++i) fmac_bcast<KK>(X[i], X[i], ncj);
    if (lane < 16 && c >= KK) Rl[KK * 16 + c] = rk - wj;
    Trow[KK] = (c < KK) ? -tfac * inv * acc : (c == KK ? tfac : 0.0);
    if (c == KK) myinv = inv;
    // the next step reads X through DPP operands of inline asm, which the hazard recognizer cannot see: nothing of it
    // may be scheduled in between this step's updates (a VALU write needs 2 wait states before a DPP read)
    __builtin_amdgcn_sched_barrier(0);
}

// Factor one panel: on return Rl holds the new diagonal block, Vl (M x kLdv) the reflectors V = X diag(inv), Tl (16 x 16,
// row-major) the T factor.
template <int RPL>
__device__ __forceinline__ void wy_factor_panel(double (&X)[RPL], double *__restrict__ Rl, double *__restrict__ red,
                                                double *__restrict__ Vl, double *__restrict__ Tl, const int lane,
                                                const int c, const int g) {
    double Trow[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) Trow[m] = 0.0;
    double myinv = 0.0;
    // X may still be in flight from the matrix pipe (the chunk update just before), and its first readers are DPP
    // operands of inline asm: the required wait states (MFMA write -> VALU read, VALU write -> DPP read) are not
    // inserted by the compiler for asm, so they are spelled out once per panel
#pragma unroll
    for (int i = 0; i < RPL; ++i) asm volatile("" : "+v"(X[i]));
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    wy_panel_step<0, RPL>(X, Trow, myinv, Rl, red, lane, c);
    wy_panel_step<1, RPL>(X, Trow, myinv, Rl, red, lane, c);
    wy_panel_step<2, RPL>(X, Trow, myinv, Rl, red, lane, c);
    wy_panel_step<3, RPL>(X, Trow, myinv, Rl, red, lane, c);
    wy_panel_step<4, RPL>(X, Trow, myinv, Rl, red, lane, c);
    wy_panel_step<5, RPL>(X, Trow, myinv, Rl, red, lane, c);
    wy_panel_step<6, RPL>(X, Trow, myinv, Rl, red, lane, c);
    wy_panel_step<7, RPL>(X, Trow, myinv, Rl, red, lane, c);
    wy_panel_step<8, RPL>(X, Trow, myinv, Rl, red, lane, c);
    wy_panel_step<9, RPL>(X, Trow, myinv, Rl, red, lane, c);
    wy_panel_step<10, RPL>(X, Trow, myinv, Rl, red, lane, c);
    wy_panel_step<11, RPL>(X, Trow, myinv, Rl, red, lane, c);
    wy_panel_step<12, RPL>(X, Trow, myinv, Rl, red, lane, c);
    wy_panel_step<13, RPL>(X, Trow, myinv, Rl, red, lane, c);
    wy_panel_step<14, RPL>(X, Trow, myinv, Rl, red, lane, c);
    wy_panel_step<15, RPL>(X, Trow, myinv, Rl, red, lane, c);
#pragma unroll
    for (int i = 0; i < RPL; ++i) Vl[(16 * (i >> 2) + 4 * (i & 3) + g) * kLdv + c] = X[i] * myinv;
    if (g == 0) {
#pragma unroll
        for (int m = 0; m < 16; ++m) Tl[c * 16 + m] = Trow[m];
    }
}

// Apply the panel's block reflector to one trailing chunk B (NRC row chunks of 16 x 16, C/D layout) and to its block of
// the triangle (Rblock: element lane + 64 r = row g + 4 r, column c).
template <int NRC>
__device__ __forceinline__ void wy_update_chunk(f64x4 (&B)[NRC], const double *__restrict__ Vl,
                                                const double *__restrict__ Tl, double *__restrict__ Rblock, const int lane,
                                                const int c, const int g) {
    f64x4 Rpt;
#pragma unroll
    for (int r = 0; r < 4; ++r) Rpt[r] = Rblock[lane + 64 * r];
    f64x4 G0 = {0.0, 0.0, 0.0, 0.0}, G1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int rc = 0; rc < NRC; ++rc) {
        // A[i = c][k = g] = V[row 16 rc + 4 r + g][c], B[k = g][j = c] = the tile entry of the same row: K-slice r
        G0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Vl[(16 * rc + 0 + g) * kLdv + c], B[rc][0], G0, 0, 0, 0);
        G1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Vl[(16 * rc + 4 + g) * kLdv + c], B[rc][1], G1, 0, 0, 0);
        G0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Vl[(16 * rc + 8 + g) * kLdv + c], B[rc][2], G0, 0, 0, 0);
        G1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Vl[(16 * rc + 12 + g) * kLdv + c], B[rc][3], G1, 0, 0, 0);
    }
    const f64x4 G = (G0 + G1) + Rpt;  // G[r] = row g + 4 r of R_p,cc + V^T B
    f64x4 Wm = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r)  // A[i = c][k = g + 4 r] = T[g + 4 r][c] (= T^T), B[k][j = c] = G[g + 4 r][c]
        Wm = __builtin_amdgcn_mfma_f64_16x16x4f64(Tl[(g + 4 * r) * 16 + c], G[r], Wm, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) Rblock[lane + 64 * r] = Rpt[r] - Wm[r];
    const f64x4 Wn = -Wm;
#pragma unroll
    for (int rc = 0; rc < NRC; ++rc)
#pragma unroll
        for (int s = 0; s < 4; ++s)  // A[i = c][k = g + 4 s] = V[row 16 rc + c][g + 4 s], B[k][j = c] = -Wm[g + 4 s][c]
            B[rc] = __builtin_amdgcn_mfma_f64_16x16x4f64(Vl[(16 * rc + c) * kLdv + g + 4 * s], Wn[s], B[rc], 0, 0, 0);
}

template <int NW, int CPW, int NRC>
__global__ __launch_bounds__(64 * NW, 2) void tsqr_wy_kernel(const double *__restrict__ W, const long rows,
                                                             const long ldw, const int *__restrict__ col_idx,
                                                             const int n, const double *__restrict__ tau,
                                                             const double *__restrict__ blkw, const long rows_per_blk,
                                                             double *__restrict__ Rblk, double *__restrict__ Rout,
                                                             const int nc) {
    static_assert((NW & (NW - 1)) == 0, "NW must be a power of two");
    constexpr int RPL = 4 * NRC, M = 16 * NRC, VBUF = M * kLdv + 256;
    __shared__ double vt[2][VBUF];      // ping-pong: V (M x kLdv) followed by T (16 x 16)
    __shared__ double rpp[NW][256];     // the diagonal block of the panel a wave is factoring (wave-private)
    __shared__ double redbuf[NW][64];   // cross-row-group sums (wave-private)
    __shared__ int fnz[2][NW];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int nch = (nc + 15) >> 4;
    double *Rb = Rblk + (long)blockIdx.x * ((long)nch * (nch + 1) / 2) * 256;
    auto block = [&](const int p, const int cc) { return Rb + ((long)cc * (cc + 1) / 2 + p) * 256; };
    double *Rl = rpp[wave];
    double *red = redbuf[wave];

    // this wave's columns of the triangle start empty
#pragma unroll
    for (int s = 0; s < CPW; ++s) {
        const int cc = wave + NW * s;
        if (cc < nch)
            for (int p = 0; p <= cc; ++p) {
                double *b = block(p, cc);
#pragma unroll
                for (int r = 0; r < 4; ++r) b[lane + 64 * r] = 0.0;
            }
    }

    // per-lane column sources.  Full tiles are read through a wave-uniform row base (SGPR pair) + a 32-bit per-lane
    // element offset g*ldw + column (the host side guarantees ldw < 2^24); tau is column n; everything beyond is a dead
    // lane-column whose registers stay exactly zero for the whole kernel (zero data, zero triangle entries).
    bool wlive[CPW], tlive[CPW];
    unsigned boff[CPW];  // BYTE offset of (row g, my column) from the row base: unsigned 32 bits = the saddr + voffset form
    int wcol[CPW];
#pragma unroll
    for (int s = 0; s < CPW; ++s) {
        const int col = 16 * (wave + NW * s) + c;
        wlive[s] = col < n;
        tlive[s] = col == n && tau != nullptr;
        wcol[s] = wlive[s] ? (col_idx ? col_idx[col] : col) : 0;
        boff[s] = 8u * ((unsigned)g * (unsigned)ldw + (unsigned)wcol[s]);
    }
    const unsigned toff = 8u * (unsigned)g;

    f64x4 T[CPW][NRC];
#pragma unroll
    for (int s = 0; s < CPW; ++s)
#pragma unroll
        for (int rc = 0; rc < NRC; ++rc) T[s][rc] = f64x4{0.0, 0.0, 0.0, 0.0};
    bool pf[CPW];  // chunk registers already hold the data of the coming tile
#pragma unroll
    for (int s = 0; s < CPW; ++s) pf[s] = false;

    // loads of one chunk of a FULL tile at row r0_: RPL independent requests per lane, dead lanes keep their zeros
#define FIGH_WY_LOAD(s, r0_)                                                                                      \
    do {                                                                                                          \
        if (wlive[s]) {                                                                                           \
            _Pragma("unroll") for (int i = 0; i < RPL; ++i)                                                       \
                T[s][i >> 2][i & 3] = *reinterpret_cast<const double *>(                                          \
                    reinterpret_cast<const char *>(W + ((r0_) + 16 * (i >> 2) + 4 * (i & 3)) * ldw) + boff[s]);   \
        }                                                                                                         \
        if (tlive[s]) {                                                                                           \
            _Pragma("unroll") for (int i = 0; i < RPL; ++i)                                                       \
                T[s][i >> 2][i & 3] = *reinterpret_cast<const double *>(                                          \
                    reinterpret_cast<const char *>(tau + (r0_) + 16 * (i >> 2) + 4 * (i & 3)) + toff);            \
        }                                                                                                         \
    } while (0)

    const long ntiles = (rows + M - 1) / M;
    int parity = 0;
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x, parity ^= 1) {
        const long r0 = t * M;
        const long r0n = (t + gridDim.x) * M;
        const bool next_full = r0n + M <= rows;  // only full tiles are prefetched
        if (r0 + M <= rows) {
#pragma unroll
            for (int s = 0; s < CPW; ++s)
                if (!pf[s]) FIGH_WY_LOAD(s, r0);
        } else {  // the last, ragged tile: rows clamped, then masked
#pragma unroll
            for (int s = 0; s < CPW; ++s)
#pragma unroll
                for (int i = 0; i < RPL; ++i) {
                    const long row = r0 + 16 * (i >> 2) + 4 * (i & 3) + g;
                    const long rowc = row < rows ? row : rows - 1;
                    double val = 0.0;
                    if (wlive[s]) val = W[rowc * ldw + wcol[s]];
                    if (tlive[s]) val = tau[rowc];
                    T[s][i >> 2][i & 3] = row < rows ? val : 0.0;
                }
        }
        if (blkw) {  // row-block weights (WLS): row r is scaled by blkw[r / rows_per_blk]
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const long row = r0 + 16 * (i >> 2) + 4 * (i & 3) + g;
                const double scale = blkw[(row < rows ? row : rows - 1) / rows_per_blk];
#pragma unroll
                for (int s = 0; s < CPW; ++s) T[s][i >> 2][i & 3] *= scale;
            }
        }
        // the first column with a non-zero in this tile: the column steps in front of it are identities (stacked
        // triangles in the merge levels, the joint-torque rows of a tree)
        int myfirst = 16 * nch;
#pragma unroll
        for (int s = CPW - 1; s >= 0; --s) {
            bool nz = false;
#pragma unroll
            for (int i = 0; i < RPL; ++i) nz |= (T[s][i >> 2][i & 3] != 0.0);
            const unsigned long long b = __ballot(nz);
            const unsigned m16 = (unsigned)((b | (b >> 16) | (b >> 32) | (b >> 48)) & 0xffffull);
            if (m16) myfirst = 16 * (wave + NW * s) + __ffs((int)m16) - 1;
            pf[s] = false;
        }
        if (lane == 0) fnz[parity][wave] = myfirst;
        __syncthreads();
        int first_nz = fnz[parity][0];
#pragma unroll
        for (int w = 1; w < NW; ++w) first_nz = min(first_nz, fnz[parity][w]);
        const int p0 = __builtin_amdgcn_readfirstlane(first_nz) >> 4;
        if (p0 >= nch) continue;  // the tile is zero

        // ---- the first panel of the tile has nobody to overlap with
        if (wave == (p0 & (NW - 1))) {
            const int so = p0 / NW;
            double X[RPL];
#pragma unroll
            for (int s = 0; s < CPW; ++s)
                if (s == so) {
#pragma unroll
                    for (int i = 0; i < RPL; ++i) X[i] = T[s][i >> 2][i & 3];
                }
            double *bpp = block(p0, p0);
#pragma unroll
            for (int r = 0; r < 4; ++r) Rl[lane + 64 * r] = bpp[lane + 64 * r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double *Vn = vt[p0 & 1];
            wy_factor_panel<RPL>(X, Rl, red, Vn, Vn + M * kLdv, lane, c, g);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int r = 0; r < 4; ++r) bpp[lane + 64 * r] = Rl[lane + 64 * r];
            if (next_full) {  // the chunk is retired: its registers take the coming tile's chunk
                long rn = r0n;
                asm volatile("" : "+s"(rn));  // opaque: the 16 row addresses must not be hoisted out of the phase loop
#pragma unroll
                for (int s = 0; s < CPW; ++s)
                    if (s == so) {
                        FIGH_WY_LOAD(s, rn);
                        pf[s] = true;
                    }
            }
        }
        __syncthreads();

        // ---- phase p: apply panel p to the trailing chunks; the owner of chunk p + 1 factors panel p + 1 meanwhile
        for (int p = p0; p + 1 < nch; ++p) {
            const double *Vl = vt[p & 1];
            const double *Tl = Vl + M * kLdv;
            const int pn = p + 1;
            if (wave == (pn & (NW - 1))) {
                const int sn = pn / NW;
                double *bpp = block(pn, pn);
                f64x4 rq;  // diagonal block of the coming panel: requested before the MFMA chain of the chunk update
#pragma unroll
                for (int r = 0; r < 4; ++r) rq[r] = bpp[lane + 64 * r];
                double X[RPL];
#pragma unroll
                for (int s = 0; s < CPW; ++s)
                    if (s == sn) {
                        wy_update_chunk<NRC>(T[s], Vl, Tl, block(p, pn), lane, c, g);
#pragma unroll
                        for (int i = 0; i < RPL; ++i) X[i] = T[s][i >> 2][i & 3];
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) Rl[lane + 64 * r] = rq[r];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                double *Vn = vt[pn & 1];
                wy_factor_panel<RPL>(X, Rl, red, Vn, Vn + M * kLdv, lane, c, g);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int r = 0; r < 4; ++r) bpp[lane + 64 * r] = Rl[lane + 64 * r];
                if (next_full) {
                    long rn = r0n;
                    asm volatile("" : "+s"(rn));  // (as above: keeps 32 VGPRs of loop-invariant addresses per chunk away)
#pragma unroll
                    for (int s = 0; s < CPW; ++s)
                        if (s == sn) {
                            FIGH_WY_LOAD(s, rn);
                            pf[s] = true;
                        }
                }
            }
#pragma unroll
            for (int s = 0; s < CPW; ++s) {
                const int cc = wave + NW * s;
                if (cc > pn && cc < nch) wy_update_chunk<NRC>(T[s], Vl, Tl, block(p, cc), lane, c, g);
                __builtin_amdgcn_sched_barrier(0);  // one chunk at a time: hoisting the next chunk's operands costs registers
            }
            __syncthreads();
        }
    }
#undef FIGH_WY_LOAD

    // ---- write this wave's columns of the nc x nc row-major triangle (zeros below the diagonal)
    double *Ro = Rout + (long)blockIdx.x * nc * nc;
#pragma unroll
    for (int s = 0; s < CPW; ++s) {
        const int cc = wave + NW * s;
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

template <int NW, int CPW, int NRC>
int wy_occupancy() {  // resident workgroups per CU (registers and LDS decide)
    static int nb = 0;
    if (!nb) {
        int v = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, tsqr_wy_kernel<NW, CPW, NRC>, 64 * NW, 0) != hipSuccess || v < 1)
            v = 1;
        nb = v;
    }
    return nb;
}

struct WyConfig {
    int nw, cpw, nrc;
};

// Geometry by column count.  Four waves per workgroup while a wave's share of the tile (CPW chunk slots of 32 registers)
// leaves room for the panel state in 256 registers -- two workgroups per CU then run independent panel chains that hide
// each other's latency; eight waves beyond.
WyConfig wy_config(const int nc) {
    const int nch = (nc + 15) >> 4;
#ifdef FIGH_ABLATION
    if (const char *e = getenv("FIGH_WY_CFG")) {  // "nw,cpw,nrc" (ablation build only)
        WyConfig cfg{0, 0, 0};
        if (sscanf(e, "%d,%d,%d", &cfg.nw, &cfg.cpw, &cfg.nrc) == 3 && cfg.nw * cfg.cpw >= nch) return cfg;
    }
#endif
    if (nch <= 8) return {4, 2, 4};
    if (nch <= 12) return {4, 3, 4};
    if (nch <= 16) return {8, 2, 4};
    if (nch <= 24) return {8, 3, 4};
    return {8, 4, 4};
}

template <class F>
bool wy_dispatch(const WyConfig cfg, F &&f) {
#define FIGH_WY_CASE(NW_, CPW_, NRC_)                                                              \
    if (cfg.nw == NW_ && cfg.cpw == CPW_ && cfg.nrc == NRC_) {                                     \
        f(std::integral_constant<int, NW_>{}, std::integral_constant<int, CPW_>{},                 \
          std::integral_constant<int, NRC_>{});                                                    \
        return true;                                                                               \
    }
    FIGH_WY_CASE(4, 2, 4)
    FIGH_WY_CASE(4, 3, 4)
    FIGH_WY_CASE(8, 2, 4)
    FIGH_WY_CASE(8, 3, 4)
    FIGH_WY_CASE(8, 4, 4)
#ifdef FIGH_ABLATION
    FIGH_WY_CASE(4, 4, 4)
    FIGH_WY_CASE(4, 4, 2)
    FIGH_WY_CASE(4, 6, 2)
    FIGH_WY_CASE(4, 8, 2)
    FIGH_WY_CASE(8, 3, 2)
#endif
#undef FIGH_WY_CASE
    return false;
}

}  // namespace

// persistent workgroups the wide kernel wants for nc columns (one private triangle each)
long tsqr_wide_workgroups(const int nc, const int cus) {
    int occ = 1;
    wy_dispatch(wy_config(nc), [&](auto NW, auto CPW, auto NRC) {
        occ = wy_occupancy<decltype(NW)::value, decltype(CPW)::value, decltype(NRC)::value>();
    });
    return (long)cus * occ;
}

// rows of (W, ldw) -> nwg triangles (nc x nc, row-major) in Rws_out; tiles are dealt round-robin to the workgroups
int launch_tsqr_wide(const double *W, long rows, long ldw, const int *col_idx, int n, const double *tau,
                     const double *d_blkw, long rows_per_blk, int nc, long nwg, double *Rws_out) {
    const int nch = (nc + 15) >> 4;
    const size_t blk_bytes = sizeof(double) * 256 * ((size_t)nch * (nch + 1) / 2) * (size_t)nwg;
    double *Rblk = static_cast<double *>(workspace(blk_bytes, 13));
    if (!Rblk) return FIGH_ERR_ALLOC;
    const bool ok = wy_dispatch(wy_config(nc), [&](auto NW, auto CPW, auto NRC) {
        hipLaunchKernelGGL((tsqr_wy_kernel<decltype(NW)::value, decltype(CPW)::value, decltype(NRC)::value>),
                           dim3((unsigned)nwg), dim3(64 * decltype(NW)::value), 0, stream(), W, rows, ldw, col_idx, n, tau,
                           d_blkw, rows_per_blk, Rblk, Rws_out, nc);
    });
    if (!ok) {
        set_error("figh_tsqr: no wide-kernel geometry for this column count");
        return FIGH_ERR_UNSUPPORTED;
    }
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}

}  // namespace figh
