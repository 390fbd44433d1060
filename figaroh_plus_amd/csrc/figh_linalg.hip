// K2 / K3 -- column norms, gathers, residual norms and the tall-skinny Householder QR on gfx950.
//
// figh_tsqr replaces the np.linalg.qr calls of src/figaroh/tools/qrdecomposition.py:105,205,238,286.  The
// reference only consumes R (|diag R| > tol rank test, R1 / R2 regrouping) and Q1^T tau, never Q itself
// (W_b = Q1 R1 is re-derived as the gathered base columns, qrdecomposition.py:268-269), so the kernels stream
// the rows of W once and keep only the n x n triangle: for each tile of rows the stacked [R; tile] is re-triangularised
// ("triangle on top of a rectangle", LAPACK tpqrt structure: 2*m*n^2 flops for m appended rows, no wasted work on the
// triangle).  Tiles whose leading columns are structurally zero (rows of joint j have zeros for links < j in the
// joint-torque layout) start at their first non-zero column.  The per-wave / per-workgroup triangles are then reduced
// by the same kernels over the stacked R factors.  Householder throughout: the rank decision |R_kk| > 1e-8 needs
// ~eps*||col|| accuracy on dependent pivots, which a Gram/Cholesky route cannot give.
//
// Two level-0 kernels:
//   nc <= 80   tsqr2_kernel (this file): one wavefront = one register tile + a private triangle in LDS, unblocked
//              wave-level steps on the fp64 VALU.  On gfx950 v_mfma_f64_16x16x4 and the fp64 VALU FMA have the SAME
//              peak (78.6 TFLOP/s) and share the FP64 datapath, so at n ~ 50 -- where a 16-wide panel is a third of
//              the work -- a blocked formulation cannot win (measured in round 1: 3.4 ms against 1.0 ms).
//   nc > 80    tsqr_wy_kernel (figh_tsqr_wide.hip): column-split workgroups, 16-column panels, compact-WY trailing
//              updates on the matrix pipe.
#include <cstdlib>
#include <algorithm>
#include <cstdio>
#include <type_traits>
#include <vector>

#include "figh_internal.h"
#include "figh_wave.h"
#include "figh_tsqr_narrow.h"
#include "figh_tsqr_narrow_kernel.h"

namespace figh {


template <int NCC, int NRC, bool LDSRED, bool RLAST = false>
__global__ __launch_bounds__(64, (NCC <= 4 || RLAST) ? 2 : 1) void tsqr2_kernel(
    const double *__restrict__ W, const long rows, const long ldw, const int *__restrict__ col_idx, const int n,
    const double *__restrict__ tau, const double *__restrict__ blkw, const long rows_per_blk,
    double *__restrict__ Rws, const int nc, const int *__restrict__ tile_first, const double null2) {
    tsqr2_level0_body<NCC, NRC, LDSRED, RLAST>(W, rows, ldw, col_idx, n, tau, blkw, rows_per_blk, Rws, nc, tile_first,
                                        (long)blockIdx.x, (long)gridDim.x, null2);
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

// dst[r][c] = scale * src[r][c] for a rows x cols block (strided 2-D copy): the np.concatenate / negation statements
// that assemble the total-least-squares regressor (regressor.py:316-412, :446-490).  A thread handles one element;
// consecutive threads run along a row.
__global__ __launch_bounds__(256) void place_block_kernel(const double *__restrict__ src, long lds, long rows, long cols,
                                                          double scale, double *__restrict__ dst, long ldd) {
    const long total = rows * cols;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long r = e / cols;
        const long c = e - r * cols;
        dst[r * ldd + c] = scale * src[r * lds + c];
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

// get_index_eliminate (regressor.py:258-279) on the device: kept = {c : not (colsq[c] < tol_e)} (NaN compares false and
// is kept, as in the reference's loop).  sel (int32): [0] number of kept columns, [1] ncols, [2 .. 2+ncols) the kept
// columns in W's own numbering (link_stride 16: the link-padded layout of figh_regressor_build_padded), zero-filled
// behind the kept ones (a consumer launched with a stale count reads valid columns), [2+ncols .. 2+2 ncols) the mask.
// tile_first (nullable): the structure hint of tsqr2_kernel -- per 64-row tile the number of kept columns in front of
// column 14 b of the row blocks b the tile touches (joint-torque layout: row block b only involves the links >= b).
// Every workgroup recomputes the 84..560-entry selection (cheaper than a second launch); workgroup 0 writes sel.
__global__ __launch_bounds__(256) void select_columns_kernel(const double *__restrict__ colsq, const int ncols,
                                                             const double tol_e, const int link_stride, const int nblocks,
                                                             const long rows, const long rows_per_block,
                                                             const long ntiles, int *__restrict__ sel,
                                                             int *__restrict__ tile_first,
                                                             const int *__restrict__ link_pos = nullptr) {
    __shared__ int kept[1024];
    __shared__ int first[kMaxJoints];
    const int tid = threadIdx.x;
    for (int c = tid; c < ncols; c += 256) kept[c] = !(colsq[c] < tol_e);
    __syncthreads();
    if (tile_first) {
        for (int b = tid; b < nblocks; b += 256) {
            const int lim = 14 * b < ncols ? 14 * b : ncols;
            int f = 0;
            for (int c = 0; c < lim; ++c) f += kept[c];
            first[b] = f;
        }
        __syncthreads();
        const long t = (long)blockIdx.x * 256 + tid;
        if (t < ntiles) {
            const long r0 = t * 64, r1 = (r0 + 63 < rows ? r0 + 63 : rows - 1);
            int f = first[r0 / rows_per_block];
            for (long b = r0 / rows_per_block + 1; b <= r1 / rows_per_block; ++b) f = min(f, first[b]);
            tile_first[t] = f;
        }
    }
    if (blockIdx.x != 0) return;
    int total = 0;
    for (int c = 0; c < ncols; ++c) total += kept[c];
    for (int c = tid; c < ncols; c += 256) {
        int pos = 0;
        for (int e = 0; e < c; ++e) pos += kept[e];
        // (link_pos: the link-compact layout of figh_regressor_build_padded + FIGH_FLAG_LINK_COMPACT -- a kept column's link
        // always has a segment there: columns of links without one have norm exactly 0)
        if (kept[c]) sel[2 + pos] = (link_pos ? max(link_pos[c / 14], 0) : c / 14) * link_stride + c % 14;
        if (c >= total) sel[2 + c] = 0;
        sel[2 + ncols + c] = kept[c];
    }
    if (tid == 0) {
        sel[0] = total;
        sel[1] = ncols;
    }
}

// External-wrench regressor of a free-flyer model: in the three FORCE row blocks the six rotational-inertia columns of
// every link are exact zeros (a force does not depend on the rotational inertia; the kernel writes J_a = 0 for those
// rows).  From the kept list (device column numbering, `n` entries) the columns that can be non-zero in force rows --
// slot >= 6 within a link: mx my mz m, Ia fv fs off -- and their positions in the kept list.  fsel: [n entries: columns]
// [n entries: positions]; the count is a pure function of the kept mask, the host derives the same number from it.
// force_compact: the columns are those of the force region of FIGH_FLAG_FORCE_COMPACT -- device column 16 p + s of the torque
// rows (p = the link's position, s >= 6) is column 16 (p / 4) + 4 (p % 4) + (s - 6) there.
__global__ __launch_bounds__(256) void split_force_columns_kernel(const int *__restrict__ kept, const int n,
                                                                  const int link_stride, int *__restrict__ fsel,
                                                                  const int force_compact) {
    __shared__ int flag[1024];
    const int tid = threadIdx.x;
    for (int c = tid; c < n; c += 256) flag[c] = (kept[c] % link_stride) >= 6;
    __syncthreads();
    for (int c = tid; c < n; c += 256) {
        if (!flag[c]) continue;
        int pos = 0;
        for (int e = 0; e < c; ++e) pos += flag[e];
        const int p = kept[c] / link_stride, s = kept[c] % link_stride;
        fsel[pos] = force_compact ? 16 * (p >> 2) + 4 * (p & 3) + (s - 6) : kept[c];
        fsel[n + pos] = c;
    }
}

// The force rows' triangle Rf (ncf x ncf over the force columns [+ tau]) as an nc x nc triangle over ALL kept columns
// [+ tau]: row r keeps its place, column c moves to the position of force column c in the kept list (tau stays last).
// Still upper triangular (the position of force column r is >= r); the other entries are zero.
__global__ __launch_bounds__(1024) void embed_force_triangle_kernel(const double *__restrict__ Rf, const int ncf,
                                                                    const int nf, const int *__restrict__ fpos,
                                                                    const int nc, const int n, double *__restrict__ out) {
    // one workgroup: zero fill, then the scatter
    for (int e = threadIdx.x; e < nc * nc; e += 1024) out[e] = 0.0;
    __syncthreads();
    for (int e = threadIdx.x; e < ncf * ncf; e += 1024) {
        const int r = e / ncf, c = e - r * ncf;
        if (c < r) continue;
        int col = c < nf ? fpos[c] : n;  // (c == nf: the tau column)
        // (a caller that guessed too many force columns -- the count is speculative, verified afterwards -- reads entries the
        // split kernel never wrote: zero-filled by the launcher, and in range whatever they hold)
        col = col < 0 ? 0 : (col >= nc ? nc - 1 : col);
        out[(long)r * nc + col] = Rf[e];
    }
}

int split_force_columns(const int *d_kept, int n, int link_stride, int *d_fsel, int force_compact) {
    // the kernel writes as many entries as there are kept columns with slot >= 6; a caller's (speculative) count may be
    // larger: what lies behind them must be valid column indices, not whatever the workspace held
    FIGH_HIP(hipMemsetAsync(d_fsel, 0, sizeof(int) * 2 * (size_t)n, stream()));
    hipLaunchKernelGGL(split_force_columns_kernel, dim3(1), dim3(256), 0, stream(), d_kept, n, link_stride, d_fsel, force_compact);
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}
int embed_force_triangle(const double *d_Rf, int ncf, int nf, const int *d_fpos, int nc, int n, double *d_out) {
    hipLaunchKernelGGL(embed_force_triangle_kernel, dim3(1), dim3(1024), 0, stream(), d_Rf, ncf, nf, d_fpos, nc, n, d_out);
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}

// Regrouped triangle -> the layout the regrouping phase of figh_tsqr_tree.hip produces: rows of qr([W1 W2 tau]) in the
// original column order under their base column, rows of dependent columns zero, then one more row with the diagonal of
// the plain factorisation.  Wide matrices only (nc > 80).
__global__ __launch_bounds__(256) void scatter_regrouped_kernel(const double *__restrict__ R, const double *__restrict__ Rr,
                                                                const int *__restrict__ perm, const int nc, const int n,
                                                                const double tol, double *__restrict__ out) {
    if (blockIdx.x == gridDim.x - 1) {
        // the last block: (tau, tau) = the residual || tau - W1 phi || = what lies below the base rows of the tau column,
        // summed by the whole block in a fixed order (one thread walking the two columns cost 90 us of dependent loads)
        if (n != nc - 1) return;
        __shared__ double part[256];
        __shared__ int cnt[256];
        int c = 0;
        for (int k = threadIdx.x; k < n; k += 256) c += fabs(R[(long)k * nc + k]) > tol;
        cnt[threadIdx.x] = c;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) cnt[threadIdx.x] += cnt[threadIdx.x + s];
            __syncthreads();
        }
        const int r = cnt[0];
        double ss = 0.0;
        for (int k = r + threadIdx.x; k < nc; k += 256) ss += Rr[(long)k * nc + n] * Rr[(long)k * nc + n];
        part[threadIdx.x] = ss;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[(long)perm[nc - 1] * nc + perm[nc - 1]] = sqrt(part[0]);
        return;
    }
    const int nb = gridDim.x - 1;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < nc * nc; e += nb * 256) {
        const int i = e / nc, j = e - i * nc;
        const int pi = perm[i], pj = perm[j];
        const bool base_row = pi >= n || fabs(R[(long)pi * nc + pi]) > tol;  // the tau column is always a pivot
        const double val = (base_row && j >= i) ? Rr[e] : 0.0;
        if (!(n == nc - 1 && e == nc * nc - 1)) out[(long)pi * nc + pj] = val;  // ((tau, tau): the last block's)
        if (i == 0) out[(long)nc * nc + j] = R[(long)j * nc + j];
    }
}

// Row rejection of the real-data scripts (examples/staubli_TX40/identification.py:207-233, examples/tiago/identification.py
// :170-187: rows whose joint velocity -- the fv column of the joint's own link -- is below a threshold are dropped from W
// and tau), as an order-preserving stream compaction.  Pass 1: kept rows per 64-row group; pass 2: exclusive scan of the
// group counts (one workgroup); pass 3: every wave copies the kept rows of its group to their final place, lanes across
// the columns.
__global__ __launch_bounds__(256) void compact_count_kernel(const double *__restrict__ W, const long rows, const long ldw,
                                                            const int key_col, const double thr, int *__restrict__ cnt) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    const bool keep = r < rows && fabs(W[r * ldw + key_col]) >= thr;
    const unsigned long long m = __ballot(keep);
    if ((threadIdx.x & 63) == 0) cnt[r >> 6] = __popcll(m);
}
__global__ __launch_bounds__(1024) void compact_scan_kernel(int *__restrict__ cnt, const long ngroups, long *__restrict__ total) {
    __shared__ long part[1024];
    const long per = (ngroups + 1023) / 1024;
    const long lo = threadIdx.x * per, hi = lo + per < ngroups ? lo + per : ngroups;
    long s = 0;
    for (long i = lo; i < hi; ++i) s += cnt[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        long run = 0;
        for (int i = 0; i < 1024; ++i) {
            const long v = part[i];
            part[i] = run;
            run += v;
        }
        *total = run;
    }
    __syncthreads();
    long run = part[threadIdx.x];
    for (long i = lo; i < hi; ++i) {
        const int v = cnt[i];
        cnt[i] = (int)run;  // (exclusive offsets; the host side bounds rows below 2^31)
        run += v;
    }
}
__global__ __launch_bounds__(256) void compact_copy_kernel(const double *__restrict__ W, const long rows, const int cols,
                                                           const long ldw, const double *__restrict__ tau, const int key_col,
                                                           const double thr, const int *__restrict__ off,
                                                           double *__restrict__ Wo, const long ldo, double *__restrict__ tauo) {
    const int lane = threadIdx.x & 63;
    const long grp = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const long r = grp * 64 + lane;
    if (grp * 64 >= rows) return;
    const bool keep = r < rows && fabs(W[r * ldw + key_col]) >= thr;
    unsigned long long m = __ballot(keep);
    long dst = off[grp];
    if (keep && tau) tauo[dst + __popcll(m & ((1ull << lane) - 1ull))] = tau[r];
    while (m) {  // wave-uniform loop over the kept rows of the group: lanes across the columns
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        const double *in = W + (grp * 64 + src) * ldw;
        double *out = Wo + dst * ldo;
        for (int c = lane; c < cols; c += 64) out[c] = in[c];
        ++dst;
    }
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


// LDS of one tsqr2 wave: 64 doubles of reduction scratch + the packed triangle minus the rows of the padding columns
static size_t tsqr2_lds_bytes(int ncc, int nc, bool rlast = false) {
    const int pad = 16 * ncc - nc, lch = rlast ? ncc - 1 : ncc;
    size_t skip = 0;
    for (int kp = 0; kp < pad; ++kp) skip += 16 * (lch - (kp >> 4) > 0 ? lch - (kp >> 4) : 0);
    return sizeof(double) * ((rlast ? 80 : 64) + 256 * (size_t)(lch * lch - (lch * (lch - 1)) / 2) - skip);
}

// per-tile structure hint of the register-tile kernel: g_tile_hint is installed by figh_tsqr_hint_begin for the level-0
// launch of the NEXT figh_tsqr_level0 call only (the merge levels run on stacked triangles and never see it)
static const int *g_tile_hint = nullptr;
// chained level-0 launches of a streamed run (blocked kernel only): workgroup count and chain flags of the next call
static long g_chain_wgs = 0;
static int g_chain_flags = 0;
void tsqr_level0_chain(long wgs, int chain_flags) {
    g_chain_wgs = wgs;
    g_chain_flags = chain_flags;
}

__global__ __launch_bounds__(256) void tile_hint_kernel(const int *__restrict__ first, const long hint_rows,
                                                        const long rows, const long ntiles, int *__restrict__ out) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= ntiles) return;
    const long r0 = t * 64, r1 = (r0 + 63 < rows ? r0 + 63 : rows - 1);
    int f = first[r0 / hint_rows];
    for (long b = r0 / hint_rows + 1; b <= r1 / hint_rows; ++b) f = min(f, first[b]);
    out[t] = f;
}

static const int *zero_tile_hint(long ntiles) {  // "no structure known": first possible non-zero = column 0 everywhere
    static size_t zeroed = 0;
    const size_t need = sizeof(int) * (size_t)(ntiles + 1);
    int *z = static_cast<int *>(workspace(need, 15));
    if (z && zeroed < need) {  // (re)allocated: workspace() hands out at least `need` bytes, growing by 25 %
        if (hipMemsetAsync(z, 0, need, stream()) != hipSuccess) return nullptr;
        zeroed = need;
    }
    return z;
}

// one TSQR level: the rows of (W, ldw) -> *nw_out triangles (compact nc x nc, row-major) in Rws_out; tiles of 64 rows
// are dealt round-robin to at most target_wgs wavefronts (nc <= 80) or workgroups (wide kernel).  hint: per-tile
// structure hint of the register-tile kernel, level 0 only (nullptr: none).
static int tsqr_level(const double *W, long rows, long ldw, const int *col_idx, int n, const double *tau,
                      const double *d_blkw, long rows_per_blk, int nc, long target_wgs, double *Rws_out, long *nw_out,
                      const int *hint, long chain_wgs = 0, int chain_flags = 0) {
    const long ntiles = (rows + 63) / 64;
    long nw = target_wgs < ntiles ? target_wgs : ntiles;
    if (nw < 1) nw = 1;
    if (nc > 80 && chain_wgs > 0) nw = chain_wgs;  // chained launches keep their workgroup count (idle ones pass through)
    *nw_out = nw;
    if (nc > 80)
        return launch_tsqr_wide(W, rows, ldw, col_idx, n, tau, d_blkw, rows_per_blk, nc, nw, Rws_out,
                                chain_wgs > 0 ? chain_flags : 0);
    // The register-tile kernel deals tile positions p = wave, wave + nw, ... and maps position p to the (p mod 8)-th
    // eighth of the row range (8-way interleave, see the kernel).  With nw a multiple of 8 a wave would stay inside one
    // eighth for its whole life -- in the joint-major row order that is all-heavy (joint 1) or all-light (joint 6) work:
    // measured 1.47 ms against 1.03 ms on the UR10 problem (nw = 2048 vs 2039).  One wave less keeps the residues moving.
    if (nw > 8 && nw % 8 == 0) nw -= 1;
    *nw_out = nw;
    const int *th = hint ? hint : zero_tile_hint(ntiles);
    if (!th) return FIGH_ERR_ALLOC;
    dim3 grid((unsigned)nw), block(64);
    bool tall48 = nc > 64 && !hint && rows >= 6L * nc * cu_count() * 8;
#ifdef FIGH_ABLATION
    if (const char *e = getenv("FIGH_T53")) tall48 = nc > 64 && !hint && atoi(e) != 0;
#endif
    if (nc <= 64)
        FIGH_LAUNCH_TIMED((tsqr2_kernel<4, 4, true>), grid, block, tsqr2_lds_bytes(4, nc), W, rows, ldw, col_idx, n, tau,
                          d_blkw, rows_per_blk, Rws_out, nc, th, null_pivot_sq());
    else if (!tall48)
        FIGH_LAUNCH_TIMED((tsqr2_kernel<5, 4, false>), grid, block, tsqr2_lds_bytes(5, nc), W, rows, ldw, col_idx, n, tau,
                          d_blkw, rows_per_blk, Rws_out, nc, th, null_pivot_sq());
    else {
        // 65 .. 80 columns without a structure hint: 48-row tiles and the last triangle chunk in registers -- 17 KB of LDS and
        // < 256 registers per wave, two waves per SIMD (the 64-row form is alone on its SIMD: latency-bound)
        const int *th3 = zero_tile_hint((rows + 47) / 48);
        if (!th3) return FIGH_ERR_ALLOC;
        FIGH_LAUNCH_TIMED((tsqr2_kernel<5, 3, false, true>), grid, block, tsqr2_lds_bytes(5, nc, true), W, rows, ldw, col_idx, n,
                          tau, d_blkw, rows_per_blk, Rws_out, nc, th3, null_pivot_sq());
    }
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}

// reduce `count` stacked compact nc x nc triangles (contiguous in Rs) to one in d_R_out.
//   nc <= 80: cooperative workgroups, 512 (256) stacked rows per sweep of nc column steps -- a merge level is
//             latency-bound, ~1 us per dependent column step;
//   nc  > 80: the stack is a tall matrix again (fan-in 4 per level through the wide kernel; tiles inside a triangle
//             start at their first non-zero column).
// all merge levels (+ rank decision and regrouping when d_rows_out) in one launch: the pipelined tree if it fits one
// resident grid, else the level-by-level tree; FIGH_ERR_UNSUPPORTED if neither does
static int merge_one_launch(const double *Rs, long count, int nc, int n_free, double tol, double *d_out, double *d_rows_out) {
    const int rc = launch_tsqr_stream(Rs, count, nc, n_free, tol, d_out, d_rows_out);
    if (rc != FIGH_ERR_UNSUPPORTED) return rc;
    return launch_tsqr_tree(Rs, count, nc, n_free, tol, d_out, d_rows_out);
}

static int tsqr_reduce(const double *Rs, long count, int nc, double *d_R_out) {
    const size_t tri = sizeof(double) * (size_t)nc * nc;
    if (nc <= 80 && count > 1) {  // every level in one launch (figh_tsqr_tree.hip) when the stack fits one resident grid
        const int rc = merge_one_launch(Rs, count, nc, 0, 0.0, d_R_out, nullptr);
        if (rc != FIGH_ERR_UNSUPPORTED) return rc;
    }
    const double *cur = Rs;
    long cnt = count;
    int slot = 2;
    while (cnt > 1) {
        ProfileScope scope("tsqr_reduce");
        long nb;
        double *dst;
        if (nc <= 80) {
            const long rows = cnt * nc;
            const int nwv = (rows > 256 && nc <= 64) ? 8 : 4;  // 5 column chunks per lane need > 256 registers: 4 waves
            nb = (rows + 64L * nwv - 1) / (64L * nwv);
            dst = nb == 1 ? d_R_out : static_cast<double *>(workspace(tri * nb, slot));
            if (!dst) return FIGH_ERR_ALLOC;
            if (nwv == 8)
                hipLaunchKernelGGL((tsqr_coop_kernel<4, 8>), dim3((unsigned)nb), dim3(512), 0, stream(), cur, rows, nc, dst);
            else if (nc > 64)
                hipLaunchKernelGGL((tsqr_coop_kernel<5, 4>), dim3((unsigned)nb), dim3(256), 0, stream(), cur, rows, nc, dst);
            else
                hipLaunchKernelGGL((tsqr_coop_kernel<4, 4>), dim3((unsigned)nb), dim3(256), 0, stream(), cur, rows, nc, dst);
            FIGH_HIP(hipGetLastError());
        } else {
            // pairs: a workgroup starts FROM its first triangle and absorbs the second (66 panel phases for TALOS'
            // 331 columns); a level costs (fan-in - 1) absorptions and there are log_fan(count) of them, so 2 is the
            // fan-in (round 2 walked four triangles into an empty one per level: 12.7 ms for 512 triangles)
            nb = (cnt + 1) / 2;
            dst = nb == 1 ? d_R_out : static_cast<double *>(workspace(tri * nb, slot));
            if (!dst) return FIGH_ERR_ALLOC;
            if (int rc = launch_tsqr_wide_pairs(cur, cnt, nc, dst)) return rc;
        }
        cur = dst;
        cnt = nb;
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

int figh_place_block(const double *d_src, int64_t ld_src, int64_t rows, int64_t cols, double scale, double *d_dst,
                     int64_t ld_dst) {
    FIGH_REQUIRE(d_src && d_dst, "NULL device pointer");
    FIGH_REQUIRE(rows >= 0 && cols >= 0 && ld_src >= cols && ld_dst >= cols, "bad shape");
    if (int rc = ensure_device()) return rc;
    if (rows == 0 || cols == 0) return FIGH_OK;
    long blocks = (rows * cols + 255) / 256;
    if (blocks > cu_count() * 16L) blocks = cu_count() * 16L;
    ProfileScope scope("place_block");
    hipLaunchKernelGGL(place_block_kernel, dim3((unsigned)blocks), dim3(256), 0, stream(), d_src, (long)ld_src, (long)rows,
                       (long)cols, scale, d_dst, (long)ld_dst);
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}

int figh_compact_rows(const double *d_W, int64_t rows, int cols, int64_t ldw, const double *d_tau, int key_col,
                      double threshold, double *d_W_out, int64_t ld_out, double *d_tau_out, int64_t *h_count_out) {
    FIGH_REQUIRE(d_W && d_W_out && h_count_out, "NULL pointer");
    FIGH_REQUIRE((d_tau == nullptr) == (d_tau_out == nullptr), "tau in and out come together");
    FIGH_REQUIRE(rows >= 0 && rows < (1L << 31) && cols >= 1 && ldw >= cols && ld_out >= cols && key_col >= 0 && key_col < cols,
                 "bad shape");
    if (int rc = ensure_device()) return rc;
    *h_count_out = 0;
    if (rows == 0) return FIGH_OK;
    const long ngroups = (rows + 63) / 64;
    int *cnt = static_cast<int *>(workspace(sizeof(int) * (size_t)ngroups + 16, 27));
    long *total = static_cast<long *>(workspace(sizeof(long), 28));
    if (!cnt || !total) return FIGH_ERR_ALLOC;
    ProfileScope scope("compact_rows");
    const unsigned grid = (unsigned)((rows + 255) / 256);
    hipLaunchKernelGGL(compact_count_kernel, dim3(grid), dim3(256), 0, stream(), d_W, (long)rows, (long)ldw, key_col, threshold,
                       cnt);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, stream(), cnt, ngroups, total);
    hipLaunchKernelGGL(compact_copy_kernel, dim3(grid), dim3(256), 0, stream(), d_W, (long)rows, cols, (long)ldw, d_tau, key_col,
                       threshold, cnt, d_W_out, (long)ld_out, d_tau_out);
    FIGH_HIP(hipGetLastError());
    long h = 0;
    FIGH_HIP(hipMemcpyAsync(&h, total, sizeof(long), hipMemcpyDeviceToHost, stream()));
    FIGH_HIP(hipStreamSynchronize(stream()));
    *h_count_out = h;
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
                     double **ws_out) {
    const long chain_wgs = g_chain_wgs;  // (consumed by this call whatever its outcome)
    const int chain_flags = g_chain_flags;
    g_chain_wgs = 0;
    g_chain_flags = 0;
    FIGH_REQUIRE(d_W && count_out, "NULL device pointer");
    FIGH_REQUIRE(rows > 0 && n > 0 && ldw > 0, "bad shape");
    const int nc = n + (d_tau ? 1 : 0);
    FIGH_REQUIRE(nc <= 512, "figh_tsqr: more than 512 columns not supported yet");
    FIGH_REQUIRE(ldw < (1L << 22), "figh_tsqr: leading dimension must be below 2^22 elements");
    // the blocked kernel addresses a tile of up to 96 rows through one buffer descriptor whose size field holds 2^31 - 1
    // bytes: 96 * ldw * 8 must stay below that, or rows of a full tile would be range-checked to zero
    FIGH_REQUIRE(nc <= 80 || ldw < (1L << 21), "figh_tsqr: more than 80 columns need a leading dimension below 2^21 elements");
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
    // persistent wavefronts / workgroups: as many as stay resident (register-tile kernel: 256 registers = two waves per
    // SIMD, and as many waves per CU as the LDS triangles admit; wide kernel: occupancy of its workgroups)
    long target;
    if (nc <= 80) {
        // (65 .. 80 columns without a structure hint: the 48-row form with its register chunk, two waves per SIMD)
        // -- when there are rows for eight waves per CU (same-box A/B in alternating order, tools/t53_ab.py: 3e6 x 77: 2.10-2.21
        // against 2.30-2.45 ms, 1e6 x 66: 0.74 against 0.85-0.92 ms, 1e6 x 80: equal; 5e5 x 80: 0.59-0.61 against 0.53-0.75)
        bool small_lds = nc > 64 && !g_tile_hint && rows >= 6L * nc * cu_count() * 8;
#ifdef FIGH_ABLATION
        if (const char *e = getenv("FIGH_T53")) small_lds = nc > 64 && !g_tile_hint && atoi(e) != 0;  // 1: always, 0: never
#endif
        long per_cu = (long)((160 * 1024) / tsqr2_lds_bytes(nc <= 64 ? 4 : 5, nc, small_lds));
        if (per_cu > ((nc <= 64 || small_lds) ? 8 : 4)) per_cu = (nc <= 64 || small_lds) ? 8 : 4;
        if (per_cu < 1) per_cu = 1;
        target = cu_count() * per_cu;
    } else {
        target = tsqr_wide_workgroups(nc, cu_count());
    }
    {   // a leaf must be much taller than wide, or the "reduction" produces more triangle rows than it consumed
        long cap = rows / (8L * nc);
        if (cap < 1) cap = 1;
        if (target > cap) target = cap;
    }
    const long nw_est = (nc > 80 && chain_wgs > target ? chain_wgs : target) + 1;
    double *Rws = d_tri_out;
    if (Rws) {
        FIGH_REQUIRE(capacity >= nw_est, "figh_tsqr_level0: triangle buffer too small");
    } else {
        Rws = static_cast<double *>(workspace(sizeof(double) * (size_t)nc * nc * nw_est, 5));
        if (!Rws) return FIGH_ERR_ALLOC;
    }
    const int *hint = g_tile_hint;  // consumed by this launch only
    g_tile_hint = nullptr;
    long nw = 0;
    {
        ProfileScope scope(rows >= 65536 ? "tsqr" : "tsqr_small", true);
        if (int rc = tsqr_level(d_W, rows, ldw, d_col_idx, n, d_tau, d_blkw, rows_per_blk, nc, target, Rws, &nw, hint,
                                chain_wgs, chain_flags))
            return rc;
    }
    *count_out = nw;
    if (ws_out) *ws_out = Rws;
    return FIGH_OK;
}

// upper bound of the triangles one figh_tsqr_level0 call can produce (for sizing the stack of a streamed run)
int64_t figh_tsqr_level0_capacity(int nc) {
    if (nc > 80) return (int64_t)tsqr_wide_workgroups(nc, cu_count()) + 1;
    return (int64_t)cu_count() * 8 + 1;
}

int figh_tsqr(const double *d_W, int64_t rows, int64_t ldw, const int32_t *d_col_idx, int n, const double *d_tau,
              const double *h_block_weight, int nblocks, double *d_R_out) {
    FIGH_REQUIRE(d_W && d_R_out, "NULL device pointer");
    int64_t nw = 0;
    double *Rws = nullptr;
    if (int rc = figh_tsqr_level0(d_W, rows, ldw, d_col_idx, n, d_tau, h_block_weight, nblocks, nullptr, 0, &nw, &Rws))
        return rc;
    const int nc = n + (d_tau ? 1 : 0);
    if (nw == 1) {
        FIGH_HIP(hipMemcpyAsync(d_R_out, Rws, sizeof(double) * (size_t)nc * nc, hipMemcpyDeviceToDevice, stream()));
        return FIGH_OK;
    }
    return tsqr_reduce(Rws, nw, nc, d_R_out);
}


// internal (figh_internal.h): install / remove the structure hint for the next level-0 launches on `rows` rows
int figh_tsqr_hint_begin(const int32_t *h_first_col, int nfirst, int64_t rows, int n, int nc) {
    FIGH_REQUIRE(h_first_col && nfirst > 0 && rows > 0 && rows % nfirst == 0, "rows must be a multiple of the hint blocks");
    for (int b = 0; b < nfirst; ++b) FIGH_REQUIRE(h_first_col[b] >= 0 && h_first_col[b] <= n, "first column out of range");
    if (int rc = ensure_device()) return rc;
    g_tile_hint = nullptr;
    if (nc > 80) return FIGH_OK;  // only the register-tile kernel uses the hint
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


int figh_select_columns(const double *d_colsq, int ncols, double tol_e, int link_stride, int32_t *d_sel) {
    FIGH_REQUIRE(d_colsq && d_sel, "NULL device pointer");
    FIGH_REQUIRE(ncols >= 1 && ncols <= 1024, "figh_select_columns: 1 .. 1024 columns");
    FIGH_REQUIRE(link_stride == 14 || link_stride == 16, "link_stride must be 14 (reference layout) or 16 (link-padded)");
    if (int rc = ensure_device()) return rc;
    ProfileScope scope("select_columns");
    hipLaunchKernelGGL(select_columns_kernel, dim3(1), dim3(256), 0, stream(), d_colsq, ncols, tol_e, link_stride, 0, 0L,
                       1L, 0L, d_sel, (int *)nullptr, (const int *)nullptr);
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}

// plain triangle -> rows of the regrouped factorisation in the original column order + the plain diagonal (see figh.h)
static int reveal_triangle(const double *d_R, int nc, int n_free, double tol_qr, double *d_rows_out) {
    if (nc <= 80) return merge_one_launch(d_R, 0, nc, n_free, tol_qr, nullptr, d_rows_out);
    int *perm = static_cast<int *>(workspace(sizeof(int) * (size_t)nc, 17));
    double *Rr = static_cast<double *>(workspace(sizeof(double) * (size_t)nc * nc, 18));
    if (!perm || !Rr) return FIGH_ERR_ALLOC;
    if (int rc = figh_base_permutation(d_R, nc, n_free, tol_qr, perm)) return rc;
    {
        ProfileScope scope("tsqr_regroup");
        if (int rc = launch_tsqr_wide_single(d_R, nc, nc, perm, nc, nc, Rr)) return rc;
    }
    ProfileScope scope("scatter_regrouped");
    hipLaunchKernelGGL(scatter_regrouped_kernel, dim3(65), dim3(256), 0, stream(), d_R, Rr, perm, nc, n_free, tol_qr,
                       d_rows_out);
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}

}  // extern "C"

// stack of `count` triangles -> one; tol_qr >= 0: + rank decision and regrouped rows ((nc+1) x nc), else the plain triangle
int figh::tsqr_reduce_stack(const double *d_Rs, long count, int nc, int n_free, double tol_qr, double *d_out) {
    if (tol_qr < 0.0) return figh_tsqr_merge(d_Rs, (int)count, nc, d_out);
    double *one = static_cast<double *>(workspace(sizeof(double) * (size_t)nc * nc, 16));
    if (!one) return FIGH_ERR_ALLOC;
    if (count == 1) {
        if (nc <= 80) return reveal_triangle(d_Rs, nc, n_free, tol_qr, d_out);  // (one launch, touches no workspace)
        // the wide regrouping runs figh_tsqr again, whose level 0 reuses the workspace d_Rs may live in
        FIGH_HIP(hipMemcpyAsync(one, d_Rs, sizeof(double) * (size_t)nc * nc, hipMemcpyDeviceToDevice, stream()));
        return reveal_triangle(one, nc, n_free, tol_qr, d_out);
    }
    if (nc <= 80) {
        const int rc = merge_one_launch(d_Rs, count, nc, n_free, tol_qr, one, d_out);
        if (rc != FIGH_ERR_UNSUPPORTED) return rc;
    }
    if (int rc = tsqr_reduce(d_Rs, count, nc, one)) return rc;
    return reveal_triangle(one, nc, n_free, tol_qr, d_out);
}

extern "C" {

static int tsqr_selected_impl(const double *d_W, int64_t rows, int64_t ldw, const double *d_colsq, int ncols, double tol_e,
                              int link_stride, int nblocks, int n_expected, const double *d_tau, double tol_qr, int32_t *d_sel,
                              double *d_R_out, const int32_t *d_link_pos);
int figh_tsqr_selected(const double *d_W, int64_t rows, int64_t ldw, const double *d_colsq, int ncols, double tol_e,
                       int link_stride, int nblocks, int n_expected, const double *d_tau, double tol_qr, int32_t *d_sel,
                       double *d_R_out) {
    return tsqr_selected_impl(d_W, rows, ldw, d_colsq, ncols, tol_e, link_stride, nblocks, n_expected, d_tau, tol_qr, d_sel,
                              d_R_out, nullptr);
}
static int tsqr_selected_impl(const double *d_W, int64_t rows, int64_t ldw, const double *d_colsq, int ncols, double tol_e,
                              int link_stride, int nblocks, int n_expected, const double *d_tau, double tol_qr, int32_t *d_sel,
                              double *d_R_out, const int32_t *d_link_pos) {
    FIGH_REQUIRE(d_W && d_colsq && d_sel, "NULL device pointer");
    FIGH_REQUIRE(ncols >= 1 && ncols <= 1024, "figh_tsqr_selected: 1 .. 1024 columns");
    FIGH_REQUIRE(link_stride == 14 || link_stride == 16, "link_stride must be 14 (reference layout) or 16 (link-padded)");
    FIGH_REQUIRE(rows > 0 && n_expected <= ncols, "bad shape");
    FIGH_REQUIRE(nblocks >= 0 && nblocks <= kMaxJoints && (nblocks == 0 || rows % nblocks == 0),
                 "rows must be a multiple of the hint blocks");
    if (int rc = ensure_device()) return rc;
    const int nc = n_expected + (d_tau ? 1 : 0);
    const bool hinted = nblocks > 0 && n_expected > 0 && nc <= 80 && rows / nblocks >= 1;
    const long ntiles = (rows + 63) / 64;
    int *d_tile = nullptr;
    if (hinted) {
        d_tile = static_cast<int *>(workspace(sizeof(int) * (size_t)(ntiles + 1), 19));
        if (!d_tile) return FIGH_ERR_ALLOC;
    }
    {
        ProfileScope scope("select_columns");
        const long grid = hinted ? (ntiles + 255) / 256 : 1;
        hipLaunchKernelGGL(select_columns_kernel, dim3((unsigned)grid), dim3(256), 0, stream(), d_colsq, ncols, tol_e,
                           link_stride, hinted ? nblocks : 0, (long)rows, hinted ? (long)(rows / nblocks) : 1L,
                           hinted ? ntiles : 0L, d_sel, d_tile, (const int *)d_link_pos);
        FIGH_HIP(hipGetLastError());
    }
    if (n_expected <= 0) return FIGH_OK;  // selection only (the caller does not know the count yet)
    FIGH_REQUIRE(d_R_out, "NULL device pointer");
    g_tile_hint = d_tile;  // consumed by the level-0 launch below
    int64_t nw = 0;
    double *Rws = nullptr;
    const int rc0 = figh_tsqr_level0(d_W, rows, ldw, d_sel + 2, n_expected, d_tau, nullptr, 0, nullptr, 0, &nw, &Rws);
    g_tile_hint = nullptr;
    if (rc0) return rc0;
    return tsqr_reduce_stack(Rws, nw, nc, n_expected, tol_qr, d_R_out);
}

// figh_tsqr_selected for the external-wrench regressor of a free-flyer model (six row blocks of rows / 6 rows: three force
// components, three torque components): the force rows are factored over the `nf_expected` kept columns that can be
// non-zero there (split_force_columns_kernel) -- 2 m nf^2 instead of 2 m n^2 flops for half of the rows -- their triangle
// is embedded into the full column set and joins the stack of the torque rows' triangles: R is the R factor of the whole
// W, at 0.5 (1 + (nf / n)^2) of the flops (TALOS, human: 58 %).
int figh_tsqr_selected_wrench(const double *d_W, int64_t rows, int64_t ldw, const double *d_colsq, int ncols, double tol_e,
                              int link_stride, int n_expected, int nf_expected, const double *d_tau, double tol_qr,
                              int32_t *d_sel, double *d_R_out, const int32_t *d_link_pos, int64_t ld_force) {
    const int nc = n_expected + (d_tau ? 1 : 0);
    FIGH_REQUIRE(!d_link_pos || link_stride == 16, "a link map goes with the link-padded layout");
    FIGH_REQUIRE(ld_force >= 0 && (ld_force == 0 || (link_stride == 16 && ld_force % 16 == 0 && rows % 6 == 0)),
                 "force-compact W: link-padded torque rows, a multiple of 16 force columns, six row blocks");
    // no split: unknown counts, the register-tile kernel's column range (no chained form), nothing to gain, odd shapes
    if (n_expected <= 0 || nf_expected <= 0 || nf_expected >= n_expected || nc <= 80 || rows % 6 != 0 ||
        rows / 2 < 16L * nc) {
        if (ld_force > 0) {
            // (the plain pass reads every row over one column list: not what a force-compact W offers.  Selection only --
            // the caller learns the counts and comes back with them -- or refuse)
            if (n_expected <= 0)
                return tsqr_selected_impl(d_W, rows, ldw, d_colsq, ncols, tol_e, link_stride, 0, n_expected, d_tau, tol_qr,
                                          d_sel, d_R_out, d_link_pos);
            set_error("force-compact W needs the force / torque split: more than 80 kept columns, some of them inertia columns, "
                      "at least 32 x columns rows");
            return FIGH_ERR_UNSUPPORTED;
        }
        return tsqr_selected_impl(d_W, rows, ldw, d_colsq, ncols, tol_e, link_stride, 0, n_expected, d_tau, tol_qr, d_sel,
                                  d_R_out, d_link_pos);
    }
    FIGH_REQUIRE(d_W && d_colsq && d_sel && d_R_out, "NULL device pointer");
    FIGH_REQUIRE(ncols >= 1 && ncols <= 1024, "figh_tsqr_selected: 1 .. 1024 columns");
    FIGH_REQUIRE(link_stride == 14 || link_stride == 16, "link_stride must be 14 (reference layout) or 16 (link-padded)");
    FIGH_REQUIRE(n_expected <= ncols, "bad shape");
    if (int rc = ensure_device()) return rc;
    {
        ProfileScope scope("select_columns");
        hipLaunchKernelGGL(select_columns_kernel, dim3(1), dim3(256), 0, stream(), d_colsq, ncols, tol_e, link_stride, 0,
                           (long)rows, 1L, 0L, d_sel, (int *)nullptr, (const int *)d_link_pos);
        FIGH_HIP(hipGetLastError());
    }
    const int n = n_expected, nf = nf_expected, ncf = nf + (d_tau ? 1 : 0);
    int *fsel = static_cast<int *>(workspace(sizeof(int) * 2 * (size_t)n, 22));
    if (!fsel) return FIGH_ERR_ALLOC;
    if (int rc = split_force_columns(d_sel + 2, n, link_stride, fsel, ld_force > 0 ? 1 : 0)) return rc;  // (zero-fills behind the count)
    const int64_t rows_f = rows / 2;
    const int64_t ldf = ld_force > 0 ? ld_force : ldw;  // force-compact: the force rows are a matrix of their own
    // ---- force rows: their own TSQR over nf columns, reduced to one triangle
    const int64_t cap_f = figh_tsqr_level0_capacity(ncf);
    double *tri_f = static_cast<double *>(workspace(sizeof(double) * (size_t)ncf * ncf * cap_f, 23));
    double *Rf = static_cast<double *>(workspace(sizeof(double) * (size_t)ncf * ncf, 24));
    if (!tri_f || !Rf) return FIGH_ERR_ALLOC;
    int64_t cnt_f = 0;
    if (int rc = figh_tsqr_level0(d_W, rows_f, ldf, fsel, nf, d_tau, nullptr, 0, tri_f, cap_f, &cnt_f, nullptr)) return rc;
    if (cnt_f == 1) FIGH_HIP(hipMemcpyAsync(Rf, tri_f, sizeof(double) * (size_t)ncf * ncf, hipMemcpyDeviceToDevice, stream()));
    else if (int rc = tsqr_reduce(tri_f, cnt_f, ncf, Rf)) return rc;
    // ---- torque rows: chained launch, workgroup 0 starts from the embedded force triangle, the others from zeros
    long wgs = tsqr_wide_workgroups(nc, cu_count());
    {
        long cap = (rows - rows_f) / (8L * nc);
        if (cap < 1) cap = 1;
        if (wgs > cap) wgs = cap;
    }
    // (the embedded triangle joins the stack as one more element.  Chaining the torque launch onto it -- workgroup 0 starts
    // from it, no extra element -- was measured first: the CHAIN instantiation of the TALOS geometry carries 180 bytes of
    // scratch against the plain one's 68 and ran the torque rows in 105.6 ms)
    const size_t tri = sizeof(double) * (size_t)nc * nc;
    double *stack = static_cast<double *>(workspace(tri * (size_t)(wgs + 3), 5));
    if (!stack) return FIGH_ERR_ALLOC;
    int64_t cnt = 0;
    if (wgs > 64) tsqr_level0_chain(wgs - 1, 0);  // one workgroup less: with the embedded triangle the stack is 2^k again
    if (int rc = figh_tsqr_level0(d_W + rows_f * ldf, rows - rows_f, ldw, d_sel + 2, n, d_tau ? d_tau + rows_f : nullptr,
                                  nullptr, 0, stack, wgs + 2, &cnt, nullptr))
        return rc;
    hipLaunchKernelGGL(embed_force_triangle_kernel, dim3(1), dim3(1024), 0, stream(), Rf, ncf, nf, fsel + n, nc, n,
                       stack + (size_t)cnt * nc * nc);
    FIGH_HIP(hipGetLastError());
    return tsqr_reduce_stack(stack, cnt + 1, nc, n, tol_qr, d_R_out);
}

// figh_tsqr_selected for the joint-torque regressor of a TREE of single-dof joints (regressor.py:45-87): row block j (the
// rows of joint j) only involves the links of j's subtree (+ the Ia fv fs off columns of link j itself) -- every other
// column is a structural zero the tape kernel writes as such.  For a chain that is the prefix structure the register-tile
// kernel's hint exploits; for a tree (TIAGo: wheels, head, arm on one base) most of the zeros are NOT a prefix: 7 .. 85 of
// the 240 kept columns are non-zero per row block.  Every row block is factored over its own column list (h_counts[j]
// entries of d_cols / d_pos: device columns and positions in the kept list, concatenated; built by the caller from the
// kept mask it expects and verified against d_sel afterwards), reduced to one triangle, embedded into the full column set,
// and the nblocks embedded triangles are merged: 2 m sum_j n_j^2 / nblocks instead of 2 m n^2 flops (TIAGo: 2.4 %).
int figh_tsqr_selected_blocks(const double *d_W, int64_t rows, int64_t ldw, const double *d_colsq, int ncols, double tol_e,
                              int link_stride, int n_expected, int nblocks, const int32_t *h_counts, const int32_t *d_cols,
                              const int32_t *d_pos, const int64_t *h_block_off, const int32_t *h_block_ld,
                              const double *d_tau, double tol_qr, int32_t *d_sel, double *d_R_out, double *d_block_tri) {
    FIGH_REQUIRE(d_W && d_colsq && d_sel && d_R_out && h_counts && d_cols && d_pos, "NULL pointer");
    FIGH_REQUIRE((h_block_off == nullptr) == (h_block_ld == nullptr), "block offsets and leading dimensions come together");
    FIGH_REQUIRE(ncols >= 1 && ncols <= 1024, "figh_tsqr_selected: 1 .. 1024 columns");
    FIGH_REQUIRE(link_stride == 14 || link_stride == 16, "link_stride must be 14 (reference layout) or 16 (link-padded)");
    FIGH_REQUIRE(n_expected >= 1 && n_expected <= ncols && n_expected < 512, "bad shape");
    FIGH_REQUIRE(nblocks >= 1 && nblocks <= kMaxJoints && rows % nblocks == 0, "rows must be a multiple of the row blocks");
    if (int rc = ensure_device()) return rc;
    {
        ProfileScope scope("select_columns");
        hipLaunchKernelGGL(select_columns_kernel, dim3(1), dim3(256), 0, stream(), d_colsq, ncols, tol_e, link_stride, 0,
                           (long)rows, 1L, 0L, d_sel, (int *)nullptr, (const int *)nullptr);
        FIGH_HIP(hipGetLastError());
    }
    const int n = n_expected, nc = n + (d_tau ? 1 : 0);
    const int64_t rows_b = rows / nblocks;
    int nmax = 1;
    for (int j = 0; j < nblocks; ++j) {
        // (-1: an INACTIVE row block -- neither its rows of W nor of tau take part; figh_model_set_active_rows)
        FIGH_REQUIRE(h_counts[j] >= -1 && h_counts[j] <= n, "block column count out of range");
        if (h_counts[j] >= 0) nmax = std::max(nmax, h_counts[j] + (d_tau ? 1 : 0));
    }
    const size_t tri = sizeof(double) * (size_t)nc * nc;
    // The per-row-block triangles, embedded into the kept column set, are stacked COMPACTLY: block j contributes its
    // n_j (+ 1 with tau) rows only -- TIAGo: 650 rows instead of 24 x 241 -- and the stack is one small tall matrix that a
    // single workgroup factors (one launch; the 24 full-size triangles took five pair-merge levels, 3.4 ms of the 17 ms step).
    // It lives in the caller's buffer when the caller wants to keep it (weighted solve afterwards: figh_block_rows_residuals
    // + figh_tsqr over the rows with per-row weights), else in a library workspace.
    long rows_total = 0;
    for (int j = 0; j < nblocks; ++j)
        if (h_counts[j] > 0 || (d_tau && h_counts[j] == 0)) rows_total += h_counts[j] + (d_tau ? 1 : 0);
    double *stack = d_block_tri ? d_block_tri
                                : static_cast<double *>(workspace(sizeof(double) * (size_t)(rows_total + nc + 1) * nc, 26));
    const int64_t cap_b = std::max(figh_tsqr_level0_capacity(nmax), figh_tsqr_level0_capacity(std::min(nmax, 80)));
    double *tri_b = static_cast<double *>(workspace(sizeof(double) * (size_t)nmax * nmax * cap_b, 23));
    // one triangle per row block (embedded at the end, in block order: an embedding zero-fills nc rows from its offset)
    double *Rb_all = static_cast<double *>(workspace(sizeof(double) * (size_t)nmax * nmax * nblocks, 24));
    double *one = static_cast<double *>(workspace(tri, 16));
    if (!stack || !tri_b || !Rb_all || !one) return FIGH_ERR_ALLOC;
    // the WIDE blocks (more than 80 columns: the blocked kernel, one triangle per workgroup) keep their level-0 triangles
    // until all of them are there and are then reduced together, level by level (reduce_wide_stacks: one launch per level for
    // all of them once a level fits the chip -- the levels are latency-bound, 65 us each whatever the number of pairs)
    size_t wide_doubles = 0;
    for (int j = 0; j < nblocks; ++j) {
        const int ncj = h_counts[j] + (d_tau ? 1 : 0);
        if (h_counts[j] > 0 && ncj > 80) wide_doubles += (size_t)ncj * ncj * (size_t)figh_tsqr_level0_capacity(ncj);
    }
    double *wide_tri = wide_doubles ? static_cast<double *>(workspace(sizeof(double) * wide_doubles, 36)) : nullptr;
    if (wide_doubles && !wide_tri) return FIGH_ERR_ALLOC;
    size_t wide_at = 0;
    std::vector<WyPairStack> wide;
    struct Embed {
        int block;
        const double *R;
        int ncj, nj;
        const int *pos;
        double *out;
    };
    std::vector<Embed> embeds;
    // The narrow blocks (at most 64 columns with tau: the register-tile kernel's range) of a problem that has several of
    // them go through GROUPED launches (figh_tsqr_group.hip: one level-0 launch, two merge launches, one embedding launch
    // for all of them) after the others -- their embedding writes exactly their own rows of the stack, the per-block
    // embedding below zero-fills nc rows from its offset.
    std::vector<Tsqr2Job> jobs;
    int narrow = 0;
    for (int j = 0; j < nblocks; ++j) narrow += (h_counts[j] >= 1 && h_counts[j] + (d_tau ? 1 : 0) <= 64) ? 1 : 0;
    const bool grouped = narrow >= 4 && rows_b >= 64 * 64;
    // With a grouped launch in the pass, the merge levels -- latency-bound chains of small launches, 2.4 ms of TIAGo's 12.3 ms
    // step when they ran one after the other behind all the level-0 launches -- are spread over two streams: the level-0
    // launches of the WIDE blocks go first, and their merge levels (reduce_wide_stacks) run on the high-priority side stream
    // from then on, beside the library stream's level-0 launches of the mid blocks (65 .. 80 columns) and of the group (little
    // of them gets in while those fill the chip: a 512-thread workgroup does not find its eight wave slots between one-wave
    // workgroups) and beside the mid blocks' and the group's own merge levels behind them.  A mid block keeps its level-0
    // triangles in its own region instead of the shared tri_b.  All embeddings come last, in block order (a per-block embedding
    // zero-fills nc rows from its offset, the grouped one writes exactly the group's rows).
    // (Also without a group: the side stream's merges then run beside the other blocks' level 0.)
    struct MidStack {
        double *tri;
        long cnt;
        int ncj;
        double *R;
    };
    std::vector<MidStack> mids;
    size_t mid_doubles = 0, mid_at = 0;
    for (int j = 0; j < nblocks; ++j) {
        const int ncj = h_counts[j] + (d_tau ? 1 : 0);
        if (h_counts[j] > 0 && ncj > 64 && ncj <= 80) mid_doubles += (size_t)ncj * ncj * (size_t)figh_tsqr_level0_capacity(ncj);
    }
    double *mid_tri = mid_doubles ? static_cast<double *>(workspace(sizeof(double) * mid_doubles, 38)) : nullptr;
    if (mid_doubles && !mid_tri) return FIGH_ERR_ALLOC;
    // what every block is and where its rows of the stack / entries of the column lists start
    enum { SKIP, JOB, WIDE, MID, PLAIN };
    int kind[kMaxJoints];
    long row_at[kMaxJoints], off_at[kMaxJoints];
    long row_off = 0, off = 0;
    for (int j = 0; j < nblocks; ++j) {
        const int nj = h_counts[j], ncj = nj + (d_tau ? 1 : 0);
        row_at[j] = row_off;
        off_at[j] = off;
        if (nj < 0 || (nj == 0 && !d_tau)) {  // (an inactive row block / nothing of this block is kept)
            kind[j] = SKIP;
            continue;
        }
        kind[j] = (grouped && nj >= 1 && ncj <= 64) ? JOB : (nj > 0 && ncj > 80) ? WIDE : (nj > 0 && ncj > 64) ? MID : PLAIN;
        row_off += ncj;
        off += nj;
    }
    hipEvent_t wide_done = nullptr;
    for (int phase = 0; phase < 2; ++phase) {  // the wide blocks first
        for (int j = 0; j < nblocks; ++j) {
            if (kind[j] == SKIP || (kind[j] == WIDE) != (phase == 0)) continue;
            const int nj = h_counts[j], ncj = nj + (d_tau ? 1 : 0);
            // (block-compact W, FIGH_FLAG_COMPACT_BLOCKS: every row block is a matrix of its own; d_cols are then columns of it)
            const double *Wj = h_block_off ? d_W + h_block_off[j] : d_W + (int64_t)j * rows_b * ldw;
            const int64_t ldj = h_block_ld ? h_block_ld[j] : ldw;
            const double *tj = d_tau ? d_tau + (int64_t)j * rows_b : nullptr;
            const int32_t *cols_j = d_cols + off_at[j], *pos_j = d_pos + off_at[j];
            double *out_j = stack + (size_t)row_at[j] * nc;
            if (kind[j] == JOB) {
                Tsqr2Job J{};
                J.W = Wj;
                J.tau = tj;
                J.col_idx = cols_j;
                J.pos = pos_j;
                J.out = out_j;
                J.rows = rows_b;
                J.ldw = ldj;
                J.n = nj;
                J.nc = ncj;
                jobs.push_back(J);
                continue;
            }
            int64_t cnt = 0;
            double *Rb = Rb_all + (size_t)j * nmax * nmax;
            if (kind[j] == WIDE || kind[j] == MID) {
                const int64_t cap_j = figh_tsqr_level0_capacity(ncj);
                double *tri_j = kind[j] == WIDE ? wide_tri + wide_at : mid_tri + mid_at;
                (kind[j] == WIDE ? wide_at : mid_at) += (size_t)ncj * ncj * (size_t)cap_j;
                if (int rc = figh_tsqr_level0(Wj, rows_b, ldj, cols_j, nj, tj, nullptr, 0, tri_j, cap_j, &cnt, nullptr)) return rc;
                if (kind[j] == WIDE) wide.push_back({tri_j, (long)cnt, ncj, Rb});
                else mids.push_back({tri_j, (long)cnt, ncj, Rb});
                embeds.push_back({j, Rb, ncj, nj, pos_j, out_j});
                continue;
            }
            if (nj > 0) {
                if (int rc = figh_tsqr_level0(Wj, rows_b, ldj, cols_j, nj, tj, nullptr, 0, tri_b, cap_b, &cnt, nullptr)) return rc;
            } else {
                // (only tau in this block: its norm still counts) a 1 x 1 "matrix", the tau rows alone, through the same
                // kernel with tau as its only column
                if (int rc = figh_tsqr_level0(tj, rows_b, 1, nullptr, 1, nullptr, nullptr, 0, tri_b, cap_b, &cnt, nullptr)) return rc;
            }
            if (cnt == 1) FIGH_HIP(hipMemcpyAsync(Rb, tri_b, sizeof(double) * (size_t)ncj * ncj, hipMemcpyDeviceToDevice, stream()));
            else if (int rc = tsqr_reduce(tri_b, cnt, ncj, Rb)) return rc;
            embeds.push_back({j, Rb, ncj, nj, pos_j, out_j});
        }
        if (phase == 0 && !wide.empty()) {
            SideStream side;  // (falls back to the library stream when a second stream cannot be had; an error return leaves
                              // the streams joined)
            if (int rc = reduce_wide_stacks(wide)) return rc;
            wide_done = side.finish();
        }
    }
    GroupEmbed group_embed;
    if (!jobs.empty())
        if (int rc = launch_tsqr_group(jobs, nc, n, cu_count(), &group_embed)) return rc;
    for (const MidStack &ms : mids) {
        if (ms.cnt == 1)
            FIGH_HIP(hipMemcpyAsync(ms.R, ms.tri, sizeof(double) * (size_t)ms.ncj * ms.ncj, hipMemcpyDeviceToDevice, stream()));
        else if (int rc = tsqr_reduce(ms.tri, ms.cnt, ms.ncj, ms.R))
            return rc;
    }
    stream_wait(wide_done);
    // (the embedding zero-fills nc rows from its offset and writes the block's ncj rows: the rows behind them belong to the next
    // block, whose own embedding follows in stream order; the buffer ends nc rows behind the last block)
    std::sort(embeds.begin(), embeds.end(), [](const Embed &x, const Embed &y) { return x.block < y.block; });
    for (const Embed &e : embeds)
        if (int rc = embed_force_triangle(e.R, e.ncj, e.nj, e.pos, nc, n, e.out)) return rc;
    if (int rc = launch_tsqr_group_embed(group_embed)) return rc;
    if (row_off == 0) {
        FIGH_HIP(hipMemsetAsync(stack, 0, tri, stream()));
        row_off = nc;
    }
    if (nc > 80) {
        {
            ProfileScope scope("tsqr_block_stack");
            if (int rc = launch_tsqr_wide_single(stack, row_off, nc, nullptr, nc, nc, one)) return rc;
        }
        if (tol_qr < 0.0) {
            FIGH_HIP(hipMemcpyAsync(d_R_out, one, tri, hipMemcpyDeviceToDevice, stream()));
            return FIGH_OK;
        }
        return reveal_triangle(one, nc, n, tol_qr, d_R_out);
    }
    int64_t nw = 0;
    double *Rws = nullptr;
    if (int rc = figh_tsqr_level0(stack, row_off, nc, nullptr, nc, nullptr, nullptr, 0, nullptr, 0, &nw, &Rws)) return rc;
    return tsqr_reduce_stack(Rws, nw, nc, n, tol_qr, d_R_out);
}

// r2[b] = sum over the rows [off[b], off[b+1]) of a stacked matrix of (row . v)^2: one workgroup per block, rows over the
// threads, fixed-order tree (deterministic)
struct BlockOffsets {
    int off[kMaxJoints + 1];
};
__global__ __launch_bounds__(256) void block_rows_residuals_kernel(const double *__restrict__ Rs, const BlockOffsets bo,
                                                                   const int nc, const double *__restrict__ v,
                                                                   double *__restrict__ r2) {
    __shared__ double sm[256];
    const int lo = bo.off[blockIdx.x], hi = bo.off[blockIdx.x + 1];
    double s = 0.0;
    for (int k = lo + threadIdx.x; k < hi; k += 256) {
        double d = 0.0;
        for (int c = 0; c < nc; ++c) d += Rs[(size_t)k * nc + c] * v[c];
        s += d * d;
    }
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) r2[blockIdx.x] = sm[0];
}

int figh_block_rows_residuals(const double *d_rows, int nblocks, const int32_t *h_row_off, int nc, const double *d_v,
                              double *d_r2) {
    FIGH_REQUIRE(d_rows && h_row_off && d_v && d_r2, "NULL pointer");
    FIGH_REQUIRE(nblocks >= 1 && nblocks <= kMaxJoints && nc >= 1 && nc <= 1024, "bad shape");
    BlockOffsets bo;
    for (int b = 0; b <= nblocks; ++b) {
        FIGH_REQUIRE(h_row_off[b] >= 0 && (b == 0 || h_row_off[b] >= h_row_off[b - 1]), "row offsets must not decrease");
        bo.off[b] = h_row_off[b];
    }
    if (int rc = ensure_device()) return rc;
    ProfileScope scope("block_rows_residuals");
    hipLaunchKernelGGL(block_rows_residuals_kernel, dim3((unsigned)nblocks), dim3(256), 0, stream(), d_rows, bo, nc, d_v, d_r2);
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}

int figh_tsqr_merge_base(const double *d_Rs, int count, int nc, int n_free, double tol_qr, double *d_Rk_out) {
    FIGH_REQUIRE(d_Rs && d_Rk_out, "NULL device pointer");
    FIGH_REQUIRE(count >= 1 && nc >= 1 && nc <= 512 && n_free >= 1 && n_free <= nc, "bad shape");
    FIGH_REQUIRE(tol_qr >= 0.0, "tol_qr must be non-negative");
    if (int rc = ensure_device()) return rc;
    return tsqr_reduce_stack(d_Rs, count, nc, n_free, tol_qr, d_Rk_out);
}

}  // extern "C"
