// Streamed ("fused" in SURVEY.md section 8b) entry points: the stacked regressor is never materialised in full.
// The samples are cut into chunks; each chunk's W lives in a library workspace just long enough for the next kernel
// (column norms, or the level-0 Householder TSQR) and is then overwritten by the next chunk.  This is what makes the
// human model at 1e7 samples (269 GB of W) or any W beyond HBM tractable, and it is the C-ABI form of
// IdentificationPipeline(chunk_samples=...).
//
//   figh_regressor_colsq  : diag(W^T W) of build_regressor_basic's W               (regressor.py:243,271)
//   figh_regressor_tsqr   : R of np.linalg.qr(W[:, col_idx] | tau), optionally row-block weighted
//                                                                                   (qrdecomposition.py:105,205,238)
//   figh_regressor_gram   : W_e^T W_e, W_e^T tau, tau^T tau from that R (host, O(n^3)) -- the normal-equation
//                           quantities of the SIP QP (identification_tools.py:528-531) and of collective (1) in
//                           SURVEY.md section 8e, without squaring the condition number on the way.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "figh_internal.h"

using namespace figh;

namespace {

__global__ __launch_bounds__(256) void vec_add_kernel(double *__restrict__ acc, const double *__restrict__ x, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) acc[i] += x[i];
}

// reference column 14 l + s -> link-padded column 16 l + s (a NULL list = all columns)
// (link_pos: the link-compact layout, FIGH_FLAG_LINK_COMPACT -- a link without a segment only has eliminated columns, which
// no caller lists; should one be listed all the same it reads segment 0: values of another column, never out of range)
__global__ __launch_bounds__(256) void pad_columns_kernel(const int32_t *__restrict__ in, int n, int32_t *__restrict__ out,
                                                          const int *__restrict__ link_pos) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const int c = in ? in[i] : i;
        const int l = link_pos ? max(link_pos[c / 14], 0) : c / 14;
        out[i] = l * 16 + c % 14;
    }
}

int64_t default_chunk(int rps, int ncols) {
    // ~2 GB of W per chunk: large enough to fill the chip (2048 waves x >= 8 tiles), small next to 288 GB
    int64_t c = (int64_t)(2.0e9 / (8.0 * rps * ncols));
    c = (c / 64) * 64;
    return c < 64 ? 64 : c;
}

}  // namespace

extern "C" int figh_regressor_colsq(figh_model_t model, int mode, int flags, int ft_mask, int64_t N, const double *d_q,
                                    const double *d_v, const double *d_a, int64_t chunk_samples, double *d_colsq) {
    int rps = 0, ncols = 0;
    if (int rc = figh_regressor_shape(model, mode, flags, &rps, &ncols)) return rc;
    FIGH_REQUIRE(N >= 0 && chunk_samples >= 0, "negative size");
    FIGH_REQUIRE(d_q && d_v && d_a && d_colsq, "NULL device pointer");
    if (int rc = ensure_device()) return rc;
    FIGH_HIP(hipMemsetAsync(d_colsq, 0, sizeof(double) * ncols, stream()));
    if (N == 0) return FIGH_OK;
    if (!(model->is_chain && mode == FIGH_MODE_JOINT_TORQUE && !(flags & FIGH_FLAG_GENERIC)) && !(flags & FIGH_FLAG_TX40)) {
        // tree kernel: the norms are accumulated from the computed segments directly, W is never written (nor chunked)
        int done = 0;
        if (int rc = launch_regressor_tree(model, mode, flags, ft_mask, N, d_q, d_v, d_a, nullptr, 0, ncols, 16, d_colsq, &done))
            return rc;
        return FIGH_OK;
    }
    if (chunk_samples == 0) chunk_samples = default_chunk(rps, ncols);
    const int64_t cs = chunk_samples < N ? chunk_samples : N;
    double *Wc = static_cast<double *>(workspace(sizeof(double) * (size_t)rps * cs * ncols, 8));
    double *part = static_cast<double *>(workspace(sizeof(double) * ncols, 9));
    if (!Wc || !part) return FIGH_ERR_ALLOC;
    const int nq = model->host.nq, nv = model->host.nv;
    for (int64_t lo = 0; lo < N; lo += cs) {
        const int64_t nc_ = (lo + cs <= N) ? cs : N - lo;
        if (int rc = figh_regressor_build(model, mode, flags, ft_mask, nc_, d_q + lo * nq, d_v + lo * nv, d_a + lo * nv, Wc,
                                          ncols, part))
            return rc;
        hipLaunchKernelGGL(vec_add_kernel, dim3((ncols + 255) / 256), dim3(256), 0, stream(), d_colsq, part, ncols);
        FIGH_HIP(hipGetLastError());
    }
    return FIGH_OK;
}

static int regressor_tsqr_impl(figh_model_t model, int mode, int flags, int ft_mask, int64_t N, const double *d_q,
                               const double *d_v, const double *d_a, const int32_t *d_col_idx, int n, const double *d_tau,
                               const double *h_block_weight, int nblocks, int64_t chunk_samples, double *d_R_out,
                               double *d_colsq_out);

extern "C" int figh_regressor_tsqr(figh_model_t model, int mode, int flags, int ft_mask, int64_t N, const double *d_q,
                                   const double *d_v, const double *d_a, const int32_t *d_col_idx, int n,
                                   const double *d_tau, const double *h_block_weight, int nblocks, int64_t chunk_samples,
                                   double *d_R_out) {
    return regressor_tsqr_impl(model, mode, flags, ft_mask, N, d_q, d_v, d_a, d_col_idx, n, d_tau, h_block_weight, nblocks,
                               chunk_samples, d_R_out, nullptr);
}

extern "C" int figh_regressor_tsqr_norms(figh_model_t model, int mode, int flags, int ft_mask, int64_t N, const double *d_q,
                                         const double *d_v, const double *d_a, const int32_t *d_col_idx, int n,
                                         const double *d_tau, const double *h_block_weight, int nblocks,
                                         int64_t chunk_samples, double *d_R_out, double *d_colsq_out) {
    FIGH_REQUIRE(d_colsq_out, "NULL device pointer");
    return regressor_tsqr_impl(model, mode, flags, ft_mask, N, d_q, d_v, d_a, d_col_idx, n, d_tau, h_block_weight, nblocks,
                               chunk_samples, d_R_out, d_colsq_out);
}

static int regressor_tsqr_impl(figh_model_t model, int mode, int flags, int ft_mask, int64_t N, const double *d_q,
                               const double *d_v, const double *d_a, const int32_t *d_col_idx, int n, const double *d_tau,
                               const double *h_block_weight, int nblocks, int64_t chunk_samples, double *d_R_out,
                               double *d_colsq_out) {
    int rps = 0, ncols = 0;
    if (int rc = figh_regressor_shape(model, mode, flags, &rps, &ncols)) return rc;
    FIGH_REQUIRE(N >= 0 && chunk_samples >= 0, "negative size");
    FIGH_REQUIRE(d_q && d_v && d_a && d_R_out, "NULL device pointer");
    FIGH_REQUIRE(n >= 1 && (d_col_idx ? n <= ncols : n == ncols), "bad column count");
    if (h_block_weight) FIGH_REQUIRE(nblocks > 0 && rps % nblocks == 0, "row-block weights must divide the rows of a sample");
    if (int rc = ensure_device()) return rc;
    const int nc = n + (d_tau ? 1 : 0);
    if (d_colsq_out) FIGH_HIP(hipMemsetAsync(d_colsq_out, 0, sizeof(double) * ncols, stream()));
    if (N == 0) {
        FIGH_HIP(hipMemsetAsync(d_R_out, 0, sizeof(double) * (size_t)nc * nc, stream()));
        return FIGH_OK;
    }
    if (chunk_samples == 0) chunk_samples = default_chunk(rps, ncols);
    const int64_t cs = chunk_samples < N ? chunk_samples : N;
    const int64_t nchunks = (N + cs - 1) / cs;
    // diag(W^T W) of all columns fused into the regressor kernel of every chunk (the elimination's input, for a caller that
    // factors the columns it EXPECTS to be kept and verifies the set afterwards: no separate norms pass over the samples)
    double *cs_part = d_colsq_out ? static_cast<double *>(workspace(sizeof(double) * ncols, 21)) : nullptr;
    if (d_colsq_out && !cs_part) return FIGH_ERR_ALLOC;
    FIGH_REQUIRE(!(flags & FIGH_FLAG_BLOCKED_INPUTS) || nchunks == 1 || cs % 64 == 0,
                 "tile-blocked inputs: chunk_samples must be a multiple of 64");
    // the chunk's W is a private workspace: tree models get the link-padded layout (16 columns per link, every row
    // segment one 128-byte line) and the caller's column list is translated to it
    const bool padded = !(model->is_chain && mode == FIGH_MODE_JOINT_TORQUE) && !(flags & FIGH_FLAG_TX40);
    // external wrench on a free-flyer root: links without entries (massless bodies) get no columns in the chunk's W
    // (FIGH_FLAG_LINK_COMPACT; the caller's column list can only name them if it keeps columns that are identically zero)
    int link_pos[kMaxJoints];
    int *d_link_pos = nullptr;
    int nlive = padded ? tree_link_positions(model, mode, flags & 7, ft_mask, link_pos) : -1;
    if (nlive <= 0 || nlive >= model->host.nlinks) nlive = -1;
    if (nlive > 0) {
        // a caller may list columns of links without entries (identically zero columns: the SIP program passes the inertial
        // columns of EVERY link, identification_tools.py:528-531): those exist in the link-padded layout only
        std::vector<int32_t> cols(n);
        if (d_col_idx) {
            if (int rc = figh_memcpy_d2h(cols.data(), d_col_idx, sizeof(int32_t) * n)) return rc;
        } else {
            for (int c = 0; c < n; ++c) cols[c] = c;
        }
        for (int c = 0; c < n && nlive > 0; ++c)
            if (cols[c] < 0 || cols[c] / 14 >= model->host.nlinks || link_pos[cols[c] / 14] < 0) nlive = -1;
    }
    if (nlive > 0) {
        d_link_pos = static_cast<int *>(workspace(sizeof(int) * kMaxJoints, 37));
        if (!d_link_pos) return FIGH_ERR_ALLOC;
        FIGH_HIP(hipMemcpyAsync(d_link_pos, link_pos, sizeof(int) * model->host.nlinks, hipMemcpyHostToDevice, stream()));
        FIGH_HIP(hipStreamSynchronize(stream()));  // (link_pos lives on this stack frame)
        flags |= FIGH_FLAG_LINK_COMPACT;
    }
    const int64_t ldc = padded ? 16 * (int64_t)(nlive > 0 ? nlive : model->host.nlinks) : ncols;
    double *Wc = static_cast<double *>(workspace(sizeof(double) * (size_t)rps * cs * ldc, 8));
    int32_t *d_cols = const_cast<int32_t *>(d_col_idx);
    if (padded) {
        d_cols = static_cast<int32_t *>(workspace(sizeof(int32_t) * (size_t)n, 9));
        if (!d_cols) return FIGH_ERR_ALLOC;
        hipLaunchKernelGGL(pad_columns_kernel, dim3((n + 255) / 256), dim3(256), 0, stream(), d_col_idx, n, d_cols,
                           (const int *)d_link_pos);
        FIGH_HIP(hipGetLastError());
    }
    double *tc = d_tau ? static_cast<double *>(workspace(sizeof(double) * (size_t)rps * cs, 10)) : nullptr;
    // level-0 triangles of ALL chunks are stacked and the merge tree runs once (a merge is latency-bound: running it
    // per chunk cost 158 ms of the 1.24 s human pass)
    const int64_t per_chunk = figh_tsqr_level0_capacity(nc);
    double *stack = static_cast<double *>(workspace(sizeof(double) * (size_t)nchunks * per_chunk * nc * nc, 11));
    if (!Wc || (d_tau && !tc) || !stack) return FIGH_ERR_ALLOC;
    const int nq = model->host.nq, nv = model->host.nv;
    // joint-torque regressor of single-dof joints: the rows of joint j only involve the links from j on, so the kept
    // columns in front of 14 j are exact zeros in row block j -- the structure hint of figh_tsqr_structured
    std::vector<int32_t> first;
    if (mode == FIGH_MODE_JOINT_TORQUE && nv == model->host.njoints - 1 && nc <= 80 && cs >= 64) {
        std::vector<int32_t> cols(n);
        if (d_col_idx) {
            if (int rc = figh_memcpy_d2h(cols.data(), d_col_idx, sizeof(int32_t) * n)) return rc;
        } else {
            for (int c = 0; c < n; ++c) cols[c] = c;
        }
        bool sorted = true;
        for (int c = 1; c < n; ++c) sorted = sorted && cols[c - 1] < cols[c];
        if (sorted) {
            first.resize(rps);
            for (int j = 0; j < rps; ++j) {
                int f = 0;
                while (f < n && cols[f] < 14 * j) ++f;
                first[j] = f;
            }
        }
    }
    // More than 80 columns and several chunks: CHAINED launches -- every workgroup of the blocked kernel keeps ONE running
    // triangle across the chunks (same workgroup count in every launch; a launch starts from the triangle the previous one
    // wrote), so the stack to merge holds one launch's triangles instead of nchunks times as many (human model, 20 chunks:
    // 512 instead of 10 240 triangles, merges 8.3 -> 1.8 ms).
#ifdef FIGH_ABLATION
    const bool chained = nc > 80 && nchunks > 1 && !getenv("FIGH_NO_CHAIN");  // same-box A/B (tools/chain_ab.sh)
#else
    const bool chained = nc > 80 && nchunks > 1;
#endif
    // External wrench on a free-flyer root: per chunk the three force row blocks are factored over the nf columns that can
    // be non-zero there (rotational-inertia columns are exact zeros in force rows, figh_tsqr_selected_wrench) and reduced to
    // one triangle per chunk; the torque row blocks go through the (chained) launches over all columns.
    int nf = 0;
    if (mode == FIGH_MODE_EXT_WRENCH && rps == 6 && model->host.njoints > 1 && model->host.jtype[1] == FIGH_JT_FREEFLYER &&
        !(flags & FIGH_FLAG_TX40) && !h_block_weight && nc > 80 && 3 * cs >= 16L * nc) {
        std::vector<int32_t> cols(n);
        if (d_col_idx) {
            if (int rc = figh_memcpy_d2h(cols.data(), d_col_idx, sizeof(int32_t) * n)) return rc;
        } else {
            for (int c = 0; c < n; ++c) cols[c] = c;
        }
        for (int c = 0; c < n; ++c) nf += (cols[c] % 14) >= 6;
        if (nf >= n) nf = 0;
    }
    const bool split = nf > 0;
    const int ncf = nf + (d_tau ? 1 : 0);
    const long rsplit = split ? 2 : 1;  // the torque rows are half of a chunk
    int *fsel = nullptr;
    double *tri_f = nullptr, *stack_f = nullptr, *Rf = nullptr;
    int64_t cap_f = 0;
    if (split) {
        cap_f = figh_tsqr_level0_capacity(ncf);
        fsel = static_cast<int *>(workspace(sizeof(int) * 2 * (size_t)n, 22));
        // the force rows' level-0 triangles of ALL chunks are stacked and merged once (a merge per chunk cost 0.175 ms x 20
        // for the human model)
        stack_f = static_cast<double *>(workspace(sizeof(double) * (size_t)ncf * ncf * cap_f * nchunks, 25));
        tri_f = stack_f;
        Rf = static_cast<double *>(workspace(sizeof(double) * (size_t)ncf * ncf, 24));
        if (!fsel || !tri_f || !stack_f || !Rf) return FIGH_ERR_ALLOC;
        if (int rc = split_force_columns(d_cols, n, padded ? 16 : 14, fsel)) return rc;
    }
    const long chain_wgs =
        chained ? std::min<long>(per_chunk - 1, std::max<long>(1, (long)rps * cs / rsplit / (8L * nc))) : 0;
    int64_t have = 0, have_f = 0;
    for (int64_t lo = 0; lo < N; lo += cs) {
        const int64_t nc_ = (lo + cs <= N) ? cs : N - lo;
        if (int rc = padded ? figh_regressor_build_padded(model, mode, flags, ft_mask, nc_, d_q + lo * nq, d_v + lo * nv,
                                                          d_a + lo * nv, Wc, ldc, cs_part)
                            : figh_regressor_build(model, mode, flags, ft_mask, nc_, d_q + lo * nq, d_v + lo * nv,
                                                   d_a + lo * nv, Wc, ldc, cs_part))
            return rc;
        if (cs_part) {
            hipLaunchKernelGGL(vec_add_kernel, dim3((ncols + 255) / 256), dim3(256), 0, stream(), d_colsq_out, cs_part, ncols);
            FIGH_HIP(hipGetLastError());
        }
        if (d_tau)  // rows j*N + [lo, lo + nc_) of tau -> the chunk's joint-major vector (rows j*nc_ + i)
            FIGH_HIP(hipMemcpy2DAsync(tc, sizeof(double) * nc_, d_tau + lo, sizeof(double) * N, sizeof(double) * nc_, rps,
                                      hipMemcpyDeviceToDevice, stream()));
        int64_t got = 0;
        if (!first.empty() && nc_ >= 64) {
            if (int rc = figh_tsqr_hint_begin(first.data(), rps, (int64_t)rps * nc_, n, nc)) return rc;
        }
        const int64_t rows_f = split ? 3 * nc_ : 0;  // the chunk's force rows (row blocks 0 .. 2 of its joint-major W)
        if (split) {
            int64_t cnt_f = 0;
            if (int rc = figh_tsqr_level0(Wc, rows_f, ldc, fsel, nf, tc, nullptr, 0, stack_f + (size_t)have_f * ncf * ncf, cap_f,
                                          &cnt_f, nullptr))
                return rc;
            have_f += cnt_f;
        }
        if (chained) tsqr_level0_chain(chain_wgs, lo > 0 ? 1 : 0);
        const int rc0 = figh_tsqr_level0(Wc + rows_f * ldc, (int64_t)rps * nc_ - rows_f, ldc, d_cols, n,
                                         tc ? tc + rows_f : nullptr, h_block_weight, nblocks,
                                         stack + (size_t)(chained ? 0 : have) * nc * nc, per_chunk, &got, nullptr);
        figh_tsqr_hint_end();
        if (rc0) return rc0;
        have = chained ? got : have + got;
    }
    FIGH_REQUIRE(have < (1LL << 31), "too many level-0 triangles");
    if (split) {  // the force rows' triangle of all chunks, over all kept columns, as one more element of the stack
        FIGH_REQUIRE(have_f < (1LL << 31), "too many level-0 triangles");
        if (int rc = figh_tsqr_merge(stack_f, (int)have_f, ncf, Rf)) return rc;
        if (int rc = embed_force_triangle(Rf, ncf, nf, fsel + n, nc, n, stack + (size_t)have * nc * nc)) return rc;
        ++have;
    }
    return figh_tsqr_merge(stack, (int)have, nc, d_R_out);
}

// triangles of the previous trajectories (common) and of trajectory b, interleaved: the stack of the pair level that
// folds the first into every one of the B results
__global__ __launch_bounds__(256) void interleave_stack_kernel(const double *__restrict__ common,
                                                               const double *__restrict__ each, const long tri,
                                                               const long B, double *__restrict__ out) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= tri * B) return;
    const long b = e / tri, k = e - b * tri;
    out[2 * b * tri + k] = common[k];
    out[(2 * b + 1) * tri + k] = each[e];
}

extern "C" int figh_regressor_tsqr_batch(figh_model_t model, int mode, int flags, int ft_mask, int64_t B, int64_t n_per,
                                         const double *d_q, const double *d_v, const double *d_a,
                                         const int32_t *d_col_idx, int n, const double *d_R_stack, double *d_R_out) {
    int rps = 0, ncols = 0;
    if (int rc = figh_regressor_shape(model, mode, flags, &rps, &ncols)) return rc;
    FIGH_REQUIRE(B >= 1 && n_per >= 1, "figh_regressor_tsqr_batch: at least one trajectory of at least one sample");
    FIGH_REQUIRE(d_q && d_v && d_a && d_R_out, "NULL device pointer");
    FIGH_REQUIRE(n >= 1 && (d_col_idx ? n <= ncols : n == ncols) && n <= 512, "bad column count");
    if (int rc = ensure_device()) return rc;
    const int nc = n;
    const size_t tri = (size_t)nc * nc;
    const int nq = model->host.nq, nv = model->host.nv;
    const int64_t N_tot = B * n_per;
    if (nc <= 80 || (int64_t)rps * n_per < 8L * nc) {
        // register-tile kernel (or trajectories too short to be cut further): a launch pair per trajectory
        FIGH_REQUIRE(!(flags & FIGH_FLAG_BLOCKED_INPUTS) || n_per % 64 == 0,
                     "tile-blocked inputs: samples per trajectory must be a multiple of 64");
        double *pair = d_R_stack ? static_cast<double *>(workspace(sizeof(double) * 2 * tri, 20)) : nullptr;
        if (d_R_stack && !pair) return FIGH_ERR_ALLOC;
        if (d_R_stack) FIGH_HIP(hipMemcpyAsync(pair, d_R_stack, sizeof(double) * tri, hipMemcpyDeviceToDevice, stream()));
        for (int64_t b = 0; b < B; ++b) {
            double *dst = d_R_stack ? pair + tri : d_R_out + (size_t)b * tri;
            if (int rc = figh_regressor_tsqr(model, mode, flags, ft_mask, n_per, d_q + b * n_per * nq, d_v + b * n_per * nv,
                                             d_a + b * n_per * nv, d_col_idx, n, nullptr, nullptr, 0, n_per, dst))
                return rc;
            if (d_R_stack)
                if (int rc = figh_tsqr_merge(pair, 2, nc, d_R_out + (size_t)b * tri)) return rc;
        }
        return FIGH_OK;
    }
    // ---- blocked kernel: ONE K1 launch over all B * n_per samples (the standard joint-major regressor: trajectory b is the
    // rps row segments [j N_tot + b n_per, + n_per)), ONE batched level-0 launch (wgs workgroups per trajectory), then
    // pair levels over the whole stack -- wgs is a power of two, so a pair never straddles two trajectories
    const bool padded = !(model->is_chain && mode == FIGH_MODE_JOINT_TORQUE) && !(flags & FIGH_FLAG_TX40);
    const int64_t ldc = padded ? 16 * (int64_t)model->host.nlinks : ncols;
    FIGH_REQUIRE(ldc < (1L << 21), "figh_tsqr: more than 80 columns need a leading dimension below 2^21 elements");
    double *Wc = static_cast<double *>(workspace(sizeof(double) * (size_t)rps * N_tot * ldc, 8));
    if (!Wc) return FIGH_ERR_ALLOC;
    int32_t *d_cols = const_cast<int32_t *>(d_col_idx);
    if (padded) {
        d_cols = static_cast<int32_t *>(workspace(sizeof(int32_t) * (size_t)n, 9));
        if (!d_cols) return FIGH_ERR_ALLOC;
        hipLaunchKernelGGL(pad_columns_kernel, dim3((n + 255) / 256), dim3(256), 0, stream(), d_col_idx, n, d_cols,
                           (const int *)nullptr);
        FIGH_HIP(hipGetLastError());
    }
    if (int rc = padded ? figh_regressor_build_padded(model, mode, flags, ft_mask, N_tot, d_q, d_v, d_a, Wc, ldc, nullptr)
                        : figh_regressor_build(model, mode, flags, ft_mask, N_tot, d_q, d_v, d_a, Wc, ldc, nullptr))
        return rc;
    const long M = tsqr_wide_tile_rows(nc);
    const long tiles = rps * ((n_per + M - 1) / M);
    // workgroups per trajectory: enough in total to fill the chip twice over, a leaf at least four tiles and 4 nc rows tall
    long cap = std::min<long>({tiles / 4, (long)(rps * n_per) / (4L * nc), std::max<long>(1, 1024 / B)});
    long wgs = 1;
    while (2 * wgs <= cap) wgs *= 2;
    double *stack = static_cast<double *>(workspace(sizeof(double) * tri * (size_t)(B * wgs), 11));
    if (!stack) return FIGH_ERR_ALLOC;
    {
        ProfileScope scope("tsqr_batch");
        if (int rc = launch_tsqr_wide_batch(Wc, ldc, d_cols, n, nc, B, n_per, rps, N_tot, wgs, stack)) return rc;
    }
    const double *cur = stack;
    int slot = 2;
    for (long w = wgs; w > 1; w /= 2) {
        ProfileScope scope("tsqr_reduce");
        double *dst = (w == 2 && !d_R_stack) ? d_R_out : static_cast<double *>(workspace(sizeof(double) * tri * (size_t)(B * w / 2), slot));
        if (!dst) return FIGH_ERR_ALLOC;
        if (int rc = launch_tsqr_wide_pairs(cur, B * w, nc, dst)) return rc;
        cur = dst;
        slot = slot == 2 ? 3 : 2;
    }
    if (!d_R_stack) {
        if (cur != d_R_out) FIGH_HIP(hipMemcpyAsync(d_R_out, cur, sizeof(double) * tri * B, hipMemcpyDeviceToDevice, stream()));
        return FIGH_OK;
    }
    double *both = static_cast<double *>(workspace(sizeof(double) * 2 * tri * (size_t)B, 20));
    if (!both) return FIGH_ERR_ALLOC;
    hipLaunchKernelGGL(interleave_stack_kernel, dim3((unsigned)((tri * B + 255) / 256)), dim3(256), 0, stream(), d_R_stack, cur,
                       (long)tri, (long)B, both);
    FIGH_HIP(hipGetLastError());
    ProfileScope scope("tsqr_reduce");
    return launch_tsqr_wide_pairs(both, 2 * B, nc, d_R_out);
}

extern "C" int figh_regressor_gram(figh_model_t model, int mode, int flags, int ft_mask, int64_t N, const double *d_q,
                                   const double *d_v, const double *d_a, const int32_t *d_col_idx, int n,
                                   const double *d_tau, int64_t chunk_samples, double *h_G, double *h_g,
                                   double *h_tau_sq) {
    FIGH_REQUIRE(h_G, "NULL output");
    FIGH_REQUIRE(!d_tau || (h_g && h_tau_sq), "tau given but no output for W^T tau / tau^T tau");
    const int nc = n + (d_tau ? 1 : 0);
    FIGH_REQUIRE(n >= 1 && nc <= 512, "bad column count");
    double *d_R = static_cast<double *>(workspace(sizeof(double) * (size_t)nc * nc, 12));
    if (!d_R) return FIGH_ERR_ALLOC;
    if (int rc = figh_regressor_tsqr(model, mode, flags, ft_mask, N, d_q, d_v, d_a, d_col_idx, n, d_tau, nullptr, 0,
                                     chunk_samples, d_R))
        return rc;
    std::vector<double> R((size_t)nc * nc);
    if (int rc = figh_memcpy_d2h(R.data(), d_R, sizeof(double) * R.size())) return rc;
    // W^T W = R1^T R1, W^T tau = R1^T z, tau^T tau = z^T z + rho^2 with R = [R1 z; 0 rho]
    for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j) {
            double s = 0.0;
            for (int k = 0; k <= i; ++k) s += R[(size_t)k * nc + i] * R[(size_t)k * nc + j];
            h_G[(size_t)i * n + j] = s;
            h_G[(size_t)j * n + i] = s;
        }
    if (d_tau) {
        double tt = 0.0;
        for (int k = 0; k < nc; ++k) tt += R[(size_t)k * nc + n] * R[(size_t)k * nc + n];
        *h_tau_sq = tt;
        for (int i = 0; i < n; ++i) {
            double s = 0.0;
            for (int k = 0; k <= i; ++k) s += R[(size_t)k * nc + i] * R[(size_t)k * nc + n];
            h_g[i] = s;
        }
    }
    return FIGH_OK;
}
