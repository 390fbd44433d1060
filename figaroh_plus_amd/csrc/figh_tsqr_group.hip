// Grouped launches for the NARROW row blocks of a tree's joint-torque regressor (figh_tsqr_selected_blocks, figh_linalg.hip).
//
// Row block j of that regressor (regressor.py:45-87: the rows of joint j) is factored over its own column list -- the links
// of the joint's subtree -- and for most joints of a mobile manipulator the list is short: TIAGo's wheels, casters, head and
// gripper joints own 7 .. 14 of the 240 kept columns.  One register-tile launch + its merge launch + the embedding per
// block made 19 x 3 dependent launches of 10 .. 300 us that fill a fraction of the chip each (2.9 ms of a 16 ms step).
// Here every such block is a JOB in a device table and all of them run in
//
//   tsqr2_group_kernel      one launch: wave -> (job, local wave); the job's tiles are dealt round-robin to its waves, every
//                           wave keeps a private triangle (the column steps of figh_tsqr_narrow.h);
//   tsqr_coop_group_kernel  one launch per merge level (two for 64 waves per job): workgroup -> (job, 512 stacked rows);
//   embed_group_kernel      one launch: every job's triangle into its rows of the compact stack over the kept columns.
//
// Same arithmetic as the per-block path (np.linalg.qr of qrdecomposition.py:205 restricted to a row block and its columns).
#include <algorithm>
#include <cstring>
#include <type_traits>
#include <vector>

#include "figh_internal.h"
#include "figh_wave.h"
#include "figh_tsqr_narrow.h"
#include "figh_tsqr_narrow_kernel.h"

namespace figh {


// ------------------------------------------------------------------------------------------------ level 0
// tsqr2_kernel<4, 4, true>'s body (figh_tsqr_narrow_kernel.h: all loads of a tile in flight, next tile's chunks requested
// into retired registers, 8-way interleaved tile order) on the job's matrix; `zeros` = a structure hint of all zeros.
// Two forms of the body in one kernel (wave-uniform choice per job): <4, 4> = up to 64 columns, 64-row tiles; <1, 16> = at
// most 16 columns (with tau) -- ONE column chunk and 256-row tiles in the same 128 registers: a quarter of the column steps
// per row (the wheel, caster, head and gripper blocks of TIAGo have 8 .. 15 columns; their tiles are issue-bound on the
// steps' fixed part).  One launch for both kinds: separately neither fills the chip.
__global__ __launch_bounds__(64, 2) void tsqr2_group_kernel(const Tsqr2Job *__restrict__ jobs,
                                                            const int *__restrict__ job_of_wave,
                                                            const int *__restrict__ zeros, const double null2) {
    const int jid = __builtin_amdgcn_readfirstlane(job_of_wave[blockIdx.x]);
    const Tsqr2Job *J = jobs + jid;
    if (J->tall)  // (tall form, chosen by the host)
        tsqr2_level0_body<1, 16, true>(J->W, J->rows, J->ldw, J->col_idx, J->n, J->tau, (const double *)nullptr, 1L, J->tri,
                                       J->nc, zeros, (long)blockIdx.x - J->wave0, (long)J->nwaves, null2);
    else
        tsqr2_level0_body<4, 4, true>(J->W, J->rows, J->ldw, J->col_idx, J->n, J->tau, (const double *)nullptr, 1L, J->tri,
                                      J->nc, zeros, (long)blockIdx.x - J->wave0, (long)J->nwaves, null2);
}

// ------------------------------------------------------------------------------------------------ merge levels
// tsqr_coop_kernel<4, 8> per job: workgroup b of the job factors the stacked rows [512 b, 512 (b + 1)) of the level's input
// (level 0's triangles, then the previous level's).
__global__ __launch_bounds__(512) void tsqr_coop_group_kernel(const Tsqr2Job *__restrict__ jobs,
                                                              const int *__restrict__ job_of_wg, const int level) {
    constexpr int NCC = 4, NW = 8;
    __shared__ double pw[2][NW][16 * NCC];
    const int jid = __builtin_amdgcn_readfirstlane(job_of_wg[blockIdx.x]);
    const Tsqr2Job *J = jobs + jid;  // (read in place: a local copy with its level-indexed arrays would live in scratch)
    const int nc = J->nc;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lane_c = lane & 15, lane_g = lane >> 4;
    const int bl = (int)blockIdx.x - J->wg0[level];
    const double *Rs = level == 0 ? J->tri : J->lvl[level - 1];
    const long rows = (long)(level == 0 ? J->nwaves : J->nb[level - 1]) * nc;
    double *Rg = J->lvl[level] + (long)bl * nc * nc;
    const long r0 = ((long)bl * NW + wave) * 64;
    const int pad = 16 * NCC - nc;
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
    __syncthreads();
    tsqr_coop_panels<0, NCC, NW>(T, nc, pad, lane_c, lane_g, wave, pw, Rg);
}

// ------------------------------------------------------------------------------------------------ embedding
// one workgroup per job: its nc x nc triangle over (columns | tau) -> nc rows over the kept columns (+ tau): row r keeps its
// place, column c moves to pos[c] (tau to the last column): embed_force_triangle_kernel for every job at once
__global__ __launch_bounds__(256) void embed_group_kernel(const Tsqr2Job *__restrict__ jobs, const int ncfull, const int nfull) {
    const Tsqr2Job *J = jobs + blockIdx.x;
    const int nc = J->nc, n = J->n;
    const double *R = J->nlevels > 0 ? J->lvl[J->nlevels - 1] : J->tri;  // (where the last merge level left it)
    double *out = J->out;
    const int *pos = J->pos;
    for (int e = threadIdx.x; e < nc * ncfull; e += 256) out[e] = 0.0;
    __syncthreads();
    for (int e = threadIdx.x; e < nc * nc; e += 256) {
        const int r = e / nc, c = e - r * nc;
        if (c < r) continue;
        int col = c < n ? pos[c] : nfull;
        col = col < 0 ? 0 : (col >= ncfull ? ncfull - 1 : col);
        out[(long)r * ncfull + col] = R[e];
    }
}

// Host side: the jobs come with W, rows, ldw, col_idx, pos, tau, n, nc, out filled in by the caller; the rest is planned
// here.  All jobs have nc <= 64 and at least 64 rows.
int launch_tsqr_group(std::vector<Tsqr2Job> &jobs, int ncfull, int nfull, int cus, GroupEmbed *later) {
    const int njobs = (int)jobs.size();
    if (njobs == 0) return FIGH_OK;
    // waves per job: as many as its tiles allow (a leaf much taller than wide), at most 512 -- three merge levels of 512
    // stacked rows per workgroup then reach one triangle for any nc <= 64 -- and together a few rounds of the chip
    // (the waves are independent: the grid need not be resident at once)
    const long per_job_cap = std::max(16L, (long)cus * 32 / njobs);
    std::vector<int> job_of_wave, job_of_wg[3];
    size_t doubles = 0;
    int max_nc = 1;
    for (int j = 0; j < njobs; ++j) {
        Tsqr2Job &J = jobs[j];
        const bool tall = J.nc <= 16 && J.rows >= 256L * 64;
        const long ntiles = tall ? (J.rows + 255) / 256 : (J.rows + 63) / 64;
        long nw = std::min(std::min(ntiles / 8, 512L), per_job_cap);
        if (nw < 1) nw = 1;
        J.nwaves = (int)nw;
        J.tall = tall ? 1 : 0;
        J.wave0 = (int)job_of_wave.size();
        for (long w = 0; w < nw; ++w) job_of_wave.push_back(j);
        doubles += (size_t)nw * J.nc * J.nc;
        long cnt = nw;
        J.nlevels = 0;
        for (int l = 0; l < 3; ++l) {
            J.nb[l] = 0;
            J.wg0[l] = 0;
        }
        while (cnt > 1) {
            if (J.nlevels == 3) {
                set_error("grouped TSQR: a job needs more than three merge levels");
                return FIGH_ERR_UNSUPPORTED;
            }
            const long nb = (cnt * J.nc + 511) / 512;
            J.nb[J.nlevels] = (int)nb;
            J.wg0[J.nlevels] = (int)job_of_wg[J.nlevels].size();
            for (long b = 0; b < nb; ++b) job_of_wg[J.nlevels].push_back(j);
            doubles += (size_t)nb * J.nc * J.nc;
            ++J.nlevels;
            cnt = nb;
        }
        max_nc = J.nc > max_nc ? J.nc : max_nc;
    }
    double *buf = static_cast<double *>(workspace(sizeof(double) * doubles, 29));
    if (!buf) return FIGH_ERR_ALLOC;
    {
        size_t at = 0;
        for (auto &J : jobs) {
            J.tri = buf + at;
            at += (size_t)J.nwaves * J.nc * J.nc;
            for (int l = 0; l < J.nlevels; ++l) {
                J.lvl[l] = buf + at;
                at += (size_t)J.nb[l] * J.nc * J.nc;
            }
        }
    }
    // the tables: uploaded when their content changed (the pipeline passes the same structure every step)
    const size_t jb = sizeof(Tsqr2Job) * jobs.size();
    std::vector<int> maps(job_of_wave);
    size_t off_l[3];
    for (int l = 0; l < 3; ++l) {
        off_l[l] = maps.size();
        maps.insert(maps.end(), job_of_wg[l].begin(), job_of_wg[l].end());
    }
    const size_t mb = sizeof(int) * maps.size();
    char *dev = static_cast<char *>(workspace(jb + mb + 64, 30));
    if (!dev) return FIGH_ERR_ALLOC;
    static std::vector<char> cached;
    static const char *cached_dev = nullptr;
    std::vector<char> blob(jb + mb);
    std::memcpy(blob.data(), jobs.data(), jb);
    std::memcpy(blob.data() + jb, maps.data(), mb);
    if (cached_dev != dev || cached != blob) {
        FIGH_HIP(hipMemcpyAsync(dev, blob.data(), blob.size(), hipMemcpyHostToDevice, stream()));
        FIGH_HIP(hipStreamSynchronize(stream()));  // (blob is host memory of this call)
        cached = blob;
        cached_dev = dev;
    }
    const Tsqr2Job *d_jobs = reinterpret_cast<const Tsqr2Job *>(dev);
    const int *d_maps = reinterpret_cast<const int *>(dev + jb);
    // LDS of one level-0 wave: the packed triangle of the widest job (64 doubles of scratch in front)
    const int padm = 64 - max_nc;
    size_t skipm = 0;
    for (int kp = 0; kp < padm; ++kp) skipm += 16 * (4 - (kp >> 4));
    const size_t lds = sizeof(double) * (64 + 256 * 10 - skipm);

    long max_tiles = 1;
    for (auto &J : jobs) max_tiles = std::max(max_tiles, (J.rows + 63) / 64);  // (enough for either tile height)
    static size_t zeroed = 0;
    const size_t zneed = sizeof(int) * (size_t)(max_tiles + 1);
    int *zeros = static_cast<int *>(workspace(zneed, 31));
    if (!zeros) return FIGH_ERR_ALLOC;
    if (zeroed < zneed) {  // ((re)allocated: workspace() grows by 25 %)
        FIGH_HIP(hipMemsetAsync(zeros, 0, zneed, stream()));
        zeroed = zneed;
    }
    {
        ProfileScope scope("tsqr_group");
        hipLaunchKernelGGL(tsqr2_group_kernel, dim3((unsigned)job_of_wave.size()), dim3(64), lds, stream(), d_jobs, d_maps,
                           (const int *)zeros, null_pivot_sq());
        for (int l = 0; l < 3; ++l)
            if (!job_of_wg[l].empty())
                hipLaunchKernelGGL(tsqr_coop_group_kernel, dim3((unsigned)job_of_wg[l].size()), dim3(512), 0, stream(), d_jobs,
                                   d_maps + off_l[l], l);
        if (later) {
            later->jobs = d_jobs;
            later->njobs = njobs;
            later->ncfull = ncfull;
            later->nfull = nfull;
        } else {
            hipLaunchKernelGGL(embed_group_kernel, dim3((unsigned)njobs), dim3(256), 0, stream(), d_jobs, ncfull, nfull);
        }
    }
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}

int launch_tsqr_group_embed(const GroupEmbed &e) {
    if (e.njobs == 0) return FIGH_OK;
    hipLaunchKernelGGL(embed_group_kernel, dim3((unsigned)e.njobs), dim3(256), 0, stream(),
                       static_cast<const Tsqr2Job *>(e.jobs), e.ncfull, e.nfull);
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}

}  // namespace figh
