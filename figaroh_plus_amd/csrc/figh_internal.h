// Internal declarations shared by the translation units of libfigh.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdint>
#include <cstdio>
#include <string>

#include "figh.h"

namespace figh {

constexpr int kMaxJoints = 64;  // flattened trees up to 63 joints + universe (human model: 41)

// Device-resident copy of the kinematic tree; every index into it is wave-uniform, so the compiler
// fetches it through the scalar cache into SGPRs.
struct DevModel {
    int njoints, nq, nv, nlinks;
    int parents[kMaxJoints];
    int jtype[kMaxJoints];
    int idx_q[kMaxJoints];
    int idx_v[kMaxJoints];
    int body_mask[kMaxJoints];
    int depth[kMaxJoints];
    double axis[kMaxJoints][3];
    double placement[kMaxJoints][12];
    double gravity[3];
};

}  // namespace figh

struct figh_model_s {
    figh::DevModel host;
    figh::DevModel *dev = nullptr;
    bool is_chain = false;  // fixed-base serial chain of revolute joints: eligible for the chain kernel
    int max_depth = 0;
    // figh_model_set_active_rows: bit j set <=> row block j (the rows of dof j, joint-torque mode of a tree) is stored by
    // figh_regressor_build_padded; the norms of diag(W^T W) always cover every row block
    unsigned long long active_rows = ~0ull;
};

namespace figh {

void set_error(const std::string &msg);
double null_pivot_sq();  // square of figh_tsqr_null_pivot_tol (0: exact zeros only)
hipStream_t stream();
int ensure_device();

// per-kernel-family hipEvent timing (figh_profile_*).  Two forms:
//   ProfileScope s("name")           events recorded on the stream around everything launched inside the scope;
//   ProfileScope s("name", true)     the dominant kernels: the event pair is handed to the ONE launch inside the scope that
//                                    goes through FIGH_LAUNCH_TIMED (hipExtLaunchKernelGGL stamps them from the dispatch
//                                    packet itself).  Event records on the stream are barrier packets of their own: four
//                                    of them per pass cost 0.1 ms of a 1.85 ms UR10 step once the pass had no host round
//                                    trips left to hide them behind.
struct ProfileScope {
    explicit ProfileScope(const char *name, bool at_launch = false);
    ~ProfileScope();
    const char *name_;
    hipEvent_t e0_ = nullptr, e1_ = nullptr;
    bool at_launch_ = false;
};
struct LaunchEvents {
    hipEvent_t start = nullptr, stop = nullptr;
};
LaunchEvents take_launch_events();  // the pair of the innermost at_launch scope (nullptr, nullptr when none / not profiling)

#define FIGH_LAUNCH_TIMED(kernel, grid, block, lds, ...)                                                        \
    do {                                                                                                        \
        const figh::LaunchEvents ev_ = figh::take_launch_events();                                              \
        hipExtLaunchKernelGGL(kernel, grid, block, lds, figh::stream(), ev_.start, ev_.stop, 0, __VA_ARGS__);    \
    } while (0)

// scratch buffer owned by the library, grown on demand (device)
void *workspace(size_t bytes, int slot);

// A second, high-priority stream for LATENCY-BOUND chains of small launches (the merge levels of a TSQR: a handful of
// workgroups per launch, 45 .. 500 us per level whatever the number of pairs) that do not depend on the throughput-bound
// launch the library stream runs meanwhile.  Inside the scope stream() is the side stream, which starts behind everything
// queued on the library stream so far; finish() switches back and returns the event that marks the end of the side work --
// the library stream waits for it (stream_wait) where it first needs the results.  Not nested.
struct SideStream {
    SideStream();
    ~SideStream();  // finish() + stream_wait() when finish() was not called
    hipEvent_t finish();
    bool ok() const { return ok_; }
    bool ok_ = false, open_ = false;
    hipStream_t main_ = nullptr;
};
void stream_wait(hipEvent_t ev);

}  // namespace figh

// internal (not part of include/figh.h, not exported by libfigh.so: hidden visibility): level 0 of the TSQR only, see
// figh_linalg.hip
#define FIGH_INTERNAL extern "C" __attribute__((visibility("hidden")))
FIGH_INTERNAL int figh_tsqr_level0(const double *d_W, int64_t rows, int64_t ldw, const int32_t *d_col_idx, int n,
                                   const double *d_tau, const double *h_block_weight, int nblocks, double *d_tri_out,
                                   int64_t capacity, int64_t *count_out, double **ws_out);
FIGH_INTERNAL int64_t figh_tsqr_level0_capacity(int nc);
FIGH_INTERNAL int figh_tsqr_hint_begin(const int32_t *h_first_col, int nfirst, int64_t rows, int n, int nc);
FIGH_INTERNAL void figh_tsqr_hint_end(void);

namespace figh {

// figh_regressor_tree.hip: K1' for kinematic trees (tape-driven); *colsq_done = 1 when diag(W^T W) was fused
int launch_regressor_tree(const figh_model_s *m, int mode, int flags, int ft_mask, long N, const double *q,
                          const double *v, const double *a, double *W, long ldw, int ncols, int link_stride,
                          double *d_colsq, int *colsq_done);
void forget_tapes(const figh_model_s *m);
// link -> segment position of the link-compact layout (FIGH_FLAG_LINK_COMPACT), -1 = no segment; returns the number of links
// with a segment or -1 when the layout does not apply
int tree_link_positions(const figh_model_s *m, int mode, int flags, int ft_mask, int *pos);
// leading dimension of the force region of the force-compact layout (FIGH_FLAG_FORCE_COMPACT), 0 when it does not apply
long tree_force_ld(const figh_model_s *m, int mode, int flags, int ft_mask);

// figh_tsqr_wide.hip: the blocked (compact-WY, MFMA) level for nc > 80 columns
long tsqr_wide_workgroups(int nc, int cus);
int launch_tsqr_wide(const double *W, long rows, long ldw, const int *col_idx, int n, const double *tau,
                     const double *d_blkw, long rows_per_blk, int nc, long nwg, double *Rws_out, int chain_flags = 0);
// figh_linalg.hip: the NEXT figh_tsqr_level0 call (more than 80 columns) runs with exactly `wgs` workgroups and these chain
// flags (launch_tsqr_wide); consumed by that call
void tsqr_level0_chain(long wgs, int chain_flags);
// figh_linalg.hip, external wrench on a free-flyer root (figh_tsqr_selected_wrench): the kept columns that can be non-zero
// in force rows (d_fsel: n columns, then n positions in the kept list) and the force rows' triangle over all kept columns
int split_force_columns(const int *d_kept, int n, int link_stride, int *d_fsel, int force_compact = 0);
int embed_force_triangle(const double *d_Rf, int ncf, int nf, const int *d_fpos, int nc, int n, double *d_out);
// figh_tsqr_group.hip: the narrow row blocks of a tree's joint-torque regressor as jobs of grouped launches
struct Tsqr2Job {
    const double *W;       // the block's rows (its own matrix in the block-compact layout)
    const double *tau;     // nullable
    const int *col_idx;    // n device columns of W
    const int *pos;        // their positions in the kept column list (embedding)
    double *tri;           // level-0 triangles of this job: nwaves x nc x nc
    double *lvl[3];        // outputs of the merge levels (nb[l] x nc x nc each)
    double *out;           // the job's rows of the compact stack (nc rows of ncfull doubles)
    long rows, ldw;
    int n, nc;             // columns, columns + tau
    int wave0, nwaves;     // level 0: waves [wave0, wave0 + nwaves) of the launch
    int wg0[3], nb[3];     // merge level l: workgroups [wg0[l], wg0[l] + nb[l]) of its launch (nb[l] == 0: level unused)
    int nlevels;
    int tall;              // level 0 in the one-chunk form with 256-row tiles (nc <= 16)
};
// figh_tsqr_wide_pair.hip: one pair-merge level of SEVERAL stacks (the wide row blocks of a tree's regressor) in one launch
struct WyPairJob {
    const double *stack;  // `count` compact nc x nc triangles
    double *Rblk;         // packed blocks of the job's workgroups
    double *Rout;         // (count + 1) / 2 triangles
    long count;
    int nc;
    int wg0;              // the job's workgroups are [wg0, wg0 + (count + 1) / 2) of the launch
};
}  // namespace figh
#include <vector>
namespace figh {
struct WyPairStack {  // one stack to reduce to one triangle
    const double *tri;
    long count;
    int nc;
    double *out;
};
int reduce_wide_stacks(std::vector<WyPairStack> &stacks);
// (later: the embedding launch -- the only launch that writes outside the group's own workspaces -- is not queued but
// described there, for launch_tsqr_group_embed to queue once what it must follow has been queued)
struct GroupEmbed {
    const void *jobs = nullptr;
    int njobs = 0, ncfull = 0, nfull = 0;
};
int launch_tsqr_group(std::vector<Tsqr2Job> &jobs, int ncfull, int nfull, int cus, GroupEmbed *later = nullptr);
int launch_tsqr_group_embed(const GroupEmbed &e);
// figh_linalg.hip: stack of `count` compact nc x nc triangles -> one; tol_qr >= 0: + rank decision and regrouped rows
// ((nc + 1) x nc doubles, layout in figh.h, figh_tsqr_selected), else the plain triangle
int tsqr_reduce_stack(const double *d_Rs, long count, int nc, int n_free, double tol_qr, double *d_out);
// figh_tsqr_wide_pair.hip: one pair-merge level, `count` stacked triangles -> (count + 1) / 2
int launch_tsqr_wide_pairs(const double *stack, long count, int nc, double *Rws_out);
int launch_tsqr_wide_single(const double *W, long rows, long ldw, const int *col_idx, int n, int nc, double *R_out);
// figh_tsqr_wide_batch.hip: B matrices (row segments of one joint-major regressor) in one launch, wgs triangles each
int launch_tsqr_wide_batch(const double *W, long ldw, const int *col_idx, int n, int nc, long B, long n_per, int rps,
                           long seg_stride, long wgs, double *Rws_out);
int tsqr_wide_tile_rows(int nc);
int launch_tsqr_wide_chain(const double *W, long rows, long ldw, const int *col_idx, int n, const double *tau,
                           const double *d_blkw, long rows_per_blk, int nc, long nwg, double *Rws_out);

// figh_tsqr_tree.hip: every merge level of the register-tile TSQR (nc <= 80) in one launch: `count` stacked triangles ->
// the plain triangle d_out; d_rows_out != nullptr appends the rank decision (columns k < n_free, threshold tol) and the
// regrouped factorisation ((nc + 1) x nc doubles, layout in figh.h).  count == 0: Rs already is the plain triangle.
// FIGH_ERR_UNSUPPORTED when the stack is too tall for one resident grid.
int launch_tsqr_tree(const double *Rs, long count, int nc, int n_free, double tol, double *d_out, double *d_rows_out);

// figh_tsqr_stream.hip: the same contract as launch_tsqr_tree, with the levels, the rank decision and the regrouped
// factorisation software-pipelined (about nc column steps in total).  FIGH_ERR_UNSUPPORTED when the tree does not fit one
// resident grid.
int launch_tsqr_stream(const double *Rs, long count, int nc, int n_free, double tol, double *d_out, double *d_rows_out);

#define FIGH_HIP(expr)                                                                          \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            figh::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                 \
            return FIGH_ERR_NO_DEVICE;                                                          \
        }                                                                                       \
    } while (0)

#define FIGH_REQUIRE(cond, msg)              \
    do {                                     \
        if (!(cond)) {                       \
            figh::set_error(msg);            \
            return FIGH_ERR_INVALID;         \
        }                                    \
    } while (0)

}  // namespace figh
