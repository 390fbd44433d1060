/* figh.h -- C-ABI of libfigh.so: the MI355X (gfx950) regressor + QR/LS hot path of FIGAROH.
 *
 * The reference (thanhndv212/figaroh-plus) is pure Python and has no FFI; its boundary for this path is
 * the set of module-level functions listed below.  Each entry point names the
 * reference statement(s) it replaces (paths relative to the reference root).  The Python shim in
 * figaroh_plus_amd/ binds exactly these symbols with ctypes; INTEGRATION.md shows the stub a FIGAROH
 * maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative figh_status; figh_last_error() gives the text;
 *   - all matrices are float64, row-major ("C order"), leading dimension in elements;
 *   - pointers named d_* are DEVICE pointers obtained from figh_malloc (or any hipMalloc), pointers named
 *     h_* are host pointers; buffers are caller-owned, inputs are never written;
 *   - the library keeps one HIP stream per process (figh_stream); calls are asynchronous on it unless they
 *     return data to the host; it is not thread-safe;
 *   - there is NO CPU implementation behind this ABI: without a HIP device every compute entry fails
 *     with FIGH_ERR_NO_DEVICE.
 */
#ifndef FIGH_H
#define FIGH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    FIGH_OK = 0,
    FIGH_ERR_INVALID = -1,    /* bad argument (the reference would raise ValueError / AssertionError) */
    FIGH_ERR_NO_DEVICE = -2,  /* no HIP device / HIP runtime error */
    FIGH_ERR_ALLOC = -3,
    FIGH_ERR_UNSUPPORTED = -4,
    FIGH_ERR_COMM = -5
} figh_status;

/* joint types of figh_model_create (Pinocchio joint models reached through URDF) */
enum { FIGH_JT_REVOLUTE = 0, FIGH_JT_PRISMATIC = 1, FIGH_JT_CONTINUOUS = 2, FIGH_JT_FREEFLYER = 3, FIGH_JT_UNIVERSE = -1 };
/* regressor modes: param["is_joint_torques"] / param["is_external_wrench"] (regressor.py:45, :89) */
enum { FIGH_MODE_JOINT_TORQUE = 0, FIGH_MODE_EXT_WRENCH = 1 };
/* flags: param["has_friction"], ["has_actuator_inertia"], ["has_joint_offset"] (regressor.py:55-70, :144-169);
 * FIGH_FLAG_TX40 appends the 3 coupling columns of add_coupling_TX40 (regressor.py:198-227);
 * FIGH_FLAG_GENERIC forces the generic-tree kernel even when the serial-chain kernel applies (tests). */
enum { FIGH_FLAG_FRICTION = 1, FIGH_FLAG_ACT_INERTIA = 2, FIGH_FLAG_OFFSET = 4, FIGH_FLAG_TX40 = 8, FIGH_FLAG_GENERIC = 256,
       /* d_q / d_v / d_a are TILE-BLOCKED copies made by figh_repack_samples ([tile of 64 samples][value][lane]) instead of
        * the reference's sample-major arrays: the generic-tree kernel then reads one 512-byte line per value and wave
        * (sample-major, a lane's values lie nq * 8 bytes from its neighbour's and every 8-byte load pulls a line of its
        * own: 7.4x the algorithmic input bytes on TALOS).  Tree models only. */
       FIGH_FLAG_BLOCKED_INPUTS = 512,
       /* figh_regressor_build_padded only, OPT-IN: the caller guarantees that the structural zeros of W (the entries of a
        * row block outside its joint's subtree, joint-torque mode of a tree) already ARE zero in d_W -- a buffer it
        * zero-filled once and that nothing but this function writes -- so the kernel does not stream them again; every
        * entry that depends on q, v, a is written as always.  90 % of TIAGo's 73.7 GB are such zeros.  Not the default
        * of anything: a pass that re-creates every byte of W each time is what the pipeline and bench.py measure unless
        * asked otherwise. */
       FIGH_FLAG_ZEROS_PRESENT = 1024,
       /* figh_regressor_build_padded only: d_W is BLOCK-COMPACT -- the joint-torque regressor of a tree of single-dof joints
        * stored row block by row block, block j (the N rows of joint j + 1) as its own N x ld_j matrix holding the columns
        * of that joint's subtree (links j + 1 .. j + size_j, depth-first numbering: a contiguous window of the dense
        * link-padded row, ld_j = 16 size_j) at element offset N * (ld_0 + .. + ld_{j-1}).  Everything outside the window is
        * a structural zero that is neither stored nor read (figh_tsqr_selected_blocks with block offsets).  ldw is ignored;
        * d_colsq keeps the reference's column numbering.  TIAGo: 7.9 instead of 73.7 GB per 1e6 samples. */
       FIGH_FLAG_COMPACT_BLOCKS = 2048,
       /* figh_regressor_build_padded only, external-wrench regressor of a model with a free-flyer root: d_W is LINK-COMPACT --
        * only the links that can have a non-zero entry at all keep their 16-column segment (a body with mass and a selected
        * wrench component, or every link when a friction / inertia / offset flag is set; regressor.py:36-39, :142-169), in
        * link order: segment of link l at column 16 pos[l], ldw >= 16 * (number of such links), pos from
        * figh_regressor_link_layout.  The other links' columns -- structural zeros in all six row blocks, eliminated by
        * get_index_eliminate (regressor.py:258-279) whatever the samples -- are neither written nor read: human model
        * 19 x 16 = 304 instead of 40 x 16 = 640 columns (146 instead of 307 GB for 1e7 samples).  d_colsq keeps the
        * reference's numbering (exact zeros for the dropped links).  Consumers take the map as d_link_pos
        * (figh_tsqr_selected_wrench). */
       FIGH_FLAG_LINK_COMPACT = 4096,
       /* figh_regressor_build_padded only, external-wrench regressor of a model with a free-flyer root, no friction /
        * actuator-inertia / offset columns: d_W is FORCE-COMPACT -- the three FORCE row blocks (regressor.py:89-192: rows
        * c N + i, c = Fx Fy Fz) are stored in a region of their own, 3 N rows of ldf columns IN FRONT of the three torque
        * row blocks, one 128-byte line per FOUR links: the force rows of link l hold mx my mz m (reference columns 14 l + 6
        * .. 14 l + 9) at columns 16 (p / 4) + 4 (p % 4) + (s - 6) of that region, p = the link's position among the links
        * with entries (figh_regressor_link_layout), ldf from figh_regressor_force_layout -- everything a force row has:
        * the six rotational-inertia entries are exact zeros (a force does not depend on them).  The torque row blocks
        * follow at d_W + 3 N ldf: 3 N rows of ldw columns, link-padded (or link-compact with FIGH_FLAG_LINK_COMPACT).  A
        * quarter of the force-row lines: TALOS 101 -> 64 GB, human 146 -> 92 GB per pass.  Consumer:
        * figh_tsqr_selected_wrench(..., ld_force). */
       FIGH_FLAG_FORCE_COMPACT = 8192 };

typedef struct figh_model_s *figh_model_t;

/* ------------------------------------------------------------------ runtime plumbing (no reference equivalent) */
/* The ABI this header declares.  Bumped whenever an entry point changes its argument list or the meaning of an argument
 * (round 5 did so for figh_tsqr_selected_wrench / figh_regressor_build_padded without a bump: a library of the older ABI
 * accepts the longer argument list under cdecl and silently ignores the new arguments).  figaroh_plus_amd/_lib.py refuses a
 * library -- in-tree or FIGH_LIB_PATH -- whose figh_version() differs from the value it was written against. */
#define FIGH_ABI_VERSION 106
int figh_version(void);
const char *figh_last_error(void);
int figh_device_count(int *count);
int figh_device_set(int device);
/* How the host waits for the device in this process: 0 (default) = the runtime's spinning wait, lowest latency, one busy
 * CPU per process; 1 = interrupt-driven (hipDeviceScheduleBlockingSync).  For several ranks per node under a CPU quota
 * (figaroh_plus_amd/dist.py launches): eight spinning ranks plus their runtime threads reach a 16-CPU cgroup quota, and a
 * process that exhausts it is frozen until the next 100 ms accounting period.  Call before figh_device_set / any
 * device call.  No reference analogue (the reference has no device). */
int figh_host_wait_mode(int blocking);
/* Null-pivot rule of the Householder TSQR kernels (figh_tsqr*, figh_regressor_tsqr*): a pivot column whose norm at and below
 * the diagonal is <= tol -- sqrt(R_kk^2 + |x|^2) <= tol for the tile's part x of the column -- gets H = I (LAPACK dlarfg's
 * rule for an exactly zero x, with a threshold): its norm is folded into R_kk, the column leaves the tile, no reflector
 * is formed and no trailing update is made.  R_kk therefore carries the running residual norm of the column and the test
 * includes it, so that INSIDE ONE LEVEL-0 TRIANGLE at most tol of a column's norm is ever folded.  Every level-0 triangle
 * -- one per wave (register-tile kernel), per workgroup (blocked kernel), per consumer wave (fused launch) and per sample
 * chunk of a streamed run -- starts from R_kk = 0 and makes the test on its own rows; the merge levels fold nothing.  With
 * T level-0 triangles R is therefore the exact R factor of W + E with |E[:, k]|_2 <= sqrt(T) tol for such columns (the
 * folded parts lie in disjoint rows), E = 0 elsewhere: T = 1536 for the fused UR10 launch, 512 for the blocked kernel,
 * i.e. up to 0.35 - 0.6 tol_qr at tol = tol_qr / 64 -- reached only by a column whose residual really is that large in
 * every triangle, and even then |R_kk| itself is exact (the norm moves into R_kk, it is not dropped); what E perturbs is
 * the component of the LATER columns along that direction.  The linearly dependent columns of a regressor
 * (qrdecomposition.py:208-221: |R_kk| <= tol_qr, 27 % of the kept columns of UR10 and 29 % of TALOS) are null in every
 * tile -- their residual is rounding noise, 1e-13 .. 1e-12 -- and cost a norm instead of a column step.  tol = 0 (the
 * default): exact zeros only.  The pipeline and the qrdecomposition mirrors set tol_qr / 64 unless told otherwise
 * (null_pivots=False: plain Householder, the reference's arithmetic step for step).  Process-wide, takes effect at the
 * next launch. */
int figh_tsqr_null_pivot_tol(double tol);
/* PCI bus id ("0000:c1:00.0") of HIP device `device`: the physical device a rank drives, whatever logical index the
 * launcher's HIP_VISIBLE_DEVICES left it with (figaroh_plus_amd/dist.py: two ranks on one GPU cannot form an RCCL
 * communicator).  Creates no context.  No reference analogue. */
int figh_device_pci_bus_id(int device, char *out, int out_len);
int figh_device_info(char *name, int name_len, int *cu_count, size_t *hbm_bytes);
int figh_malloc(void **d_ptr, size_t bytes);
int figh_free(void *d_ptr);
/* Host <-> HBM copies on the library stream; both return when the data has arrived (h2d below 1 MB: when it is queued in
 * stream order).  This is where the reference's host-resident NumPy arrays cross the boundary: q, v, a in, and the stacked
 * regressor out when a caller keeps W = build_regressor_basic(...) as an ndarray (regressor.py:20-194: 4.03 GB for UR10 at 1e6
 * samples).  Transfers of 24 MB and more between PAGEABLE memory and HBM run in 32 MB chunks through two page-locked staging
 * buffers -- the DMA of chunk k + 1 beside the host copy of chunk k, shared by eight threads (which also take a fresh
 * destination's page faults in parallel): D2H 50 GB/s, H2D 27 - 55 GB/s on the GPU box (round 5, one hipMemcpyAsync per array:
 * 24 and 7.5).  A figh_host_alloc buffer is copied by plain DMA (57 GB/s).  Smaller transfers stage through one 1 MB buffer. */
int figh_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes);
int figh_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes);
/* Page-locked host memory for results that come back every pass (the (nc + 1) x nc rows of the rank step: 0.9 MB for
 * TALOS): figh_memcpy_d2h into such a buffer is one DMA, without the staging copy and without first-touch page faults of a
 * fresh NumPy array.  No reference analogue (the reference's arrays never leave the host). */
int figh_host_alloc(void **h_ptr, size_t bytes);
int figh_host_free(void *h_ptr);
int figh_memcpy_d2d(void *d_dst, const void *d_src, size_t bytes);
int figh_memset(void *d_dst, int value, size_t bytes);
int figh_synchronize(void);
/* per-kernel live timing with hipEvents on the library stream (bench.py roofline): enable, run, then query.
 * figh_profile_enable(level): 0 off; 1 times only the dominant kernels ("regressor_chain" / "regressor_tree" /
 * "tsqr": a few event records per pass, cheap enough to stay on inside a timed region); 2 times every launch.
 * figh_profile_get: name is the kernel family ("regressor_chain", "regressor_tree", "tsqr", "colsq", ...);
 * returns launches and total milliseconds since the last figh_profile_reset(). */
int figh_profile_enable(int level);
int figh_profile_reset(void);
int figh_profile_get(const char *name, int *launches, double *total_ms);

/* ------------------------------------------------------------------ model
 * Replaces what regressor.py reads from `robot.model` (regressor.py:36-42) and what
 * pin.computeJointTorqueRegressor reads from the Pinocchio Model (call sites regressor.py:49-51, :93-95).
 * Arrays have njoints entries, joint 0 = universe.  placement: 12 doubles per joint, row-major rotation (9)
 * then translation (3), child coords -> parent coords.  body_mask[j] != 0 iff inertias[j].mass != 0
 * (id_inertias, regressor.py:36-39).  The standard-parameter VALUES are not needed by the kernels. */
int figh_model_create(int njoints, const int32_t *parents, const int32_t *jtype, const double *axis,
                      const double *placement, const int32_t *idx_q, const int32_t *idx_v, const double *gravity,
                      const int32_t *body_mask, figh_model_t *out);
int figh_model_destroy(figh_model_t model);
/* ACTIVE JOINTS (examples/tiago/identification.py:148-187, :406-424: the script builds the regressor of all 24 dofs,
 * eliminates columns on the norms of the full matrix, and then decimates, stacks and factors only the row blocks of the
 * eight joints that carry measurements, act_idxv).  h_rows: the n dof indices whose row blocks figh_regressor_build_padded
 * STORES from now on (joint-torque mode of a tree of single-dof joints; link-padded or block-compact layout -- in the
 * block-compact one the other blocks take no memory at all, ld_j = 0); d_colsq keeps covering every row block, so
 * get_index_eliminate sees the reference's norms.  n <= 0 / NULL: every row block again (the default).  The chain kernel,
 * figh_regressor_build and the external-wrench mode ignore the setting. */
int figh_model_set_active_rows(figh_model_t model, const int32_t *h_rows, int n);
/* number of rows per sample and number of columns of the stacked regressor for (mode, flags) */
int figh_regressor_shape(figh_model_t model, int mode, int flags, int *rows_per_sample, int *ncols);

/* ------------------------------------------------------------------ K1: regressor assembly
 * Replaces build_regressor_basic (regressor.py:20-194) and, with FIGH_FLAG_TX40, add_coupling_TX40
 * (regressor.py:198-227): d_W[(j*N + i)*ldw + col] in the reference's joint-major row order and FIGAROH
 * column order (14 columns per link: Ixx Ixy Ixz Iyy Iyz Izz mx my mz m Ia fv fs off).
 * d_q: N x nq, d_v / d_a: N x nv, row-major.  ft_mask: bit c set <=> wrench component c (Fx Fy Fz Mx My Mz)
 * is named in param["force_torque"] ('All' = 63); ignored in joint-torque mode.
 * d_colsq (nullable): ncols doubles, receives diag(W^T W) of this call (the quantity of regressor.py:243,271). */
int figh_regressor_build(figh_model_t model, int mode, int flags, int ft_mask, int64_t N, const double *d_q,
                         const double *d_v, const double *d_a, double *d_W, int64_t ldw, double *d_colsq);

/* figh_regressor_build_padded: the same regressor for W that STAYS ON THE DEVICE (IdentificationPipeline, the streamed
 * entry points), tree models only.  Link-padded layout: 16 columns per link -- the 14 of the reference followed by two
 * zero columns -- so reference column 14 l + s is column 16 l + s; ldw must be a multiple of 16 (>= 16 (njoints-1)) and
 * d_W 128-byte aligned: every (row, link) segment is then exactly one cache line and the kernel writes whole lines
 * (4.4-5.0 TB/s against 2.3-2.6 TB/s for the 112-byte segments of the dense layout, rows of which are not line
 * aligned).  d_W may be NULL: column norms only.  d_colsq (nullable) is in the REFERENCE's column numbering
 * (14 (njoints-1) entries).  figh_tsqr / figh_matvec / figh_gather_cols take such a W through their column lists. */
int figh_regressor_build_padded(figh_model_t model, int mode, int flags, int ft_mask, int64_t N, const double *d_q,
                                const double *d_v, const double *d_a, double *d_W, int64_t ldw, double *d_colsq);
/* The link -> segment map of FIGH_FLAG_LINK_COMPACT for (model, mode, flags, ft_mask): h_link_pos[l] (njoints - 1 entries,
 * host) = position of link l + 1's segment, -1 for a link without one; *nlive = number of links with a segment.
 * FIGH_ERR_UNSUPPORTED when the layout does not apply (not an external-wrench regressor on a free-flyer root).  Which
 * links drop out follows from the model alone -- id_inertias of regressor.py:36-39 -- not from the samples. */
int figh_regressor_link_layout(figh_model_t model, int mode, int flags, int ft_mask, int32_t *h_link_pos, int *nlive);
/* Leading dimension of the force region of FIGH_FLAG_FORCE_COMPACT for (model, mode, flags, ft_mask): 16 * ceil(nlive / 4).
 * FIGH_ERR_UNSUPPORTED when the layout does not apply (see the flag). */
int figh_regressor_force_layout(figh_model_t model, int mode, int flags, int ft_mask, int64_t *ld_force);

/* figh_repack_samples: d_dst[(t * width + k) * 64 + l] = d_src[min(64 t + l, N - 1) * width + k] -- a sample-major N x width
 * array (q, v or a exactly as the reference holds them) re-laid per tile of 64 samples, value-major inside the tile, the
 * last tile padded with its last sample.  d_dst: ceil(N / 64) * 64 * width doubles.  The copies are what
 * FIGH_FLAG_BLOCKED_INPUTS announces; a chunk of samples [lo, lo + n) with lo % 64 == 0 starts at d_dst + lo * width, like
 * in the original. */
int figh_repack_samples(const double *d_src, int64_t N, int width, double *d_dst);

/* add_coupling_TX40 as a separate call (regressor.py:198-227), for callers that append the three columns
 * [Iam6 fvm6 fsm6] to an existing W: d_out is (6N x 3) row-major, rows in the same joint-major order; only the
 * rows of joints 5 and 6 are non-zero: [a6 v6 sign(v5+v6)] and [a5 v5 sign(v5+v6)].  d_v, d_a: N x nv, nv >= 6. */
int figh_coupling_tx40(int64_t N, int nv, const double *d_v, const double *d_a, double *d_out);

/* ------------------------------------------------------------------ K2: column norms / gathers / residuals
 * figh_colsq: diag(W^T W) of a materialised matrix -- np.diag(np.dot(W.T, W)) at regressor.py:243 and :271. */
int figh_colsq(const double *d_W, int64_t rows, int cols, int64_t ldw, double *d_out);
/* figh_gather_cols: out[:, c] = W[:, col_idx[c]] -- np.delete(W, idx_e, 1) (regressor.py:254,292) and the
 * column copies of build_baseRegressor / get_baseParams (qrdecomposition.py:223-236, :299-313). */
int figh_gather_cols(const double *d_W, int64_t rows, int64_t ldw, const int32_t *d_col_idx, int n, double *d_out,
                     int64_t ldo);
/* figh_place_block: dst[r][c] = scale * src[r][c], a rows x cols block between two row-major device matrices (the
 * caller offsets the pointers) -- the np.concatenate / unary-minus / np.zeros statements that assemble W_tot in
 * build_total_regressor_current / _wrench (regressor.py:316-412, :446-490). */
int figh_place_block(const double *d_src, int64_t ld_src, int64_t rows, int64_t cols, double scale, double *d_dst,
                     int64_t ld_dst);
/* figh_matvec: y = W[:, col_idx] . x  (tau_base = np.dot(W_b, phi_b), e.g. staubli_TX40/identification.py:244) */
int figh_matvec(const double *d_W, int64_t rows, int64_t ldw, const int32_t *d_col_idx, int n, const double *d_x,
                double *d_y);
/* figh_block_sqnorm: out[b] = sum over the b-th block of `rows/nblocks` consecutive entries of (a - b)^2;
 * the per-joint residual norms of the WLS weights (identification_tools.py:312-315,
 * staubli_TX40/identification.py:310-313) and the residual of relative_stdev (identification_tools.py:222). */
int figh_block_sqnorm(const double *d_a, const double *d_b, int64_t rows, int nblocks, double *d_out);

/* ------------------------------------------------------------------ K3: tall-skinny Householder QR
 * Replaces the np.linalg.qr calls of qrdecomposition.py:105,205,238,286 for everything the reference uses
 * them for (|diag R| rank test, R1/R2 regrouping, Q1^T tau): Householder TSQR of the gathered columns
 * W[:, col_idx] (n of them), optionally with tau appended as one more column and with per-row-block weights
 * (row r is scaled by h_block_weight[r / (rows/nblocks)]; NULL = unweighted) -- the weighted LS of
 * staubli_TX40/identification.py:305-327 and identification_tools.py:317-325 is the QR of the scaled rows.
 * d_R_out: nc x nc row-major upper triangle, nc = n + (d_tau ? 1 : 0); with tau its last column holds
 * Q^T tau (first n entries) and the residual norm (entry nc-1).  R is defined up to row signs, like LAPACK's. */
int figh_tsqr(const double *d_W, int64_t rows, int64_t ldw, const int32_t *d_col_idx, int n, const double *d_tau,
              const double *h_block_weight, int nblocks, double *d_R_out);
/* figh_tsqr with a structure hint: the rows are nfirst consecutive blocks of rows/nfirst rows, and in block b the
 * gathered columns 0 .. h_first_col[b]-1 are EXACTLY zero (the caller's guarantee; 0 <= h_first_col[b] <= n).  That is
 * the shape of build_regressor_basic's W in joint-torque mode (regressor.py:45-87): row block j (the rows of joint j)
 * only involves the links of j's subtree, i.e. the columns from 14 j on.  The register-tile kernel (n <= 79) then does
 * not read those entries at all (41 % of the reads for UR10).  Same result as figh_tsqr. */
int figh_tsqr_structured(const double *d_W, int64_t rows, int64_t ldw, const int32_t *d_col_idx, int n,
                         const double *d_tau, const double *h_block_weight, int nblocks, const int32_t *h_first_col,
                         int nfirst, double *d_R_out);
/* Order-preserving row rejection (examples/staubli_TX40/identification.py:207-233, examples/tiago/identification.py:170-187:
 * samples whose joint velocity is below a threshold are removed from a joint's block of W and tau): the rows r of d_W
 * (rows x cols, leading dimension ldw) with |d_W[r, key_col]| >= threshold are copied, in order, to d_W_out (leading
 * dimension ld_out; room for `rows` rows) and, if given, d_tau[r] to d_tau_out; *h_count_out = the number of kept rows
 * (the call synchronises to return it). */
int figh_compact_rows(const double *d_W, int64_t rows, int cols, int64_t ldw, const double *d_tau, int key_col,
                      double threshold, double *d_W_out, int64_t ld_out, double *d_tau_out, int64_t *h_count_out);
/* The rank decision and the regrouped column order on the device (qrdecomposition.py:215-236): d_perm receives
 * [i : |R_ii| > tol_qr, ascending] followed by the remaining i < n, followed by n .. nc-1 (the tau column); the regrouped
 * factorisation qr([W1 W2 tau]) is then figh_tsqr(d_R, nc, nc, d_perm, nc, NULL, ...) with no host round trip. */
int figh_base_permutation(const double *d_R, int nc, int n, double tol_qr, int32_t *d_perm);
/* Sample-sharded reduction step: QR of `count` stacked nc x nc R factors (d_Rs: count*nc x nc) into one. */
int figh_tsqr_merge(const double *d_Rs, int count, int nc, double *d_R_out);

/* ---- the elimination and the base-parameter factorisation without a host round trip (IdentificationPipeline)
 * figh_select_columns: get_index_eliminate (regressor.py:258-279) on the device.  d_sel (2 + 2 ncols int32):
 * [0] the number of kept columns {c : not (colsq[c] < tol_e)}, [1] ncols, [2 .. 2+ncols) the kept columns in W's own
 * numbering (link_stride 14: the reference layout; 16: the link-padded layout, column 16 (c / 14) + c % 14), zeros behind
 * them, [2+ncols .. 2+2 ncols) the 0/1 mask in the reference's numbering. */
int figh_select_columns(const double *d_colsq, int ncols, double tol_e, int link_stride, int32_t *d_sel);
/* figh_tsqr_selected: figh_select_columns followed by the TSQR of W[:, kept] (+ tau), launched back to back: the column
 * list never visits the host.  The launch shape needs the column COUNT, which the caller supplies as n_expected (the
 * count of the previous pass; <= 0: selection only) and verifies afterwards against d_sel[0] -- on a mismatch the result
 * is to be discarded and the call repeated with the right count (entries of the list behind the device's count are 0,
 * so a stale count reads valid columns).  nblocks > 0: the structure hint of figh_tsqr_structured, derived from the list
 * on the device (row block b of rows/nblocks rows only involves the reference columns >= 14 b).
 * tol_qr < 0: d_R_out is the plain nc x nc triangle of figh_tsqr.
 * tol_qr >= 0: the rank decision and the regrouped factorisation of get_baseParams / double_QR
 * (qrdecomposition.py:205-244, :105-160) follow in the same launch as the last merge level.  d_R_out then is
 * (nc + 1) x nc: rows 0 .. nc-1 hold, in the ORIGINAL column order, row k = the row of qr([W1 W2 tau]) that belongs to
 * base column k (W1 = the columns with |R_kk| > tol_qr in the plain factorisation qr([W_e tau]), W2 the others), i.e.
 * R1 = out[base][:, base] (its upper triangle), R2 = out[base][:, rest], Q1^T tau = out[base][:, n]; the rows of
 * dependent columns are zero, and with exactly one extra column (tau) out[n][n] = || tau - W1 phi ||, the residual of the
 * least-squares problem in base coordinates.  Row nc is the diagonal of the plain factorisation: the numbers the rank
 * decision was taken on. */
int figh_tsqr_selected(const double *d_W, int64_t rows, int64_t ldw, const double *d_colsq, int ncols, double tol_e,
                       int link_stride, int nblocks, int n_expected, const double *d_tau, double tol_qr, int32_t *d_sel,
                       double *d_R_out);
/* figh_regressor_build (+ column norms) and the level-0 TSQR of W[:, d_kept] [+ tau] in ONE launch, for fixed-base serial
 * chains in joint-torque mode (UR10; regressor.py:20-194 with the pin.computeJointTorqueRegressor loop :45-87,
 * get_index_eliminate's norms :258-279, np.linalg.qr of get_baseParams qrdecomposition.py:205): W is written to HBM exactly
 * as figh_regressor_build writes it (ldw must equal the column count, d_W 16-byte aligned), d_colsq receives diag(W^T W),
 * and every 64-row tile is factored while it is still in the LDS of the CU that produced it -- the TSQR does not read W
 * back.  d_kept: the n kept columns (reference numbering, ascending) the caller expects -- those of the previous pass; the
 * caller verifies them afterwards against d_colsq (figh_select_columns) and repeats the pass through figh_regressor_build +
 * figh_tsqr_selected on a mismatch.  d_R_out as in figh_tsqr_selected (tol_qr < 0: the plain nc x nc triangle; tol_qr >= 0:
 * (nc + 1) x nc rows of the regrouped factorisation + the plain diagonal).  The tiles of a CU are handed to whichever
 * consumer wave is free, so R is reproduced up to rounding (not bit for bit) from run to run; W and d_colsq are
 * bit-reproducible.  FIGH_ERR_UNSUPPORTED (nothing launched): not a 6-joint chain, TX40 coupling columns, more than 64
 * columns with tau, fewer than 4096 samples, padded or unaligned W. */
int figh_regressor_tsqr_fused(figh_model_t model, int flags, int64_t N, const double *d_q, const double *d_v,
                              const double *d_a, double *d_W, int64_t ldw, double *d_colsq, const int32_t *d_kept, int n,
                              const double *d_tau, double tol_qr, double *d_R_out);
/* figh_tsqr_selected for the external-wrench regressor of a model with a free-flyer root (regressor.py:89-192: six row
 * blocks of rows / 6 rows, force components first): in the force rows the six rotational-inertia columns of every link
 * are exact zeros, so those rows are factored over the nf_expected kept columns with slot >= 6 only (2 m nf^2 instead of
 * 2 m n^2 flops for half of W), and the torque rows by a launch whose first workgroup starts from the force rows'
 * triangle.  Same outputs as figh_tsqr_selected (R is the R factor of the whole W[:, kept | tau]); nf_expected is the
 * number of kept columns c with c % 14 >= 6 -- the caller derives it from the same mask as n_expected and verifies both
 * afterwards.  nf_expected <= 0, at most 80 columns or rows % 6 != 0: plain figh_tsqr_selected.  d_link_pos (nullable,
 * device, njoints - 1 int32): d_W is link-compact (FIGH_FLAG_LINK_COMPACT), reference column 14 l + s is column
 * 16 d_link_pos[l] + s.  ld_force > 0: d_W is force-compact (FIGH_FLAG_FORCE_COMPACT) -- rows / 2 force rows of ld_force columns
 * at d_W, the torque rows of ldw columns behind them; needs the split (nf_expected > 0, more than 80 columns). */
int figh_tsqr_selected_wrench(const double *d_W, int64_t rows, int64_t ldw, const double *d_colsq, int ncols, double tol_e,
                              int link_stride, int n_expected, int nf_expected, const double *d_tau, double tol_qr,
                              int32_t *d_sel, double *d_R_out, const int32_t *d_link_pos, int64_t ld_force);
/* figh_tsqr_selected for the joint-torque regressor of a tree of single-dof joints (regressor.py:45-87, rows j*N + i): row
 * block j only involves the links of joint j's subtree and the Ia / fv / fs / off columns of link j itself; all other
 * entries are structural zeros.  Row block j (rows / nblocks rows) is factored over its own column list -- h_counts[j]
 * entries of d_cols (device column numbering) and d_pos (positions in the kept list), concatenated block after block,
 * built by the caller from the kept mask it expects (and verified against d_sel afterwards) -- reduced, embedded into the
 * kept column set, and the nblocks triangles are merged.  Same outputs as figh_tsqr_selected.  h_block_off / h_block_ld
 * (host, nblocks entries each, both or neither): the block-compact W of FIGH_FLAG_COMPACT_BLOCKS -- row block j is the
 * rows / nblocks x h_block_ld[j] matrix at d_W + h_block_off[j] and d_cols index ITS columns; ldw is then unused.
 * The nblocks embedded triangles are stacked compactly -- block j contributes its h_counts[j] (+ 1 with tau) rows over the
 * nc kept columns [+ tau] -- and factored as one small tall matrix.  d_block_tri (nullable; room for
 * (sum_j (h_counts[j] + 1) + nc + 1) nc doubles) receives that stack, for a weighted solve afterwards
 * (figh_block_rows_residuals, then figh_tsqr over its rows with one weight per row: W itself is not read again).
 * h_counts[j] == -1: row block j is INACTIVE -- neither its rows of W nor its rows of tau take part (the joints without
 * measurements of examples/tiago/identification.py:148-187; figh_model_set_active_rows keeps them out of W as well); it has
 * no entries in d_cols / d_pos and no rows in the stack. */
int figh_tsqr_selected_blocks(const double *d_W, int64_t rows, int64_t ldw, const double *d_colsq, int ncols, double tol_e,
                              int link_stride, int n_expected, int nblocks, const int32_t *h_counts, const int32_t *d_cols,
                              const int32_t *d_pos, const int64_t *h_block_off, const int32_t *h_block_ld,
                              const double *d_tau, double tol_qr, int32_t *d_sel, double *d_R_out, double *d_block_tri);
/* Per-row-block residual norms from the compact stack figh_tsqr_selected_blocks leaves in d_block_tri: block b occupies the
 * rows [h_row_off[b], h_row_off[b+1]) (h_row_off[b] = sum over the blocks in front of b of their column count + 1 for tau;
 * nblocks + 1 entries), every row over the nc = n + 1 kept columns [+ tau].  With v = [phi over the kept columns; -1],
 * d_r2[b] = sum of (row . v)^2 = || tau_b - W_b phi ||^2 -- the per-joint variances of the weighted least squares
 * (examples/staubli_TX40/identification.py:305-316, identification_tools.py:291-331) without another pass over W. */
int figh_block_rows_residuals(const double *d_rows, int nblocks, const int32_t *h_row_off, int nc, const double *d_v,
                              double *d_r2);
/* figh_tsqr_merge followed by the rank decision and the regrouped factorisation as in figh_tsqr_selected (the cross-rank
 * reduction of the all-gathered per-rank triangles): columns k < n_free take part in the rank decision, the others (tau)
 * always count as base columns.  d_rows_out: (nc + 1) x nc. */
int figh_tsqr_merge_base(const double *d_Rs, int count, int nc, int n_free, double tol_qr, double *d_rows_out);

/* ------------------------------------------------------------------ streamed entry points (W never stored in full)
 * The "fused" forms of SURVEY.md section 8b: the samples are processed in chunks of `chunk_samples` (0 = library
 * default, about 2 GB of W); each chunk's W lives in a library workspace only until the next kernel has consumed it.
 * Same mode / flags / ft_mask / q, v, a conventions as figh_regressor_build.  d_tau keeps the reference's layout
 * (rows_per_sample * N entries, row j*N + i).
 *
 * figh_regressor_colsq: diag(W^T W) of the W that build_regressor_basic would return (regressor.py:243,271). */
int figh_regressor_colsq(figh_model_t model, int mode, int flags, int ft_mask, int64_t N, const double *d_q,
                         const double *d_v, const double *d_a, int64_t chunk_samples, double *d_colsq);
/* figh_regressor_tsqr: the R factor that figh_tsqr would return for that W (columns d_col_idx, optional tau column and
 * per-row-block weights; nblocks must divide rows_per_sample, e.g. one weight per joint) -- qrdecomposition.py:105,
 * 205,238,286 on top of regressor.py:20-227 without the 6N x ncols intermediate. */
int figh_regressor_tsqr(figh_model_t model, int mode, int flags, int ft_mask, int64_t N, const double *d_q,
                        const double *d_v, const double *d_a, const int32_t *d_col_idx, int n, const double *d_tau,
                        const double *h_block_weight, int nblocks, int64_t chunk_samples, double *d_R_out);
/* figh_regressor_tsqr_norms: figh_regressor_tsqr that also returns diag(W^T W) of ALL columns (d_colsq_out, ncols doubles,
 * the reference's numbering) from the same pass -- the norms are fused into the regressor kernel of every chunk.  For a
 * caller that factors the columns it expects get_index_eliminate (regressor.py:258-279) to keep, e.g. the set of the
 * previous pass, and verifies the set afterwards: one pass over the samples instead of two. */
int figh_regressor_tsqr_norms(figh_model_t model, int mode, int flags, int ft_mask, int64_t N, const double *d_q,
                              const double *d_v, const double *d_a, const int32_t *d_col_idx, int n, const double *d_tau,
                              const double *h_block_weight, int nblocks, int64_t chunk_samples, double *d_R_out,
                              double *d_colsq_out);
/* figh_regressor_tsqr_batch: B independent trajectories of n_per samples each (d_q, d_v, d_a hold the B * n_per samples
 * back to back) -> B triangles, d_R_out[b] = the n x n R factor of W[:, d_col_idx] of trajectory b.  With d_R_stack (one
 * n x n triangle, nullable) every result is the R factor of vstack((W_stack, W_b)) instead, R_stack being the triangle
 * of W_stack.  This is one finite-difference gradient of the excitation objective, np.linalg.cond(W_b) per perturbed
 * trajectory (examples/tiago/optimal_trajectory.py:100-133, 296-313), in one K1 launch + one batched TSQR launch + the
 * pair-merge levels for more than 80 columns (trajectory by trajectory through figh_regressor_tsqr otherwise). */
int figh_regressor_tsqr_batch(figh_model_t model, int mode, int flags, int ft_mask, int64_t B, int64_t n_per,
                              const double *d_q, const double *d_v, const double *d_a, const int32_t *d_col_idx, int n,
                              const double *d_R_stack, double *d_R_out);
/* figh_regressor_gram: h_G = W_e^T W_e (n x n, row-major, host), h_g = W_e^T tau (n), *h_tau_sq = tau^T tau, formed
 * from the Householder R (G = R1^T R1): the normal-equation quantities of the SIP QP (identification_tools.py:528-531)
 * and of the weighted LS statements (staubli_TX40/identification.py:320-327).  h_g / h_tau_sq may be NULL iff d_tau
 * is NULL. */
int figh_regressor_gram(figh_model_t model, int mode, int flags, int ft_mask, int64_t N, const double *d_q,
                        const double *d_v, const double *d_a, const int32_t *d_col_idx, int n, const double *d_tau,
                        int64_t chunk_samples, double *h_G, double *h_g, double *h_tau_sq);

/* ------------------------------------------------------------------ zero-phase filtering / decimation
 * The step on either side of the hot path on real data: scipy.signal.decimate(x, q, zero_phase=True) over every
 * column of W_b and over tau, joint block by joint block (examples/staubli_TX40/identification.py:186-204,
 * examples/tiago/identification.py:142-187), and signal.filtfilt of the joint positions (identification_tools.py:
 * 390-424).  d_X is rows x cols (ldx), made of nblocks row blocks of rows/nblocks samples; every (block, column)
 * sequence is filtered forward and backward with odd padding of `padlen` samples (SciPy's method='pad') and every
 * q-th sample is kept: d_Y receives nblocks blocks of ceil(L/q) rows (*rows_out in total).
 * form 0: second-order sections -- h_b / h_a are nsec x 3 (sos[:, :3] and sos[:, 3:]), order = 2, h_zi = sosfilt_zi
 * (nsec x 2).  form 1: transfer function -- nsec = 1, h_b / h_a have order + 1 entries (a[0] = 1), h_zi = lfilter_zi.
 * The recurrences follow SciPy's loops operation by operation (no FMA contraction). */
int figh_filtfilt_cols(const double *d_X, int64_t rows, int cols, int64_t ldx, int nblocks, int form, const double *h_b,
                       const double *h_a, int nsec, int order, const double *h_zi, int padlen, int q, double *d_Y,
                       int64_t ldy, int64_t *rows_out);

/* ------------------------------------------------------------------ multi-GPU (RCCL over xGMI), SURVEY.md section 8e
 * One process per GPU.  Rank 0 calls figh_comm_unique_id and ships the 128 bytes to the other ranks by any
 * means (the Python side uses the torch.distributed store); every rank then calls figh_comm_init.
 * figh_comm_available is the local preflight (librccl loads, every symbol resolves, a HIP device is there): the ranks
 * agree on its outcome BEFORE anybody enters ncclCommInitRank, so that a rank without RCCL cannot leave the others
 * blocked in the rendezvous. */
int figh_comm_available(void);
int figh_comm_unique_id(void *h_id128);
int figh_comm_init(int nranks, int rank, const void *h_id128);
int figh_comm_destroy(void);
/* all-gather of each rank's nc x nc R factor into d_all (nranks*nc x nc), in rank order */
int figh_comm_allgather(const double *d_send, double *d_all, int64_t count_per_rank);
/* in-place sum all-reduce (column norms, Gram blocks, residual norms) */
int figh_comm_allreduce_sum(double *d_buf, int64_t count);

#ifdef __cplusplus
}
#endif
#endif /* FIGH_H */
