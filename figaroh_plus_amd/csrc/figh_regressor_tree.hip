// K1' -- regressor assembly for kinematic TREES (TIAGo, TALOS, human; free-flyer, prismatic and continuous joints;
// external-wrench mode) on gfx950.
//
// Replaces, for every model the chain kernel (figh_regressor.hip) does not take, the per-sample Python loop around
// pin.computeJointTorqueRegressor and the scatter / friction / permutation statements of build_regressor_basic
// (src/figaroh/tools/regressor.py:45-87 joint torques, :89-192 external wrench).  Output in the reference's layout:
// row r = rb*N + i (row block rb = dof index or wrench component), 14 columns per link in FIGAROH order.
//
// Design (one sample per lane, 64 consecutive samples per wavefront):
//   * TAPE.  The host flattens the work for (model, mode, flags, ft_mask) into a short program of wave-uniform ops --
//     RESET / FETCH joint inputs / STEP joint / EMIT link segment / ZERO column run -- read through the scalar cache.
//     Joints are numbered depth-first (Pinocchio), so a walk needs ONE current state per lane; where it backs up to a
//     branch point (TALOS: 4 times, human: 5) the path from the root is simply stepped again.  No per-joint arrays, no
//     scratch: the state is the spatial velocity / acceleration of the current link (12 doubles) plus
//       - external wrench on a free-flyer root (one walk, six rows per link): the placement of the link in the
//         root-joint frame (12 doubles); row c is J_c^T B with J_c = unit twist c of the root frame seen from the link,
//       - otherwise (one walk per row block): the motion axis J of the row's joint seen from the link (6 doubles),
//         pushed down the subtree (J <- liMi^-1 J).
//     The ten inertial entries of a (row, link) pair are J^T bodyRegressor(v, a) in closed form (figh_spatial.h) -- no
//     6 x 10 body regressor is formed or carried up the chain.
//   * FULL CACHE LINES, NO MEMSET.  Every (row block, link) segment is written exactly once: a computed segment goes
//     through a 64 x LS LDS tile (lane = sample writes its row; the wave then streams the tile out with 16-byte buffer
//     stores, consecutive lanes along a row segment), structural zeros (links outside the subtree, massless links) are
//     streamed from registers as long runs.  The old kernel zero-filled all of W first (88.7 GB for TALOS) and then
//     scattered 8-byte stores.  Measured on the TALOS / human shapes (tools/tree_kernel_bench.py, store-only tapes): the
//     reference's 112-byte link segments (LS = 14) reach 2.3-2.6 TB/s whatever the order -- rows of W are not line
//     aligned and every 128-byte line is shared by two links; 8-link runs reach 3.2 TB/s (rows not aligned) / 5.7 TB/s
//     (aligned), whole contiguous rows 6.0-6.8 TB/s: a store instruction has to cover whole cache lines.  Hence the
//     LINK-PADDED layout (LS = 16: columns 14, 15 of every link are zero, leading dimension a multiple of 16 doubles):
//     every row segment is exactly one aligned 128-byte line, 4.4-5.0 TB/s of algorithmic bytes with one link per flush.
//     The device-resident pipeline and the streamed entry points use it for their private W (figh_tsqr takes a column
//     list anyway); the drop-in functions, which hand W to the caller, keep the reference's layout.
//   * COLUMN NORMS FUSED.  In the stream-out a lane always handles the same column pair of the group, so diag(W^T W)
//     costs two FMAs per 16 bytes; the sums are folded per group in fixed order (bit-reproducible) into per-wave column
//     sums in LDS.  STORE = false: the norms alone (pass 1 of the streamed entry points): same arithmetic, no W at all.
//   * ONE DRAIN PER FIVE JOINTS.  Loads and stores retire through one in-order counter on this hardware, so waiting for
//     a load issued after a store also waits for the store's acknowledgement; the joint inputs q, qd, qdd are therefore
//     fetched five joints at a time (FETCH) instead of inside every STEP.
//
// HBM-bound: algorithmic bytes per sample = 8 (nq + 2 nv) read + 8 rows_per_sample ncols written (SURVEY 8d:
// TIAGo 65 168 B, TALOS 23 096 B, human 27 968 B).
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "figh_internal.h"
#include "figh_spatial.h"
#include "figh_wave.h"

#ifndef FIGH_TREE_LSP
#define FIGH_TREE_LSP 18  // LDS row stride of the segment tile in doubles (see regressor_tape_kernel)
#endif

namespace figh {

enum : int32_t { OP_RESET = 0, OP_STEP = 1, OP_EMIT = 2, OP_ZERO = 3, OP_TX40 = 4, OP_FETCH = 5, OP_RESTORE = 6 };
// STEP flags (field b): bit 0 = the row's joint (J starts here), bits 4..6 = dof inside the joint (free-flyer),
// bits 8.. = FETCH slot
// STEP_SAVE (free-flyer walk, force-compact W): the state behind this step is kept (one copy) -- the joint has several children, and OP_RESTORE
// brings the walk back to it for the next child instead of a RESET and the steps down from the root again (human model: 57 -> 45
// forward steps per sample, K1' 26.5 -> 25.4 ms)
enum : int32_t { STEP_JSTART = 1, STEP_SAVE = 2 };
// EMIT flags (field c)
enum : int32_t { EMIT_INERT = 1, EMIT_EXTRA = 2, EMIT_OWN = 4, EMIT_FLUSH = 8, EMIT_NOSTORE = 16 };
// joints per FETCH group (at most five: the payload fields of an op).  The free-flyer walk (one EMIT per STEP, state in ~250 registers
// at one wave per SIMD) is fastest with five; the joint-torque walks carry kRowSlots axes and gain more from the registers
// (same-box A/B, tools/var_run.sh: TIAGo K1' 2.71 / 2.43 / 2.51 ms with 4 / 3 / 2; human K1' 26.7 -> 30.8 ms with three)
__host__ __device__ constexpr int fetch_group(bool extff) { return extff ? 5 : 3; }
// Row slots of a walk (joint-torque mode): one walk over the subtree of a joint serves the row blocks of that joint AND of the
// joints up to kRowSlots - 1 levels below it -- the axis of every such row joint is carried down the tree in its slot (= its depth
// below the top of the walk), and a link is emitted once per slot it lies under.  With one slot a tree of depth d steps onto
// O(d) joints per row block and link (TIAGo: 195 forward steps per sample for 24 joints; three slots: 69, five: 51).
constexpr int kRowSlots = 5;

struct TapeOp {
    int32_t op, a, b, c, d, e;
};

namespace {

#ifdef FIGH_ABLATION
__device__ int g_tree_hotin = 0;
__device__ int g_tree_half = 0;  // FIGH_TREE_HALF: force rows store only the upper 64 bytes of every line (timing probe)
#endif

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Stream one link segment -- the tile, 64 rows x LS doubles -- to rows rowbase .. + nvalid - 1, columns col0 .. col0 + LS - 1
// of W and/or add its squares to the wave's column sums.  LS = 16: 8 lanes per row, every store instruction writes 8
// whole 128-byte lines; LS = 14: 7 lanes per 112-byte row segment (63 lanes active).
template <int LS, bool STORE, bool COLSQ>
__device__ __forceinline__ void flush_tile(const double *__restrict__ tile, double *__restrict__ red,
                                           double *__restrict__ colacc, const int lane, const int nvalid,
                                           double *__restrict__ W, const long ldw, const unsigned ldw8, const long rowbase,
                                           const int col0, double &acc0, double &acc1, const bool fold,
                                           const bool skip_lo = false, const int col0_acc = -1,
                                           const bool nostore = false, const int force_pos0 = -1) {
    constexpr int CP = LS / 2, RPI = 64 / CP;  // 16-byte chunks per row, rows per store instruction
    constexpr int LSP = FIGH_TREE_LSP;         // LDS row stride of the tile (see regressor_tape_kernel)
    const int rg = lane / CP, ch = lane - rg * CP;
    const bool active = rg < RPI;
    // (acc0, acc1: this lane's column pair, summed over its rows -- and, external-wrench mode, over the six row blocks of the
    // link, which share their columns: ONE fold per link instead of six)
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(W + rowbase * ldw + col0, (short)0, 0x7fffffff, 0x00020000);
    const unsigned voff = (unsigned)rg * ldw8 + 16u * (unsigned)ch;
    if (active) {
#pragma unroll
        for (int it = 0; it < (64 + RPI - 1) / RPI; ++it) {
            const int row = RPI * it + rg;
            if (row < nvalid) {
                const double2 x = *reinterpret_cast<const double2 *>(tile + row * LSP + 2 * ch);
                if constexpr (COLSQ) {
                    acc0 = fma(x.x, x.x, acc0);
                    acc1 = fma(x.y, x.y, acc1);
                }
                if (STORE && !nostore && !(skip_lo && ch < CP / 2)) {
                    u32x4 d;
                    d[0] = (unsigned)__double2loint(x.x);
                    d[1] = (unsigned)__double2hiint(x.x);
                    d[2] = (unsigned)__double2loint(x.y);
                    d[3] = (unsigned)__double2hiint(x.y);
                    __builtin_amdgcn_raw_buffer_store_b128(d, rs, voff, (unsigned)(RPI * it) * ldw8, 0);
                }
            }
        }
    }
    if (COLSQ && fold) {  // fold the row groups in fixed order: bit-reproducible
        red[2 * lane] = active ? acc0 : 0.0;
        red[2 * lane + 1] = active ? acc1 : 0.0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < CP) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int r = 0; r < RPI; ++r) {
                s0 += red[2 * (CP * r + lane)];
                s1 += red[2 * (CP * r + lane) + 1];
            }
            if (force_pos0 >= 0) {
                // a line of the force-compact region: four links x (mx my mz m) -- the pair of this lane belongs to link
                // force_pos0 + lane / 2, slots 6 + 2 (lane % 2) and the next one, of the kernel's own column numbering
                const int ca = LS * (force_pos0 + (lane >> 1)) + 6 + 2 * (lane & 1);
                colacc[ca] += s0;
                colacc[ca + 1] += s1;
            } else {
                const int ca = col0_acc >= 0 ? col0_acc : col0;  // (block-compact W: the norms keep the dense numbering)
                colacc[ca + 2 * lane] += s0;
                colacc[ca + 2 * lane + 1] += s1;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        acc0 = acc1 = 0.0;
    }
}

// zeros to rows rowbase .. + nvalid - 1, columns c0 .. c0 + w - 1 (magic = ceil(2^32 / chunks per row))
template <bool VEC2>
__device__ __forceinline__ void stream_zeros(const int lane, const int nvalid, double *__restrict__ W, const long ldw,
                                             const unsigned ldw8, const long rowbase, const int c0, const int w,
                                             const unsigned magic2, const unsigned magic1) {
    if constexpr (VEC2) {
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(W + rowbase * ldw + c0, (short)0, 0x7fffffff, 0x00020000);
        const unsigned ch = (unsigned)w >> 1, total = (unsigned)nvalid * ch;
        const u32x4 z = {0u, 0u, 0u, 0u};
        // (row, chunk) of a lane's store advance by a fixed step with a carry: additions only inside the loop (the division by
        // multiplication per store cost three quarter-rate integer multiplies for every 16-byte store of a zero run)
        unsigned row = __umulhi((unsigned)lane, magic2);
        unsigned k = (unsigned)lane - row * ch;
        unsigned off = row * ldw8 + 16u * k;
        const unsigned dr = __umulhi(64u, magic2), dk = 64u - dr * ch;  // 64 = dr ch + dk
        const unsigned doff = dr * ldw8 + 16u * dk, wrap = ldw8 - 16u * ch;
        for (unsigned id = (unsigned)lane; id < total; id += 64u) {
            __builtin_amdgcn_raw_buffer_store_b128(z, rs, off, 0u, 0);
            k += dk;
            off += doff;
            const bool carry = k >= ch;
            k -= carry ? ch : 0u;
            off += carry ? wrap : 0u;
        }
    } else {
        const unsigned total = (unsigned)nvalid * (unsigned)w;
        for (unsigned id = (unsigned)lane; id < total; id += 64u) {
            const unsigned row = __umulhi(id, magic1);
            const unsigned k = id - row * (unsigned)w;
            W[(rowbase + row) * ldw + c0 + k] = 0.0;
        }
    }
}

// LS: columns per link in W (14 = the reference's layout; 16 = link-padded, one 128-byte line per row segment).
// EXTFF: external wrench on a free-flyer root -- one walk, state = placement of the current link in the root-joint frame,
// six rows per link; otherwise the row's joint axis is carried down the subtree, one walk per row block.
// VEC2: 16-byte stores.  STORE = false: column norms only.
// FC (FIGH_FLAG_FORCE_COMPACT, EXTFF with LS = 16, no friction / inertia / offset columns): the three FORCE row blocks go to
// their own region in front of the torque rows -- 3 N rows of ldf columns, one 128-byte line per FOUR links (mx my mz m of
// each: everything a force row has, the rotational-inertia entries being exact zeros) instead of one line per link.  A lane
// keeps the twelve force entries of its sample for up to four links in registers (96 VGPRs: the kernel has them, 124 of 256)
// and every fourth link the three lines go through the segment tile and out as whole lines like any other segment: a quarter
// of the force-row lines, 5/8 of W (TALOS 101 -> 64 GB, human 146 -> 92 GB), and the force rows' TSQR reads 16-column groups
// that are all payload.  W = force region, torque region at W + 3 N ldf with leading dimension ldw.
template <int LS, bool EXTFF, bool VEC2, bool STORE, bool COLSQ, bool FC = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FC ? 1 : 2))) void regressor_tape_kernel(const DevModel *__restrict__ M,
                                                            const TapeOp *__restrict__ tape, const int ntape,
                                                            const int flags, const long N,
                                                            const double *__restrict__ q, const double *__restrict__ v,
                                                            const double *__restrict__ a, double *__restrict__ W,
                                                            const long ldw, const int ncols_int,
                                                            double *__restrict__ colsq_part, const long ldf = 0) {
    static_assert(!FC || (EXTFF && LS == 16 && VEC2), "force-compact: external wrench, link-padded columns");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    // The tile's rows are LS + 2 doubles apart in LDS: a lane writes ITS row (one sample), and with the rows 128 bytes apart
    // (LS = 16) all 64 lanes of a ds_write hit the same four banks -- 64 cycles per instruction instead of 8, fourteen doubles
    // per segment, and the flush's ds_read_b128 (8 rows x 8 chunks per instruction) 8-way on top: the kernel spent its time
    // in LDS (TALOS: 198 segments per tile, ~1200 LDS cycles each = the whole 24 ms; a store-only tape ran 2.2 x faster).
    // 144 bytes (both segment widths): rows stay 16-byte aligned, eight consecutive rows cover all 32 banks.
    constexpr int LSP = FIGH_TREE_LSP;
    double *tile = lds;            // 64 x LSP
    double *red = lds + 64 * LSP;  // 64 x 2
    double *colacc = red + 128;   // ncols_int (COLSQ): column sums in the kernel's own (LS-strided) numbering
    const int lane = threadIdx.x;
    const unsigned ldw8 = 8u * (unsigned)ldw;
    const int nq = M->nq, nv = M->nv;
    const bool fric = flags & FIGH_FLAG_FRICTION, actin = flags & FIGH_FLAG_ACT_INERTIA, offs = flags & FIGH_FLAG_OFFSET;
    const bool blocked = flags & FIGH_FLAG_BLOCKED_INPUTS;
    const int is = blocked ? 64 : 1;
    if constexpr (COLSQ) {
        for (int e = lane; e < ncols_int; e += 64) colacc[e] = 0.0;
    }
    if constexpr (LS == 16) {  // the two padding columns of the tile are written once
        tile[LSP * lane + 14] = 0.0;
        tile[LSP * lane + 15] = 0.0;
    }
    const double g0 = M->gravity[0], g1 = M->gravity[1], g2 = M->gravity[2];
    const long ntiles = (N + 63) / 64;
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long i0 = t * 64;
        const int nvalid = (int)((N - i0) < 64 ? (N - i0) : 64);
        const long i = i0 + (lane < nvalid ? lane : nvalid - 1);
        // value k of this lane's sample is qi[k * is]: is = 1 for the reference's sample-major arrays, 64 for the
        // tile-blocked copies of figh_repack_samples (FIGH_FLAG_BLOCKED_INPUTS: the wave's 64 values of one k are one line)
#ifdef FIGH_ABLATION
        // FIGH_TREE_HOTIN: every tile reads the inputs of the first 64 samples (wrong numbers, same instruction stream):
        // what the kernel would cost if q, v, a came from cache (tools/tree_hotin.sh)
        const long iin = g_tree_hotin ? (lane < nvalid ? lane : nvalid - 1) : i;
        const long tin = g_tree_hotin ? 0 : t;
#else
        const long iin = i, tin = t;
#endif
        const double *qi = blocked ? q + tin * 64 * nq + lane : q + iin * nq;
        const double *vi = blocked ? v + tin * 64 * nv + lane : v + iin * nv;
        const double *ai = blocked ? a + tin * 64 * nv + lane : a + iin * nv;
        // state of the current link
        double V[6] = {0, 0, 0, 0, 0, 0}, A[6] = {-g0, -g1, -g2, 0, 0, 0};
        double Rc[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pc[3] = {0, 0, 0};  // EXTFF: link -> root-joint frame
        // otherwise: the axes of the walk's row joints (slot = depth below the top of the walk), link frame; jmask = slots in use
        constexpr int NS = EXTFF ? 1 : kRowSlots;
        double Jl[NS][3], Ja[NS][3];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int d = 0; d < 3; ++d) Jl[s][d] = Ja[s][d] = 0.0;
        int jmask = 0;
        // the state kept at a branch joint (STEP_SAVE / OP_RESTORE; the force-compact free-flyer walk only: at one wave per SIMD
        // the copy lives in AGPRs -- the walks that run at two waves per SIMD spill for it or drop to one: TIAGo K1' 2.43 -> 2.90 ms,
        // human link-compact 5.50 -> 6.35 .. 6.96 ms, same-box A/B with tools/var_run.sh / tools/k1_layout_ab.py)
        double sV[6], sA[6], sRc[9], sPc[3];
#pragma unroll
        for (int d = 0; d < 6; ++d) sV[d] = sA[d] = 0.0;
#pragma unroll
        for (int d = 0; d < 9; ++d) sRc[d] = 0.0;
#pragma unroll
        for (int d = 0; d < 3; ++d) sPc[d] = 0.0;
        // inputs of the coming single-dof joints (FETCH): q (cos q for a continuous joint), sin q, qd, qdd
        constexpr int kFetch = fetch_group(EXTFF);
        double sq0[kFetch], sq1[kFetch], sqd[kFetch], sqdd[kFetch];
#pragma unroll
        for (int s = 0; s < kFetch; ++s) sq0[s] = sq1[s] = sqd[s] = sqdd[s] = 0.0;
        double last_qd = 0.0, last_qdd = 0.0;  // of the joint stepped last (Ia / fv / fs of its own row)
        double F[FC ? 3 : 1][FC ? 16 : 1];     // FC: the force entries of up to four links, per wrench component
        double fs0 = 0.0, fs1 = 0.0;           // FC: column norms of the force lines (flush_tile)
        for (int pcnt = 0; pcnt < ntape; ++pcnt) {
            int op = tape[pcnt].op, oa = tape[pcnt].a, ob = tape[pcnt].b, oc = tape[pcnt].c, od = tape[pcnt].d,
                oe = tape[pcnt].e;
            if (op == OP_STEP) {
                // ---- forward step onto joint k (restates the first loop of pinocchio::computeJointTorqueRegressor)
                const int k = oa;
                const int jt = M->jtype[k], iq = M->idx_q[k], iv = M->idx_v[k];
                const double ax[3] = {M->axis[k][0], M->axis[k][1], M->axis[k][2]};
                double jq0 = 0.0, jq1 = 0.0, jqd = 0.0, jqdd = 0.0;
                {
                    const int slot = ob >> 8;
#pragma unroll
                    for (int s = 0; s < kFetch; ++s)
                        if (s == slot) {
                            jq0 = sq0[s];
                            jq1 = sq1[s];
                            jqd = sqd[s];
                            jqdd = sqdd[s];
                        }
                }
                last_qd = jqd;
                last_qdd = jqdd;
                double Rj[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pj[3] = {0, 0, 0}, vj[6] = {0, 0, 0, 0, 0, 0},
                       aj[6] = {0, 0, 0, 0, 0, 0};
                if (jt == FIGH_JT_REVOLUTE || jt == FIGH_JT_CONTINUOUS) {
                    double s, c;
                    if (jt == FIGH_JT_REVOLUTE) {
                        sincos(jq0, &s, &c);
                    } else {
                        c = jq0;
                        s = jq1;
                    }
                    rodrigues(ax, c, s, Rj);
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        vj[3 + d] = ax[d] * jqd;
                        aj[3 + d] = ax[d] * jqdd;
                    }
                } else if (jt == FIGH_JT_PRISMATIC) {
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        pj[d] = ax[d] * jq0;
                        vj[d] = ax[d] * jqd;
                        aj[d] = ax[d] * jqdd;
                    }
                } else {  // free-flyer: q = [p, qx qy qz qw], v in the joint's local frame (read here: once per walk)
                    const double x = qi[(iq + 3) * is], y = qi[(iq + 4) * is], z = qi[(iq + 5) * is], ww = qi[(iq + 6) * is];
                    Rj[0] = 1 - 2 * (y * y + z * z); Rj[1] = 2 * (x * y - z * ww); Rj[2] = 2 * (x * z + y * ww);
                    Rj[3] = 2 * (x * y + z * ww); Rj[4] = 1 - 2 * (x * x + z * z); Rj[5] = 2 * (y * z - x * ww);
                    Rj[6] = 2 * (x * z - y * ww); Rj[7] = 2 * (y * z + x * ww); Rj[8] = 1 - 2 * (x * x + y * y);
#pragma unroll
                    for (int d = 0; d < 3; ++d) pj[d] = qi[(iq + d) * is];
#pragma unroll
                    for (int d = 0; d < 6; ++d) {
                        vj[d] = vi[(iv + d) * is];
                        aj[d] = ai[(iv + d) * is];
                    }
                }
                double Rk[9], pk[3];  // liMi = placement * M_joint(q)
                matmul3(M->placement[k], Rj, Rk);
                rot(M->placement[k], pj, pk);
#pragma unroll
                for (int d = 0; d < 3; ++d) pk[d] += M->placement[k][9 + d];
                double t1[3], t2[3], Vk[6], Ak[6];
                cross3(pk, V + 3, t1);
#pragma unroll
                for (int d = 0; d < 3; ++d) t2[d] = V[d] - t1[d];
                rotT(Rk, t2, Vk);
                rotT(Rk, V + 3, Vk + 3);
                cross3(pk, A + 3, t1);
#pragma unroll
                for (int d = 0; d < 3; ++d) t2[d] = A[d] - t1[d];
                rotT(Rk, t2, Ak);
                rotT(Rk, A + 3, Ak + 3);
#pragma unroll
                for (int d = 0; d < 6; ++d) Vk[d] += vj[d];
                double c1[3], c2[3], c3[3];  // Vk x vj
                cross3(Vk + 3, vj, c1);
                cross3(Vk, vj + 3, c2);
                cross3(Vk + 3, vj + 3, c3);
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    Ak[d] += aj[d] + c1[d] + c2[d];
                    Ak[3 + d] += aj[3 + d] + c3[d];
                }
#pragma unroll
                for (int d = 0; d < 6; ++d) {
                    V[d] = Vk[d];
                    A[d] = Ak[d];
                }
                if constexpr (EXTFF) {
                    if (ob & STEP_JSTART) {  // the root joint: its frame is where the wrench is expressed
#pragma unroll
                        for (int d = 0; d < 9; ++d) Rc[d] = (d % 4 == 0) ? 1.0 : 0.0;
                        pc[0] = pc[1] = pc[2] = 0.0;
                    } else {
                        double Rn[9], pn[3];
                        matmul3(Rc, Rk, Rn);
                        rot(Rc, pk, pn);
#pragma unroll
                        for (int d = 0; d < 9; ++d) Rc[d] = Rn[d];
#pragma unroll
                        for (int d = 0; d < 3; ++d) pc[d] += pn[d];
                    }
                } else {
                    // push the axes of the row joints above one link down: J <- liMi^-1 J
#pragma unroll
                    for (int s = 0; s < NS; ++s)
                        if ((jmask >> s) & 1) {
                            double nJl[3], nJa[3];
                            cross3(pk, Ja[s], t1);
#pragma unroll
                            for (int d = 0; d < 3; ++d) t2[d] = Jl[s][d] - t1[d];
                            rotT(Rk, t2, nJl);
                            rotT(Rk, Ja[s], nJa);
#pragma unroll
                            for (int d = 0; d < 3; ++d) {
                                Jl[s][d] = nJl[d];
                                Ja[s][d] = nJa[d];
                            }
                        }
                    if (ob & STEP_JSTART) {  // a row's own joint (slot oc): the axis of its dof, in its own frame
                        const bool pris = jt == FIGH_JT_PRISMATIC;
#pragma unroll
                        for (int s = 0; s < NS; ++s)
                            if (s == oc) {
#pragma unroll
                                for (int d = 0; d < 3; ++d) {
                                    Jl[s][d] = pris ? ax[d] : 0.0;
                                    Ja[s][d] = pris ? 0.0 : ax[d];
                                }
                            }
                        jmask = (jmask & ((1 << oc) - 1)) | (1 << oc);  // (deeper slots belong to a branch that was left)
                    }
                }
                if constexpr (FC) {
                    if (ob & STEP_SAVE) {
#pragma unroll
                        for (int d = 0; d < 6; ++d) {
                            sV[d] = V[d];
                            sA[d] = A[d];
                        }
#pragma unroll
                        for (int d = 0; d < 9; ++d) sRc[d] = Rc[d];
#pragma unroll
                        for (int d = 0; d < 3; ++d) sPc[d] = pc[d];
                    }
                }
                // A STEP that is followed by the EMIT of its link runs it in the SAME iteration: every trip through the loop
                // latch copies the whole kinematic state (50 doubles: the register allocator does not keep it in place across
                // the op branches), and half of the ops of a tape are such EMITs
                if (pcnt + 1 < ntape && tape[pcnt + 1].op == OP_EMIT) {
                    ++pcnt;
                    op = OP_EMIT;
                    oa = tape[pcnt].a;
                    ob = tape[pcnt].b;
                    oc = tape[pcnt].c;
                    od = tape[pcnt].d;
                    oe = tape[pcnt].e;
                }
            }
            if (op == OP_EMIT) {
              // (the EMITs of one link -- one per row slot it lies under -- run back to back inside this iteration, for the same reason)
              bool more = false;
              do {
                // ---- the segment of link oa: EXTFF in all six row blocks (od = components with inertial entries),
                // otherwise in row block ob
                // (external wrench with FIGH_FLAG_LINK_COMPACT: the link's segment sits at its position among the links that
                // have one, oe - 1; otherwise at its place in the link-padded row)
                const int b = oa, col0 = (EXTFF && oe > 0) ? oe - 1 : LS * (b - 1);
                double ex[4] = {0.0, 0.0, 0.0, 0.0};
                if (oc & EMIT_EXTRA) {  // regressor.py:55-70 (own row) / :142-169 (all six rows): Ia fv fs off of link b
                    // joint-torque mode writes them on the link's own row, right after its STEP: dof b - 1 is that joint
                    const bool own = oc & EMIT_OWN;
                    if (actin) ex[0] = own ? last_qdd : ai[(b - 1) * is];
                    if (fric) {
                        const double vv = own ? last_qd : vi[(b - 1) * is];
                        ex[1] = vv;
                        ex[2] = sgn(vv);
                    }
                    if (offs) ex[3] = 1.0;
                }
                double accv[3];
                {
                    double tt[3];
                    cross3(V + 3, V, tt);
#pragma unroll
                    for (int d = 0; d < 3; ++d) accv[d] = A[d] + tt[d];
                }
                double *my = tile + LSP * lane;
                double cs0 = 0.0, cs1 = 0.0;  // column norms of the segment (flush_tile)
                if constexpr (FC) {
                    // ---- the three force components of link b: into the lane's registers, slot ob & 3 of the group of four
                    const int fslot = ob & 0xff;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        double o[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                        if ((od >> c) & 1) {
                            const double jl[3] = {Rc[3 * c], Rc[3 * c + 1], Rc[3 * c + 2]};
                            axis_times_body_regressor_lin(jl, accv, A + 3, V + 3, o);
                        }
                        if (fslot == 0) {  // a new group: the slots of a last, incomplete group stay zero
#pragma unroll
                            for (int k = 4; k < 16; ++k) F[c][k] = 0.0;
                        }
#pragma unroll
                        for (int sl = 0; sl < 4; ++sl)
                            if (fslot == sl) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) F[c][4 * sl + k] = o[6 + k];
                            }
                    }
                    if (oc & EMIT_FLUSH) {  // the group is complete (or the walk ends): three lines per sample
                        const int fcol0 = ob >> 8;
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
#pragma unroll
                            for (int k = 0; k < 16; ++k) my[k] = F[c][k];
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                            flush_tile<LS, STORE, COLSQ>(tile, red, colacc, lane, nvalid, W, ldf, 8u * (unsigned)ldf, (long)c * N + i0,
                                                         fcol0, fs0, fs1, c == 2, false, -1, false, fcol0 / 4);
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                        }
                        // (the tile's padding columns 14, 15 were overwritten by the line: zero again for the torque segments)
                        my[14] = 0.0;
                        my[15] = 0.0;
                    }
                }
#pragma unroll 1
                for (int c = FC ? 3 : 0; c < (EXTFF ? 6 : 1); ++c) {
                    const bool fold = !EXTFF || c == 5;
                    double o[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                    if constexpr (EXTFF) {
                        if ((od >> c) & 1) {  // unit twist c of the root-joint frame, seen from the link
                            double jl[3], ja[3];
                            if (c < 3) {  // (a force row: Ja = 0, the rotational-inertia entries stay exact zeros)
                                jl[0] = Rc[3 * c];
                                jl[1] = Rc[3 * c + 1];
                                jl[2] = Rc[3 * c + 2];
                                axis_times_body_regressor_lin(jl, accv, A + 3, V + 3, o);
                            } else {
                                const int kk = c - 3;
                                ja[0] = Rc[3 * kk];
                                ja[1] = Rc[3 * kk + 1];
                                ja[2] = Rc[3 * kk + 2];
                                double tt[3];  // Jl = Rc^T (e_k x pc)
                                tt[0] = kk == 0 ? 0.0 : (kk == 1 ? pc[2] : -pc[1]);
                                tt[1] = kk == 0 ? -pc[2] : (kk == 1 ? 0.0 : pc[0]);
                                tt[2] = kk == 0 ? pc[1] : (kk == 1 ? -pc[0] : 0.0);
                                rotT(Rc, tt, jl);
                                axis_times_body_regressor(jl, ja, accv, A + 3, V + 3, o);
                            }
                        }
                    } else {
                        if (oc & EMIT_INERT) {
                            double jl[3] = {0, 0, 0}, ja[3] = {0, 0, 0};
#pragma unroll
                            for (int s = 0; s < NS; ++s)
                                if (s == (oc >> 8)) {
#pragma unroll
                                    for (int d = 0; d < 3; ++d) {
                                        jl[d] = Jl[s][d];
                                        ja[d] = Ja[s][d];
                                    }
                                }
                            axis_times_body_regressor(jl, ja, accv, A + 3, V + 3, o);
                        }
                    }
#pragma unroll
                    for (int d = 0; d < 10; ++d) my[d] = o[d];
#pragma unroll
                    for (int d = 0; d < 4; ++d) my[10 + d] = ex[d];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    // (FC: the torque region lies behind the 3 N x ldf force region, its row blocks are numbered 0 .. 2 there)
                    const long rowbase = (long)(EXTFF ? (FC ? c - 3 : c) : ob) * N + i0;
                    double *const Wseg = FC ? W + 3 * N * ldf : W;
                    if constexpr (VEC2) {
                        if (!EXTFF && od != 0) {
                            // FIGH_FLAG_COMPACT_BLOCKS: row block ob is its own N x ld matrix -- the columns of its joint's
                            // subtree, a contiguous window of the dense row -- at W + N * (ld's prefix sum); od = the
                            // segment's column in it | ld << 16, oe = the prefix sum
                            const long ldc = od >> 16;
                            flush_tile<LS, STORE, COLSQ>(tile, red, colacc, lane, nvalid, W + (long)N * oe, ldc,
                                                         8u * (unsigned)ldc, i0, (od & 0xffff) - 1, cs0, cs1, fold, false, col0,
                                                         (oc & EMIT_NOSTORE) != 0);
                        } else {
#ifdef FIGH_ABLATION
                            flush_tile<LS, STORE, COLSQ>(tile, red, colacc, lane, nvalid, Wseg, ldw, ldw8, rowbase, col0, cs0, cs1, fold,
                                                         EXTFF && g_tree_half && c < 3);
#else
                            flush_tile<LS, STORE, COLSQ>(tile, red, colacc, lane, nvalid, Wseg, ldw, ldw8, rowbase, col0, cs0, cs1, fold,
                                                         false, -1, !EXTFF && (oc & EMIT_NOSTORE) != 0);
#endif
                        }
                    } else if (!(oc & EMIT_NOSTORE)) {  // odd column count / unaligned W: plain 8-byte stores, no fused norms
                        for (int id = lane; id < nvalid * 14; id += 64) {
                            const int row = id / 14, col = id - 14 * row;
                            W[(rowbase + row) * ldw + col0 + col] = tile[row * LSP + col];
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                more = !EXTFF && pcnt + 1 < ntape && tape[pcnt + 1].op == OP_EMIT;  // (the free-flyer walk emits a link once)
                if (more) {
                    ++pcnt;
                    oa = tape[pcnt].a;
                    ob = tape[pcnt].b;
                    oc = tape[pcnt].c;
                    od = tape[pcnt].d;
                    oe = tape[pcnt].e;
                }
              } while (more);
            } else if (op == OP_ZERO) {
                if constexpr (STORE)
                    stream_zeros<VEC2>(lane, nvalid, W, ldw, ldw8, (long)oa * N + i0, ob, oc, (unsigned)od, (unsigned)oe);
            } else if (op == OP_FETCH) {
                const int js[5] = {oa, ob, oc, od, oe};
#pragma unroll
                for (int s = 0; s < kFetch; ++s)
                    if (js[s] > 0) {
                        const int iq = M->idx_q[js[s]], iv = M->idx_v[js[s]];
                        sq0[s] = qi[iq * is];
                        sq1[s] = M->jtype[js[s]] == FIGH_JT_CONTINUOUS ? qi[(iq + 1) * is] : 0.0;
                        sqd[s] = vi[iv * is];
                        sqdd[s] = ai[iv * is];
                    }
            } else if (op == OP_RESET) {
#pragma unroll
                for (int d = 0; d < 6; ++d) V[d] = 0.0;
                A[0] = -g0; A[1] = -g1; A[2] = -g2; A[3] = A[4] = A[5] = 0.0;
#pragma unroll
                for (int d = 0; d < 9; ++d) Rc[d] = (d % 4 == 0) ? 1.0 : 0.0;
                pc[0] = pc[1] = pc[2] = 0.0;
                jmask = 0;
            } else if (FC && op == OP_RESTORE) {
#pragma unroll
                for (int d = 0; d < 6; ++d) {
                    V[d] = sV[d];
                    A[d] = sA[d];
                }
#pragma unroll
                for (int d = 0; d < 9; ++d) Rc[d] = sRc[d];
#pragma unroll
                for (int d = 0; d < 3; ++d) pc[d] = sPc[d];
            } else if (op == OP_TX40) {  // (regressor.py:198-227, fused): columns 14 nl .. + 2 on the six joint rows
                if constexpr (STORE) {
                    if (lane < nvalid) {
                        const double sc = sgn(vi[4 * is] + vi[5 * is]);
                        for (int r = 0; r < 6; ++r) {
                            double *row = W + ((long)r * N + i) * ldw + oa;
                            row[0] = r == 4 ? ai[5 * is] : (r == 5 ? ai[4 * is] : 0.0);
                            row[1] = r == 4 ? vi[5 * is] : (r == 5 ? vi[4 * is] : 0.0);
                            row[2] = (r == 4 || r == 5) ? sc : 0.0;
                        }
                    }
                }
            }
        }
    }
    if constexpr (COLSQ) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int e = lane; e < ncols_int; e += 64) colsq_part[(long)blockIdx.x * ncols_int + e] = colacc[e];
    }
}

// partial[b][LS l + s] -> out[14 l + s]: one workgroup per reference column, strided partial sums + LDS tree (fixed
// order: deterministic)
__global__ __launch_bounds__(256) void reduce_tree_partials_kernel(const double *__restrict__ part, int nblocks,
                                                                   int ncols_int, int ls, const int *__restrict__ link_pos,
                                                                   double *__restrict__ out) {
    __shared__ double sm[256];
    const int c = blockIdx.x;
    // link_pos (FIGH_FLAG_LINK_COMPACT): position of a link's segment in W, -1 for a link without one (all its entries are
    // structural zeros: norm exactly 0)
    const int lp = link_pos ? link_pos[c / 14] : c / 14;
    const int ci = lp * ls + c % 14;
    double s = 0.0;
    if (lp >= 0)
        for (int b = threadIdx.x; b < nblocks; b += 256) s += part[(long)b * ncols_int + ci];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = sm[0];
}

// ---------------------------------------------------------------------------------------------- tape builder (host)
unsigned magic_for(unsigned d) { return d ? (unsigned)((0x100000000ull + d - 1) / d) : 0u; }

struct TapeBuilder {
    const DevModel &h;
    std::vector<TapeOp> ops;
    explicit TapeBuilder(const DevModel &m) : h(m) {}
    void push(int op, int a = 0, int b = 0, int c = 0, int d = 0, int e = 0) { ops.push_back({op, a, b, c, d, e}); }
    void zero(int rowblock, int col0, int w) {
        if (w <= 0) return;
        push(OP_ZERO, rowblock, col0, w, (int)magic_for((unsigned)w / 2), (int)magic_for((unsigned)w));
    }
    std::vector<int> path_to(int k) const {  // joints root -> k (universe excluded)
        std::vector<int> p;
        for (int j = k; j > 0; j = h.parents[j]) p.insert(p.begin(), j);
        return p;
    }
    // Which joints of the walk over the subtree of t keep the state behind their step (STEP_SAVE): there is ONE copy, so no two of
    // them may lie on one root path; a joint with c children at depth d (d + 1 steps from the root) saves (c - 1)(d + 1)
    // steps.  best(j) = max(own gain, sum over the children): the classic independent-set-on-paths recursion.
    std::vector<char> choose_saves(int t, int end) const {
        const int n = h.njoints;
        std::vector<long> gain(n, 0), best(n, 0);
        std::vector<int> nchild(n, 0);
        for (int k = t + 1; k < end; ++k) ++nchild[h.parents[k]];
        for (int j = end - 1; j >= t; --j) {
            gain[j] = nchild[j] > 1 ? (long)(nchild[j] - 1) * (long)path_to(j).size() : 0;
            long below = 0;
            for (int k = j + 1; k < end; ++k)
                if (h.parents[k] == j) below += best[k];
            best[j] = gain[j] > below ? gain[j] : below;
        }
        std::vector<char> save(n, 0);
        std::vector<int> todo = {t};
        while (!todo.empty()) {
            const int j = todo.back();
            todo.pop_back();
            long below = 0;
            for (int k = j + 1; k < end; ++k)
                if (h.parents[k] == j) below += best[k];
            if (gain[j] > 0 && gain[j] >= below) {
                save[j] = 1;
                continue;
            }
            for (int k = j + 1; k < end; ++k)
                if (h.parents[k] == j) todo.push_back(k);
        }
        return save;
    }
    int subtree_end(int j) const {  // joints j .. end-1 form the subtree of j (depth-first numbering)
        int e = j + 1;
        while (e < h.njoints) {
            int k = e;
            while (k > j) k = h.parents[k];
            if (k != j) break;
            ++e;
        }
        return e;
    }
};

// Insert the FETCH ops: the single-dof STEPs are grouped by kFetch in tape order, each group's joint inputs are loaded
// by one FETCH in front of its first STEP, and every STEP learns its slot (bits 8.. of its flags field).
std::vector<TapeOp> with_fetches(const DevModel &h, const std::vector<TapeOp> &in, const int kFetch) {
    std::vector<TapeOp> out;
    out.reserve(in.size() + in.size() / kFetch + 1);
    auto fetched = [&](const TapeOp &op) { return op.op == OP_STEP && h.jtype[op.a] != FIGH_JT_FREEFLYER; };
    size_t i = 0;
    while (i < in.size()) {
        int js[5] = {0, 0, 0, 0, 0};
        int cnt = 0;
        size_t j = i;
        for (; j < in.size() && cnt < kFetch; ++j)
            if (fetched(in[j])) js[cnt++] = in[j].a;
        if (cnt == 0) {  // no step left: copy the rest
            out.insert(out.end(), in.begin() + i, in.end());
            break;
        }
        size_t first = i;
        while (!fetched(in[first])) ++first;
        out.insert(out.end(), in.begin() + i, in.begin() + first);
        out.push_back({OP_FETCH, js[0], js[1], js[2], js[3], js[4]});
        int slot = 0;
        for (size_t k = first; k < j; ++k) {
            TapeOp op = in[k];
            if (fetched(op)) op.b |= (slot++) << 8;
            out.push_back(op);
        }
        i = j;
    }
    return out;
}

// external wrench, free-flyer root: one walk over the tree, six row segments per link (ls = columns per link in W)
std::vector<TapeOp> build_tape_extff(const DevModel &h, int flags, int ft_mask, int ls, const int *link_pos,
                                     bool force_compact) {
    TapeBuilder T(h);
    int fpos = 0, last_emit = -1;  // force-compact: running position of the links with a segment; index of the last EMIT
    const bool extras = flags & (FIGH_FLAG_FRICTION | FIGH_FLAG_ACT_INERTIA | FIGH_FLAG_OFFSET);
    int prev = 0, zero_from = -1;
    auto flush_zero = [&](int upto_link) {  // links zero_from .. upto_link-1 (1-based joints) are all-zero segments
        if (zero_from < 0) return;
        // (link-compact W: such links have no columns at all)
        if (!link_pos)
            for (int c = 0; c < 6; ++c) T.zero(c, ls * (zero_from - 1), ls * (upto_link - zero_from));
        zero_from = -1;
    };
    T.push(OP_RESET);
    const std::vector<char> save = force_compact ? T.choose_saves(1, h.njoints) : std::vector<char>(h.njoints, 0);
    for (int b = 1; b < h.njoints; ++b) {
        if (h.parents[b] != prev) {
            if (save[h.parents[b]]) {
                T.push(OP_RESTORE);
            } else {
                T.push(OP_RESET);
                for (int k : T.path_to(h.parents[b])) T.push(OP_STEP, k, (k == 1 ? STEP_JSTART : 0) | (save[k] ? STEP_SAVE : 0));
            }
        }
        T.push(OP_STEP, b, (b == 1 ? STEP_JSTART : 0) | (save[b] ? STEP_SAVE : 0));
        prev = b;
        const int inert = h.body_mask[b] ? (ft_mask & 63) : 0;
        if (inert || extras) {
            flush_zero(b);
            // force-compact: field b = the link's slot in its group of four | column of the group's line in the force region
            // << 8; EMIT_FLUSH on every fourth link (and on the last one, below)
            // (force positions count the links that HAVE a segment, whatever the torque rows' layout: groups are then always
            // complete but the last)
            const int p = fpos++;
            T.push(OP_EMIT, b, force_compact ? ((p & 3) | ((16 * (p >> 2)) << 8)) : 0,
                   (inert ? EMIT_INERT : 0) | (extras ? EMIT_EXTRA : 0) | ((force_compact && (p & 3) == 3) ? EMIT_FLUSH : 0),
                   inert, link_pos ? ls * link_pos[b - 1] + 1 : 0);
            last_emit = (int)T.ops.size() - 1;
        } else if (zero_from < 0) {
            zero_from = b;
        }
    }
    flush_zero(h.njoints);
    if (force_compact && last_emit >= 0) T.ops[last_emit].c |= EMIT_FLUSH;  // (a last group of fewer than four links)
    return with_fetches(h, T.ops, fetch_group(true));
}

// One row block per dof.  Joint-torque mode: every joint is single-dof and has its row.  External wrench on a fixed
// base: rows = the first six joints; inertial entries only for links with mass and components in ft_mask, Ia / fv / fs /
// off on all six rows of every link (regressor.py:142-169).
std::vector<TapeOp> build_tape_rows(const DevModel &h, int mode, int flags, int ft_mask, int ls,
                                    unsigned long long active_rows) {
    TapeBuilder T(h);
    const bool ext = mode == FIGH_MODE_EXT_WRENCH;
    const bool extras = flags & (FIGH_FLAG_FRICTION | FIGH_FLAG_ACT_INERTIA | FIGH_FLAG_OFFSET);
    const int nrows = ext ? 6 : h.nv;
    const int nl = h.nlinks;
    int compact_prefix = 0;  // block-compact W: sum of the leading dimensions of the row blocks in front
    for (int row = 0; row < nrows; ++row) {
        int j = 0;  // the joint that owns dof `row`
        for (int k = 1; k < h.njoints; ++k)
            if (h.idx_v[k] == row) j = k;
        const bool row_inert = j > 0 && (!ext || ((ft_mask >> row) & 1));
        const int s0 = j > 0 ? j : nl + 1, s1 = j > 0 ? T.subtree_end(j) : nl + 1;  // links s0 .. s1-1 = subtree
        int prev = -1, zero_from = -1;
        const bool compact = (flags & FIGH_FLAG_COMPACT_BLOCKS) != 0;
        // figh_model_set_active_rows: a row block that is not stored is still walked (its entries count in diag(W^T W))
        const bool stored = ext || row >= 64 || ((active_rows >> row) & 1ull);  // (rows >= 64: always stored, see figh_model_set_active_rows)
        auto flush_zero = [&](int upto_link) {
            if (zero_from < 0) return;
            if (stored && !(flags & (FIGH_FLAG_ZEROS_PRESENT | FIGH_FLAG_COMPACT_BLOCKS)))
                T.zero(row, ls * (zero_from - 1), ls * (upto_link - zero_from));
            zero_from = -1;
        };
        // block-compact: the row block's own leading dimension (its subtree's links); 0 for a block that is not stored
        const int ld_row = stored ? ls * (s1 - s0) : 0;
        for (int b = 1; b <= nl; ++b) {
            const bool in_sub = b >= s0 && b < s1;
            const bool inert = row_inert && in_sub && (!ext || h.body_mask[b]);
            const bool extra = extras && (ext || b == j);
            if (in_sub && row_inert) {  // the walk over the subtree keeps the state current even for massless links
                if (b == j || h.parents[b] != prev) {
                    T.push(OP_RESET);
                    for (int k : T.path_to(h.parents[b])) T.push(OP_STEP, k, k == j ? STEP_JSTART : 0);
                }
                T.push(OP_STEP, b, b == j ? STEP_JSTART : 0);
                prev = b;
            }
            if (inert || extra) {
                flush_zero(b);
                T.push(OP_EMIT, b, row,
                       (inert ? EMIT_INERT : 0) | (extra ? EMIT_EXTRA : 0) | ((!ext && b == j && row_inert) ? EMIT_OWN : 0) |
                           (stored ? 0 : EMIT_NOSTORE),
                       (compact && stored) ? ((ls * (b - s0) + 1) | (ld_row << 16)) : 0, compact ? compact_prefix : 0);
            } else if (zero_from < 0) {
                zero_from = b;
            }
        }
        flush_zero(nl + 1);
        compact_prefix += ld_row;
    }
    if (flags & FIGH_FLAG_TX40) T.push(OP_TX40, ls * nl);
    return with_fetches(h, T.ops, fetch_group(false));
}

// Joint-torque mode (every joint single-dof, one row block per joint): walks with kRowSlots row slots.  A walk has a top
// joint t; it steps down from the root to t and then over the subtree of t, and serves the row blocks of the joints of that
// subtree that lie fewer than kRowSlots levels below t (slot = levels below t).  A link is emitted once per served row joint
// above it (its own row first: Ia / fv / fs / off read the inputs of the joint stepped last).  Joints that are deeper become
// the tops of their own walks.  The zero runs of a row block (links outside its joint's subtree) do not depend on the walks.
std::vector<TapeOp> build_tape_torque_rows(const DevModel &h, int flags, int ls, unsigned long long active_rows) {
    TapeBuilder T(h);
    const bool extras = flags & (FIGH_FLAG_FRICTION | FIGH_FLAG_ACT_INERTIA | FIGH_FLAG_OFFSET);
    const bool compact = (flags & FIGH_FLAG_COMPACT_BLOCKS) != 0;
    const int nl = h.nlinks, nj = h.njoints;
    std::vector<int> depth(nj, 0), s1(nj, 0), ld_row(h.nv, 0), prefix(h.nv, 0), joint_of_row(h.nv, 0);
    for (int j = 1; j < nj; ++j) {
        depth[j] = h.parents[j] > 0 ? depth[h.parents[j]] + 1 : 0;
        s1[j] = T.subtree_end(j);
        joint_of_row[h.idx_v[j]] = j;
    }
    auto stored = [&](int row) { return row >= 64 || ((active_rows >> row) & 1ull) != 0; };
    int compact_prefix = 0;
    for (int row = 0; row < h.nv; ++row) {  // block-compact: a row block's own leading dimension, 0 when it is not stored
        const int j = joint_of_row[row];
        ld_row[row] = stored(row) ? ls * (s1[j] - j) : 0;
        prefix[row] = compact_prefix;
        compact_prefix += ld_row[row];
    }
    if (!(flags & (FIGH_FLAG_ZEROS_PRESENT | FIGH_FLAG_COMPACT_BLOCKS)))
        for (int row = 0; row < h.nv; ++row) {
            if (!stored(row)) continue;
            const int j = joint_of_row[row];
            T.zero(row, 0, ls * (j - 1));
            T.zero(row, ls * (s1[j] - 1), ls * (nl + 1 - s1[j]));
        }
    std::vector<char> served(nj, 0);
    for (int t = 1; t < nj; ++t) {
        if (served[t]) continue;
        auto in_walk = [&](int k) { return k >= t && k < s1[t] && depth[k] - depth[t] < kRowSlots; };
        for (int k = t; k < s1[t]; ++k)
            if (in_walk(k)) served[k] = 1;
        auto step = [&](int k) { T.push(OP_STEP, k, in_walk(k) ? STEP_JSTART : 0, in_walk(k) ? depth[k] - depth[t] : 0); };
        int prev = -1;
        for (int b = t; b < s1[t]; ++b) {
            if (b == t || h.parents[b] != prev) {  // a new branch: down from the root again
                T.push(OP_RESET);
                for (int k : T.path_to(h.parents[b])) step(k);
            }
            step(b);
            prev = b;
            std::vector<int> rows;  // the served row joints above the link, its own one first
            if (in_walk(b)) rows.push_back(b);
            for (int k = h.parents[b]; k >= t && k > 0; k = h.parents[k])
                if (in_walk(k)) rows.push_back(k);
            for (int j : rows) {
                const int row = h.idx_v[j];
                const bool extra = extras && b == j;
                T.push(OP_EMIT, b, row,
                       EMIT_INERT | (extra ? EMIT_EXTRA : 0) | (b == j ? EMIT_OWN : 0) | (stored(row) ? 0 : EMIT_NOSTORE) |
                           ((depth[j] - depth[t]) << 8),
                       (compact && stored(row)) ? ((ls * (b - j) + 1) | (ld_row[row] << 16)) : 0, compact ? prefix[row] : 0);
            }
        }
    }
    if (flags & FIGH_FLAG_TX40) T.push(OP_TX40, ls * nl);
    return with_fetches(h, T.ops, fetch_group(false));
}

struct DeviceTape {
    TapeOp *dev = nullptr;
    int n = 0;
    bool extff = false;
    int *link_pos = nullptr;  // FIGH_FLAG_LINK_COMPACT: device copy of the link -> segment position map (nlinks entries)
    int nlive = 0;
};
std::map<std::vector<long>, DeviceTape> g_tapes;  // (model handle, mode, flags, ft_mask, ls) -> tape

}  // namespace

// FIGH_FLAG_LINK_COMPACT (external wrench on a free-flyer root): position of every link's 16-column segment among the links
// that have one -- a link with mass (and a wrench component selected) or, with friction / inertia / offset flags, every link;
// -1 for the others, whose entries are structural zeros in all six row blocks (regressor.py:36-39: massless bodies are
// skipped).  Returns the number of links with a segment, or -1 when the layout does not apply to (model, mode).
int tree_link_positions(const figh_model_s *m, int mode, int flags, int ft_mask, int *pos) {
    const DevModel &h = m->host;
    if (!(mode == FIGH_MODE_EXT_WRENCH && h.njoints > 1 && h.jtype[1] == FIGH_JT_FREEFLYER) || (flags & FIGH_FLAG_TX40)) return -1;
    const bool extras = flags & (FIGH_FLAG_FRICTION | FIGH_FLAG_ACT_INERTIA | FIGH_FLAG_OFFSET);
    int live = 0;
    for (int b = 1; b < h.njoints; ++b) {
        const bool inert = h.body_mask[b] && (ft_mask & 63);
        pos[b - 1] = (inert || extras) ? live++ : -1;
    }
    return live;
}

// FIGH_FLAG_FORCE_COMPACT: leading dimension of the force region -- 16 columns per group of four links that have a segment
// (tree_link_positions: the force region is compacted over the links whatever the layout of the torque rows) -- or 0 when the layout does
// not apply: not an external-wrench regressor on a free-flyer root, or friction / actuator-inertia / offset columns (which
// every row of a link carries, regressor.py:142-169: eight entries per force row and link instead of four).
long tree_force_ld(const figh_model_s *m, int mode, int flags, int ft_mask) {
    int pos[kMaxJoints];
    int live = tree_link_positions(m, mode, flags, ft_mask, pos);
    if (live < 0 || (flags & (FIGH_FLAG_FRICTION | FIGH_FLAG_ACT_INERTIA | FIGH_FLAG_OFFSET))) return 0;
    if (live <= 0) return 0;
    return 16L * ((live + 3) / 4);
}

// internal: rows of W for a tree model.  ls = columns per link in W: 14 (the reference's layout, ncols = 14 nlinks [+ 3])
// or 16 (link-padded: columns 14, 15 of every link are zero, every row segment is one 128-byte line; needs ldw % 16 == 0
// and a 128-byte aligned W).  W == nullptr: column norms only.  d_colsq (nullable) receives diag(W^T W) in the
// reference's column numbering; *colsq_done = 0 when the norms could not be fused (odd column count / unaligned W: the
// caller runs figh_colsq).
int launch_regressor_tree(const figh_model_s *m, int mode, int flags, int ft_mask, long N, const double *q,
                          const double *v, const double *a, double *W, long ldw, int ncols, int ls, double *d_colsq,
                          int *colsq_done) {
    const DevModel &h = m->host;
    const bool store = W != nullptr;
    FIGH_REQUIRE(store || d_colsq, "nothing to compute");
    FIGH_REQUIRE(ls == 14 || ls == 16, "link stride must be 14 or 16");
    FIGH_REQUIRE(ls == 14 || !(flags & FIGH_FLAG_TX40), "the link-padded layout has no TX40 coupling columns");
    int link_pos[kMaxJoints];
    int nlive = -1;
    if (flags & FIGH_FLAG_LINK_COMPACT) {
        nlive = tree_link_positions(m, mode, flags, ft_mask, link_pos);
        FIGH_REQUIRE(nlive >= 0 && ls == 16, "link-compact W: external-wrench regressor of a free-flyer model, link-padded columns");
    }
    const int ncols_int = nlive >= 0 ? ls * (nlive > 0 ? nlive : 1) : ls * h.nlinks + ((flags & FIGH_FLAG_TX40) ? 3 : 0);
    // FIGH_FLAG_FORCE_COMPACT: the force row blocks in their own region, one line per four links (regressor_tape_kernel, FC)
    const bool fc = (flags & FIGH_FLAG_FORCE_COMPACT) != 0;
    long ldf = 0;
    if (fc) {
        ldf = tree_force_ld(m, mode, flags, ft_mask);
        FIGH_REQUIRE(ldf > 0 && ls == 16, "force-compact W: external-wrench regressor of a free-flyer model without friction / "
                                          "actuator-inertia / offset columns, link-padded torque rows");
        // (a link's place in the force region is its position among the links with entries: the torque rows must number
        // the links the same way, i.e. be link-compact unless every link has entries)
        int pos_all[kMaxJoints];
        FIGH_REQUIRE((flags & FIGH_FLAG_LINK_COMPACT) || tree_link_positions(m, mode, flags, ft_mask, pos_all) == h.nlinks,
                     "force-compact W of a model with links without entries goes with FIGH_FLAG_LINK_COMPACT");
    }
    if (!store) ldw = ncols_int;
    FIGH_REQUIRE(ldw >= ncols_int, "ldw smaller than the number of columns");
    FIGH_REQUIRE(ldw < (1L << 22), "figh_regressor_build: leading dimension must be below 2^22 elements");
    if (store && ls == 16)
        FIGH_REQUIRE(ldw % 16 == 0 && reinterpret_cast<uintptr_t>(W) % 128 == 0, "link-padded W must be 128-byte aligned");
    const bool extff = mode == FIGH_MODE_EXT_WRENCH && h.jtype[1] == FIGH_JT_FREEFLYER;
    for (int k = extff ? 2 : 1; k < h.njoints; ++k)
        FIGH_REQUIRE(h.jtype[k] != FIGH_JT_FREEFLYER, "a free-flyer joint is only supported as the root joint of the "
                                                      "external-wrench mode");
    if (flags & FIGH_FLAG_COMPACT_BLOCKS) {
        FIGH_REQUIRE(mode == FIGH_MODE_JOINT_TORQUE && ls == 16 && h.nv == h.njoints - 1 && !(flags & FIGH_FLAG_TX40),
                     "block-compact W: joint-torque regressor of single-dof joints, link-padded columns");
        FIGH_REQUIRE(16L * h.nlinks < (1L << 15), "block-compact W: too many links");
    }
    const std::vector<long> key = {(long)reinterpret_cast<uintptr_t>(m), mode,
                                   flags & (7 | FIGH_FLAG_TX40 | FIGH_FLAG_ZEROS_PRESENT | FIGH_FLAG_COMPACT_BLOCKS |
                                            FIGH_FLAG_LINK_COMPACT | FIGH_FLAG_FORCE_COMPACT), ft_mask,
                                   ls};
    auto it = g_tapes.find(key);
    if (it == g_tapes.end()) {
        std::vector<TapeOp> ops = extff ? build_tape_extff(h, flags, ft_mask, ls, nlive >= 0 ? link_pos : nullptr, fc)
                                        : (mode == FIGH_MODE_JOINT_TORQUE && h.nv == h.njoints - 1
                                               ? build_tape_torque_rows(h, flags, ls, m->active_rows)
                                               : build_tape_rows(h, mode, flags, ft_mask, ls, m->active_rows));
#ifdef FIGH_ABLATION
        {
            const int hot = getenv("FIGH_TREE_HOTIN") != nullptr;
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tree_hotin), &hot, sizeof(int));
            const int half = getenv("FIGH_TREE_HALF") != nullptr;
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tree_half), &half, sizeof(int));
        }
        if (const char *e = getenv("FIGH_TREE_TAPE")) {  // store-pattern ceilings: W is all zeros, timing only
            TapeBuilder T(h);
            const int nrows = mode == FIGH_MODE_EXT_WRENCH ? 6 : h.nv;
            if (!strncmp(e, "zrun", 4)) {  // runs of k links, row-block-major
                const int k = atoi(e + 4);
                for (int r = 0; r < nrows; ++r)
                    for (int b = 1; b <= h.nlinks; b += k)
                        T.zero(r, ls * (b - 1), ls * (b + k - 1 <= h.nlinks ? k : h.nlinks - b + 1));
            } else if (!strcmp(e, "zlink")) {  // one segment per (link, row block), link-major
                for (int b = 1; b <= h.nlinks; ++b)
                    for (int r = 0; r < nrows; ++r) T.zero(r, ls * (b - 1), ls);
            } else {  // whole rows
                for (int r = 0; r < nrows; ++r) T.zero(r, 0, ncols_int);
            }
            ops = T.ops;
        }
#endif
        DeviceTape dt;
        dt.n = (int)ops.size();
        dt.extff = extff;
        if (hipMalloc(&dt.dev, sizeof(TapeOp) * (ops.empty() ? 1 : ops.size())) != hipSuccess) {
            set_error("hipMalloc(tape) failed");
            return FIGH_ERR_ALLOC;
        }
        FIGH_HIP(hipMemcpy(dt.dev, ops.data(), sizeof(TapeOp) * ops.size(), hipMemcpyHostToDevice));
        if (nlive >= 0) {
            dt.nlive = nlive;
            if (hipMalloc(&dt.link_pos, sizeof(int) * kMaxJoints) != hipSuccess) {
                set_error("hipMalloc(link map) failed");
                return FIGH_ERR_ALLOC;
            }
            FIGH_HIP(hipMemcpy(dt.link_pos, link_pos, sizeof(int) * h.nlinks, hipMemcpyHostToDevice));
        }
        it = g_tapes.emplace(key, dt).first;
    }
    const DeviceTape &tp = it->second;
    const bool vec2 = !store || ls == 16 ||
                      ((ncols_int % 2 == 0) && (ldw % 2 == 0) && (reinterpret_cast<uintptr_t>(W) % 16 == 0));
    const bool fuse = d_colsq != nullptr && vec2;
    *colsq_done = fuse ? 1 : 0;
    (void)ncols;
    const long ntiles = (N + 63) / 64;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    // (force-compact: the norm fold of a partial last group of links touches up to 3 x 16 entries behind the last column)
    const size_t lds = sizeof(double) * (64 * (size_t)FIGH_TREE_LSP + 128 + (size_t)(fuse ? ncols_int + (fc ? 64 : 0) : 0));
    long grid = (long)cus * 8;
#ifdef FIGH_ABLATION
    if (const char *e = getenv("FIGH_TREE_WAVES")) grid = (long)cus * atoi(e);
#endif
    if (grid > ntiles) grid = ntiles;
    if (grid < 1) grid = 1;
    double *part = nullptr;
    if (fuse) {
        part = static_cast<double *>(workspace(sizeof(double) * grid * ncols_int, 0));
        if (!part) return FIGH_ERR_ALLOC;
    }
    ProfileScope scope("regressor_tree", true);
#define FIGH_TAPE_LAUNCH(LS, E, V2, ST, C)                                                                         \
    FIGH_LAUNCH_TIMED((regressor_tape_kernel<LS, E, V2, ST, C>), dim3((unsigned)grid), dim3(64), lds, m->dev, tp.dev, \
                      tp.n, flags, N, q, v, a, W, ldw, ncols_int, part, 0L)
#define FIGH_TAPE_LAUNCH_FC(ST, C)                                                                                       \
    FIGH_LAUNCH_TIMED((regressor_tape_kernel<16, true, true, ST, C, true>), dim3((unsigned)grid), dim3(64), lds, m->dev, \
                      tp.dev, tp.n, flags, N, q, v, a, W, ldw, ncols_int, part, ldf)
#define FIGH_TAPE_MODES(LS, E)                                      \
    do {                                                            \
        if (!store) FIGH_TAPE_LAUNCH(LS, E, true, false, true);     \
        else if (vec2 && fuse) FIGH_TAPE_LAUNCH(LS, E, true, true, true);  \
        else if (vec2) FIGH_TAPE_LAUNCH(LS, E, true, true, false);  \
        else FIGH_TAPE_LAUNCH(14, E, false, true, false);           \
    } while (0)
    if (fc) {
        FIGH_REQUIRE(tp.extff && vec2, "force-compact W needs the free-flyer walk");
        if (!store) FIGH_TAPE_LAUNCH_FC(false, true);
        else if (fuse) FIGH_TAPE_LAUNCH_FC(true, true);
        else FIGH_TAPE_LAUNCH_FC(true, false);
    } else if (ls == 16) {
        if (tp.extff) FIGH_TAPE_MODES(16, true);
        else FIGH_TAPE_MODES(16, false);
    } else {
        if (tp.extff) FIGH_TAPE_MODES(14, true);
        else FIGH_TAPE_MODES(14, false);
    }
#undef FIGH_TAPE_MODES
#undef FIGH_TAPE_LAUNCH
#undef FIGH_TAPE_LAUNCH_FC
    FIGH_HIP(hipGetLastError());
    if (fuse) {
        const int nref = 14 * h.nlinks;  // (TX40 tail: never fused, odd column count)
        hipLaunchKernelGGL(reduce_tree_partials_kernel, dim3(nref), dim3(256), 0, stream(), part, (int)grid, ncols_int, ls,
                           (const int *)tp.link_pos, d_colsq);
        FIGH_HIP(hipGetLastError());
    }
    return FIGH_OK;
}

void forget_tapes(const figh_model_s *m) {  // figh_model_destroy
    for (auto it = g_tapes.begin(); it != g_tapes.end();) {
        if (it->first[0] == (long)reinterpret_cast<uintptr_t>(m)) {
            (void)hipFree(it->second.dev);
            if (it->second.link_pos) (void)hipFree(it->second.link_pos);
            it = g_tapes.erase(it);
        } else {
            ++it;
        }
    }
}

}  // namespace figh
