// Zero-phase IIR filtering and decimation of the columns of a row-block matrix (the step on
// either side of the hot path on real data):
//   - scipy.signal.decimate(x, q, zero_phase=True) over every column of W_b and over tau, per joint block
//     (examples/staubli_TX40/identification.py:186-204, examples/tiago/identification.py:142-187): Chebyshev-I order 8
//     as four second-order sections, sosfiltfilt with odd padding, every q-th sample kept;
//   - signal.filtfilt(b, a, q, padtype='odd', padlen=...) of the joint positions (identification_tools.py:390-424).
// One thread per (row block, column) sequence runs the recurrences exactly as SciPy's C / Cython loops do (same
// operation order, no FMA contraction) so that the result is bit-comparable with the reference's dependency: the
// forward pass goes through a library workspace [sample][sequence] (coalesced across sequences), the backward pass
// writes the kept samples.  The recursion is sequential in time by nature; the parallelism is the number of sequences
// (blocks x columns, 360-522 for the TX40 data) -- the matrices of this step are 10^4-10^5 samples long.
#include <vector>

#include "figh_internal.h"

using namespace figh;

namespace {

constexpr int kMaxSections = 8, kMaxOrder = 16;

struct FilterParams {
    int form;      // 0 = second-order sections (sosfilt), 1 = transfer function (lfilter)
    int nsec;      // sections (form 0) or 1
    int order;     // 2 (form 0) or len(b) - 1 (form 1)
    double b[kMaxSections][kMaxOrder + 1];
    double a[kMaxSections][kMaxOrder + 1];
    double zi[kMaxSections][kMaxOrder];
};

// one sample through the cascade; z is the per-thread state.  __dmul_rn / __dadd_rn / __dsub_rn are never contracted
// into FMAs (the file-level -ffp-contract=fast would fuse plain a * b + c regardless of pragmas): the roundings are
// those of SciPy's compiled loops, which makes the device result bit-equal to scipy.signal on the same input.
// FORM / NS / NO are compile-time for the common designs (state in registers, coefficients in SGPRs); NS = 0 selects
// the generic run-time loops (state in scratch: slow, any order up to kMaxOrder).
template <int FORM, int NS, int NO, int ZS, int ZO>
__device__ __forceinline__ double filter_step(const FilterParams &F, double (&z)[ZS][ZO], double x) {
    const int nsec = NS ? NS : F.nsec, order = NO ? NO : F.order;
    if (FORM == 0) {
#pragma unroll
        for (int s = 0; s < nsec; ++s) {  // scipy/signal/_sosfilt.pyx
            const double y = __dadd_rn(__dmul_rn(F.b[s][0], x), z[s][0]);
            z[s][0] = __dadd_rn(__dsub_rn(__dmul_rn(F.b[s][1], x), __dmul_rn(F.a[s][1], y)), z[s][1]);
            z[s][1] = __dsub_rn(__dmul_rn(F.b[s][2], x), __dmul_rn(F.a[s][2], y));
            x = y;
        }
        return x;
    }
    // scipy/signal/_lfilter.c.in (coefficients already divided by a[0])
    const double y = __dadd_rn(z[0][0], __dmul_rn(F.b[0][0], x));
#pragma unroll
    for (int i = 0; i + 1 < order; ++i)
        z[0][i] = __dsub_rn(__dadd_rn(z[0][i + 1], __dmul_rn(x, F.b[0][i + 1])), __dmul_rn(y, F.a[0][i + 1]));
    z[0][order - 1] = __dsub_rn(__dmul_rn(x, F.b[0][order]), __dmul_rn(y, F.a[0][order]));
    return y;
}

template <int FORM, int NS, int NO>
__global__ __launch_bounds__(64) void filtfilt_cols_kernel(const double *__restrict__ X, const long L, const int cols,
                                                           const long ldx, const int nblocks, const FilterParams F,
                                                           const int edge, const int q, double *__restrict__ work,
                                                           double *__restrict__ Y, const long ldy, const long Lout) {
    const long seq = (long)blockIdx.x * 64 + threadIdx.x;
    const long nseq = (long)nblocks * cols;
    if (seq >= nseq) return;
    const int blk = (int)(seq / cols), col = (int)(seq - (long)blk * cols);
    const double *x = X + (long)blk * L * ldx + col;  // x[n] = x[n * ldx]
    const long Lext = L + 2L * edge;
    const double x_first = x[0], x_last = x[(L - 1) * ldx];
    auto ext = [&](const long n) -> double {  // odd extension (scipy _arraytools.odd_ext)
        if (n < edge) return __dsub_rn(__dmul_rn(2.0, x_first), x[(edge - n) * ldx]);
        if (n >= edge + L) return __dsub_rn(__dmul_rn(2.0, x_last), x[(L - 2 - (n - edge - L)) * ldx]);
        return x[(n - edge) * ldx];
    };
    constexpr int ZS = NS ? (FORM == 0 ? NS : 1) : kMaxSections, ZO = NS ? NO : kMaxOrder;
    double z[ZS][ZO];
    const int ns = NS ? ZS : (F.form == 0 ? F.nsec : 1), no = NS ? ZO : (F.form == 0 ? 2 : F.order);
    const double x0 = ext(0);
    for (int s = 0; s < ns; ++s)
        for (int i = 0; i < no; ++i) z[s][i] = __dmul_rn(F.zi[s][i], x0);
    // The recurrence is a dependent chain per sample; the loads are not.  Both passes therefore fetch TB samples with
    // independent (coalesced across the 64 sequences of the wave) loads first and then run the TB filter steps from
    // registers, instead of paying one HBM round trip per sample.
    constexpr int TB = 32;
    double xb[TB];
    double ylast = 0.0;
    for (long n0 = 0; n0 < Lext; n0 += TB) {
        const int cnt = (int)((Lext - n0) < TB ? (Lext - n0) : TB);
        if (n0 >= edge && n0 + TB <= edge + L) {  // interior: plain strided loads
#pragma unroll
            for (int k = 0; k < TB; ++k) xb[k] = x[(n0 + k - edge) * ldx];
        } else {
#pragma unroll
            for (int k = 0; k < TB; ++k) xb[k] = (k < cnt) ? ext(n0 + k) : 0.0;
        }
#pragma unroll
        for (int k = 0; k < TB; ++k) {
            if (k < cnt) {
                ylast = filter_step<FORM, NS, NO>(F, z, xb[k]);
                xb[k] = ylast;
            }
        }
#pragma unroll
        for (int k = 0; k < TB; ++k)
            if (k < cnt) work[(n0 + k) * nseq + seq] = xb[k];
    }
    for (int s = 0; s < ns; ++s)
        for (int i = 0; i < no; ++i) z[s][i] = __dmul_rn(F.zi[s][i], ylast);
    double *y = Y + (long)blk * Lout * ldy + col;
    for (long n1 = Lext; n1 > 0; n1 -= TB) {  // samples n1-1 down to max(n1-TB, 0)
        const int cnt = (int)(n1 < TB ? n1 : TB);
#pragma unroll
        for (int k = 0; k < TB; ++k) xb[k] = (k < cnt) ? work[(n1 - 1 - k) * nseq + seq] : 0.0;
#pragma unroll
        for (int k = 0; k < TB; ++k) {
            if (k < cnt) {
                const double v = filter_step<FORM, NS, NO>(F, z, xb[k]);
                const long m = n1 - 1 - k - edge;
                if (m >= 0 && m < L && m % q == 0) y[(m / q) * ldy] = v;
            }
        }
    }
}

}  // namespace

extern "C" int figh_filtfilt_cols(const double *d_X, int64_t rows, int cols, int64_t ldx, int nblocks, int form,
                                  const double *h_b, const double *h_a, int nsec, int order, const double *h_zi,
                                  int padlen, int q, double *d_Y, int64_t ldy, int64_t *rows_out) {
    FIGH_REQUIRE(d_X && d_Y && h_b && h_a && h_zi, "NULL pointer");
    FIGH_REQUIRE(rows > 0 && cols > 0 && ldx >= cols && ldy >= cols && nblocks > 0 && rows % nblocks == 0, "bad shape");
    FIGH_REQUIRE(form == 0 || form == 1, "form must be 0 (sos) or 1 (tf)");
    FIGH_REQUIRE(q >= 1 && padlen >= 0, "bad decimation factor / padlen");
    if (form == 0) FIGH_REQUIRE(nsec >= 1 && nsec <= kMaxSections && order == 2, "bad section count");
    else FIGH_REQUIRE(nsec == 1 && order >= 1 && order <= kMaxOrder, "bad filter order");
    const int64_t L = rows / nblocks;
    // scipy _validate_pad: "The length of the input vector x must be greater than padlen"
    FIGH_REQUIRE(L > padlen, "The length of the input vector x must be greater than padlen");
    if (int rc = ensure_device()) return rc;
    FilterParams F;
    F.form = form;
    F.nsec = nsec;
    F.order = order;
    for (int s = 0; s < nsec; ++s) {
        for (int i = 0; i <= order; ++i) {
            F.b[s][i] = h_b[s * (order + 1) + i];
            F.a[s][i] = h_a[s * (order + 1) + i];
        }
        for (int i = 0; i < order; ++i) F.zi[s][i] = h_zi[s * order + i];
    }
    const int64_t Lout = (L + q - 1) / q;
    if (rows_out) *rows_out = Lout * nblocks;
    const int64_t nseq = (int64_t)nblocks * cols;
    double *work = static_cast<double *>(workspace(sizeof(double) * (size_t)(L + 2 * (int64_t)padlen) * nseq, 13));
    if (!work) return FIGH_ERR_ALLOC;
    ProfileScope scope("filtfilt_cols");
    const dim3 grid((unsigned)((nseq + 63) / 64)), block(64);
#define FIGH_FF_LAUNCH(FORM, NS, NO)                                                                                 \
    hipLaunchKernelGGL((filtfilt_cols_kernel<FORM, NS, NO>), grid, block, 0, stream(), d_X, (long)L, cols, (long)ldx, \
                       nblocks, F, padlen, q, work, d_Y, (long)ldy, (long)Lout)
    if (form == 0) {
        switch (nsec) {
            case 1: FIGH_FF_LAUNCH(0, 1, 2); break;
            case 2: FIGH_FF_LAUNCH(0, 2, 2); break;
            case 3: FIGH_FF_LAUNCH(0, 3, 2); break;
            case 4: FIGH_FF_LAUNCH(0, 4, 2); break;  // scipy.signal.decimate default: Chebyshev-I order 8
            case 5: FIGH_FF_LAUNCH(0, 5, 2); break;
            case 6: FIGH_FF_LAUNCH(0, 6, 2); break;
            default: FIGH_FF_LAUNCH(0, 0, 0); break;
        }
    } else {
        switch (order) {
            case 1: FIGH_FF_LAUNCH(1, 1, 1); break;
            case 2: FIGH_FF_LAUNCH(1, 1, 2); break;
            case 3: FIGH_FF_LAUNCH(1, 1, 3); break;
            case 4: FIGH_FF_LAUNCH(1, 1, 4); break;  // the TX40 script's Butterworth
            case 5: FIGH_FF_LAUNCH(1, 1, 5); break;  // low_pass_filter_data default
            case 6: FIGH_FF_LAUNCH(1, 1, 6); break;
            case 8: FIGH_FF_LAUNCH(1, 1, 8); break;
            default: FIGH_FF_LAUNCH(1, 0, 0); break;
        }
    }
#undef FIGH_FF_LAUNCH
    FIGH_HIP(hipGetLastError());
    return FIGH_OK;
}
