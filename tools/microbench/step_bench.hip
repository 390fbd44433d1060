// What does a column step of the register-tile TSQR cost per wave -- alone on its SIMD and next to a second wave?
// The last-panel step of the fused kernel (RLAST form, one live chunk) on a 16-row-per-lane tile (one 64-row tile) and on a
// 32-row-per-lane tile (the last panel of two tiles at once), 16 steps per pass, the tile restored from registers between
// passes.  usage: step_bench   (prints ticks per step and wave for 1 and 2 waves per SIMD)
#include <hip/hip_runtime.h>

#include <cstdio>

#include "../../figaroh_plus_amd/csrc/figh_tsqr_narrow.h"
#include "../../figaroh_plus_amd/csrc/figh_tsqr_wide_kernel.h"

using namespace figh;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NRC, int LIVE_CHUNKS>
__global__ __launch_bounds__(512) void k_steps(double *out, long long *cyc, const int passes, const double seed,
                                               const int stagger) {
    __shared__ double lds[8][2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NCC = LIVE_CHUNKS;
    Tsqr2State<NCC, NRC, true> S;
    double *mine = lds[wave];
    for (int e = lane; e < 2048; e += 64) mine[e] = 0.0;
    S.red = mine;
    S.bc = mine + 64;
    S.Rl = mine + 80;
    S.lane_c = lane & 15;
    S.lane_g = lane >> 4;
    S.nc = 16 * NCC;
    S.null2 = 0.0;
#pragma unroll
    for (int s = 0; s < 4 * NCC; ++s) S.Rq[s] = 0.0;
    double keep[NCC][4 * NRC];
#pragma unroll
    for (int cc = 0; cc < NCC; ++cc)
#pragma unroll
        for (int i = 0; i < 4 * NRC; ++i) keep[cc][i] = seed + 1e-3 * ((lane * 31 + i * 7 + cc * 3) % 97) - 0.04;
    __syncthreads();
    // stagger > 0: wave w starts w * stagger ticks late, so that the waves of the workgroup sit at different places of the
    // (straight-line, 16 * NCC step bodies long) code instead of marching through it together -- what the consumers of the
    // fused kernel do; the difference to the lock-step run is what instruction fetch costs
    if (stagger > 0) {
        const long long until = __builtin_readcyclecounter() + (long long)wave * stagger;
        while (__builtin_readcyclecounter() < until) __builtin_amdgcn_s_sleep(8);
    }
    const long long t0 = __builtin_readcyclecounter();
    for (int p = 0; p < passes; ++p) {
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc)
#pragma unroll
            for (int i = 0; i < 4 * NRC; ++i) {
                S.T[cc][i] = keep[cc][i];
                asm volatile("" : "+v"(S.T[cc][i]));
            }
        tsqr2_panels<0, NCC, NRC, false, true>(S, 0, [](auto) {});
    }
    const long long t1 = __builtin_readcyclecounter();
    double acc = 0.0;
#pragma unroll
    for (int s = 0; s < 4 * NCC; ++s) acc += S.Rq[s];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + S.T[0][0];
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NRC, int NCC>
int run(const char *label, double *out, long long *cyc) {
    const int passes = 400;
    for (int mode = 0; mode < 4; ++mode) {
        const int waves = mode == 0 ? 1 : (mode == 1 ? 4 : 8);
        // staggered: the eight waves spread evenly over one pass of the code
        const int stagger = mode == 3 ? (int)(2200.0 * 16 * NCC / 8) : 0;
        long long best = 1LL << 60;
        for (int r = 0; r < 3; ++r) {
            hipLaunchKernelGGL((k_steps<NRC, NCC>), dim3(1), dim3(64 * waves), 0, 0, out, cyc, passes, 1.0 + r, stagger);
            long long h[8];
            CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
            long long worst = 0;
            for (int w = 0; w < waves; ++w) worst = h[w] > worst ? h[w] : worst;
            best = worst < best ? worst : best;
        }
        const int steps = 16 * NCC;
        printf("%-44s %d waves in the workgroup (%s%s): %7.1f ticks per column step and wave, %6.2f per tile-row-step\n", label, waves,
               waves == 8 ? "2 per SIMD" : "1 per SIMD", stagger ? ", staggered" : "", (double)best / (passes * steps),
               (double)best / (passes * steps) / (NRC / 4.0));
    }
    return 0;
}

// The PANEL CHAIN of the blocked kernel (wy_factor_panel: 16 dependent column steps on a 64 x 16 chunk) next to what shares
// its SIMD in tsqr_wy_kernel: nothing, a second panel chain, or a wave of the other workgroup streaming v_mfma_f64_16x16x4
// (a trailing sweep).  partner: 0 = chain waves only (1 per SIMD), 1 = every chain wave shares its SIMD with a second chain
// wave, 2 = ... with an MFMA wave.
typedef double f64x4_t __attribute__((ext_vector_type(4)));
template <int TMODE>
__global__ __launch_bounds__(512) void k_panel(double *out, long long *cyc, const int passes, const int partner, const double seed) {
    __shared__ double rl[8][256], red[8][64], vl[8][64 * 17 + 16 * 17];
    __shared__ int stop;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) stop = 0;
    __syncthreads();
    const int c = lane & 15, g = lane >> 4;
    const bool chain = wave < 4 || partner == 1;
    if (chain) {
        if (partner == 3) __builtin_amdgcn_s_setprio(3);  // (what tsqr_wy_kernel gives the owner of a panel)
        double keep[16], X[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) keep[i] = seed + 1e-3 * ((lane * 31 + i * 7) % 97) - 0.04;
        const long long t0 = __builtin_readcyclecounter();
        for (int p = 0; p < passes; ++p) {
            for (int e = lane; e < 256; e += 64) rl[wave][e] = (e / 16 == e % 16) ? 3.0 : 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                X[i] = keep[i];
                asm volatile("" : "+v"(X[i]));
            }
            wy_factor_panel<16, TMODE>(X, rl[wave], red[wave], vl[wave], vl[wave] + 64 * 17, lane, c, g, 0.0);
        }
        const long long t1 = __builtin_readcyclecounter();
        out[threadIdx.x] = X[0] + rl[wave][lane];
        if (lane == 0) cyc[wave] = t1 - t0;
        if (wave == 0 && lane == 0) __hip_atomic_store(&stop, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        f64x4_t a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        double x = seed + lane, y = 0.5 - lane;
        while (__hip_atomic_load(&stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
            }
        }
        out[threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
        if (lane == 0) cyc[wave] = 0;
    }
}

template <int TMODE>
int run_panel(double *out, long long *cyc) {
    const int passes = 300;
    const char *what[4] = {"alone on its SIMD", "next to a second panel chain", "next to a wave streaming v_mfma_f64_16x16x4",
                           "next to an MFMA wave, s_setprio 3"};
    for (int partner = 0; partner < 4; ++partner) {
        long long best = 1LL << 60;
        for (int r = 0; r < 3; ++r) {
            hipLaunchKernelGGL(k_panel<TMODE>, dim3(1), dim3(partner == 0 ? 256 : 512), 0, 0, out, cyc, passes, partner, 1.0 + r);
            long long h[8];
            CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
            long long worst = 0;
            for (int w = 0; w < 4; ++w) worst = h[w] > worst ? h[w] : worst;
            best = worst < best ? worst : best;
        }
        printf("panel chain of the blocked kernel (16 steps on 64 x 16; %s) %-46s %7.1f ticks per column step\n",
               TMODE == 0 ? "T inside the steps" : TMODE == 1 ? "Gram entries only, NO T" : "Gram entries + one larft per panel", what[partner],
               (double)best / (passes * 16));
    }
    return 0;
}

int main() {
    double *out;
    long long *cyc;
    CHECK(hipMalloc(&out, sizeof(double) * 4096));
    CHECK(hipMalloc(&cyc, sizeof(long long) * 64));
    if (run_panel<0>(out, cyc)) return 1;
    if (run_panel<1>(out, cyc)) return 1;
    if (run_panel<2>(out, cyc)) return 1;
    if (run<4, 1>("one chunk, 64-row tile (last panel)", out, cyc)) return 1;
    if (run<8, 1>("one chunk, 128 rows (last panel of two tiles)", out, cyc)) return 1;
    if (run<4, 2>("two chunks, 64-row tile (panels 2, 3)", out, cyc)) return 1;
    if (run<4, 4>("four chunks, 64-row tile (whole UR10 tile)", out, cyc)) return 1;
    return 0;
}
