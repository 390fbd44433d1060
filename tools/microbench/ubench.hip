// gfx950 instruction-rate microbenchmarks used to price the TSQR kernel:
// fp64 FMA, v_mfma_f64_16x16x4, ds_bpermute_b32, v_readlane_b32, fp64 rcp/rsq/sqrt/div.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 4096;
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_fma(double *out, double a, double b) {
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; ++i) {
        x0 = x0 * a + b; x1 = x1 * a + b; x2 = x2 * a + b; x3 = x3 * a + b;
        x4 = x4 * a + b; x5 = x5 * a + b; x6 = x6 * a + b; x7 = x7 * a + b;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_mfma(double *out, double a, double b) {
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double x = a + threadIdx.x, y = b - threadIdx.x;
    for (int i = 0; i < ITERS; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ void k_bperm(double *out, int src) {
    int v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    int addr = src << 2;
    for (int i = 0; i < ITERS; ++i) {
        v0 = __builtin_amdgcn_ds_bpermute(addr, v0) + 1; v1 = __builtin_amdgcn_ds_bpermute(addr, v1) + 1;
        v2 = __builtin_amdgcn_ds_bpermute(addr, v2) + 1; v3 = __builtin_amdgcn_ds_bpermute(addr, v3) + 1;
        v4 = __builtin_amdgcn_ds_bpermute(addr, v4) + 1; v5 = __builtin_amdgcn_ds_bpermute(addr, v5) + 1;
        v6 = __builtin_amdgcn_ds_bpermute(addr, v6) + 1; v7 = __builtin_amdgcn_ds_bpermute(addr, v7) + 1;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}
// bpermute (LDS pipe) interleaved with independent fp64 FMAs (VALU): do they overlap?
__global__ void k_bperm_fma(double *out, int src, double a, double b) {
    int v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    int addr = src << 2;
    for (int i = 0; i < ITERS; ++i) {
        v0 = __builtin_amdgcn_ds_bpermute(addr, v0); x0 = x0 * a + b;
        v1 = __builtin_amdgcn_ds_bpermute(addr, v1); x1 = x1 * a + b;
        v2 = __builtin_amdgcn_ds_bpermute(addr, v2); x2 = x2 * a + b;
        v3 = __builtin_amdgcn_ds_bpermute(addr, v3); x3 = x3 * a + b;
        v0 = __builtin_amdgcn_ds_bpermute(addr, v0); x0 = x0 * a + b;
        v1 = __builtin_amdgcn_ds_bpermute(addr, v1); x1 = x1 * a + b;
        v2 = __builtin_amdgcn_ds_bpermute(addr, v2); x2 = x2 * a + b;
        v3 = __builtin_amdgcn_ds_bpermute(addr, v3); x3 = x3 * a + b;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + x0 + x1 + x2 + x3;
}
__global__ void k_readlane_fma(double *out, int src, double a) {
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, y0 = 1, y1 = 2, y2 = 3, y3 = 4;
    for (int i = 0; i < ITERS; ++i) {
#define RL(v) __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src))
        y0 = y0 * a + RL(x0); y1 = y1 * a + RL(x1); y2 = y2 * a + RL(x2); y3 = y3 * a + RL(x3);
        x0 += y3; x1 += y2; x2 += y1; x3 += y0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = y0 + y1 + y2 + y3;
}
__global__ void k_sqrtdiv(double *out, double a) {
    double x = a + threadIdx.x;
    for (int i = 0; i < ITERS / 16; ++i) {
        double s = sqrt(x * x + a);
        double inv = 1.0 / (x + s);
        double t = (x + s) / s;
        x = inv * t + a;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ void k_fastrcp(double *out, double a) {
    double x = a + threadIdx.x;
    for (int i = 0; i < ITERS / 16; ++i) {
        double q = x * x + a;
        double r = __builtin_amdgcn_rsq(q);
        r = r * (1.5 - 0.5 * q * r * r);
        r = r * (1.5 - 0.5 * q * r * r);
        double s = q * r;
        double d = x + s;
        double inv = __builtin_amdgcn_rcp(d);
        inv = inv * (2.0 - d * inv);
        inv = inv * (2.0 - d * inv);
        double t = d * r;
        x = inv * t + a;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

template <typename F> static float time_it(F launch, int reps = 5) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    return best;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double clk = prop.clockRate * 1e3;  // Hz
    printf("device %s CUs %d clock %.0f MHz\n", prop.gcnArchName, cus, clk / 1e6);
    double *out; CHECK(hipMalloc(&out, sizeof(double) * cus * 8 * 256));
    for (int wpc : {4, 8, 16}) {   // waves per CU
        dim3 grid(cus * wpc / 4), block(256);
        const double nw = (double)cus * wpc;
        float ms = time_it([&] { hipLaunchKernelGGL(k_fma, grid, block, 0, 0, out, 1.0000001, 1e-9); });
        printf("waves/CU %2d  fp64 FMA   : %.2f TFLOP/s  (%.2f cyc/wave-instr/SIMD @%.0fMHz)\n", wpc, nw * 64 * ITERS * 8 * 2 / ms / 1e9, ms * 1e-3 * clk / (ITERS * 8.0 * wpc / 4), clk / 1e6);
        ms = time_it([&] { hipLaunchKernelGGL(k_mfma, grid, block, 0, 0, out, 1.0, 2.0); });
        printf("waves/CU %2d  mfma f64   : %.2f TFLOP/s  (%.2f cyc/instr/SIMD)\n", wpc, nw * ITERS * 4 * 2048.0 / ms / 1e9, ms * 1e-3 * clk / (ITERS * 4.0 * wpc / 4));
        ms = time_it([&] { hipLaunchKernelGGL(k_bperm, grid, block, 0, 0, out, 5); });
        printf("waves/CU %2d  bpermute   : %.2f cyc/instr/CU  (%.2f cyc per wave-instr/SIMD-equivalent)\n", wpc, ms * 1e-3 * clk / (ITERS * 8.0 * wpc), ms * 1e-3 * clk / (ITERS * 8.0 * wpc / 4));
        ms = time_it([&] { hipLaunchKernelGGL(k_bperm_fma, grid, block, 0, 0, out, 5, 1.0000001, 1e-9); });
        printf("waves/CU %2d  bperm+fma  : %.2f cyc per (bperm,fma) pair per SIMD\n", wpc, ms * 1e-3 * clk / (ITERS * 8.0 * wpc / 4));
        ms = time_it([&] { hipLaunchKernelGGL(k_readlane_fma, grid, block, 0, 0, out, 5, 1.0000001); });
        printf("waves/CU %2d  2readlane+fma+add: %.2f cyc per group per SIMD\n", wpc, ms * 1e-3 * clk / (ITERS * 4.0 * wpc / 4));
        ms = time_it([&] { hipLaunchKernelGGL(k_sqrtdiv, grid, block, 0, 0, out, 1.5); });
        printf("waves/CU %2d  sqrt+2div  : %.1f cyc per chain per SIMD (latency-bound at low occupancy)\n", wpc, ms * 1e-3 * clk / (ITERS / 16.0 * wpc / 4));
        ms = time_it([&] { hipLaunchKernelGGL(k_fastrcp, grid, block, 0, 0, out, 1.5); });
        printf("waves/CU %2d  rsq+rcp NR : %.1f cyc per chain per SIMD\n", wpc, ms * 1e-3 * clk / (ITERS / 16.0 * wpc / 4));
    }
    return 0;
}
