// gfx950 dependent-issue latencies of the instructions on the TSQR column-step critical path:
// one wave per SIMD, a chain of N dependent instructions, cycles per link = elapsed s_memtime / N.
// build: hipcc -O3 --offload-arch=gfx950 latency.hip -o latency
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int N = 2048;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double allreduce_rowgroups(double x) {
    unsigned lo = __double2loint(x), hi = __double2hiint(x);
    u32x2 a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    u32x2 b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const double y = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    lo = __double2loint(y);
    hi = __double2hiint(y);
    a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
template <int K>
__device__ __forceinline__ double row_bcast(double x) {
    return __longlong_as_double(__builtin_amdgcn_update_dpp((long long)0, __double_as_longlong(x), 0x150 + K, 0xf, 0xf, true));
}

#define CHAIN_KERNEL(name, body)                                                        \
    __global__ void name(double *out, long long *cyc, double a, double b) {             \
        double x = a + threadIdx.x * 1e-3, y = b;                                       \
        const long long t0 = __builtin_readcyclecounter();                              \
        _Pragma("unroll 16") for (int i = 0; i < N; ++i) { body; }                      \
        const long long t1 = __builtin_readcyclecounter();                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x + y;                             \
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                      \
    }

CHAIN_KERNEL(k_fma, x = fma(x, a, b))
CHAIN_KERNEL(k_fma2, x = fma(x, a, b); y = fma(y, a, b))
CHAIN_KERNEL(k_fma4, x = fma(x, a, b); y = fma(y, a, b); a = fma(a, 1.0, 1e-30); b = fma(b, 1.0, 1e-30))
CHAIN_KERNEL(k_mul, x = x * a)
CHAIN_KERNEL(k_add, x = x + a)
CHAIN_KERNEL(k_rsq, x = __builtin_amdgcn_rsq(x) + a)
CHAIN_KERNEL(k_rcp, x = __builtin_amdgcn_rcp(x) + a)
CHAIN_KERNEL(k_dppmov, x = row_bcast<3>(x) + a)
CHAIN_KERNEL(k_fmacdpp, asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y), "v"(a)))
CHAIN_KERNEL(k_fmacdpp_self, asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(a)))
CHAIN_KERNEL(k_allreduce, x = allreduce_rowgroups(x) * a)
CHAIN_KERNEL(k_readfirst, x = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x))) + a)

__global__ void k_lds(double *out, long long *cyc, double a, double b) {
    __shared__ double buf[1024];
    double x = a + threadIdx.x * 1e-3;
    buf[threadIdx.x] = x;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) {
        buf[threadIdx.x] = x;
        x = buf[threadIdx.x ^ 1] + a;   // write -> read of a neighbour's value: LDS round trip
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_ldsread(double *out, long long *cyc, double a, double b) {
    __shared__ double buf[1024];
    buf[threadIdx.x] = (double)((threadIdx.x + 1) & 63);
    __syncthreads();
    int idx = threadIdx.x;
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) idx = (int)buf[idx];   // pointer chase: ds_read latency + cvt
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = idx;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// uniform branch in the chain: value -> readfirstlane -> s_cmp -> s_cbranch
__global__ void k_branch(double *out, long long *cyc, double a, double b) {
    double x = a + threadIdx.x * 1e-3;
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < N; ++i) {
        const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
        if (hi == 0x7ff00000) break;
        x = x + a;
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    double *out; long long *cyc;
    CHECK(hipMalloc(&out, sizeof(double) * 4096));
    CHECK(hipMalloc(&cyc, sizeof(long long)));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s  (cycle counter = s_memtime; shader clock %d MHz)\n", prop.gcnArchName, prop.clockRate / 1000);
#define RUN(k, label, per)                                                                                   \
    {                                                                                                        \
        long long best = 1LL << 60;                                                                          \
        for (int r = 0; r < 5; ++r) {                                                                        \
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, 1.0000001, 0.5);                        \
            long long c; CHECK(hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost));                        \
            if (c < best) best = c;                                                                          \
        }                                                                                                    \
        printf("%-46s %7.1f counter ticks per link\n", label, (double)best / (N * (per)));                   \
    }
    RUN(k_fma, "v_fma_f64 dependent chain", 1);
    RUN(k_fma2, "v_fma_f64 two interleaved chains (per instr)", 2);
    RUN(k_fma4, "v_fma_f64 four interleaved chains (per instr)", 4);
    RUN(k_mul, "v_mul_f64 dependent chain", 1);
    RUN(k_add, "v_add_f64 dependent chain", 1);
    RUN(k_rsq, "v_rsq_f64 + v_add_f64", 1);
    RUN(k_rcp, "v_rcp_f64 + v_add_f64", 1);
    RUN(k_dppmov, "v_mov_b64_dpp row_newbcast + v_add_f64", 1);
    RUN(k_fmacdpp, "v_fmac_f64_dpp (acc chain)", 1);
    RUN(k_fmacdpp_self, "s_nop 1 + v_fmac_f64_dpp (acc = dpp source)", 1);
    RUN(k_allreduce, "allreduce_rowgroups (4 permlane swaps, 2 adds) + mul", 1);
    RUN(k_readfirst, "2 v_readfirstlane + v_add_f64 (SGPR round trip)", 1);
    RUN(k_lds, "ds_write_b64 -> ds_read_b64 + v_add_f64", 1);
    RUN(k_ldsread, "ds_read_b64 pointer chase (+cvt)", 1);
    RUN(k_branch, "readfirstlane + s_cmp + s_cbranch + v_add_f64 loop", 1);
    return 0;
}
