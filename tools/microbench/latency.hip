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

// throughput: 8 independent chains per iteration
#define TP_KERNEL(name, decl, body)                                                     \
    __global__ void name(double *out, long long *cyc, double a, double b) {             \
        double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;  \
        double y = b + threadIdx.x * 1e-3; decl;                                         \
        const long long t0 = __builtin_readcyclecounter();                              \
        _Pragma("unroll 4") for (int i = 0; i < N / 8; ++i) { body; }                   \
        const long long t1 = __builtin_readcyclecounter();                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + y; \
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                      \
    }
#define FD(x) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y), "v"(a))
TP_KERNEL(k_tp_fmacdpp, , FD(x0); FD(x1); FD(x2); FD(x3); FD(x4); FD(x5); FD(x6); FD(x7))
#define FP(x) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(x) : "v"(y), "v"(a))
TP_KERNEL(k_tp_fmac, , FP(x0); FP(x1); FP(x2); FP(x3); FP(x4); FP(x5); FP(x6); FP(x7))
#define MD(x) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(x) : "v"(y))
TP_KERNEL(k_tp_movdpp, , MD(x0); MD(x1); MD(x2); MD(x3); MD(x4); MD(x5); MD(x6); MD(x7))
#define RS(x) x = __builtin_amdgcn_rsq(x)
TP_KERNEL(k_tp_rsq, , RS(x0); RS(x1); RS(x2); RS(x3); RS(x4); RS(x5); RS(x6); RS(x7))
#define SW(x) { unsigned lo = __double2loint(x), hi = __double2hiint(x); u32x2 p = __builtin_amdgcn_permlane32_swap(lo, hi, false, false); x = __hiloint2double(p[1], p[0]); }
TP_KERNEL(k_tp_swap, , SW(x0); SW(x1); SW(x2); SW(x3); SW(x4); SW(x5); SW(x6); SW(x7))
// does the f64 MFMA overlap with independent fp64 VALU work of the same wave?  1 MFMA (64 ticks alone) + 12 FMAs (60 ticks alone)
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k_tp_mfma_fma(double *out, long long *cyc, double a, double b) {
    double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, y = b + threadIdx.x * 1e-3;
    d4 c0 = {0, 0, 0, 0}, c1 = c0;
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 4
    for (int i = 0; i < N / 8; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, a, c0, 0, 0, 0);
        FP(x0); FP(x1); FP(x2); FP(x3); FP(x4); FP(x5);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, a, c1, 0, 0, 0);
        FP(x0); FP(x1); FP(x2); FP(x3); FP(x4); FP(x5);
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + c0[0] + c1[1];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_tp_mfma(double *out, long long *cyc, double a, double b) {
    double y = b + threadIdx.x * 1e-3;
    d4 c0 = {0, 0, 0, 0}, c1 = c0;
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 4
    for (int i = 0; i < N / 8; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, a, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, a, c1, 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = c0[0] + c1[1];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
#define M32(x) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(x) : "v"(yy))
__global__ void k_tp_mov32(double *out, long long *cyc, double a, double b) {
    int x0, x1, x2, x3, x4, x5, x6, x7; int yy = threadIdx.x;
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 4
    for (int i = 0; i < N / 8; ++i) { M32(x0); M32(x1); M32(x2); M32(x3); M32(x4); M32(x5); M32(x6); M32(x7); }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// seed accuracy of v_rsq_f64 / v_rcp_f64 (max relative error over a sweep), and after one / two Newton steps
__global__ void k_seed_accuracy(double *out) {
    double e_rsq = 0, e_rsq1 = 0, e_rsq2 = 0, e_rcp = 0, e_rcp1 = 0, e_rcp2 = 0;
    for (int i = 0; i < 20000; ++i) {
        const double q = (1.0 + (threadIdx.x * 20000 + i) * 7.8125e-7) * (1 + (i % 7));
        const double exact_rs = 1.0 / sqrt(q), exact_rc = 1.0 / q;
        double r = __builtin_amdgcn_rsq(q);
        e_rsq = fmax(e_rsq, fabs(r - exact_rs) / exact_rs);
        r = r * fma(-0.5 * q * r, r, 1.5);
        e_rsq1 = fmax(e_rsq1, fabs(r - exact_rs) / exact_rs);
        r = r * fma(-0.5 * q * r, r, 1.5);
        e_rsq2 = fmax(e_rsq2, fabs(r - exact_rs) / exact_rs);
        double c = __builtin_amdgcn_rcp(q);
        e_rcp = fmax(e_rcp, fabs(c - exact_rc) / exact_rc);
        c = c * fma(-q, c, 2.0);
        e_rcp1 = fmax(e_rcp1, fabs(c - exact_rc) / exact_rc);
        c = c * fma(-q, c, 2.0);
        e_rcp2 = fmax(e_rcp2, fabs(c - exact_rc) / exact_rc);
    }
    double *o = out + threadIdx.x * 6;
    o[0] = e_rsq; o[1] = e_rsq1; o[2] = e_rsq2; o[3] = e_rcp; o[4] = e_rcp1; o[5] = e_rcp2;
}

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
    RUN(k_tp_fmac, "THROUGHPUT v_fmac_f64 (8 chains, per instr)", 1);
    RUN(k_tp_fmacdpp, "THROUGHPUT v_fmac_f64_dpp (8 chains, per instr)", 1);
    RUN(k_tp_movdpp, "THROUGHPUT v_mov_b64_dpp (per instr)", 1);
    RUN(k_tp_rsq, "THROUGHPUT v_rsq_f64 (8 chains, per instr)", 1);
    RUN(k_tp_mfma, "THROUGHPUT v_mfma_f64_16x16x4 alone (per pair of MFMA)", 1.0 / 8);
    RUN(k_tp_mfma_fma, "2 MFMA f64 + 12 independent v_fmac_f64 (per group)", 1.0 / 8);
    RUN(k_tp_swap, "THROUGHPUT v_permlane32_swap_b32 (per instr)", 1);
    RUN(k_tp_mov32, "THROUGHPUT v_mov_b32 (per instr)", 1);
    {
        hipLaunchKernelGGL(k_seed_accuracy, dim3(1), dim3(64), 0, 0, out);
        double h[64 * 6]; CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        double m[6] = {0};
        for (int t = 0; t < 64; ++t) for (int k = 0; k < 6; ++k) if (h[t * 6 + k] > m[k]) m[k] = h[t * 6 + k];
        printf("max rel. error  v_rsq_f64 seed %.3g, +1 Newton %.3g, +2 Newton %.3g;  v_rcp_f64 seed %.3g, +1 %.3g, +2 %.3g\n",
               m[0], m[1], m[2], m[3], m[4], m[5]);
    }
    RUN(k_lds, "ds_write_b64 -> ds_read_b64 + v_add_f64", 1);
    RUN(k_ldsread, "ds_read_b64 pointer chase (+cvt)", 1);
    RUN(k_branch, "readfirstlane + s_cmp + s_cbranch + v_add_f64 loop", 1);
    return 0;
}
