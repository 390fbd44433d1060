// Store patterns of the tree regressor kernel (K1'): every wave writes, for each of its 64-sample tiles, NB row blocks x NL
// link segments of 64 rows x 128 bytes.  Pattern 0: row-major W (row stride ldw * 8 bytes: a store instruction covers eight
// 128-byte lines 4 KB apart).  Pattern 1: 16-row blocked layout (block = 16 rows x 16 columns = 2 KB contiguous: a store
// instruction covers 1 KB contiguous).  Pattern 2: whole 8 KB tile contiguous (upper bound).  Pattern 5 (round 5): LANE-OWNED
// lines -- lane i writes the whole 128-byte line of ITS row with eight consecutive 16-byte stores (each store instruction
// touches 64 different lines, 16 bytes each; the eight pieces of a line meet in L2): what a sample-per-lane kernel that stages
// a line in registers would issue.
//   hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern && ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int PAT, int NT>
__global__ __launch_bounds__(64) void store_kernel(double *W, long N, int NB, int NL, long ntiles) {
    const int lane = threadIdx.x;
    const int rg = lane >> 3, ch = lane & 7;  // 8 rows per instruction, 8 x 16 bytes per row
    const long ldw = 16L * NL;
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        for (int j = 0; j < NB; ++j) {
            const long rowbase = (long)j * N + 64 * t;
            for (int l = 0; l < NL; ++l) {
                u32x4 d = {(unsigned)l, (unsigned)j, (unsigned)lane, 0u};
                if (PAT == 5) {
                    double *p = W + (rowbase + lane) * ldw + 16 * l;
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        if (NT) __builtin_nontemporal_store(d, reinterpret_cast<u32x4 *>(p + 2 * it));
                        else *reinterpret_cast<u32x4 *>(p + 2 * it) = d;
                    }
                    continue;
                }
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const long r = rowbase + 8 * it + rg;
                    double *p;
                    if (PAT == 0) p = W + r * ldw + 16 * l + 2 * ch;
                    else if (PAT == 1) p = W + (r >> 4) * (16 * ldw) + (long)l * 256 + (r & 15) * 16 + 2 * ch;
                    else p = W + ((long)j * NL + l) * (N * 16) + (64 * t + 8 * it + rg) * 16 + 2 * ch;
                    if (NT) __builtin_nontemporal_store(d, reinterpret_cast<u32x4 *>(p));
                    else *reinterpret_cast<u32x4 *>(p) = d;
                }
            }
        }
    }
}

int main(int argc, char **argv) {
    const long N = argc > 1 ? atol(argv[1]) : 2000000;
    const int NB = 6, NL = argc > 2 ? atoi(argv[2]) : 33;
    const long ntiles = N / 64;
    const size_t bytes = (size_t)NB * N * NL * 128;
    double *W;
    if (hipMalloc(&W, bytes) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int waves : {2048, 2816}) {
        for (int pat = 0; pat < 7; ++pat) {  // 3, 4: patterns 0, 1 with ordinary (cached) stores; 5, 6: lane-owned lines (nt / cached)
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                if (pat == 0) hipLaunchKernelGGL((store_kernel<0, 1>), dim3(waves), dim3(64), 0, 0, W, N, NB, NL, ntiles);
                if (pat == 1) hipLaunchKernelGGL((store_kernel<1, 1>), dim3(waves), dim3(64), 0, 0, W, N, NB, NL, ntiles);
                if (pat == 2) hipLaunchKernelGGL((store_kernel<2, 1>), dim3(waves), dim3(64), 0, 0, W, N, NB, NL, ntiles);
                if (pat == 3) hipLaunchKernelGGL((store_kernel<0, 0>), dim3(waves), dim3(64), 0, 0, W, N, NB, NL, ntiles);
                if (pat == 4) hipLaunchKernelGGL((store_kernel<1, 0>), dim3(waves), dim3(64), 0, 0, W, N, NB, NL, ntiles);
                if (pat == 5) hipLaunchKernelGGL((store_kernel<5, 1>), dim3(waves), dim3(64), 0, 0, W, N, NB, NL, ntiles);
                if (pat == 6) hipLaunchKernelGGL((store_kernel<5, 0>), dim3(waves), dim3(64), 0, 0, W, N, NB, NL, ntiles);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("waves %d pattern %d: %.3f ms = %.2f TB/s (%.1f GB)\n", waves, pat, best, bytes / best / 1e9, bytes / 1e9);
        }
    }
    return 0;
}
