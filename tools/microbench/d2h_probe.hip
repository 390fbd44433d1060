// D2H of a 4 GB matrix into memory a NumPy caller owns: which way is fastest on this box?  (VERDICT r05 "weak" 8: the drop-in
// build_regressor_basic returns W.numpy() at 24 GB/s.)   hipcc --offload-arch=gfx950 -O2 -o d2h_probe d2h_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <sys/mman.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void par_copy(char *dst, const char *src, size_t n, int nt) {
    if (nt <= 1) { memcpy(dst, src, n); return; }
    std::vector<std::thread> th;
    size_t per = ((n / nt) + 4095) & ~size_t(4095);
    for (int t = 0; t < nt; ++t) {
        size_t lo = t * per, hi = lo + per > n ? n : lo + per;
        if (lo >= n) break;
        th.emplace_back([=] { memcpy(dst + lo, src + lo, hi - lo); });
    }
    for (auto &t : th) t.join();
}

int main(int argc, char **argv) {
    size_t bytes = (argc > 1 ? atof(argv[1]) : 4.032) * 1e9;
    bytes &= ~size_t(4095);
    char *d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 1, bytes));
    hipStream_t s; CK(hipStreamCreate(&s));
    // (a) fresh pageable
    { char *h = (char *)malloc(bytes); double t = now(); CK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost)); t = now() - t;
      printf("a fresh pageable hipMemcpy          %.1f ms  %.1f GB/s\n", 1e3 * t, bytes / t / 1e9);
      t = now(); CK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost)); t = now() - t;
      printf("b touched pageable hipMemcpy        %.1f ms  %.1f GB/s\n", 1e3 * t, bytes / t / 1e9); free(h); }
    // (c) pinned
    { char *h; double t = now(); CK(hipHostMalloc(&h, bytes, hipHostMallocDefault)); double ta = now() - t;
      t = now(); CK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost)); t = now() - t;
      printf("c hipHostMalloc %.1f ms; copy        %.1f ms  %.1f GB/s\n", 1e3 * ta, 1e3 * t, bytes / t / 1e9);
      t = now(); CK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost)); t = now() - t;
      printf("c again                              %.1f ms  %.1f GB/s\n", 1e3 * t, bytes / t / 1e9);
      t = now(); CK(hipHostFree(h)); printf("c hipHostFree %.1f ms\n", 1e3 * (now() - t)); }
    // (f) register fresh pageable
    { char *h = (char *)malloc(bytes); double t = now(); CK(hipHostRegister(h, bytes, hipHostRegisterDefault)); double tr = now() - t;
      t = now(); CK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost)); t = now() - t;
      double tu = now(); CK(hipHostUnregister(h)); tu = now() - tu;
      printf("f hipHostRegister %.1f ms; copy %.1f ms (%.1f GB/s); unregister %.1f ms; total %.1f GB/s\n", 1e3 * tr, 1e3 * t, bytes / t / 1e9,
             1e3 * tu, bytes / (tr + t + tu) / 1e9); free(h); }
    // (d) chunked through two pinned staging buffers, nt copy threads
    for (size_t chunk : {size_t(16) << 20, size_t(64) << 20}) for (int nt : {1, 2, 4, 8, 12}) for (int fresh : {1, 0}) {
        char *st[2]; CK(hipHostMalloc(&st[0], chunk, hipHostMallocDefault)); CK(hipHostMalloc(&st[1], chunk, hipHostMallocDefault));
        hipEvent_t ev[2]; CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
        char *h = (char *)malloc(bytes);
        if (!fresh) memset(h, 0, bytes);
        double t = now();
        size_t nch = (bytes + chunk - 1) / chunk;
        CK(hipMemcpyAsync(st[0], d, chunk < bytes ? chunk : bytes, hipMemcpyDeviceToHost, s)); CK(hipEventRecord(ev[0], s));
        for (size_t k = 0; k < nch; ++k) {
            size_t lo = k * chunk, n = lo + chunk > bytes ? bytes - lo : chunk;
            if (k + 1 < nch) { size_t lo2 = lo + chunk, n2 = lo2 + chunk > bytes ? bytes - lo2 : chunk;
                CK(hipMemcpyAsync(st[(k + 1) & 1], d + lo2, n2, hipMemcpyDeviceToHost, s)); CK(hipEventRecord(ev[(k + 1) & 1], s)); }
            CK(hipEventSynchronize(ev[k & 1]));
            par_copy(h + lo, st[k & 1], n, nt);
            // (the next DMA into st[k & 1] is only queued in the next iteration, after this copy: safe)
        }
        t = now() - t;
        printf("d chunk %3zu MB threads %2d %s dst   %.1f ms  %.1f GB/s\n", chunk >> 20, nt, fresh ? "fresh  " : "touched", 1e3 * t, bytes / t / 1e9);
        free(h); CK(hipHostFree(st[0])); CK(hipHostFree(st[1]));
    }
    // (g) a persistent pool of copy threads would save thread start-up: report what that costs
    { double t = now(); for (int k = 0; k < 100; ++k) { std::thread a([] {}); a.join(); } printf("g thread create+join %.1f us\n", 1e4 * (now() - t)); }
    return 0;
}
