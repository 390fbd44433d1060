// second D2H probe: (h) chunked async DMA straight into a pinned destination, one / two streams; (i) fresh destination backed by
// transparent huge pages; (j) the library's current path (one hipMemcpyAsync into fresh pageable memory); (k) populate cost.
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
    std::vector<std::thread> th;
    size_t per = ((n / nt) + 4095) & ~size_t(4095);
    for (int t = 0; t < nt; ++t) { size_t lo = t * per, hi = lo + per > n ? n : lo + per; if (lo >= n) break;
        th.emplace_back([=] { memcpy(dst + lo, src + lo, hi - lo); }); }
    for (auto &t : th) t.join();
}
static double staged(char *h, const char *d, size_t bytes, size_t chunk, int nt, hipStream_t s) {
    char *st[2]; CK(hipHostMalloc(&st[0], chunk, hipHostMallocDefault)); CK(hipHostMalloc(&st[1], chunk, hipHostMallocDefault));
    hipEvent_t ev[2]; CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    double t = now();
    size_t nch = (bytes + chunk - 1) / chunk;
    CK(hipMemcpyAsync(st[0], d, chunk < bytes ? chunk : bytes, hipMemcpyDeviceToHost, s)); CK(hipEventRecord(ev[0], s));
    for (size_t k = 0; k < nch; ++k) {
        size_t lo = k * chunk, n = lo + chunk > bytes ? bytes - lo : chunk;
        if (k + 1 < nch) { size_t lo2 = lo + chunk, n2 = lo2 + chunk > bytes ? bytes - lo2 : chunk;
            CK(hipMemcpyAsync(st[(k + 1) & 1], d + lo2, n2, hipMemcpyDeviceToHost, s)); CK(hipEventRecord(ev[(k + 1) & 1], s)); }
        CK(hipEventSynchronize(ev[k & 1]));
        par_copy(h + lo, st[k & 1], n, nt);
    }
    t = now() - t;
    CK(hipHostFree(st[0])); CK(hipHostFree(st[1]));
    return t;
}
int main(int argc, char **argv) {
    size_t bytes = (argc > 1 ? atof(argv[1]) : 4.032) * 1e9;
    bytes &= ~size_t((2 << 20) - 1);
    char *d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 1, bytes));
    hipStream_t s, s2; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
    { FILE *f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r"); char buf[128] = "?"; if (f) { fgets(buf, 128, f); fclose(f); } printf("THP enabled: %s", buf); }
    // (j) current library path
    { char *h = (char *)malloc(bytes); double t = now(); CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); t = now() - t;
      printf("j one hipMemcpyAsync into fresh pageable          %.1f ms  %.1f GB/s\n", 1e3 * t, bytes / t / 1e9);
      t = now(); CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); t = now() - t;
      printf("j again (touched)                                 %.1f ms  %.1f GB/s\n", 1e3 * t, bytes / t / 1e9); free(h); }
    // (h) pinned destination, chunked async
    { char *h; CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
      for (size_t chunk : {size_t(16) << 20, size_t(64) << 20, size_t(256) << 20, bytes}) for (int ns : {1, 2}) {
        double t = now(); size_t k = 0;
        for (size_t lo = 0; lo < bytes; lo += chunk, ++k) { size_t n = lo + chunk > bytes ? bytes - lo : chunk;
            CK(hipMemcpyAsync(h + lo, d + lo, n, hipMemcpyDeviceToHost, (ns == 2 && (k & 1)) ? s2 : s)); }
        CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2)); t = now() - t;
        printf("h pinned dst, async chunks of %4zu MB, %d stream(s)   %.1f ms  %.1f GB/s\n", chunk >> 20, ns, 1e3 * t, bytes / t / 1e9); }
      CK(hipHostFree(h)); }
    // (i) fresh THP-backed destination
    for (int nt : {4, 8}) {
        double t0 = now();
        char *h = (char *)mmap(nullptr, bytes + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        char *al = (char *)(((size_t)h + (2 << 20) - 1) & ~size_t((2 << 20) - 1));
        int rc = madvise(al, bytes, MADV_HUGEPAGE);
        double t = staged(al, d, bytes, size_t(64) << 20, nt, s);
        printf("i fresh mmap + MADV_HUGEPAGE (rc %d), staged 64 MB, %d threads   %.1f ms  %.1f GB/s (incl. mmap %.1f ms)\n", rc, nt, 1e3 * t, bytes / t / 1e9, 1e3 * (now() - t0));
        munmap(h, bytes + (2 << 20));
    }
    // (k) populate alone
    { double t = now(); char *h = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0); t = now() - t;
      printf("k mmap MAP_POPULATE %.1f ms (%.1f GB/s)\n", 1e3 * t, bytes / t / 1e9);
      double t2 = staged(h, d, bytes, size_t(64) << 20, 8, s); printf("k then staged 8 threads %.1f ms %.1f GB/s\n", 1e3 * t2, bytes / t2 / 1e9); munmap(h, bytes); }
    // (l) fresh malloc, 8 threads pre-touch one byte per page in parallel, then staged
    { char *h = (char *)malloc(bytes); double t = now(); { std::vector<std::thread> th; int nt = 8; size_t per = bytes / nt;
        for (int q = 0; q < nt; ++q) th.emplace_back([=] { for (size_t o = q * per; o < (q + 1) * per; o += 4096) h[o] = 0; }); for (auto &x : th) x.join(); }
      double tt = now() - t; double t2 = staged(h, d, bytes, size_t(64) << 20, 8, s);
      printf("l parallel pre-touch %.1f ms + staged %.1f ms = %.1f GB/s\n", 1e3 * tt, 1e3 * t2, bytes / (tt + t2) / 1e9); free(h); }
    return 0;
}
