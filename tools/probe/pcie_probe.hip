// Host <-> device copy rates: pageable, pinned (hipHostMalloc), and pageable staged through pinned buffers by host threads.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static void par_copy(char* dst, const char* src, size_t n, int threads) {
    std::vector<std::thread> t;
    size_t per = (n + threads - 1) / threads;
    for (int i = 0; i < threads; i++) {
        size_t lo = per * i, hi = lo + per < n ? lo + per : n;
        if (lo < hi) t.emplace_back([=] { memcpy(dst + lo, src + lo, hi - lo); });
    }
    for (auto& x : t) x.join();
}
int main() {
    const size_t N = (size_t)1 << 30;
    char* pageable = (char*)malloc(N);
    memset(pageable, 1, N);
    char* d;
    CK(hipMalloc((void**)&d, N));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now();
        CK(hipMemcpyAsync(d, pageable, N, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        double t1 = now();
        CK(hipMemcpyAsync(pageable, d, N, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        double t2 = now();
        printf("pageable: H2D %.1f GB/s, D2H %.1f GB/s\n", N / (t1 - t0) * 1e-9, N / (t2 - t1) * 1e-9);
    }
    char* pinned;
    hipError_t e = hipHostMalloc((void**)&pinned, N, hipHostMallocDefault);
    printf("hipHostMalloc(1 GiB): %s\n", hipGetErrorString(e));
    if (e == hipSuccess) {
        memset(pinned, 2, N);
        for (int rep = 0; rep < 2; rep++) {
            double t0 = now();
            CK(hipMemcpyAsync(d, pinned, N, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            double t1 = now();
            CK(hipMemcpyAsync(pinned, d, N, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            double t2 = now();
            printf("pinned:   H2D %.1f GB/s, D2H %.1f GB/s\n", N / (t1 - t0) * 1e-9, N / (t2 - t1) * 1e-9);
        }
        for (int th : {1, 4, 8, 16}) {
            double t0 = now();
            par_copy(pinned, pageable, N, th);
            double t1 = now();
            par_copy(pageable, pinned, N, th);
            double t2 = now();
            printf("host memcpy %2d threads: to pinned %.1f GB/s, from pinned %.1f GB/s\n", th, N / (t1 - t0) * 1e-9, N / (t2 - t1) * 1e-9);
        }
        // staged pipeline: 2 pinned halves of 64 MiB each
        const size_t C = (size_t)64 << 20;
        hipEvent_t ev[2];
        CK(hipEventCreate(&ev[0])); CK(hipEventCreate(&ev[1]));
        for (int th : {4, 8, 16}) {
            double t0 = now();
            for (size_t o = 0, k = 0; o < N; o += C, k++) {
                char* buf = pinned + (k & 1) * C;
                if (k >= 2) CK(hipEventSynchronize(ev[k & 1]));
                par_copy(buf, pageable + o, C, th);
                CK(hipMemcpyAsync(d + o, buf, C, hipMemcpyHostToDevice, st));
                CK(hipEventRecord(ev[k & 1], st));
            }
            CK(hipStreamSynchronize(st));
            double t1 = now();
            // D2H staged
            size_t k = 0;
            for (size_t o = 0; o < N + C; o += C, k++) {
                if (o < N) { CK(hipMemcpyAsync(pinned + (k & 1) * C, d + o, C, hipMemcpyDeviceToHost, st)); CK(hipEventRecord(ev[k & 1], st)); }
                if (k >= 1) { CK(hipEventSynchronize(ev[(k - 1) & 1])); par_copy(pageable + o - C, pinned + ((k - 1) & 1) * C, C, th); }
            }
            double t2 = now();
            printf("staged %2d threads: H2D %.1f GB/s, D2H %.1f GB/s\n", th, N / (t1 - t0) * 1e-9, N / (t2 - t1) * 1e-9);
        }
    }
    hipError_t r = hipHostRegister(pageable, N, hipHostRegisterDefault);
    printf("hipHostRegister(1 GiB malloc): %s\n", hipGetErrorString(r));
    if (r == hipSuccess) {
        double t0 = now();
        CK(hipMemcpyAsync(d, pageable, N, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        double t1 = now();
        CK(hipMemcpyAsync(pageable, d, N, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        double t2 = now();
        printf("registered: H2D %.1f GB/s, D2H %.1f GB/s\n", N / (t1 - t0) * 1e-9, N / (t2 - t1) * 1e-9);
    }
    return 0;
}
