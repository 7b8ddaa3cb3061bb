// Register-only inner-loop benchmark for gfx950: how many shader cycles one wavefront-wide
// field multiplication / NTT butterfly / BLAKE2s compression costs when every SIMD is full,
// with no memory traffic in the loop. Used to separate "instruction issue" from "memory" when
// reading a kernel's rocprofv3 numbers, and to A/B candidate formulations of the arithmetic.
//
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -o kernbench kernbench.hip
// Usage:  kernbench [waves_per_simd]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "blake2s.cuh"
#include "field.cuh"
#include "field29.cuh"

using namespace starkhip;

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

constexpr int ITERS = 512;

struct Stamp {
    unsigned long long cyc, rt;
};

__device__ __forceinline__ void stamp_begin(unsigned long long& c, unsigned long long& r) {
    c = __builtin_amdgcn_s_memtime();
    r = __builtin_amdgcn_s_memrealtime();
}

__device__ __forceinline__ void stamp_end(Stamp* out, unsigned long long c0, unsigned long long r0) {
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        out[w].cyc = c1 - c0;
        out[w].rt = r1 - r0;
    }
}

__device__ __forceinline__ Fe seed_fe(uint32_t s) {
    Fe x;
    for (int i = 0; i < 8; i++) x.l[i] = s * (2654435761u + 40503u * i) + i;
    x.l[7] &= 0x7FFFFFFFu;
    return x;
}

__global__ void __launch_bounds__(256) k_femul(Stamp* st, uint32_t* sink, uint32_t s) {
    Fe a = seed_fe(s + threadIdx.x), b = seed_fe(s * 3 + blockIdx.x);
    unsigned long long c0, r0;
    stamp_begin(c0, r0);
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) a = fe_mul(a, b);
    stamp_end(st, c0, r0);
    uint32_t acc = 0;
    for (int i = 0; i < 8; i++) acc ^= a.l[i];
    if (acc == 0x1234567) sink[0] = acc;
}

__global__ void __launch_bounds__(256) k_butterfly(Stamp* st, uint32_t* sink, uint32_t s) {
    Fe x = seed_fe(s + threadIdx.x), y = seed_fe(s * 5 + threadIdx.x), w = seed_fe(s * 3 + blockIdx.x);
    x = fe_canon(x); y = fe_canon(y); w = fe_canon(w);
    unsigned long long c0, r0;
    stamp_begin(c0, r0);
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
        Fe t = fe_mul(y, w);
        Fe u = fe_add(x, t);
        y = fe_sub(x, t);
        x = u;
    }
    stamp_end(st, c0, r0);
    uint32_t acc = 0;
    for (int i = 0; i < 8; i++) acc ^= x.l[i] ^ y.l[i];
    if (acc == 0x1234567) sink[0] = acc;
}

__global__ void __launch_bounds__(256) k_femul29(Stamp* st, uint32_t* sink, uint32_t s) {
    F29 a = f29_from_fe(fe_canon(seed_fe(s + threadIdx.x))), b = f29_from_fe(fe_canon(seed_fe(s * 3 + blockIdx.x)));
    unsigned long long c0, r0;
    stamp_begin(c0, r0);
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) a = f29_mul(a, b);
    stamp_end(st, c0, r0);
    uint32_t acc = 0;
    for (int i = 0; i < 9; i++) acc ^= a.l[i];
    if (acc == 0x1234567) sink[0] = acc;
}

// per-lane operands of the split product: y = b, y2 = b * 2^145 mod p (benchmark only, see field29.cuh)
#define F29_NEXT(dst)                       \
    dst = (uint32_t)acc & F29_MASK;         \
    acc >>= 29;
__device__ __forceinline__ F29 f29_mul_split(const F29& x, const F29& y, const F29& y2) {
    const uint32_t* b = y.l;
    const uint32_t* c = y2.l;
    F29_MUL_SPLIT_BODY(F29_P)
}
#undef F29_NEXT

// the split product with per-lane second operands (b, b * 2^145): 85 multiply-adds
__global__ void __launch_bounds__(256) k_femul29_split(Stamp* st, uint32_t* sink, uint32_t s) {
    F29 a = f29_from_fe(fe_canon(seed_fe(s + threadIdx.x))), b = f29_from_fe(fe_canon(seed_fe(s * 3 + blockIdx.x)));
    F29 c = f29_from_fe(fe_canon(seed_fe(s * 7 + blockIdx.x)));
    unsigned long long c0, r0;
    stamp_begin(c0, r0);
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) a = f29_mul_split(a, b, c);
    stamp_end(st, c0, r0);
    uint32_t acc = 0;
    for (int i = 0; i < 9; i++) acc ^= a.l[i];
    if (acc == 0x1234567) sink[0] = acc;
}

// the same with the second operands in scalar registers (a wave-uniform constant)
__global__ void __launch_bounds__(256) k_femul29_uniform(Stamp* st, uint32_t* sink, uint32_t s) {
    F29 a = f29_from_fe(fe_canon(seed_fe(s + threadIdx.x)));
    uint32_t b[9], c[9];
    for (int i = 0; i < 9; i++) {
        b[i] = __builtin_amdgcn_readfirstlane((s * (i + 3) + blockIdx.x * 77u) & F29_MASK);
        c[i] = __builtin_amdgcn_readfirstlane((s * (i + 11) + blockIdx.x * 131u) & F29_MASK);
    }
    b[8] &= 0xFFFFFFu;
    c[8] &= 0xFFFFFFu;
    unsigned long long c0, r0;
    stamp_begin(c0, r0);
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) a = f29_mul_uniform2(a, b, c);
    stamp_end(st, c0, r0);
    uint32_t acc = 0;
    for (int i = 0; i < 9; i++) acc ^= a.l[i];
    if (acc == 0x1234567) sink[0] = acc;
}

__global__ void __launch_bounds__(256) k_butterfly29(Stamp* st, uint32_t* sink, uint32_t s) {
    F29 x = f29_from_fe(fe_canon(seed_fe(s + threadIdx.x))), y = f29_from_fe(fe_canon(seed_fe(s * 5 + threadIdx.x)));
    F29 w = f29_from_fe(fe_canon(seed_fe(s * 3 + blockIdx.x)));
    unsigned long long c0, r0;
    stamp_begin(c0, r0);
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
        F29 t = f29_mul(y, w);
        F29 u = f29_add(x, t);
        y = f29_sub(x, t);
        x = f29_carry(u);
        y = f29_carry(y);
    }
    stamp_end(st, c0, r0);
    uint32_t acc = 0;
    for (int i = 0; i < 9; i++) acc ^= x.l[i] ^ y.l[i];
    if (acc == 0x1234567) sink[0] = acc;
}

__global__ void __launch_bounds__(256) k_blake2s(Stamp* st, uint32_t* sink, uint32_t s) {
    uint32_t h[8], m[16];
    for (int i = 0; i < 16; i++) m[i] = s * (i + 7) + threadIdx.x * 2654435761u;
    unsigned long long c0, r0;
    stamp_begin(c0, r0);
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
        b2s_hash64(h, m);
        for (int i = 0; i < 8; i++) m[i] = h[i];
    }
    stamp_end(st, c0, r0);
    uint32_t acc = 0;
    for (int i = 0; i < 8; i++) acc ^= h[i];
    if (acc == 0x1234567) sink[0] = acc;
}

typedef void (*kern_t)(Stamp*, uint32_t*, uint32_t);

static void run(const char* name, kern_t k, int blocks, Stamp* d_st, Stamp* h_st, uint32_t* d_sink) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_st, d_sink, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 3;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_st, d_sink, (uint32_t)r + 2);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    int waves = blocks * 4;
    CHECK(hipMemcpy(h_st, d_st, sizeof(Stamp) * waves, hipMemcpyDeviceToHost));
    double cyc = 0, rt = 0;
    for (int i = 0; i < waves; i++) {
        cyc += (double)h_st[i].cyc;
        rt += (double)h_st[i].rt;
    }
    cyc /= waves;
    rt /= waves;
    double ghz = cyc / (rt * 10.0);  // realtime ticks are 10 ns
    double ops = (double)waves * 64 * ITERS * reps;
    printf("%-14s wave-latency %8.1f cyc/op  clock %.2f GHz  chip rate %8.2f Gop/s  (%.1f us/launch)\n", name,
           cyc / ITERS, ghz, ops / (ms * 1e-3) * 1e-9, ms * 1e3 / reps);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
    int wps = argc > 1 ? atoi(argv[1]) : 4;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int blocks = prop.multiProcessorCount * wps;
    Stamp *d_st, *h_st = (Stamp*)malloc(sizeof(Stamp) * blocks * 4);
    uint32_t* d_sink;
    CHECK(hipMalloc(&d_st, sizeof(Stamp) * blocks * 4));
    CHECK(hipMalloc(&d_sink, 64));
    printf("waves/SIMD requested %d (256-thread blocks, %d blocks)\n", wps, blocks);
    run("fe_mul", k_femul, blocks, d_st, h_st, d_sink);
    run("butterfly", k_butterfly, blocks, d_st, h_st, d_sink);
    run("f29_mul", k_femul29, blocks, d_st, h_st, d_sink);
    run("f29_mul_split", k_femul29_split, blocks, d_st, h_st, d_sink);
    run("f29_mul_unif2", k_femul29_uniform, blocks, d_st, h_st, d_sink);
    run("butterfly29", k_butterfly29, blocks, d_st, h_st, d_sink);
    run("blake2s", k_blake2s, blocks, d_st, h_st, d_sink);
    return 0;
}
