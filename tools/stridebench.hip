// Memory skeleton of the transform's pass kernel, without any arithmetic: workgroups move tiles of R rows x C
// elements between the layouts the passes use (pieces of C elements at a row stride, or C*R contiguous elements),
// in the 32-byte packed and the 36-byte unpacked element formats, in the pass kernel's XCD-grouped work order.
// Shows what the access pattern alone allows on gfx950, as a function of the piece width C.
//
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -o stridebench stridebench.hip
// Usage:  stridebench            (runs the built-in list of cases at 64 x 2^20 elements)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

struct P {
    uint32_t log_c, log_r;      // piece width and rows (log2)
    uint32_t in_w, out_w;       // dwords per element (8 or 9)
    uint32_t in_strided, out_strided;
    uint64_t row_stride;        // elements between rows of a strided layout
    uint64_t poly_stride;       // elements per polynomial
    uint32_t tiles, polys, grp_log;
    uint32_t tile_first;        // grouped order: sweep the tiles of a polynomial before the next polynomial
    uint32_t log_c1, log_r1;    // in_strided == 2: the producer's tile shape (its tiles are stored back to back)
};

struct __attribute__((aligned(4))) U3 {
    uint32_t x, y, z;
};

template <int E, int IN_W, int OUT_W>
__global__ void __launch_bounds__(1024) tilecopy(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, const P p) {
    extern __shared__ uint32_t lds[];
    uint32_t tau, pidx;
    if (p.grp_log) {
        const uint32_t g = p.grp_log - 1, xcd = blockIdx.x & 7u, s = blockIdx.x >> 3, s2 = s >> g;
        if (p.tile_first) {
            const uint32_t per = p.tiles >> (3 + g);  // groups of one XCD per polynomial
            pidx = s2 / per;
            tau = ((((s2 % per) << 3) | xcd) << g) | (s & ((1u << g) - 1u));
        } else {
            pidx = s2 % p.polys;
            tau = ((((s2 / p.polys) << 3) | xcd) << g) | (s & ((1u << g) - 1u));
        }
    } else {
        tau = blockIdx.x % p.tiles;
        pidx = blockIdx.x / p.tiles;
    }
    const uint32_t C = 1u << p.log_c, R = 1u << p.log_r;
    uint32_t v[E][9];
    // element q of the tile: strided layouts take the piece index fastest, contiguous ones the row index
#pragma unroll
    for (int k = 0; k < E; k++) {
        const uint32_t q = threadIdx.x + k * blockDim.x;
        uint64_t idx;
        if (p.in_strided == 2) {
            // rows tau*C .. tau*C+C-1 of a matrix whose columns were written tile by tile: [t][row][c1]
            const uint32_t c = q & (C - 1), j = q >> p.log_c;
            idx = ((uint64_t)(j >> p.log_c1) << (p.log_r1 + p.log_c1)) + (((uint64_t)tau * C + c) << p.log_c1) +
                  (j & ((1u << p.log_c1) - 1u));
        } else if (p.in_strided) idx = (uint64_t)(q >> p.log_c) * p.row_stride + (uint64_t)tau * C + (q & (C - 1));
        else idx = (uint64_t)tau * C * R + q;
        const uint32_t* src = in + IN_W * ((uint64_t)pidx * p.poly_stride + idx);
        if (IN_W == 8) {
            const uint4 a = ((const uint4*)src)[0], b = ((const uint4*)src)[1];
            v[k][0] = a.x; v[k][1] = a.y; v[k][2] = a.z; v[k][3] = a.w;
            v[k][4] = b.x; v[k][5] = b.y; v[k][6] = b.z; v[k][7] = b.w;
            v[k][8] = 0;
        } else {
            const U3 a = ((const U3*)src)[0], b = ((const U3*)src)[1], c = ((const U3*)src)[2];
            v[k][0] = a.x; v[k][1] = a.y; v[k][2] = a.z;
            v[k][3] = b.x; v[k][4] = b.y; v[k][5] = b.z;
            v[k][6] = c.x; v[k][7] = c.y; v[k][8] = c.z;
        }
    }
    if (v[0][0] == 0xDEADBEEFu && v[E - 1][7] == 0x12345u) lds[threadIdx.x] = 1;  // keeps the LDS allocation alive
#pragma unroll
    for (int k = 0; k < E; k++) {
        const uint32_t q = threadIdx.x + k * blockDim.x;
        uint64_t idx;
        if (p.out_strided) idx = (uint64_t)(q >> p.log_c) * p.row_stride + (uint64_t)tau * C + (q & (C - 1));
        else idx = (uint64_t)tau * C * R + q;
        uint32_t* dst = out + OUT_W * ((uint64_t)pidx * p.poly_stride + idx);
        if (OUT_W == 8) {
            ((uint4*)dst)[0] = make_uint4(v[k][0], v[k][1], v[k][2], v[k][3]);
            ((uint4*)dst)[1] = make_uint4(v[k][4], v[k][5], v[k][6], v[k][7]);
        } else {
            ((U3*)dst)[0] = U3{v[k][0], v[k][1], v[k][2]};
            ((U3*)dst)[1] = U3{v[k][3], v[k][4], v[k][5]};
            ((U3*)dst)[2] = U3{v[k][6], v[k][7], v[k][8]};
        }
    }
}

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 nt_load(const uint4* p) {
    const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store(const uint4& v, uint4* p) {
    v4u w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<v4u*>(p));
}

// calibration: plain copies of the same 2 GiB, lane-contiguous 16-byte chunks
template <int NT>
__global__ void __launch_bounds__(256) copy_gridstride(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (NT) {
            const uint4 v = nt_load(in + i);
            nt_store(v, out + i);
        } else {
            out[i] = in[i];
        }
    }
}

__global__ void __launch_bounds__(256) read_only(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = in[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x1234567u) out[0] = acc;
}

__global__ void __launch_bounds__(256) write_only(uint4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = make_uint4((uint32_t)i, 1, 2, 3);
}

// every thread moves K chunks, loads first; MODE 0: chunk = tid + k * threads (coalesced), 1: chunks 2*tid, 2*tid+1
// of each half (the 32-byte-per-lane shape of the pass kernel's element loads)
template <int K, int MODE, int NT>
__global__ void __launch_bounds__(512) copy_oneshot(const uint4* __restrict__ in, uint4* __restrict__ out) {
    const size_t base = (size_t)blockIdx.x * blockDim.x * K;
    uint4 v[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const size_t i = MODE == 0 ? base + threadIdx.x + (size_t)k * blockDim.x
                                   : base + 2 * (threadIdx.x + (size_t)(k / 2) * blockDim.x) + (k & 1);
        v[k] = NT ? nt_load(in + i) : in[i];
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        const size_t i = MODE == 0 ? base + threadIdx.x + (size_t)k * blockDim.x
                                   : base + 2 * (threadIdx.x + (size_t)(k / 2) * blockDim.x) + (k & 1);
        if (NT) nt_store(v[k], out + i);
        else out[i] = v[k];
    }
}

template <typename F>
static void timed(const char* name, double bytes, F launch) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipDeviceSynchronize());
    const int reps = 5;
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-60s %7.3f ms  %5.2f TB/s\n", name, ms, bytes / (ms * 1e-3) * 1e-12);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

static void calibrate(uint32_t* d_in, uint32_t* d_out) {
    const size_t n = (size_t)2 << 30 >> 4;  // 2 GiB of 16-byte chunks
    const double bytes = 2.0 * 16.0 * (double)n;
    const uint4* in = (const uint4*)d_in;
    uint4* out = (uint4*)d_out;
    for (int blocks : {2048, 8192, 32768})
        for (int nt = 0; nt < 2; nt++) {
            char nm[96];
            snprintf(nm, sizeof nm, "grid-stride uint4 copy, %d blocks x 256%s", blocks, nt ? ", nontemporal" : "");
            if (nt) timed(nm, bytes, [&] { hipLaunchKernelGGL(copy_gridstride<1>, dim3(blocks), dim3(256), 0, 0, in, out, n); });
            else timed(nm, bytes, [&] { hipLaunchKernelGGL(copy_gridstride<0>, dim3(blocks), dim3(256), 0, 0, in, out, n); });
        }
    const unsigned b8 = (unsigned)(n / (512 * 8));
    timed("one-shot, 8 chunks per lane, coalesced", bytes, [&] { hipLaunchKernelGGL((copy_oneshot<8, 0, 0>), dim3(b8), dim3(512), 0, 0, in, out); });
    timed("one-shot, 8 chunks per lane, coalesced, nontemporal", bytes, [&] { hipLaunchKernelGGL((copy_oneshot<8, 0, 1>), dim3(b8), dim3(512), 0, 0, in, out); });
    timed("one-shot, 8 chunks per lane, 32 B per lane", bytes, [&] { hipLaunchKernelGGL((copy_oneshot<8, 1, 0>), dim3(b8), dim3(512), 0, 0, in, out); });
    timed("one-shot, 8 chunks per lane, 32 B per lane, nontemporal", bytes, [&] { hipLaunchKernelGGL((copy_oneshot<8, 1, 1>), dim3(b8), dim3(512), 0, 0, in, out); });
    const unsigned b4 = (unsigned)(n / (512 * 4));
    timed("one-shot, 4 chunks per lane, coalesced", bytes, [&] { hipLaunchKernelGGL((copy_oneshot<4, 0, 0>), dim3(b4), dim3(512), 0, 0, in, out); });
    timed("one-shot, 4 chunks per lane, 32 B per lane", bytes, [&] { hipLaunchKernelGGL((copy_oneshot<4, 1, 0>), dim3(b4), dim3(512), 0, 0, in, out); });
    timed("grid-stride read only", bytes / 2, [&] { hipLaunchKernelGGL(read_only, dim3(8192), dim3(256), 0, 0, in, out, n); });
    timed("grid-stride write only", bytes / 2, [&] { hipLaunchKernelGGL(write_only, dim3(8192), dim3(256), 0, 0, out, n); });
}

static void run(const char* name, int log_c, int in_w, int out_w, int in_str, int out_str, int lds_kb, int grp,
                int threads_cap, uint32_t* d_in, uint32_t* d_out, int tile_first = 0, int log_c1 = 1) {
    const int log_n = 20, log_r = 10, polys = 64;
    P p{};
    p.log_c = log_c;
    p.log_r = log_r;
    p.in_w = in_w;
    p.out_w = out_w;
    p.in_strided = in_str;
    p.out_strided = out_str;
    p.row_stride = 1ull << (log_n - log_r);
    p.poly_stride = 1ull << log_n;
    p.tiles = (1u << (log_n - log_r)) >> log_c;
    p.polys = polys;
    p.tile_first = tile_first;
    p.log_c1 = log_c1;
    p.log_r1 = log_n - log_r;
    p.grp_log = 0;
    for (int g = grp; g >= 0; g--)
        if (p.tiles % (8u << g) == 0) {
            p.grp_log = g + 1;
            break;
        }
    const int elems = 1 << (log_r + log_c);
    int threads = elems / 4;
    if (threads > threads_cap) threads = threads_cap;
    const int E = elems / threads;
    void (*k)(const uint32_t*, uint32_t*, P) = nullptr;
    const int key = E * 100 + in_w * 10 + out_w;
    switch (key) {
        case 488: k = tilecopy<4, 8, 8>; break;
        case 489: k = tilecopy<4, 8, 9>; break;
        case 498: k = tilecopy<4, 9, 8>; break;
        case 499: k = tilecopy<4, 9, 9>; break;
        case 888: k = tilecopy<8, 8, 8>; break;
        case 889: k = tilecopy<8, 8, 9>; break;
        case 898: k = tilecopy<8, 9, 8>; break;
        default: printf("%s: unsupported E=%d\n", name, E); return;
    }
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const unsigned blocks = p.tiles * polys;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), (size_t)lds_kb * 1024, 0, d_in, d_out, p);
    CHECK(hipDeviceSynchronize());
    const int reps = 5;
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), (size_t)lds_kb * 1024, 0, d_in, d_out, p);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes = (double)polys * (1 << log_n) * 4.0 * (in_w + out_w);
    printf("%-34s C=%-2d thr=%-4d E=%-2d lds=%3dK grp=%d%s  %7.3f ms  %5.2f TB/s\n", name, 1 << log_c, threads, E, lds_kb,
           (int)p.grp_log - 1, tile_first ? "t" : " ", ms, bytes / (ms * 1e-3) * 1e-12);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

int main() {
    const size_t bytes = (size_t)64 << 20 << 2;  // 64 x 2^20 elements x 9 dwords
    uint32_t *d_in, *d_out;
    CHECK(hipMalloc(&d_in, bytes * 9));
    CHECK(hipMalloc(&d_out, bytes * 9));
    CHECK(hipMemset(d_in, 1, bytes * 9));
    CHECK(hipMemset(d_out, 0, bytes * 9));
    calibrate(d_in, d_out);
    // work orders, contiguous tiles
    for (int g : {-1, 0, 4})
        for (int tf = 0; tf < 2; tf++) {
            if (g < 0 && tf) continue;
            run("contig 32 -> contig 32", 1, 8, 8, 0, 0, 77, g, 512, d_in, d_out, tf);
            run("contig 36 -> contig 36", 1, 9, 9, 0, 0, 77, g, 512, d_in, d_out, tf);
        }
    // the passes as they are, and with the intermediate stored tile by tile
    for (int g : {-1, 2, 4, 6})
        for (int tf = 0; tf < 2; tf++) {
            if (g < 0 && tf) continue;
            run("pass 1 now: strided 32 -> strided 36", 1, 8, 9, 1, 1, 77, g, 512, d_in, d_out, tf);
            run("pass 2 now: contig 36 -> strided 32", 1, 9, 8, 0, 1, 77, g, 512, d_in, d_out, tf);
            run("pass 1 new: strided 32 -> tiles 36", 1, 8, 9, 1, 0, 77, g, 512, d_in, d_out, tf);
            run("pass 2 new: tiles 36 -> strided 32", 1, 9, 8, 2, 1, 77, g, 512, d_in, d_out, tf);
        }
    // wider pieces (one workgroup per CU)
    for (int c = 2; c <= 3; c++) {
        run("pass 1 new: strided 32 -> tiles 36", c, 8, 9, 1, 0, 154, 4, 1024, d_in, d_out, 0);
        run("pass 2 new: tiles 36 -> strided 32", c, 9, 8, 2, 1, 154, 4, 1024, d_in, d_out, 0, c);
    }
    return 0;
}
