// Instruction-rate microbenchmark for gfx950 (MI355X).
//
// Measures the sustained chip-wide rate of the VALU instructions the 256-bit field arithmetic
// and the BLAKE2s compression are built from, plus an HBM copy, so that roofline fractions can
// be quoted against peaks measured on the box rather than datasheet figures.
//
// Build:  hipcc --offload-arch=gfx950 -O3 -o microbench microbench.hip
// Output: one JSON object on stdout, rates in G lane-ops/s (one "op" = one instruction executed
//         for one lane) and GB/s.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

constexpr int ITERS = 2048;      // loop trips per thread
constexpr int PER_ITER = 32;     // instructions of the measured kind per trip

// Eight independent 32-bit chains (a0..a7) and eight 64-bit chains keep every instruction's
// inputs ready, so the measured rate is issue-bound rather than latency-bound.
#define REP4(X, a, b, c, d) X(a) X(b) X(c) X(d)
#define BODY32(OP) \
    REP4(OP, r[0], r[1], r[2], r[3]) REP4(OP, r[4], r[5], r[6], r[7]) \
    REP4(OP, r[0], r[1], r[2], r[3]) REP4(OP, r[4], r[5], r[6], r[7]) \
    REP4(OP, r[0], r[1], r[2], r[3]) REP4(OP, r[4], r[5], r[6], r[7]) \
    REP4(OP, r[0], r[1], r[2], r[3]) REP4(OP, r[4], r[5], r[6], r[7])
#define BODY64(OP) \
    REP4(OP, q[0], q[1], q[2], q[3]) REP4(OP, q[4], q[5], q[6], q[7]) \
    REP4(OP, q[0], q[1], q[2], q[3]) REP4(OP, q[4], q[5], q[6], q[7]) \
    REP4(OP, q[0], q[1], q[2], q[3]) REP4(OP, q[4], q[5], q[6], q[7]) \
    REP4(OP, q[0], q[1], q[2], q[3]) REP4(OP, q[4], q[5], q[6], q[7])

#define KERNEL32(NAME, OPMACRO)                                             \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t s) { \
        uint32_t r[8];                                                      \
        uint32_t x = s ^ threadIdx.x, y = s * 2654435761u + blockIdx.x;     \
        for (int i = 0; i < 8; i++) r[i] = x * (i + 3) + y;                 \
        for (int it = 0; it < ITERS; it++) { BODY32(OPMACRO) }              \
        uint32_t acc = 0;                                                   \
        for (int i = 0; i < 8; i++) acc ^= r[i];                            \
        if (acc == 0x12345) out[0] = acc;                                   \
    }
#define KERNEL64(NAME, OPMACRO)                                             \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t s) { \
        uint64_t q[8];                                                      \
        uint32_t x = s ^ threadIdx.x, y = s * 2654435761u + blockIdx.x;     \
        for (int i = 0; i < 8; i++) q[i] = ((uint64_t)(x * (i + 3)) << 32) | (y + i); \
        for (int it = 0; it < ITERS; it++) { BODY64(OPMACRO) }              \
        uint64_t acc = 0;                                                   \
        for (int i = 0; i < 8; i++) acc ^= q[i];                            \
        if (acc == 0x12345) out[0] = (uint32_t)acc;                         \
    }

#define OP_ADD(a)      asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(x));
#define OP_ADD3(a)     asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
#define OP_XOR(a)      asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(x));
#define OP_XOR3(a)     asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a) : "v"(x), "v"(y));
#define OP_ALIGNBIT(a) asm volatile("v_alignbit_b32 %0, %0, %0, 12" : "+v"(a));
#define OP_PERM(a)     asm volatile("v_perm_b32 %0, %0, %0, %1" : "+v"(a) : "v"(y));
#define OP_CNDMASK(a)  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(x));
#define OP_ADDCO(a)    asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a) : "v"(x) : "vcc");
#define OP_ADDC(a)     asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a) : "v"(x) : "vcc");
#define OP_MULLO(a)    asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(x));
#define OP_MULHI(a)    asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "v"(x));
#define OP_MUL24(a)    asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(x));
#define OP_MULHI24(a)  asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a) : "v"(x));
#define OP_MAD24(a)    asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
#define OP_DOT4(a)     asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
#define OP_DOT2(a)     asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
#define OP_MAD64(a)    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y) : "vcc");
#define OP_MAD64C(a)   asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc" : "+v"(a), "+v"(z) : "v"(x), "v"(y) : "vcc");
#define OP_LSHLADD64(a) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a) : "v"(w64));
#define OP_FMA64(a)    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(w64), "v"(v64));
#define OP_MUL64F(a)   asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(w64));
#define OP_ADD64F(a)   asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(w64));
#define OP_FMA32(a)    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));

KERNEL32(k_add, OP_ADD)
KERNEL32(k_add3, OP_ADD3)
KERNEL32(k_xor, OP_XOR)
KERNEL32(k_xor3, OP_XOR3)
KERNEL32(k_alignbit, OP_ALIGNBIT)
KERNEL32(k_perm, OP_PERM)
KERNEL32(k_cndmask, OP_CNDMASK)
KERNEL32(k_addco, OP_ADDCO)
KERNEL32(k_addc, OP_ADDC)
KERNEL32(k_mullo, OP_MULLO)
KERNEL32(k_mulhi, OP_MULHI)
KERNEL32(k_mul24, OP_MUL24)
KERNEL32(k_mulhi24, OP_MULHI24)
KERNEL32(k_mad24, OP_MAD24)
KERNEL32(k_dot4, OP_DOT4)
KERNEL32(k_dot2, OP_DOT2)
KERNEL32(k_fma32, OP_FMA32)
KERNEL64(k_mad64, OP_MAD64)

__global__ void __launch_bounds__(256) k_mad64c(uint32_t* out, uint32_t s) {
    uint64_t q[8];
    uint32_t x = s ^ threadIdx.x, y = s * 2654435761u + blockIdx.x, z = 0;
    for (int i = 0; i < 8; i++) q[i] = ((uint64_t)(x * (i + 3)) << 32) | (y + i);
    for (int it = 0; it < ITERS; it++) { BODY64(OP_MAD64C) }
    uint64_t acc = z;
    for (int i = 0; i < 8; i++) acc ^= q[i];
    if (acc == 0x12345) out[0] = (uint32_t)acc;
}

#define KERNEL64W(NAME, OPMACRO)                                            \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t s) { \
        uint64_t q[8];                                                      \
        uint32_t x = s ^ threadIdx.x, y = s * 2654435761u + blockIdx.x;     \
        uint64_t w64 = 0x3FF0000000000001ull + x, v64 = 0x3FE0000000000003ull + y; \
        for (int i = 0; i < 8; i++) q[i] = 0x3FF0000000000000ull + ((uint64_t)(x * (i + 3)) << 8) + y; \
        for (int it = 0; it < ITERS; it++) { BODY64(OPMACRO) }              \
        uint64_t acc = 0;                                                   \
        for (int i = 0; i < 8; i++) acc ^= q[i];                            \
        if (acc == 0x12345) out[0] = (uint32_t)acc;                         \
    }
KERNEL64W(k_lshladd64, OP_LSHLADD64)
KERNEL64W(k_fma64, OP_FMA64)
KERNEL64W(k_mul64f, OP_MUL64F)
KERNEL64W(k_add64f, OP_ADD64F)

__global__ void __launch_bounds__(256) k_copy(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

typedef void (*kern_t)(uint32_t*, uint32_t);

static double time_kernel(kern_t k, int blocks, uint32_t* d_out, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, (uint32_t)r);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return ms * 1e-3 / reps;
}

int main(int argc, char** argv) {
    int waves_per_simd = argc > 1 ? atoi(argv[1]) : 8;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one wave per SIMD per block
    uint32_t* d_out;
    CHECK(hipMalloc(&d_out, 4096));
    struct Entry { const char* name; kern_t k; int insts_per_op; };
    std::vector<Entry> es = {
        {"v_add_u32", k_add, 1}, {"v_add3_u32", k_add3, 1}, {"v_xor_b32", k_xor, 1},
        {"v_bitop3_b32(xor3)", k_xor3, 1}, {"v_alignbit_b32", k_alignbit, 1}, {"v_perm_b32", k_perm, 1},
        {"v_cndmask_b32", k_cndmask, 1}, {"v_add_co_u32", k_addco, 1}, {"v_addc_co_u32", k_addc, 1},
        {"v_mul_lo_u32", k_mullo, 1}, {"v_mul_hi_u32", k_mulhi, 1}, {"v_mul_u32_u24", k_mul24, 1},
        {"v_mul_hi_u32_u24", k_mulhi24, 1}, {"v_mad_u32_u24", k_mad24, 1}, {"v_dot4_u32_u8", k_dot4, 1},
        {"v_dot2_u32_u16", k_dot2, 1}, {"v_fma_f32", k_fma32, 1}, {"v_mad_u64_u32", k_mad64, 1},
        {"v_mad_u64_u32+v_addc_co_u32", k_mad64c, 1}, {"v_lshl_add_u64", k_lshladd64, 1},
        {"v_fma_f64", k_fma64, 1}, {"v_mul_f64", k_mul64f, 1}, {"v_add_f64", k_add64f, 1}};
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"waves_per_simd\": %d, \"rates_Gops\": {",
           prop.gcnArchName, cus, prop.clockRate / 1000, waves_per_simd);
    bool first = true;
    for (auto& e : es) {
        double t = time_kernel(e.k, blocks, d_out, 5);
        double ops = (double)blocks * 256.0 * ITERS * PER_ITER;
        printf("%s\"%s\": %.1f", first ? "" : ", ", e.name, ops / t * 1e-9);
        first = false;
        fflush(stdout);
    }
    printf("}");
    // HBM copy: 1 GiB in, 1 GiB out
    size_t bytes = (size_t)1 << 30;
    uint4 *a, *b;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&b, bytes));
    CHECK(hipMemset(a, 1, bytes));
    CHECK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_copy, dim3(cus * 8), dim3(256), 0, 0, a, b, bytes / 16);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_copy, dim3(cus * 8), dim3(256), 0, 0, a, b, bytes / 16);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf(", \"hbm_copy_GBps\": %.1f}\n", 2.0 * bytes * 5 / (ms * 1e-3) * 1e-9);
    return 0;
}
