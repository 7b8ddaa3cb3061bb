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

#define OP_MOV(a)      asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(x));
#define OP_AND(a)      asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(x));
#define OP_OR(a)       asm volatile("v_or_b32 %0, %0, %1" : "+v"(a) : "v"(x));
#define OP_SUB(a)      asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a) : "v"(x));
#define OP_LSHL(a)     asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a));
#define OP_LSHR(a)     asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a));
#define OP_CND_S(a)    asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a) : "v"(x) : "s20", "s21");
#define OP_CMP(a)      asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a), "v"(x) : "vcc");
#define OP_CMP_S(a)    asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1" : : "v"(a), "v"(x) : "s20", "s21");
#define OP_ADDCO_S(a)  asm volatile("v_add_co_u32_e64 %0, s[20:21], %0, %1" : "+v"(a) : "v"(x) : "s20", "s21");
#define OP_LSHLOR(a)   asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a) : "v"(x));
#define OP_ANDOR(a)    asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
#define OP_OR3(a)      asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
#define OP_LSHLADD(a)  asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a) : "v"(x));
#define OP_ADDLSHL(a)  asm volatile("v_add_lshl_u32 %0, %0, %1, 3" : "+v"(a) : "v"(x));
#define OP_XAD(a)      asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
#define OP_BFE(a)      asm volatile("v_bfe_u32 %0, %0, 3, 17" : "+v"(a));
#define OP_MULF32(a)   asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(x));
#define OP_MADI24(a)   asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
#define OP_PKADD16(a)  asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a) : "v"(x));
#define OP_PKMAD16(a)  asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
#define OP_PKMUL16(a)  asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a) : "v"(x));
#define OP_SDWA_ROT16(a) asm volatile("v_mov_b32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(a));
#define OP_MADU16(a)   asm volatile("v_mad_u16 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
#define OP_MADU32U16(a) asm volatile("v_mad_u32_u16 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
#define OP_MINU(a)     asm volatile("v_min_u32 %0, %0, %1" : "+v"(a) : "v"(x));
#define OP_PKFMA32(a)  asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(w64), "v"(v64));
#define OP_CVTF64U32(a) asm volatile("v_cvt_f64_u32 %0, %1" : "+v"(a) : "v"(x));
#define OP_LSHL64(a)   asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(a));
#define OP_LSHR64(a)   asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(a));
#define OP_CMPCND(a)   asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(x) : "vcc");
#define OP_CMPCND4(a)  asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %1, %0, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %1, %0, vcc" : "+v"(a) : "v"(x) : "vcc");
#define OP_CMPCND4S(a) asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]\n\tv_cndmask_b32_e64 %0, %1, %0, s[20:21]\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]\n\tv_cndmask_b32_e64 %0, %1, %0, s[20:21]" : "+v"(a) : "v"(x) : "s20", "s21");
#define OP_SELARITH(a) asm volatile("v_xor_b32 %0, %0, %1\n\tv_and_b32 %0, %0, %2\n\tv_xor_b32 %0, %0, %1" : "+v"(a) : "v"(x), "v"(y));
#define OP_BITOP_SEL(a) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xca" : "+v"(a) : "v"(x), "v"(y));
#define OP_XORALIGN(a) asm volatile("v_xor_b32 %0, %0, %1\n\tv_alignbit_b32 %0, %0, %0, 12" : "+v"(a) : "v"(x));
#define OP_XOR_DEP(a)  asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[0]) : "v"(x));
#define OP_ALIGN_DEP(a) asm volatile("v_alignbit_b32 %0, %0, %0, 12" : "+v"(r[0]));
#define OP_XOR2(a)     asm volatile("v_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %2" : "+v"(a) : "v"(x), "v"(y));
#define OP_XORADD(a)   asm volatile("v_xor_b32 %0, %0, %1\n\tv_add_u32 %0, %0, %2" : "+v"(a) : "v"(x), "v"(y));
#define OP_MAD64_DEP(a) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[0]) : "v"(x), "v"(y) : "vcc");
#define OP_MADAND(a)   asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_and_b32 %1, %1, %2" : "+v"(a), "+v"(x) : "v"(y) : "vcc");

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
KERNEL32(k_mov, OP_MOV)
KERNEL32(k_and, OP_AND)
KERNEL32(k_or, OP_OR)
KERNEL32(k_sub, OP_SUB)
KERNEL32(k_lshl, OP_LSHL)
KERNEL32(k_lshr, OP_LSHR)
KERNEL32(k_cnd_s, OP_CND_S)
KERNEL32(k_cmp, OP_CMP)
KERNEL32(k_cmp_s, OP_CMP_S)
KERNEL32(k_addco_s, OP_ADDCO_S)
KERNEL32(k_lshlor, OP_LSHLOR)
KERNEL32(k_andor, OP_ANDOR)
KERNEL32(k_or3, OP_OR3)
KERNEL32(k_lshladd, OP_LSHLADD)
KERNEL32(k_addlshl, OP_ADDLSHL)
KERNEL32(k_xad, OP_XAD)
KERNEL32(k_bfe, OP_BFE)
KERNEL32(k_mulf32, OP_MULF32)
KERNEL32(k_madi24, OP_MADI24)
KERNEL32(k_pkadd16, OP_PKADD16)
KERNEL32(k_pkmad16, OP_PKMAD16)
KERNEL32(k_pkmul16, OP_PKMUL16)
KERNEL32(k_sdwarot, OP_SDWA_ROT16)
KERNEL32(k_madu16, OP_MADU16)
KERNEL32(k_madu32u16, OP_MADU32U16)
KERNEL32(k_minu, OP_MINU)
KERNEL32(k_cmpcnd, OP_CMPCND)
KERNEL32(k_cmpcnd4, OP_CMPCND4)
KERNEL32(k_cmpcnd4s, OP_CMPCND4S)
KERNEL32(k_selarith, OP_SELARITH)
KERNEL32(k_bitopsel, OP_BITOP_SEL)
KERNEL32(k_xoralign, OP_XORALIGN)
KERNEL32(k_xordep, OP_XOR_DEP)
KERNEL32(k_aligndep, OP_ALIGN_DEP)
KERNEL32(k_xor2, OP_XOR2)
KERNEL32(k_xoradd, OP_XORADD)
KERNEL64(k_mad64dep, OP_MAD64_DEP)
KERNEL64(k_madand, OP_MADAND)
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
KERNEL64W(k_pkfma32, OP_PKFMA32)
KERNEL64W(k_cvtf64u32, OP_CVTF64U32)
KERNEL64W(k_lshl64, OP_LSHL64)
KERNEL64W(k_lshr64, OP_LSHR64)

__global__ void __launch_bounds__(256) k_copy(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

// class-transition test: 16 v_xor_b32 (2-cycle class) + 16 v_alignbit_b32 (4-cycle class) per trip,
// alternating one by one, in groups of four, or in groups of sixteen; chains are independent.
#define XR(i) "v_xor_b32 %" #i ", %" #i ", %8\n\t"
#define AL(i) "v_alignbit_b32 %" #i ", %" #i ", %" #i ", 12\n\t"
#define MIXKERNEL(NAME, BODY)                                                         \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t s) {          \
        uint32_t r[8];                                                                \
        uint32_t x = s ^ threadIdx.x, y = s * 2654435761u + blockIdx.x;               \
        for (int i = 0; i < 8; i++) r[i] = x * (i + 3) + y;                           \
        for (int it = 0; it < ITERS; it++) {                                          \
            asm volatile(BODY : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(x)); \
        }                                                                             \
        uint32_t acc = 0;                                                             \
        for (int i = 0; i < 8; i++) acc ^= r[i];                                      \
        if (acc == 0x12345) out[0] = acc;                                             \
    }
MIXKERNEL(k_mix_alt, XR(0) AL(1) XR(2) AL(3) XR(4) AL(5) XR(6) AL(7) XR(1) AL(0) XR(3) AL(2) XR(5) AL(4) XR(7) AL(6) XR(0) AL(1) XR(2) AL(3) XR(4) AL(5) XR(6) AL(7) XR(1) AL(0) XR(3) AL(2) XR(5) AL(4) XR(7) AL(6))
MIXKERNEL(k_mix_g4, XR(0) XR(1) XR(2) XR(3) AL(4) AL(5) AL(6) AL(7) XR(4) XR(5) XR(6) XR(7) AL(0) AL(1) AL(2) AL(3) XR(0) XR(1) XR(2) XR(3) AL(4) AL(5) AL(6) AL(7) XR(4) XR(5) XR(6) XR(7) AL(0) AL(1) AL(2) AL(3))
MIXKERNEL(k_mix_g16, XR(0) XR(1) XR(2) XR(3) XR(4) XR(5) XR(6) XR(7) XR(0) XR(1) XR(2) XR(3) XR(4) XR(5) XR(6) XR(7) AL(0) AL(1) AL(2) AL(3) AL(4) AL(5) AL(6) AL(7) AL(0) AL(1) AL(2) AL(3) AL(4) AL(5) AL(6) AL(7))

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
        {"v_fma_f64", k_fma64, 1}, {"v_mul_f64", k_mul64f, 1}, {"v_add_f64", k_add64f, 1}, {"mix 16xor+16alignbit alternating (per 32)", k_mix_alt, 1}, {"mix groups of 4 (per 32)", k_mix_g4, 1}, {"mix groups of 16 (per 32)", k_mix_g16, 1}, {"[xor;alignbit dependent pair] x8 chains", k_xoralign, 1}, {"v_xor_b32 one dependent chain", k_xordep, 1}, {"v_alignbit one dependent chain", k_aligndep, 1}, {"[xor;xor dependent pair]", k_xor2, 1}, {"[xor;add dependent pair]", k_xoradd, 1}, {"v_mad_u64_u32 one dependent chain", k_mad64dep, 1}, {"[mad64; and independent pair]", k_madand, 1}, {"[v_cmp;v_cndmask vcc]", k_cmpcnd, 1}, {"[v_cmp;4x v_cndmask vcc]", k_cmpcnd4, 1}, {"[v_cmp_e64;4x v_cndmask_e64 sgpr]", k_cmpcnd4s, 1}, {"[xor;and;xor select]", k_selarith, 1}, {"v_bitop3_b32(select 0xca)", k_bitopsel, 1}, {"v_mov_b32", k_mov, 1}, {"v_and_b32", k_and, 1}, {"v_or_b32", k_or, 1}, {"v_sub_u32", k_sub, 1}, {"v_lshlrev_b32", k_lshl, 1}, {"v_lshrrev_b32", k_lshr, 1}, {"v_cndmask_b32_e64(sgpr)", k_cnd_s, 1}, {"v_cmp_lt_u32(vcc)", k_cmp, 1}, {"v_cmp_lt_u32_e64(sgpr)", k_cmp_s, 1}, {"v_add_co_u32_e64(sgpr)", k_addco_s, 1}, {"v_lshl_or_b32", k_lshlor, 1}, {"v_and_or_b32", k_andor, 1}, {"v_or3_b32", k_or3, 1}, {"v_lshl_add_u32", k_lshladd, 1}, {"v_add_lshl_u32", k_addlshl, 1}, {"v_xad_u32", k_xad, 1}, {"v_bfe_u32", k_bfe, 1}, {"v_mul_f32", k_mulf32, 1}, {"v_mad_i32_i24", k_madi24, 1}, {"v_pk_add_u16", k_pkadd16, 1}, {"v_pk_mad_u16", k_pkmad16, 1}, {"v_pk_mul_lo_u16", k_pkmul16, 1}, {"v_mov_b32_sdwa(rot16)", k_sdwarot, 1}, {"v_mad_u16", k_madu16, 1}, {"v_mad_u32_u16", k_madu32u16, 1}, {"v_min_u32", k_minu, 1}, {"v_pk_fma_f32", k_pkfma32, 1}, {"v_cvt_f64_u32", k_cvtf64u32, 1}, {"v_lshlrev_b64", k_lshl64, 1}, {"v_lshrrev_b64", k_lshr64, 1}};
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
