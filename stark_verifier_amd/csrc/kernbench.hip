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

#include <vector>

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

// W: the multiplier's limb form (0 narrow everywhere, 2 wide in low columns 0..2, 3 wide in the high columns too)
template <int W>
__global__ void __launch_bounds__(256) k_femul29(Stamp* st, uint32_t* sink, uint32_t s) {
    F29 a = f29_from_fe(fe_canon(seed_fe(s + threadIdx.x))), b = f29_from_fe(fe_canon(seed_fe(s * 3 + blockIdx.x)));
    unsigned long long c0, r0;
    stamp_begin(c0, r0);
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) a = f29_mul<W>(a, b);
    stamp_end(st, c0, r0);
    uint32_t acc = 0;
    for (int i = 0; i < 9; i++) acc ^= a.l[i];
    if (acc == 0x1234567) sink[0] = acc;
}

// per-lane operands of the split product: y = b, y2 = b * 2^145 mod p (benchmark only, see field29.cuh)
template <int WIDE = 3>
__device__ __forceinline__ F29 f29_mul_split(const F29& x, const F29& y, const F29& y2) {
    const uint32_t* b = y.l;
    const uint32_t* c = y2.l;
    F29_MUL_SPLIT_BODY(F29_P)
}

// the split product with per-lane second operands (b, b * 2^145): 85 multiply-adds
template <int W>
__global__ void __launch_bounds__(256) k_femul29_split(Stamp* st, uint32_t* sink, uint32_t s) {
    F29 a = f29_from_fe(fe_canon(seed_fe(s + threadIdx.x))), b = f29_from_fe(fe_canon(seed_fe(s * 3 + blockIdx.x)));
    F29 c = f29_from_fe(fe_canon(seed_fe(s * 7 + blockIdx.x)));
    unsigned long long c0, r0;
    stamp_begin(c0, r0);
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) a = f29_mul_split<W>(a, b, c);
    stamp_end(st, c0, r0);
    uint32_t acc = 0;
    for (int i = 0; i < 9; i++) acc ^= a.l[i];
    if (acc == 0x1234567) sink[0] = acc;
}

// the same with the second operands in scalar registers (a wave-uniform constant)
template <int W>
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
    for (int it = 0; it < ITERS; it++) a = f29_mul_uniform2<W>(a, b, c);
    stamp_end(st, c0, r0);
    uint32_t acc = 0;
    for (int i = 0; i < 9; i++) acc ^= a.l[i];
    if (acc == 0x1234567) sink[0] = acc;
}


// ---- product-only bake-off: 9 x 9 limbs of 29 bits -> 18 normalized limbs, no reduction mod p ----------------------
// Three ways to form the 17 column sums of a 261-bit x 261-bit product, timed alone and checked against each other:
// schoolbook (81 multiply-adds), one level of Karatsuba over three blocks of three limbs (54 signed multiply-adds,
// 18 operand differences, 23 64-bit additions to place the shared block products), and FP64 fused multiply-adds on
// five 52-bit limbs (25 products, each split into a high and a low half by two v_fma_f64 and one v_add_f64 under
// round-toward-zero, the halves summed as integer bit patterns).
struct Prod18 {
    uint32_t l[18];
};

__device__ __forceinline__ void norm_cols(const uint64_t (&col)[17], Prod18& r) {
    uint64_t acc = 0;
#pragma unroll
    for (int c = 0; c < 17; c++) {
        acc += col[c];
        r.l[c] = (uint32_t)acc & F29_MASK;
        acc >>= 29;
    }
    r.l[17] = (uint32_t)acc;
}

__device__ __forceinline__ Prod18 prod_school(const F29& x, const F29& y) {
    Prod18 r;
    uint64_t acc = 0;
#pragma unroll
    for (int c = 0; c < 17; c++) {
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = c - i;
            if (j >= 0 && j < 9) acc = mad64(x.l[i], y.l[j], acc);
        }
        r.l[c] = (uint32_t)acc & F29_MASK;
        acc >>= 29;
    }
    r.l[17] = (uint32_t)acc;
    return r;
}

__device__ __forceinline__ int64_t smad64(int32_t a, int32_t b, int64_t c) { return (int64_t)a * b + c; }

// columns of a 3 x 3 block product, each chain started from seed[i]
__device__ __forceinline__ void block_u(const uint32_t* a, const uint32_t* b, const uint64_t (&seed)[5], uint64_t (&out)[5]) {
    out[0] = mad64(a[0], b[0], seed[0]);
    out[1] = mad64(a[1], b[0], mad64(a[0], b[1], seed[1]));
    out[2] = mad64(a[2], b[0], mad64(a[1], b[1], mad64(a[0], b[2], seed[2])));
    out[3] = mad64(a[2], b[1], mad64(a[1], b[2], seed[3]));
    out[4] = mad64(a[2], b[2], seed[4]);
}
__device__ __forceinline__ void block_s(const int32_t* a, const int32_t* b, const uint64_t (&seed)[5], uint64_t (&out)[5]) {
    out[0] = (uint64_t)smad64(a[0], b[0], (int64_t)seed[0]);
    out[1] = (uint64_t)smad64(a[1], b[0], smad64(a[0], b[1], (int64_t)seed[1]));
    out[2] = (uint64_t)smad64(a[2], b[0], smad64(a[1], b[1], smad64(a[0], b[2], (int64_t)seed[2])));
    out[3] = (uint64_t)smad64(a[2], b[1], smad64(a[1], b[2], (int64_t)seed[3]));
    out[4] = (uint64_t)smad64(a[2], b[2], (int64_t)seed[4]);
}

__device__ __forceinline__ Prod18 prod_karatsuba(const F29& x, const F29& y) {
    const uint32_t *a = x.l, *b = y.l;
    const uint64_t zero[5] = {0, 0, 0, 0, 0};
    uint64_t D0[5], D1[5], D2[5];
    block_u(a, b, zero, D0);
    block_u(a + 3, b + 3, zero, D1);
    block_u(a + 6, b + 6, zero, D2);
    // A_i B_j + A_j B_i = D_i + D_j + (A_j - A_i)(B_i - B_j)
    int32_t da01[3], db01[3], da02[3], db02[3], da12[3], db12[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        da01[k] = (int32_t)(a[3 + k] - a[k]);
        db01[k] = (int32_t)(b[k] - b[3 + k]);
        da02[k] = (int32_t)(a[6 + k] - a[k]);
        db02[k] = (int32_t)(b[k] - b[6 + k]);
        da12[k] = (int32_t)(a[6 + k] - a[3 + k]);
        db12[k] = (int32_t)(b[3 + k] - b[6 + k]);
    }
    uint64_t S01[5], S012[5], S12[5], T1[5], T2[5], T3[5], seed[5];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        S01[i] = D0[i] + D1[i];
        S012[i] = S01[i] + D2[i];
        S12[i] = D1[i] + D2[i];
    }
    // block t of the result starts at column 3 t: neighbours overlap in two columns, added through the chain seeds
#pragma unroll
    for (int i = 0; i < 5; i++) seed[i] = S01[i] + (i < 2 ? D0[3 + i] : 0);
    block_s(da01, db01, seed, T1);
#pragma unroll
    for (int i = 0; i < 5; i++) seed[i] = S012[i] + (i < 2 ? T1[3 + i] : 0);
    block_s(da02, db02, seed, T2);
#pragma unroll
    for (int i = 0; i < 5; i++) seed[i] = S12[i] + (i < 2 ? T2[3 + i] : 0);
    block_s(da12, db12, seed, T3);
    uint64_t col[17];
    col[0] = D0[0]; col[1] = D0[1]; col[2] = D0[2];
    col[3] = T1[0]; col[4] = T1[1]; col[5] = T1[2];
    col[6] = T2[0]; col[7] = T2[1]; col[8] = T2[2];
    col[9] = T3[0]; col[10] = T3[1]; col[11] = T3[2];
    col[12] = D2[0] + T3[3]; col[13] = D2[1] + T3[4]; col[14] = D2[2]; col[15] = D2[3]; col[16] = D2[4];
    Prod18 r;
    norm_cols(col, r);
    return r;
}

// 261 bits as five 52-bit limbs held as doubles (exact integers below 2^53)
struct F52 {
    double l[5];
};
__device__ __forceinline__ double u52_to_double(uint64_t k) {  // k < 2^52
    return __longlong_as_double((long long)(0x4330000000000000ull | k)) - 4503599627370496.0;
}
__device__ __forceinline__ F52 f52_from_29(const uint32_t* a) {  // nine 29-bit limbs -> five 52-bit limbs
    uint64_t w[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, k = bit / 52, sh = bit % 52;
        w[k] |= ((uint64_t)a[i] << sh) & 0xFFFFFFFFFFFFFull;
        if (sh + 29 > 52 && k + 1 < 5) w[k + 1] |= (uint64_t)a[i] >> (52 - sh);
    }
    F52 r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.l[i] = u52_to_double(w[i]);
    return r;
}
// ten 52-bit columns (normalized) of the product; needs round-toward-zero for f64 (set by the kernel)
__device__ __forceinline__ void prod_fp64(const F52& x, const F52& y, uint64_t (&out)[10]) {
    const double C1 = 20282409603651670423947251286016.0;                  // 2^104
    const double C2 = 20282409603651670423947251286016.0 + 4503599627370496.0;  // 2^104 + 2^52
    uint64_t hi_sum[10], lo_sum[10];
#pragma unroll
    for (int c = 0; c < 10; c++) hi_sum[c] = lo_sum[c] = 0;
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int j = 0; j < 5; j++) {
            // in assembly: the compiler restores round-to-nearest in front of the f64 instructions it emits itself
            double hi, sub, lo;
            asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(hi) : "v"(x.l[i]), "v"(y.l[j]), "v"(C1));
            asm volatile("v_add_f64 %0, %1, -%2" : "=v"(sub) : "v"(C2), "v"(hi));
            asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(lo) : "v"(x.l[i]), "v"(y.l[j]), "v"(sub));
            hi_sum[i + j + 1] += (uint64_t)__double_as_longlong(hi);
            lo_sum[i + j] += (uint64_t)__double_as_longlong(lo);
        }
    // bit patterns: hi = 0x4670.. | h (h = floor(ab / 2^52) < 2^52), lo = 0x4330.. | l
    uint64_t acc = 0;
#pragma unroll
    for (int c = 0; c < 10; c++) {
        const int nlo = c < 5 ? c + 1 : 9 - c, nhi = c == 0 ? 0 : (c <= 5 ? c : 10 - c);
        acc += lo_sum[c] - (uint64_t)nlo * 0x4330000000000000ull + hi_sum[c] - (uint64_t)nhi * 0x4670000000000000ull;
        out[c] = acc & 0xFFFFFFFFFFFFFull;
        acc >>= 52;
    }
}

// Round-toward-zero for f64 in this wave's MODE register, written in assembly (the compiler tracks its own writes of the
// register and would put round-to-nearest back); the operands pass through an empty volatile asm behind it.
__device__ __forceinline__ void f64_round_toward_zero(F52& a, F52& b) {
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");
#pragma unroll
    for (int i = 0; i < 5; i++) {
        asm volatile("" : "+v"(a.l[i]));
        asm volatile("" : "+v"(b.l[i]));
    }
}

__device__ __forceinline__ F29 seed_tight(uint32_t s) {
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = (s * (2654435761u + 40503u * i) + i * 977u) & F29_MASK;
    return r;
}

__global__ void __launch_bounds__(256) k_prod_school(Stamp* st, uint32_t* sink, uint32_t s) {
    F29 a = seed_tight(s + threadIdx.x), b = seed_tight(s * 3 + blockIdx.x);
    unsigned long long c0, r0;
    stamp_begin(c0, r0);
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
        const Prod18 p = prod_school(a, b);
#pragma unroll
        for (int i = 0; i < 9; i++) a.l[i] = p.l[i] ^ p.l[9 + i];
    }
    stamp_end(st, c0, r0);
    uint32_t acc = 0;
    for (int i = 0; i < 9; i++) acc ^= a.l[i];
    if (acc == 0x1234567) sink[0] = acc;
}

__global__ void __launch_bounds__(256) k_prod_kara(Stamp* st, uint32_t* sink, uint32_t s) {
    F29 a = seed_tight(s + threadIdx.x), b = seed_tight(s * 3 + blockIdx.x);
    unsigned long long c0, r0;
    stamp_begin(c0, r0);
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
        const Prod18 p = prod_karatsuba(a, b);
#pragma unroll
        for (int i = 0; i < 9; i++) a.l[i] = p.l[i] ^ p.l[9 + i];
    }
    stamp_end(st, c0, r0);
    uint32_t acc = 0;
    for (int i = 0; i < 9; i++) acc ^= a.l[i];
    if (acc == 0x1234567) sink[0] = acc;
}

__global__ void __launch_bounds__(256) k_prod_fp64(Stamp* st, uint32_t* sink, uint32_t s) {
    const F29 a29 = seed_tight(s + threadIdx.x), b29 = seed_tight(s * 3 + blockIdx.x);
    F52 a = f52_from_29(a29.l);
    F52 b = f52_from_29(b29.l);
    f64_round_toward_zero(a, b);
    unsigned long long c0, r0;
    stamp_begin(c0, r0);
    uint64_t col[10];
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
        prod_fp64(a, b, col);
#pragma unroll
        for (int i = 0; i < 5; i++) a.l[i] = u52_to_double(col[i] ^ col[5 + i]);  // back to limbs for the next product
    }
    stamp_end(st, c0, r0);
    uint64_t acc = 0;
    for (int i = 0; i < 10; i++) acc ^= col[i];
    if (acc == 0x1234567) sink[0] = (uint32_t)acc;
}

// The three forms on the same 65536 operand pairs, one kernel each (form 0 schoolbook, 1 Karatsuba, 2 FP64): the product
// as 17 words of 32 bits per pair, compared on the host.
__device__ __forceinline__ void check_operands(uint32_t t, uint32_t s, F29& a, F29& b) {
    a = seed_tight(s + t * 7919u);
    b = seed_tight(s * 3 + t * 104729u);
    if (t % 5 == 0)
        for (int i = 0; i < 9; i++) a.l[i] = F29_MASK;  // all-ones operands
    if (t % 7 == 0)
        for (int i = 0; i < 9; i++) b.l[i] = (i & 1) ? F29_MASK : 0;
    a.l[8] &= 0x00FFFFFFu;  // 256-bit operands (five 52-bit limbs hold 260 bits)
    b.l[8] &= 0x00FFFFFFu;
    // hipcc 7.2 turns (x & 0xFFFFFF) * (y & 0xFFFFFF) + acc into a 24-bit multiply, drops the masks as no longer needed,
    // and then selects the full 32-bit v_mad_u64_u32 on the unmasked registers: the masked values are hidden from it
    asm volatile("" : "+v"(a.l[8]), "+v"(b.l[8]));
}

template <int FORM>
__global__ void __launch_bounds__(256) k_prod_check(uint32_t* out, uint32_t s) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    F29 a, b;
    check_operands(t, s, a, b);
    uint32_t w[17];
    for (int i = 0; i < 17; i++) w[i] = 0;
    if (FORM == 2) {
        F52 fa = f52_from_29(a.l), fb = f52_from_29(b.l);
        f64_round_toward_zero(fa, fb);
        uint64_t col[10];
        prod_fp64(fa, fb, col);
        for (int k = 0; k < 520; k++) w[k / 32] |= ((uint32_t)(col[k / 52] >> (k % 52)) & 1u) << (k % 32);
    } else {
        const Prod18 p = FORM == 0 ? prod_school(a, b) : prod_karatsuba(a, b);
        for (int k = 0; k < 522; k++) w[k / 32] |= ((p.l[k / 29] >> (k % 29)) & 1u) << (k % 32);
    }
    for (int i = 0; i < 17; i++) out[17 * (size_t)t + i] = w[i];
}

template <int W>
__global__ void __launch_bounds__(256) k_butterfly29(Stamp* st, uint32_t* sink, uint32_t s) {
    F29 x = f29_from_fe(fe_canon(seed_fe(s + threadIdx.x))), y = f29_from_fe(fe_canon(seed_fe(s * 5 + threadIdx.x)));
    F29 w = f29_from_fe(fe_canon(seed_fe(s * 3 + blockIdx.x)));
    unsigned long long c0, r0;
    stamp_begin(c0, r0);
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
        F29 t = f29_mul<W>(y, w);
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

// Instruction-level forms of the compression, timed against the library's (form 0) and checked against it:
//   bit 0: the rotation by 16 folded into its xor as two sub-dword xors (SDWA: each writes one half of the destination
//          from the opposite halves of the sources) instead of v_xor + v_alignbit
//   bit 1: the rotations by 16 and by 8 as byte permutations (v_perm_b32 with a constant selector) instead of v_alignbit
//   bit 2: a + b + m as two v_add_u32 instead of one v_add3_u32
template <int F>
__device__ __forceinline__ uint32_t xor_ror16(uint32_t d, uint32_t a) {
    if constexpr ((F & 1) != 0) {
        uint32_t t;
        asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0" : "=v"(t) : "v"(d), "v"(a));
        asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(t) : "v"(d), "v"(a));
        return t;
    } else if constexpr ((F & 2) != 0) {
        const uint32_t x = d ^ a;
        return __builtin_amdgcn_perm(x, x, 0x01000302u);
    } else {
        return rotr32(d ^ a, 16);
    }
}
template <int F>
__device__ __forceinline__ uint32_t xor_ror8(uint32_t d, uint32_t a) {
    const uint32_t x = d ^ a;
    if constexpr ((F & 2) != 0) return __builtin_amdgcn_perm(x, x, 0x00030201u);
    else return rotr32(x, 8);
}
template <int F>
__device__ __forceinline__ uint32_t add_3(uint32_t a, uint32_t b, uint32_t m) {
    if constexpr ((F & 4) != 0) {
        uint32_t t, r;
        asm("v_add_u32 %0, %1, %2" : "=v"(t) : "v"(a), "v"(b));
        asm("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(t), "v"(m));
        return r;
    } else {
        return a + b + m;
    }
}
template <int F>
__device__ __forceinline__ void g_form(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d, uint32_t x, uint32_t y) {
    a = add_3<F>(a, b, x);
    d = xor_ror16<F>(d, a);
    c = c + d;
    b = rotr32(b ^ c, 12);
    a = add_3<F>(a, b, y);
    d = xor_ror8<F>(d, a);
    c = c + d;
    b = rotr32(b ^ c, 7);
}
template <int F>
__device__ __forceinline__ void b2s_hash64_form(uint32_t h[8], const uint32_t m[16]) {
    static const uint8_t SIGMA[10][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
    b2s_init(h);
    uint32_t v[16] = {h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], B2S_IV0, B2S_IV1, B2S_IV2, B2S_IV3,
                      B2S_IV4 ^ 64u, B2S_IV5, ~B2S_IV6, B2S_IV7};
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint8_t* s = SIGMA[r];
        g_form<F>(v[0], v[4], v[8], v[12], m[s[0]], m[s[1]]);
        g_form<F>(v[1], v[5], v[9], v[13], m[s[2]], m[s[3]]);
        g_form<F>(v[2], v[6], v[10], v[14], m[s[4]], m[s[5]]);
        g_form<F>(v[3], v[7], v[11], v[15], m[s[6]], m[s[7]]);
        g_form<F>(v[0], v[5], v[10], v[15], m[s[8]], m[s[9]]);
        g_form<F>(v[1], v[6], v[11], v[12], m[s[10]], m[s[11]]);
        g_form<F>(v[2], v[7], v[8], v[13], m[s[12]], m[s[13]]);
        g_form<F>(v[3], v[4], v[9], v[14], m[s[14]], m[s[15]]);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[8 + i];
}
template <int F>
__global__ void __launch_bounds__(256) k_blake2s_form(Stamp* st, uint32_t* sink, uint32_t s) {
    uint32_t h[8], m[16];
    for (int i = 0; i < 16; i++) m[i] = s * (i + 7) + threadIdx.x * 2654435761u;
    unsigned long long c0, r0;
    stamp_begin(c0, r0);
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
        b2s_hash64_form<F>(h, m);
        for (int i = 0; i < 8; i++) m[i] = h[i];
    }
    stamp_end(st, c0, r0);
    uint32_t acc = 0;
    for (int i = 0; i < 8; i++) acc ^= h[i];
    if (acc == 0x1234567) sink[0] = acc;
}
// every form on the same messages: out[8 f .. 8 f + 7] = digest of form f (0 = the library's)
__global__ void k_blake2s_forms(uint32_t* out, uint32_t s) {
    uint32_t h[8], m[16];
    for (int i = 0; i < 16; i++) m[i] = s * (i + 7) + (threadIdx.x + blockIdx.x * blockDim.x) * 2654435761u;
    uint32_t* o = out + 64 * (size_t)(threadIdx.x + blockIdx.x * blockDim.x);
    b2s_hash64(h, m);
    for (int i = 0; i < 8; i++) o[i] = h[i];
    b2s_hash64_form<1>(h, m);
    for (int i = 0; i < 8; i++) o[8 + i] = h[i];
    b2s_hash64_form<2>(h, m);
    for (int i = 0; i < 8; i++) o[16 + i] = h[i];
    b2s_hash64_form<4>(h, m);
    for (int i = 0; i < 8; i++) o[24 + i] = h[i];
    b2s_hash64_form<6>(h, m);
    for (int i = 0; i < 8; i++) o[32 + i] = h[i];
    b2s_hash64_form<0>(h, m);
    for (int i = 0; i < 8; i++) o[40 + i] = h[i];
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
    // The chip's clock tops out at 2.4 GHz: a higher reading (seen at eight wavefronts per SIMD for kernels that do not
    // fit eight per SIMD) means the two counters did not cover the same interval, and the row's cycle count is marked.
    printf("%-14s wave-latency %8.1f cyc/op  clock %.2f GHz  chip rate %8.2f Gop/s  (%.1f us/launch)%s\n", name,
           cyc / ITERS, ghz, ops / (ms * 1e-3) * 1e-9, ms * 1e3 / reps,
           ghz > 2.45 ? "  [clock reading implausible: use the rate, not the cycles, of this row]" : "");
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
    // the clock takes a second or so of load to settle (the first kernels of a cold run report 2.0 GHz, later ones 2.3-2.4):
    // run the multiplier untimed for that long first, so that the rates below are comparable with each other
    for (int r = 0; r < 400; r++) hipLaunchKernelGGL(k_femul29<0>, dim3(blocks), dim3(256), 0, 0, d_st, d_sink, 1u);
    CHECK(hipDeviceSynchronize());
    run("fe_mul", k_femul, blocks, d_st, h_st, d_sink);
    run("butterfly", k_butterfly, blocks, d_st, h_st, d_sink);
    // (twice, the second time in the opposite order, as a check that the order does not matter)
    run("f29_mul narrow", k_femul29<0>, blocks, d_st, h_st, d_sink);
    run("f29_mul w2", k_femul29<2>, blocks, d_st, h_st, d_sink);
    run("f29_mul w1", k_femul29<1>, blocks, d_st, h_st, d_sink);
    run("f29_mul w3", k_femul29<3>, blocks, d_st, h_st, d_sink);
    run("f29_mul w3", k_femul29<3>, blocks, d_st, h_st, d_sink);
    run("f29_mul w1", k_femul29<1>, blocks, d_st, h_st, d_sink);
    run("f29_mul w2", k_femul29<2>, blocks, d_st, h_st, d_sink);
    run("f29_mul narrow", k_femul29<0>, blocks, d_st, h_st, d_sink);
    run("f29_mul_split", k_femul29_split<3>, blocks, d_st, h_st, d_sink);
    run("unif2 narrow", k_femul29_uniform<0>, blocks, d_st, h_st, d_sink);
    run("unif2 w2", k_femul29_uniform<2>, blocks, d_st, h_st, d_sink);
    run("unif2 w3", k_femul29_uniform<3>, blocks, d_st, h_st, d_sink);
    run("butterfly29 n", k_butterfly29<0>, blocks, d_st, h_st, d_sink);
    run("butterfly29 w3", k_butterfly29<3>, blocks, d_st, h_st, d_sink);
    {
        const size_t n = 65536, words = 17 * n;
        uint32_t* d_out;
        std::vector<uint32_t> h[3];
        CHECK(hipMalloc(&d_out, 4 * words));
        for (int f = 0; f < 3; f++) {
            h[f].resize(words);
            if (f == 0) hipLaunchKernelGGL(k_prod_check<0>, dim3(256), dim3(256), 0, 0, d_out, 12345u);
            if (f == 1) hipLaunchKernelGGL(k_prod_check<1>, dim3(256), dim3(256), 0, 0, d_out, 12345u);
            if (f == 2) hipLaunchKernelGGL(k_prod_check<2>, dim3(256), dim3(256), 0, 0, d_out, 12345u);
            CHECK(hipMemcpy(h[f].data(), d_out, 4 * words, hipMemcpyDeviceToHost));
        }
        if (getenv("KERNBENCH_DUMP")) {
            const size_t t = (size_t)atoi(getenv("KERNBENCH_DUMP"));
            for (int f = 0; f < 3; f++) {
                printf("form %d:", f);
                for (int i = 0; i < 17; i++) printf(" %08x", h[f][17 * t + i]);
                printf("\n");
            }
        }
        size_t bad_k = 0, bad_f = 0, nonzero = 0;
        for (size_t t = 0; t < n; t++) {
            bool k = true, f = true, nz = false;
            for (int i = 0; i < 17; i++) {
                k = k && h[0][17 * t + i] == h[1][17 * t + i];
                f = f && h[0][17 * t + i] == h[2][17 * t + i];
                nz = nz || h[0][17 * t + i] != 0;
            }
            bad_k += !k;
            bad_f += !f;
            nonzero += nz;
        }
        printf("product forms agree on %zu operand pairs (%zu non-zero products): %s (Karatsuba form: %zu mismatches, FP64 form: %zu)\n",
               n, nonzero, bad_k + bad_f ? "NO" : "yes", bad_k, bad_f);
        CHECK(hipFree(d_out));
    }
    run("prod_school", k_prod_school, blocks, d_st, h_st, d_sink);
    run("prod_karatsuba", k_prod_kara, blocks, d_st, h_st, d_sink);
    run("prod_fp64", k_prod_fp64, blocks, d_st, h_st, d_sink);
    run("blake2s", k_blake2s, blocks, d_st, h_st, d_sink);
    run("blake2s sdwa16", k_blake2s_form<1>, blocks, d_st, h_st, d_sink);
    run("blake2s perm", k_blake2s_form<2>, blocks, d_st, h_st, d_sink);
    run("blake2s 2adds", k_blake2s_form<4>, blocks, d_st, h_st, d_sink);
    run("blake2s perm+2a", k_blake2s_form<6>, blocks, d_st, h_st, d_sink);
    {
        const size_t n = 16384;
        uint32_t* d_out;
        std::vector<uint32_t> h(64 * n);
        CHECK(hipMalloc(&d_out, 256 * n));
        hipLaunchKernelGGL(k_blake2s_forms, dim3(n / 256), dim3(256), 0, 0, d_out, 777u);
        CHECK(hipMemcpy(h.data(), d_out, 256 * n, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t t = 0; t < n; t++)
            for (int f = 1; f < 6; f++)
                for (int i = 0; i < 8; i++) bad += h[64 * t + i] != h[64 * t + 8 * f + i];
        printf("blake2s forms agree on %zu messages: %s\n", n, bad ? "NO" : "yes");
        CHECK(hipFree(d_out));
    }
    return 0;
}
