// Carry-free ("unsaturated") arithmetic mod p = 2^256 - 351*2^32 + 1 for gfx950.
//
// A value is held as nine limbs of 29 bits, X = sum l[i] * 2^(29 i). Limbs live in 32-bit
// registers, so up to three spare bits per limb absorb additions without carry chains:
//   * add / sub are nine independent v_add_u32 / v_sub_u32 (the 2-cycle VALU class) instead
//     of two 8-limb v_addc_co_u32 chains (4-cycle class) plus selects;
//   * in the product, a column of nine 61-bit partial products fits a 64-bit accumulator, so
//     v_mad_u64_u32 accumulates a whole column with no carry-out handling; the column's limb
//     is the low 29 bits and the rest is the next column's starting value.
// On this chip v_mad_u64_u32 and v_addc_co_u32 issue at the same rate, so dropping the carry
// instruction that a saturated 8x32-bit product needs after every multiply-add removes close
// to half of the multiplier's issue slots.
//
// Forms
//   "lazy"  : limbs < 3 * 2^30, any value below 2^261 that is congruent to the element.
//   "tight" : limbs 0..7 < 2^29 (limb 3 may exceed it by less than 2^5), limb 8 < 2^24, i.e. value
//             < 2^256 + 2^92. f29_mul returns tight values; table entries (twiddles) are canonical,
//             which is tight.
// Results leave this representation through f29_to_fe, which returns the canonical residue, so
// everything observable is bit-identical to exact integer arithmetic mod p (what the
// reference's BigInt `%` and utils::negative_to_positive produce, reference src/fft.rs:28,53-57,84).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "field.cuh"

namespace starkhip {

struct F29 {
    uint32_t l[9];
};

#define F29_MASK 0x1FFFFFFFu
// 2^261 == F29_FOLD1 * 2^29 + F29_FOLD0 (mod p)   [= 351*2^37 - 32]
#define F29_FOLD0 0x1FFFFFE0u /* 2^29 - 32 */
#define F29_FOLD1 89855u      /* 351*2^8 - 1 */
// 2^256 == F29_C1 * 2^29 + F29_C0 (mod p)         [= 351*2^32 - 1]
#define F29_C0 0x1FFFFFFFu /* 2^29 - 1 */
#define F29_C1 2807u

__device__ __forceinline__ uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) { return (uint64_t)a * b + c; }

// 8 x 32-bit canonical (or any 256-bit) value -> nine 29-bit limbs
__device__ __forceinline__ F29 f29_from_fe(const Fe& x) {
    const uint32_t* w = x.l;
    F29 r;
    r.l[0] = w[0] & F29_MASK;
    r.l[1] = __builtin_amdgcn_alignbit(w[1], w[0], 29) & F29_MASK;
    r.l[2] = __builtin_amdgcn_alignbit(w[2], w[1], 26) & F29_MASK;
    r.l[3] = __builtin_amdgcn_alignbit(w[3], w[2], 23) & F29_MASK;
    r.l[4] = __builtin_amdgcn_alignbit(w[4], w[3], 20) & F29_MASK;
    r.l[5] = __builtin_amdgcn_alignbit(w[5], w[4], 17) & F29_MASK;
    r.l[6] = __builtin_amdgcn_alignbit(w[6], w[5], 14) & F29_MASK;
    r.l[7] = __builtin_amdgcn_alignbit(w[7], w[6], 11) & F29_MASK;
    r.l[8] = w[7] >> 8;
    return r;
}

// One sequential carry pass: limbs 0..7 end below 2^29, limb 8 takes what is left.
__device__ __forceinline__ F29 f29_carry(const F29& a) {
    F29 r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t v = a.l[i] + c;
        r.l[i] = v & F29_MASK;
        c = v >> 29;
    }
    r.l[8] = a.l[8] + c;
    return r;
}

__device__ __forceinline__ F29 f29_add(const F29& a, const F29& b) {
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}

// a - b + 2p for tight b: 2p written with limbs that dominate a tight value's limbs
// {2^29+2, 2^30-5617, 2^30-2 (x6), 2^25-2}, so no limb goes negative.
__device__ __forceinline__ F29 f29_sub(const F29& a, const F29& b) {
    F29 r;
    r.l[0] = a.l[0] + (0x20000002u - b.l[0]);
    r.l[1] = a.l[1] + (0x3FFFEA0Fu - b.l[1]);
#pragma unroll
    for (int i = 2; i < 8; i++) r.l[i] = a.l[i] + (0x3FFFFFFEu - b.l[i]);
    r.l[8] = a.l[8] + (0x01FFFFFEu - b.l[8]);
    return r;
}

// a - b + 4p for b a sum of two tight values (limbs < 2^30, limb 8 <= 2^25):
// 4p = {2^30+4, 2^31-11233, 2^31-3 ... } rewritten so every limb dominates:
// {2^30+4, 2^31-11234, 2^31-4 (x6), 2^26-4}
__device__ __forceinline__ F29 f29_sub2(const F29& a, const F29& b) {
    F29 r;
    r.l[0] = a.l[0] + (0x40000004u - b.l[0]);
    r.l[1] = a.l[1] + (0x7FFFD41Eu - b.l[1]);
#pragma unroll
    for (int i = 2; i < 8; i++) r.l[i] = a.l[i] + (0x7FFFFFFCu - b.l[i]);
    r.l[8] = a.l[8] + (0x03FFFFFCu - b.l[8]);
    return r;
}

// acc += sum of N products, as one strictly serial v_mad_u64_u32 chain (the compiler would
// otherwise start a second accumulator per column and pay a 64-bit add to merge it).
// F29_M0 opens a chain at zero (it writes the accumulator without reading it).
#define F29_M(i, j) "v_mad_u64_u32 %0, vcc, %" #i ", %" #j ", %0\n\t"
#define F29_M0(i, j) "v_mad_u64_u32 %0, vcc, %" #i ", %" #j ", 0\n\t"
#define F29_P(x, y) "v"(x), "v"(y)
#define F29_K(x, c) "v"(x), "s"(c)
#define F29_PS(x, y) "v"(x), "s"(y)
// Leaving a column. F29_NEXT: its limb is the low 29 bits, the rest starts the next column (one v_and, one 64-bit
// shift). F29_WIDE: its limb is the whole low register, and the high register times 8 starts the next column
// (acc = lo + 2^32 hi = lo + 2^29 * 8 hi), set by one multiply-add of its own with the constants 8 and 0: no
// v_and, and a multiply-add in place of the shift. Wide limbs are below 2^32 instead of 2^29, which is fine wherever
// the limb is only a multiplier operand or an addend of a later 64-bit sum: the high limbs th[] (a column then holds
// at most nine products below 3 * 2^59 each plus th * FOLD0 < 2^61 plus th * FOLD1 < 2^49: below 2^64) and the low
// limbs 0..2, which the second fold masks anyway.
#define F29_NEXT(dst)               \
    dst = (uint32_t)acc & F29_MASK; \
    acc >>= 29;
#define F29_WIDE(dst)                                                                             \
    dst = (uint32_t)acc;                                                                          \
    hi = (uint32_t)(acc >> 32);                                                                   \
    asm("v_mad_u64_u32 %0, vcc, %1, 8, 0" : "=v"(acc) : "v"(hi) : "vcc"); /* may reuse acc's registers */

#define F29_STEP_HI(dst)                 \
    if constexpr ((WIDE & 1) != 0) {     \
        F29_WIDE(dst)                    \
    } else {                             \
        F29_NEXT(dst)                    \
    }
#define F29_STEP_LO(dst)                 \
    if constexpr ((WIDE & 2) != 0) {     \
        F29_WIDE(dst)                    \
    } else {                             \
        F29_NEXT(dst)                    \
    }

// a * b mod p (not fully reduced): a lazy (limbs < 3*2^30), b tight. Returns a tight value.
// WIDE: bit 0 = wide limbs leaving the high columns, bit 1 = leaving low columns 0..2. The wide form saves a tenth of
// the instructions but keeps a few more registers alive (a wide limb stays in the low half of the pair its column
// ended in): bit 0 costs about five registers, bit 1 none, so bit 0 is for kernels with registers to spare.
template <int WIDE = 2>
__device__ __forceinline__ F29 f29_mul(const F29& x, const F29& y) {
    const uint32_t* a = x.l;
    const uint32_t* b = y.l;
    const uint32_t f0 = F29_FOLD0, f1 = F29_FOLD1;
    // columns 9..16 first: their limbs th[0..8] sit at 2^261 * 2^(29 k)
    [[maybe_unused]] uint32_t hi;
    uint32_t th[9];
    uint64_t acc;
    asm(F29_M0(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10) F29_M(11, 12) F29_M(13, 14) F29_M(15, 16)
        : "=&v"(acc)
        : F29_P(a[1], b[8]), F29_P(a[2], b[7]), F29_P(a[3], b[6]), F29_P(a[4], b[5]), F29_P(a[5], b[4]),
          F29_P(a[6], b[3]), F29_P(a[7], b[2]), F29_P(a[8], b[1])
        : "vcc");
    F29_STEP_HI(th[0])
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10) F29_M(11, 12) F29_M(13, 14)
        : "+v"(acc)
        : F29_P(a[2], b[8]), F29_P(a[3], b[7]), F29_P(a[4], b[6]), F29_P(a[5], b[5]), F29_P(a[6], b[4]),
          F29_P(a[7], b[3]), F29_P(a[8], b[2])
        : "vcc");
    F29_STEP_HI(th[1])
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10) F29_M(11, 12)
        : "+v"(acc)
        : F29_P(a[3], b[8]), F29_P(a[4], b[7]), F29_P(a[5], b[6]), F29_P(a[6], b[5]), F29_P(a[7], b[4]),
          F29_P(a[8], b[3])
        : "vcc");
    F29_STEP_HI(th[2])
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10)
        : "+v"(acc)
        : F29_P(a[4], b[8]), F29_P(a[5], b[7]), F29_P(a[6], b[6]), F29_P(a[7], b[5]), F29_P(a[8], b[4])
        : "vcc");
    F29_STEP_HI(th[3])
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8)
        : "+v"(acc)
        : F29_P(a[5], b[8]), F29_P(a[6], b[7]), F29_P(a[7], b[6]), F29_P(a[8], b[5])
        : "vcc");
    F29_STEP_HI(th[4])
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6)
        : "+v"(acc)
        : F29_P(a[6], b[8]), F29_P(a[7], b[7]), F29_P(a[8], b[6])
        : "vcc");
    F29_STEP_HI(th[5])
    asm(F29_M(1, 2) F29_M(3, 4) : "+v"(acc) : F29_P(a[7], b[8]), F29_P(a[8], b[7]) : "vcc");
    F29_STEP_HI(th[6])
    asm(F29_M(1, 2) : "+v"(acc) : F29_P(a[8], b[8]) : "vcc");
    F29_STEP_HI(th[7])
    th[8] = (uint32_t)acc;  // what the last column (below 2^61) hands on: below 2^32
    // columns 0..8 with the high limbs folded in: 2^261 * t == (FOLD1 * 2^29 + FOLD0) * t
    uint32_t r[9];
    asm(F29_M0(1, 2) F29_M(3, 4) : "=&v"(acc) : F29_P(a[0], b[0]), F29_K(th[0], f0) : "vcc");
    F29_STEP_LO(r[0])
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8)
        : "+v"(acc)
        : F29_P(a[0], b[1]), F29_P(a[1], b[0]), F29_K(th[1], f0), F29_K(th[0], f1)
        : "vcc");
    F29_STEP_LO(r[1])
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10)
        : "+v"(acc)
        : F29_P(a[0], b[2]), F29_P(a[1], b[1]), F29_P(a[2], b[0]), F29_K(th[2], f0), F29_K(th[1], f1)
        : "vcc");
    F29_STEP_LO(r[2])
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10) F29_M(11, 12)
        : "+v"(acc)
        : F29_P(a[0], b[3]), F29_P(a[1], b[2]), F29_P(a[2], b[1]), F29_P(a[3], b[0]), F29_K(th[3], f0),
          F29_K(th[2], f1)
        : "vcc");
    F29_NEXT(r[3])
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10) F29_M(11, 12) F29_M(13, 14)
        : "+v"(acc)
        : F29_P(a[0], b[4]), F29_P(a[1], b[3]), F29_P(a[2], b[2]), F29_P(a[3], b[1]), F29_P(a[4], b[0]),
          F29_K(th[4], f0), F29_K(th[3], f1)
        : "vcc");
    F29_NEXT(r[4])
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10) F29_M(11, 12) F29_M(13, 14) F29_M(15, 16)
        : "+v"(acc)
        : F29_P(a[0], b[5]), F29_P(a[1], b[4]), F29_P(a[2], b[3]), F29_P(a[3], b[2]), F29_P(a[4], b[1]),
          F29_P(a[5], b[0]), F29_K(th[5], f0), F29_K(th[4], f1)
        : "vcc");
    F29_NEXT(r[5])
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10) F29_M(11, 12) F29_M(13, 14) F29_M(15, 16)
            F29_M(17, 18)
        : "+v"(acc)
        : F29_P(a[0], b[6]), F29_P(a[1], b[5]), F29_P(a[2], b[4]), F29_P(a[3], b[3]), F29_P(a[4], b[2]),
          F29_P(a[5], b[1]), F29_P(a[6], b[0]), F29_K(th[6], f0), F29_K(th[5], f1)
        : "vcc");
    F29_NEXT(r[6])
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10) F29_M(11, 12) F29_M(13, 14) F29_M(15, 16)
            F29_M(17, 18) F29_M(19, 20)
        : "+v"(acc)
        : F29_P(a[0], b[7]), F29_P(a[1], b[6]), F29_P(a[2], b[5]), F29_P(a[3], b[4]), F29_P(a[4], b[3]),
          F29_P(a[5], b[2]), F29_P(a[6], b[1]), F29_P(a[7], b[0]), F29_K(th[7], f0), F29_K(th[6], f1)
        : "vcc");
    F29_NEXT(r[7])
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10) F29_M(11, 12) F29_M(13, 14) F29_M(15, 16)
            F29_M(17, 18) F29_M(19, 20) F29_M(21, 22)
        : "+v"(acc)
        : F29_P(a[0], b[8]), F29_P(a[1], b[7]), F29_P(a[2], b[6]), F29_P(a[3], b[5]), F29_P(a[4], b[4]),
          F29_P(a[5], b[3]), F29_P(a[6], b[2]), F29_P(a[7], b[1]), F29_P(a[8], b[0]), F29_K(th[8], f0),
          F29_K(th[7], f1)
        : "vcc");
    F29_NEXT(r[8])
    acc = mad64(th[8], F29_FOLD1, acc);  // what remains at 2^261: below 2^45
    // second fold at 2^256: top = everything above bit 256 = (acc << 5) | (r[8] >> 24)
    uint64_t top = (acc << 5) | (uint64_t)(r[8] >> 24);
    r[8] &= 0x00FFFFFFu;
    uint32_t u0 = (uint32_t)top & F29_MASK, u1 = (uint32_t)(top >> 29);  // u1 < 2^21
    F29 o;
    acc = mad64(u0, F29_C0, (uint64_t)r[0]);
    o.l[0] = (uint32_t)acc & F29_MASK;
    acc >>= 29;
    acc = mad64(u0, F29_C1, acc + r[1]);
    acc = mad64(u1, F29_C0, acc);
    o.l[1] = (uint32_t)acc & F29_MASK;
    acc >>= 29;
    acc = mad64(u1, F29_C1, acc + r[2]);
    o.l[2] = (uint32_t)acc & F29_MASK;
    // what is left above limb 2 is below 2^5 (u1*C1 < 2^33, shifted down by 29): it stays in limb 3
    // unpropagated, which every consumer of a tight value tolerates
    o.l[3] = r[3] + (uint32_t)(acc >> 29);
#pragma unroll
    for (int i = 4; i < 9; i++) o.l[i] = r[i];
    return o;
}

// x * x mod p for a tight x; returns a tight value. The columns are those of f29_mul(x, x) with the two equal
// products a_i a_j, a_j a_i of every pair i < j taken once against the doubled limb d_i = 2 a_i (below 2^30 + 2^6 for a
// tight x, so the column sums are the same numbers as in f29_mul and keep its bounds): 45 multiply-adds and 8 shifts
// in place of 81 multiply-adds. For the one caller whose time is its instruction count, the prover's trace walk
// (x -> x^3 + k, a wavefront alone on its SIMD).
__device__ __forceinline__ F29 f29_sqr(const F29& x) {
    constexpr int WIDE = 2;
    const uint32_t* a = x.l;
    uint32_t d[8];
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = a[i] << 1;
    const uint32_t f0 = F29_FOLD0, f1 = F29_FOLD1;
    [[maybe_unused]] uint32_t hi;
    uint32_t th[9];
    uint64_t acc;
    asm(F29_M0(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8)
        : "=&v"(acc)
        : F29_P(d[1], a[8]), F29_P(d[2], a[7]), F29_P(d[3], a[6]), F29_P(d[4], a[5])
        : "vcc");
    F29_STEP_HI(th[0])
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8)
        : "+v"(acc)
        : F29_P(d[2], a[8]), F29_P(d[3], a[7]), F29_P(d[4], a[6]), F29_P(a[5], a[5])
        : "vcc");
    F29_STEP_HI(th[1])
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) : "+v"(acc) : F29_P(d[3], a[8]), F29_P(d[4], a[7]), F29_P(d[5], a[6]) : "vcc");
    F29_STEP_HI(th[2])
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) : "+v"(acc) : F29_P(d[4], a[8]), F29_P(d[5], a[7]), F29_P(a[6], a[6]) : "vcc");
    F29_STEP_HI(th[3])
    asm(F29_M(1, 2) F29_M(3, 4) : "+v"(acc) : F29_P(d[5], a[8]), F29_P(d[6], a[7]) : "vcc");
    F29_STEP_HI(th[4])
    asm(F29_M(1, 2) F29_M(3, 4) : "+v"(acc) : F29_P(d[6], a[8]), F29_P(a[7], a[7]) : "vcc");
    F29_STEP_HI(th[5])
    asm(F29_M(1, 2) : "+v"(acc) : F29_P(d[7], a[8]) : "vcc");
    F29_STEP_HI(th[6])
    asm(F29_M(1, 2) : "+v"(acc) : F29_P(a[8], a[8]) : "vcc");
    F29_STEP_HI(th[7])
    th[8] = (uint32_t)acc;
    uint32_t r[9];
    asm(F29_M0(1, 2) F29_M(3, 4) : "=&v"(acc) : F29_P(a[0], a[0]), F29_K(th[0], f0) : "vcc");
    F29_STEP_LO(r[0])
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) : "+v"(acc) : F29_P(d[0], a[1]), F29_K(th[1], f0), F29_K(th[0], f1) : "vcc");
    F29_STEP_LO(r[1])
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8)
        : "+v"(acc)
        : F29_P(d[0], a[2]), F29_P(a[1], a[1]), F29_K(th[2], f0), F29_K(th[1], f1)
        : "vcc");
    F29_STEP_LO(r[2])
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8)
        : "+v"(acc)
        : F29_P(d[0], a[3]), F29_P(d[1], a[2]), F29_K(th[3], f0), F29_K(th[2], f1)
        : "vcc");
    F29_NEXT(r[3])
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10)
        : "+v"(acc)
        : F29_P(d[0], a[4]), F29_P(d[1], a[3]), F29_P(a[2], a[2]), F29_K(th[4], f0), F29_K(th[3], f1)
        : "vcc");
    F29_NEXT(r[4])
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10)
        : "+v"(acc)
        : F29_P(d[0], a[5]), F29_P(d[1], a[4]), F29_P(d[2], a[3]), F29_K(th[5], f0), F29_K(th[4], f1)
        : "vcc");
    F29_NEXT(r[5])
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10) F29_M(11, 12)
        : "+v"(acc)
        : F29_P(d[0], a[6]), F29_P(d[1], a[5]), F29_P(d[2], a[4]), F29_P(a[3], a[3]), F29_K(th[6], f0), F29_K(th[5], f1)
        : "vcc");
    F29_NEXT(r[6])
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10) F29_M(11, 12)
        : "+v"(acc)
        : F29_P(d[0], a[7]), F29_P(d[1], a[6]), F29_P(d[2], a[5]), F29_P(d[3], a[4]), F29_K(th[7], f0), F29_K(th[6], f1)
        : "vcc");
    F29_NEXT(r[7])
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10) F29_M(11, 12) F29_M(13, 14)
        : "+v"(acc)
        : F29_P(d[0], a[8]), F29_P(d[1], a[7]), F29_P(d[2], a[6]), F29_P(d[3], a[5]), F29_P(a[4], a[4]), F29_K(th[8], f0),
          F29_K(th[7], f1)
        : "vcc");
    F29_NEXT(r[8])
    acc = mad64(th[8], F29_FOLD1, acc);
    uint64_t top = (acc << 5) | (uint64_t)(r[8] >> 24);
    r[8] &= 0x00FFFFFFu;
    uint32_t u0 = (uint32_t)top & F29_MASK, u1 = (uint32_t)(top >> 29);
    F29 o;
    acc = mad64(u0, F29_C0, (uint64_t)r[0]);
    o.l[0] = (uint32_t)acc & F29_MASK;
    acc >>= 29;
    acc = mad64(u0, F29_C1, acc + r[1]);
    acc = mad64(u1, F29_C0, acc);
    o.l[1] = (uint32_t)acc & F29_MASK;
    acc >>= 29;
    acc = mad64(u1, F29_C1, acc + r[2]);
    o.l[2] = (uint32_t)acc & F29_MASK;
    o.l[3] = r[3] + (uint32_t)(acc >> 29);
#pragma unroll
    for (int i = 4; i < 9; i++) o.l[i] = r[i];
    return o;
}

// The product with the first operand split a = a_lo + 2^145 a_hi (limbs 0..4 and 5..8) and the second operand given
// twice, as b and as c = b * 2^145 mod p (both tight): a*b == a_lo*b + a_hi*c, whose columns reach only 2^(29*12), so
// four high limbs are folded back instead of nine (85 multiply-adds instead of 103 and four limb extractions fewer
// than f29_mul). For a wave-uniform constant (the 4th root of unity, the twiddles of a round in which a wavefront's
// lanes all run the same butterfly index) b and c sit in scalar registers: no vector registers and no load for them.
// The body is a macro over the operand-pair constraint so that the register-only benchmark (kernbench.hip) can also
// build the variant with per-lane second operands, which the transform does not use: inside the pass kernel the
// second table load per twiddle costs more than the 18 multiply-adds it saves.
#define F29_MUL_SPLIT_BODY(PAIR) \
    const uint32_t* a = x.l; \
    const uint32_t f0 = F29_FOLD0, f1 = F29_FOLD1; \
    [[maybe_unused]] uint32_t hi; \
    uint32_t th[5]; \
    uint64_t acc; \
    asm(F29_M0(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10) F29_M(11, 12) F29_M(13, 14) : "=&v"(acc) : PAIR(a[1], b[8]), PAIR(a[2], b[7]), PAIR(a[3], b[6]), PAIR(a[4], b[5]), PAIR(a[6], c[8]), PAIR(a[7], c[7]), PAIR(a[8], c[6]) : "vcc"); \
    F29_STEP_HI(th[0]) \
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10) : "+v"(acc) : PAIR(a[2], b[8]), PAIR(a[3], b[7]), PAIR(a[4], b[6]), PAIR(a[7], c[8]), PAIR(a[8], c[7]) : "vcc"); \
    F29_STEP_HI(th[1]) \
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) : "+v"(acc) : PAIR(a[3], b[8]), PAIR(a[4], b[7]), PAIR(a[8], c[8]) : "vcc"); \
    F29_STEP_HI(th[2]) \
    asm(F29_M(1, 2) : "+v"(acc) : PAIR(a[4], b[8]) : "vcc"); \
    F29_STEP_HI(th[3]) \
    th[4] = (uint32_t)acc; \
    uint32_t r[9]; \
    asm(F29_M0(1, 2) F29_M(3, 4) F29_M(5, 6) : "=&v"(acc) : PAIR(a[0], b[0]), PAIR(a[5], c[0]), F29_K(th[0], f0) : "vcc"); \
    F29_STEP_LO(r[0]) \
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10) F29_M(11, 12) : "+v"(acc) : PAIR(a[0], b[1]), PAIR(a[1], b[0]), PAIR(a[5], c[1]), PAIR(a[6], c[0]), F29_K(th[1], f0), F29_K(th[0], f1) : "vcc"); \
    F29_STEP_LO(r[1]) \
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10) F29_M(11, 12) F29_M(13, 14) F29_M(15, 16) : "+v"(acc) : PAIR(a[0], b[2]), PAIR(a[1], b[1]), PAIR(a[2], b[0]), PAIR(a[5], c[2]), PAIR(a[6], c[1]), PAIR(a[7], c[0]), F29_K(th[2], f0), F29_K(th[1], f1) : "vcc"); \
    F29_STEP_LO(r[2]) \
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10) F29_M(11, 12) F29_M(13, 14) F29_M(15, 16) F29_M(17, 18) F29_M(19, 20) : "+v"(acc) : PAIR(a[0], b[3]), PAIR(a[1], b[2]), PAIR(a[2], b[1]), PAIR(a[3], b[0]), PAIR(a[5], c[3]), PAIR(a[6], c[2]), PAIR(a[7], c[1]), PAIR(a[8], c[0]), F29_K(th[3], f0), F29_K(th[2], f1) : "vcc"); \
    F29_NEXT(r[3]) \
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10) F29_M(11, 12) F29_M(13, 14) F29_M(15, 16) F29_M(17, 18) F29_M(19, 20) F29_M(21, 22) : "+v"(acc) : PAIR(a[0], b[4]), PAIR(a[1], b[3]), PAIR(a[2], b[2]), PAIR(a[3], b[1]), PAIR(a[4], b[0]), PAIR(a[5], c[4]), PAIR(a[6], c[3]), PAIR(a[7], c[2]), PAIR(a[8], c[1]), F29_K(th[4], f0), F29_K(th[3], f1) : "vcc"); \
    F29_NEXT(r[4]) \
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10) F29_M(11, 12) F29_M(13, 14) F29_M(15, 16) F29_M(17, 18) F29_M(19, 20) : "+v"(acc) : PAIR(a[0], b[5]), PAIR(a[1], b[4]), PAIR(a[2], b[3]), PAIR(a[3], b[2]), PAIR(a[4], b[1]), PAIR(a[5], c[5]), PAIR(a[6], c[4]), PAIR(a[7], c[3]), PAIR(a[8], c[2]), F29_K(th[4], f1) : "vcc"); \
    F29_NEXT(r[5]) \
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10) F29_M(11, 12) F29_M(13, 14) F29_M(15, 16) F29_M(17, 18) : "+v"(acc) : PAIR(a[0], b[6]), PAIR(a[1], b[5]), PAIR(a[2], b[4]), PAIR(a[3], b[3]), PAIR(a[4], b[2]), PAIR(a[5], c[6]), PAIR(a[6], c[5]), PAIR(a[7], c[4]), PAIR(a[8], c[3]) : "vcc"); \
    F29_NEXT(r[6]) \
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10) F29_M(11, 12) F29_M(13, 14) F29_M(15, 16) F29_M(17, 18) : "+v"(acc) : PAIR(a[0], b[7]), PAIR(a[1], b[6]), PAIR(a[2], b[5]), PAIR(a[3], b[4]), PAIR(a[4], b[3]), PAIR(a[5], c[7]), PAIR(a[6], c[6]), PAIR(a[7], c[5]), PAIR(a[8], c[4]) : "vcc"); \
    F29_NEXT(r[7]) \
    asm(F29_M(1, 2) F29_M(3, 4) F29_M(5, 6) F29_M(7, 8) F29_M(9, 10) F29_M(11, 12) F29_M(13, 14) F29_M(15, 16) F29_M(17, 18) : "+v"(acc) : PAIR(a[0], b[8]), PAIR(a[1], b[7]), PAIR(a[2], b[6]), PAIR(a[3], b[5]), PAIR(a[4], b[4]), PAIR(a[5], c[8]), PAIR(a[6], c[7]), PAIR(a[7], c[6]), PAIR(a[8], c[5]) : "vcc"); \
    F29_NEXT(r[8]) \
    uint64_t top = (acc << 5) | (uint64_t)(r[8] >> 24); \
    r[8] &= 0x00FFFFFFu; \
    uint32_t u0 = (uint32_t)top & F29_MASK, u1 = (uint32_t)(top >> 29); \
    F29 o; \
    acc = mad64(u0, F29_C0, (uint64_t)r[0]); \
    o.l[0] = (uint32_t)acc & F29_MASK; \
    acc >>= 29; \
    acc = mad64(u0, F29_C1, acc + r[1]); \
    acc = mad64(u1, F29_C0, acc); \
    o.l[1] = (uint32_t)acc & F29_MASK; \
    acc >>= 29; \
    acc = mad64(u1, F29_C1, acc + r[2]); \
    o.l[2] = (uint32_t)acc & F29_MASK; \
    o.l[3] = r[3] + (uint32_t)(acc >> 29); \
    _Pragma("unroll") \
    for (int i = 4; i < 9; i++) o.l[i] = r[i]; \
    return o;

template <int WIDE = 2>
__device__ __forceinline__ F29 f29_mul_uniform2(const F29& x, const uint32_t (&b)[9], const uint32_t (&c)[9]) {
    F29_MUL_SPLIT_BODY(F29_PS)
}

// (the F29_ macros stay defined: kernbench.hip builds the split body with per-lane operands)

// Lazy -> tight without a multiplication: one carry pass that also folds what lies above bit 256 (taken
// from limb 8 before the carries below it arrive): e * 2^256 == e * (2808 * 2^29 - 1), e < 2^8, entered as
// 2^29 - e at limb 0 and 2808 e - 1 at limb 1. Input limbs are below 3 * 2^30. The value ends below
// 2^256 + 2^236 (limbs 0..7 below 2^29, limb 8 below 2^24 + 8).
__device__ __forceinline__ F29 f29_tighten(const F29& x) {
    F29 v;
    const uint32_t e = x.l[8] >> 24;
    uint32_t t = x.l[0] + (0x20000000u - e);
    v.l[0] = t & F29_MASK;
    uint32_t cc = t >> 29;  // at least 1
    t = x.l[1] + 2808u * e + cc - 1u;
    v.l[1] = t & F29_MASK;
    cc = t >> 29;
#pragma unroll
    for (int i = 2; i < 8; i++) {
        t = x.l[i] + cc;
        v.l[i] = t & F29_MASK;
        cc = t >> 29;
    }
    v.l[8] = (x.l[8] & 0x00FFFFFFu) + cc;
    return v;
}

// Any lazy value -> canonical residue as 8 x 32-bit limbs; `tight` receives the same value as tight limbs (what
// f29_tighten gives: below 2^256 + 2^236, not necessarily below p), for callers that go on computing with it.
__device__ __forceinline__ Fe f29_to_fe_keep(const F29& x, F29& tight);
__device__ __forceinline__ Fe f29_to_fe(const F29& x) {
    F29 unused;
    return f29_to_fe_keep(x, unused);
}
__device__ __forceinline__ Fe f29_to_fe_keep(const F29& x, F29& tight) {
    // it may still be at or above 2^256, which the subtraction of p below covers
    F29 v = f29_tighten(x);
    tight = v;
    // canonical: with delta = 2^256 - p, value >= p exactly when value + delta reaches 2^256, and then
    // value - p = value + delta - 2^256 (value + delta < 2^257, so bit 256 is the whole test)
    F29 u;
    {
        uint32_t t = v.l[0] + F29_C0;
        u.l[0] = t & F29_MASK;
        uint32_t cc = t >> 29;
        t = v.l[1] + F29_C1 + cc;
        u.l[1] = t & F29_MASK;
        cc = t >> 29;
#pragma unroll
        for (int i = 2; i < 8; i++) {
            t = v.l[i] + cc;
            u.l[i] = t & F29_MASK;
            cc = t >> 29;
        }
        u.l[8] = v.l[8] + cc;
    }
    const uint32_t take = 0u - (u.l[8] >> 24);  // all ones: value >= p, use u minus 2^256
    u.l[8] &= 0x00FFFFFFu;
    uint32_t l[9];
#pragma unroll
    for (int i = 0; i < 9; i++) l[i] = sel32(take, u.l[i], v.l[i]);
    Fe r;
    r.l[0] = l[0] | (l[1] << 29);
    r.l[1] = (l[1] >> 3) | (l[2] << 26);
    r.l[2] = (l[2] >> 6) | (l[3] << 23);
    r.l[3] = (l[3] >> 9) | (l[4] << 20);
    r.l[4] = (l[4] >> 12) | (l[5] << 17);
    r.l[5] = (l[5] >> 15) | (l[6] << 14);
    r.l[6] = (l[6] >> 18) | (l[7] << 11);
    r.l[7] = (l[7] >> 21) | (l[8] << 8);
    return r;
}

}  // namespace starkhip
