// 256-bit prime-field arithmetic for gfx950, p = 2^256 - 351*2^32 + 1.
//
// This is the arithmetic the reference gets from num-bigint (`+ - * %` on BigInt and
// utils::negative_to_positive, reference src/utils.rs:25-42, used by src/fft.rs:28,53-57,84):
// every result here is the least non-negative residue, so outputs are bit-identical.
//
// Representation: 8 little-endian 32-bit limbs held in VGPRs, always canonical (< p) between
// operations.  The 512-bit product is built column by column with v_mad_u64_u32 into a
// three-word accumulator (the third word collects the carry-outs through v_addc_co_u32), and
// reduced with the prime's structure: 2^256 == 351*2^32 - 1 (mod p), so the high half folds
// back with 8 small multiplies instead of a 64-multiply Montgomery pass.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace starkhip {

struct Fe {
    uint32_t l[8];
};

// 2^256 - p = 351*2^32 - 1 = {0xFFFFFFFF, 0x15E}
#define FE_C0 0xFFFFFFFFu
#define FE_C1 0x0000015Eu
// p = {1, 0xFFFFFEA1, 0xFFFFFFFF x 6}
#define FE_P0 0x00000001u
#define FE_P1 0xFFFFFEA1u

__device__ __forceinline__ uint32_t addc(uint32_t a, uint32_t b, uint32_t& carry) {
    uint32_t co;
    uint32_t r = __builtin_addc(a, b, carry, &co);
    carry = co;
    return r;
}

__device__ __forceinline__ uint32_t subb(uint32_t a, uint32_t b, uint32_t& borrow) {
    uint32_t bo;
    uint32_t r = __builtin_subc(a, b, borrow, &bo);
    borrow = bo;
    return r;
}

// Lane-wise select through a 0 / 0xFFFFFFFF mask: one v_bitop3_b32. (v_cndmask_b32 reading VCC
// issues several times slower than this on gfx950 when used back to back.)
__device__ __forceinline__ uint32_t sel32(uint32_t mask, uint32_t if_set, uint32_t if_clear) {
    uint32_t r;  // truth table 0xca: mask ? if_set : if_clear, bit by bit
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xca" : "=v"(r) : "v"(mask), "v"(if_set), "v"(if_clear));
    return r;
}

__device__ __forceinline__ Fe fe_zero() {
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = 0;
    return r;
}

__device__ __forceinline__ Fe fe_one() {
    Fe r = fe_zero();
    r.l[0] = 1;
    return r;
}

// Any 256-bit value -> canonical residue. 2^256 < 2p, so one conditional subtraction of p
// (= addition of 2^256 - p, keeping the low 256 bits) is enough.
__device__ __forceinline__ Fe fe_canon(const Fe& a) {
    Fe u;
    uint32_t c = 0;
    u.l[0] = addc(a.l[0], FE_C0, c);
    u.l[1] = addc(a.l[1], FE_C1, c);
#pragma unroll
    for (int i = 2; i < 8; i++) u.l[i] = addc(a.l[i], 0u, c);
    Fe r;
    const uint32_t mask = 0u - c;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = sel32(mask, u.l[i], a.l[i]);
    return r;
}

// (a + b) mod p for canonical a, b.
__device__ __forceinline__ Fe fe_add(const Fe& a, const Fe& b) {
    Fe s, u;
    uint32_t c1 = 0, c2 = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s.l[i] = addc(a.l[i], b.l[i], c1);
    u.l[0] = addc(s.l[0], FE_C0, c2);
    u.l[1] = addc(s.l[1], FE_C1, c2);
#pragma unroll
    for (int i = 2; i < 8; i++) u.l[i] = addc(s.l[i], 0u, c2);
    const uint32_t mask = 0u - (c1 | c2);  // a+b >= p  <=>  a+b+(2^256-p) >= 2^256
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = sel32(mask, u.l[i], s.l[i]);
    return r;
}

// (a - b) mod p for canonical a, b.
__device__ __forceinline__ Fe fe_sub(const Fe& a, const Fe& b) {
    Fe d;
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d.l[i] = subb(a.l[i], b.l[i], bw);
    // borrowed: add p, i.e. subtract 2^256 - p and drop the 2^256
    uint32_t m = 0u - bw, b2 = 0;
    Fe r;
    r.l[0] = subb(d.l[0], FE_C0 & m, b2);
    r.l[1] = subb(d.l[1], FE_C1 & m, b2);
#pragma unroll
    for (int i = 2; i < 8; i++) r.l[i] = subb(d.l[i], 0u, b2);
    return r;
}

// Column accumulation for the 512-bit product. (c2:acc) is a three-word accumulator; each
// product is one v_mad_u64_u32 whose carry-out is collected by one v_addc_co_u32. A whole
// column is one asm statement so the compiler's post-asm pad is paid once per column.
#define SH_MAC(x, y) "v_mad_u64_u32 %0, vcc, " x ", " y ", %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
#define SH_IN2(a0, b0) "v"(a0), "v"(b0)

__device__ __forceinline__ void col1(uint64_t& acc, uint32_t& c2, uint32_t a0, uint32_t b0) {
    asm(SH_MAC("%2", "%3") : "+v"(acc), "+v"(c2) : SH_IN2(a0, b0) : "vcc");
}
__device__ __forceinline__ void col2(uint64_t& acc, uint32_t& c2, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1) {
    asm(SH_MAC("%2", "%3") SH_MAC("%4", "%5") : "+v"(acc), "+v"(c2) : SH_IN2(a0, b0), SH_IN2(a1, b1) : "vcc");
}
__device__ __forceinline__ void col3(uint64_t& acc, uint32_t& c2, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1,
                                     uint32_t a2, uint32_t b2) {
    asm(SH_MAC("%2", "%3") SH_MAC("%4", "%5") SH_MAC("%6", "%7")
        : "+v"(acc), "+v"(c2)
        : SH_IN2(a0, b0), SH_IN2(a1, b1), SH_IN2(a2, b2)
        : "vcc");
}
__device__ __forceinline__ void col4(uint64_t& acc, uint32_t& c2, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1,
                                     uint32_t a2, uint32_t b2, uint32_t a3, uint32_t b3) {
    asm(SH_MAC("%2", "%3") SH_MAC("%4", "%5") SH_MAC("%6", "%7") SH_MAC("%8", "%9")
        : "+v"(acc), "+v"(c2)
        : SH_IN2(a0, b0), SH_IN2(a1, b1), SH_IN2(a2, b2), SH_IN2(a3, b3)
        : "vcc");
}
__device__ __forceinline__ void col5(uint64_t& acc, uint32_t& c2, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1,
                                     uint32_t a2, uint32_t b2, uint32_t a3, uint32_t b3, uint32_t a4, uint32_t b4) {
    asm(SH_MAC("%2", "%3") SH_MAC("%4", "%5") SH_MAC("%6", "%7") SH_MAC("%8", "%9") SH_MAC("%10", "%11")
        : "+v"(acc), "+v"(c2)
        : SH_IN2(a0, b0), SH_IN2(a1, b1), SH_IN2(a2, b2), SH_IN2(a3, b3), SH_IN2(a4, b4)
        : "vcc");
}
__device__ __forceinline__ void col6(uint64_t& acc, uint32_t& c2, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1,
                                     uint32_t a2, uint32_t b2, uint32_t a3, uint32_t b3, uint32_t a4, uint32_t b4,
                                     uint32_t a5, uint32_t b5) {
    asm(SH_MAC("%2", "%3") SH_MAC("%4", "%5") SH_MAC("%6", "%7") SH_MAC("%8", "%9") SH_MAC("%10", "%11")
            SH_MAC("%12", "%13")
        : "+v"(acc), "+v"(c2)
        : SH_IN2(a0, b0), SH_IN2(a1, b1), SH_IN2(a2, b2), SH_IN2(a3, b3), SH_IN2(a4, b4), SH_IN2(a5, b5)
        : "vcc");
}
__device__ __forceinline__ void col7(uint64_t& acc, uint32_t& c2, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1,
                                     uint32_t a2, uint32_t b2, uint32_t a3, uint32_t b3, uint32_t a4, uint32_t b4,
                                     uint32_t a5, uint32_t b5, uint32_t a6, uint32_t b6) {
    asm(SH_MAC("%2", "%3") SH_MAC("%4", "%5") SH_MAC("%6", "%7") SH_MAC("%8", "%9") SH_MAC("%10", "%11")
            SH_MAC("%12", "%13") SH_MAC("%14", "%15")
        : "+v"(acc), "+v"(c2)
        : SH_IN2(a0, b0), SH_IN2(a1, b1), SH_IN2(a2, b2), SH_IN2(a3, b3), SH_IN2(a4, b4), SH_IN2(a5, b5),
          SH_IN2(a6, b6)
        : "vcc");
}
__device__ __forceinline__ void col8(uint64_t& acc, uint32_t& c2, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1,
                                     uint32_t a2, uint32_t b2, uint32_t a3, uint32_t b3, uint32_t a4, uint32_t b4,
                                     uint32_t a5, uint32_t b5, uint32_t a6, uint32_t b6, uint32_t a7, uint32_t b7) {
    asm(SH_MAC("%2", "%3") SH_MAC("%4", "%5") SH_MAC("%6", "%7") SH_MAC("%8", "%9") SH_MAC("%10", "%11")
            SH_MAC("%12", "%13") SH_MAC("%14", "%15") SH_MAC("%16", "%17")
        : "+v"(acc), "+v"(c2)
        : SH_IN2(a0, b0), SH_IN2(a1, b1), SH_IN2(a2, b2), SH_IN2(a3, b3), SH_IN2(a4, b4), SH_IN2(a5, b5),
          SH_IN2(a6, b6), SH_IN2(a7, b7)
        : "vcc");
}

// 16-limb product of two 8-limb numbers (product scanning, column by column).
__device__ __forceinline__ void mul_wide(uint32_t t[16], const Fe& x, const Fe& y) {
    const uint32_t* a = x.l;
    const uint32_t* b = y.l;
    uint64_t acc = (uint64_t)a[0] * b[0];  // column 0: a single product
    uint32_t c2 = 0;
#define SH_NEXT(k)                                        \
    t[k] = (uint32_t)acc;                                 \
    acc = (acc >> 32) | ((uint64_t)c2 << 32);             \
    c2 = 0;
    SH_NEXT(0)
    col2(acc, c2, a[0], b[1], a[1], b[0]);
    SH_NEXT(1)
    col3(acc, c2, a[0], b[2], a[1], b[1], a[2], b[0]);
    SH_NEXT(2)
    col4(acc, c2, a[0], b[3], a[1], b[2], a[2], b[1], a[3], b[0]);
    SH_NEXT(3)
    col5(acc, c2, a[0], b[4], a[1], b[3], a[2], b[2], a[3], b[1], a[4], b[0]);
    SH_NEXT(4)
    col6(acc, c2, a[0], b[5], a[1], b[4], a[2], b[3], a[3], b[2], a[4], b[1], a[5], b[0]);
    SH_NEXT(5)
    col7(acc, c2, a[0], b[6], a[1], b[5], a[2], b[4], a[3], b[3], a[4], b[2], a[5], b[1], a[6], b[0]);
    SH_NEXT(6)
    col8(acc, c2, a[0], b[7], a[1], b[6], a[2], b[5], a[3], b[4], a[4], b[3], a[5], b[2], a[6], b[1], a[7], b[0]);
    SH_NEXT(7)
    col7(acc, c2, a[1], b[7], a[2], b[6], a[3], b[5], a[4], b[4], a[5], b[3], a[6], b[2], a[7], b[1]);
    SH_NEXT(8)
    col6(acc, c2, a[2], b[7], a[3], b[6], a[4], b[5], a[5], b[4], a[6], b[3], a[7], b[2]);
    SH_NEXT(9)
    col5(acc, c2, a[3], b[7], a[4], b[6], a[5], b[5], a[6], b[4], a[7], b[3]);
    SH_NEXT(10)
    col4(acc, c2, a[4], b[7], a[5], b[6], a[6], b[5], a[7], b[4]);
    SH_NEXT(11)
    col3(acc, c2, a[5], b[7], a[6], b[6], a[7], b[5]);
    SH_NEXT(12)
    col2(acc, c2, a[6], b[7], a[7], b[6]);
    SH_NEXT(13)
    acc = (uint64_t)a[7] * b[7] + acc;  // last column: the total fits 512 bits, no carry out
    t[14] = (uint32_t)acc;
    t[15] = (uint32_t)(acc >> 32);
#undef SH_NEXT
}

// Reduce a 512-bit value t (16 limbs) to the canonical residue mod p.
__device__ __forceinline__ Fe reduce_wide(const uint32_t t[16]) {
    // fold 1: S = L + H*(351*2^32 - 1) = L - H + ((H*351) << 32), a 10-limb value
    uint32_t m[9];
    {
        uint64_t acc = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            acc = (uint64_t)t[8 + i] * 351u + (acc >> 32);
            m[i] = (uint32_t)acc;
        }
        m[8] = (uint32_t)(acc >> 32);
    }
    uint32_t s[10];
    {
        uint32_t c = 0;
        s[0] = t[0];
#pragma unroll
        for (int i = 1; i < 8; i++) s[i] = addc(t[i], m[i - 1], c);
        s[8] = addc(m[7], 0u, c);
        s[9] = addc(m[8], 0u, c);
        uint32_t bw = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) s[i] = subb(s[i], t[8 + i], bw);
        s[8] = subb(s[8], 0u, bw);
        s[9] = subb(s[9], 0u, bw);
    }
    // fold 2: hh = s[9]:s[8] < 2^42; q = hh*(351*2^32 - 1) = ((hh*351) << 32) - hh < 2^83
    uint32_t q0, q1, q2;
    {
        uint64_t n = (uint64_t)s[8] * 351u + ((uint64_t)(s[9] * 351u) << 32);
        uint32_t bw = 0;
        q0 = subb(0u, s[8], bw);
        q1 = subb((uint32_t)n, s[9], bw);
        q2 = subb((uint32_t)(n >> 32), 0u, bw);
    }
    Fe r;
    uint32_t c = 0;
    r.l[0] = addc(s[0], q0, c);
    r.l[1] = addc(s[1], q1, c);
    r.l[2] = addc(s[2], q2, c);
#pragma unroll
    for (int i = 3; i < 8; i++) r.l[i] = addc(s[i], 0u, c);
    // fold 3: a carry out means the remainder is below 2^83; adding 2^256 - p cannot carry
    // past limb 2.
    {
        uint32_t mk = 0u - c, c3 = 0;
        r.l[0] = addc(r.l[0], FE_C0 & mk, c3);
        r.l[1] = addc(r.l[1], FE_C1 & mk, c3);
        r.l[2] = addc(r.l[2], 0u, c3);
    }
    return fe_canon(r);
}

__device__ __forceinline__ Fe fe_mul(const Fe& a, const Fe& b) {
    uint32_t t[16];
    mul_wide(t, a, b);
    return reduce_wide(t);
}

// ---- memory helpers: a field element is 32 contiguous bytes = two 16-byte accesses ----
__device__ __forceinline__ Fe fe_load(const uint32_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 lo = q[0], hi = q[1];
    Fe r;
    r.l[0] = lo.x; r.l[1] = lo.y; r.l[2] = lo.z; r.l[3] = lo.w;
    r.l[4] = hi.x; r.l[5] = hi.y; r.l[6] = hi.z; r.l[7] = hi.w;
    return r;
}

__device__ __forceinline__ void fe_store(uint32_t* p, const Fe& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(a.l[0], a.l[1], a.l[2], a.l[3]);
    q[1] = make_uint4(a.l[4], a.l[5], a.l[6], a.l[7]);
}

}  // namespace starkhip
