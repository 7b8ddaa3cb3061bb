// Scalar field checks of the STARK protocol layer, one lane per sampled position:
//   * the FRI low-degree consistency check of verify_low_degree_proof (reference src/lib.rs:73-86):
//     the cubic through the four row values at x1 * {1, q, q^2, q^3} (q a primitive 4th root of
//     unity) evaluated at special_x must equal the column value;
//   * the transition, boundary and linear-combination constraints of verify_mimc_proof
//     (reference src/lib.rs:163-192).
// The reference reaches the same field elements through generic Lagrange interpolation
// (multi_interp_4, src/utils.rs:196-244) and a modular division (divmod, src/utils.rs:122); here the
// interpolation over a coset of the 4th roots of unity is written as a size-4 inverse DFT
// (c_k * x1^k = 1/4 * sum_j row_j * q^(-jk)), x1^-1 is a power of the root of unity, and the division
// by (x - last_step_position), which is never zero for a sampled position, is cleared by
// multiplying the identity through. Equalities are decided on canonical residues.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "field.cuh"
#include "field29.cuh"
#include "internal.hpp"

namespace starkhip {

namespace {

__device__ __forceinline__ F29 f29_one() {
    F29 r;
    r.l[0] = 1;
#pragma unroll
    for (int i = 1; i < 9; i++) r.l[i] = 0;
    return r;
}

// canonical residue back in limb form: tight, and equal values have equal limbs
__device__ __forceinline__ F29 norm(const F29& x) { return f29_from_fe(f29_to_fe(x)); }

// 32 big-endian bytes (BigInt::from_bytes_be) -> tight limbs of the number as it stands (it may be p or more: the
// arithmetic below is mod p whatever representative it starts from, every equality is decided on canonical residues,
// and the one place where the reference's unreduced comparison shows, be_is_canonical, looks at the bytes itself)
__device__ __forceinline__ F29 load_be(const uint8_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1];
    Fe x;
    x.l[7] = __builtin_bswap32(a.x);
    x.l[6] = __builtin_bswap32(a.y);
    x.l[5] = __builtin_bswap32(a.z);
    x.l[4] = __builtin_bswap32(a.w);
    x.l[3] = __builtin_bswap32(b.x);
    x.l[2] = __builtin_bswap32(b.y);
    x.l[1] = __builtin_bswap32(b.z);
    x.l[0] = __builtin_bswap32(b.w);
    return f29_from_fe(x);
}

// Do these 32 big-endian bytes spell a number below p? (BigInt::from_bytes_be does not reduce.)
__device__ __forceinline__ bool be_is_canonical(const uint8_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1];
    Fe x;
    x.l[7] = __builtin_bswap32(a.x);
    x.l[6] = __builtin_bswap32(a.y);
    x.l[5] = __builtin_bswap32(a.z);
    x.l[4] = __builtin_bswap32(a.w);
    x.l[3] = __builtin_bswap32(b.x);
    x.l[2] = __builtin_bswap32(b.y);
    x.l[1] = __builtin_bswap32(b.z);
    x.l[0] = __builtin_bswap32(b.w);
    const Fe c = f29_to_fe(f29_from_fe(x));
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d |= c.l[i] ^ x.l[i];
    return d == 0;
}

__device__ __forceinline__ F29 load_limbs(const uint32_t* l) {
    Fe x;
#pragma unroll
    for (int i = 0; i < 8; i++) x.l[i] = l[i];
    return f29_from_fe(x);  // parameters are canonical
}

__device__ __forceinline__ F29 mul_t(const F29& a, const F29& b) { return f29_mul(a, b); }  // a lazy, b tight -> tight
__device__ __forceinline__ F29 add_t(const F29& a, const F29& b) { return norm(f29_add(a, b)); }
__device__ __forceinline__ F29 sub_t(const F29& a, const F29& b) { return norm(f29_sub(a, b)); }  // b tight

// base^e for a tight base, e < 2^bits
__device__ __noinline__ F29 pow_t(const F29& base, uint32_t e, int bits) {
    F29 acc = f29_one();
#pragma unroll 1
    for (int i = bits - 1; i >= 0; i--) {
        acc = mul_t(acc, acc);
        if ((e >> i) & 1u) acc = mul_t(acc, base);
    }
    return acc;
}

__device__ __forceinline__ bool is_zero_t(const F29& x) {
    Fe v = f29_to_fe(x);
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= v.l[i];
    return o == 0;
}

__device__ __forceinline__ void store_canonical(uint32_t* o, const F29& x) {
    const Fe v = f29_to_fe(x);
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = v.l[i];
}

// k_of_x = constants_mini_polynomial(xk) (src/lib.rs:178), Horner from the top coefficient
__device__ __forceinline__ F29 mini_poly_at(const uint32_t* __restrict__ mini_poly, uint32_t n_coeffs, const F29& xk) {
    F29 kx = load_limbs(mini_poly + 8 * (n_coeffs - 1));
#pragma unroll 1
    for (int c = (int)n_coeffs - 2; c >= 0; c--) kx = add_t(mul_t(kx, xk), load_limbs(mini_poly + 8 * c));
    return kx;
}

}  // namespace

// The tables of CheckParams, by the arithmetic a lane would otherwise run itself: lane i < precision writes g2^i, lane
// j < precision / skips2 also the mini polynomial at g2^(skips2 * j).
__global__ void __launch_bounds__(256) check_tables_kernel(CheckParams P, const uint32_t* __restrict__ mini_poly,
                                                           uint32_t* __restrict__ pw, uint32_t* __restrict__ kext) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.precision) return;
    const F29 g2 = load_limbs(P.g2);
    store_canonical(pw + 8 * (size_t)i, pow_t(g2, i, 32));
    if (i < (P.precision >> P.log_skips2))
        store_canonical(kext + 8 * (size_t)i, mini_poly_at(mini_poly, P.n_coeffs, norm(pow_t(g2, i << P.log_skips2, 32))));
}

hipError_t launch_check_tables(const CheckParams& P, const uint32_t* mini_poly, uint32_t* pw, uint32_t* kext, hipStream_t stream) {
    if (P.precision == 0 || (P.precision & (P.precision - 1))) return hipErrorInvalidValue;
    hipLaunchKernelGGL(check_tables_kernel, dim3((P.precision + 255) / 256), dim3(256), 0, stream, P, mini_poly, pw, kext);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) fri_check_kernel(const FriLane* __restrict__ lanes, size_t n, CheckParams P,
                                                        uint8_t* __restrict__ status) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const FriLane& L = lanes[i];
    const F29 r0 = load_be(L.rows[0]), r1 = load_be(L.rows[1]), r2 = load_be(L.rows[2]), r3 = load_be(L.rows[3]);
    const F29 col = load_be(L.col), special = load_be(L.special);
    const uint32_t layer = L.layer, y = L.y;
    // root of unity of this layer: g2^(4^layer), of order precision / 4^layer (src/lib.rs:88-90); x1 = rou^y (:74), so
    // 1 / x1 = rou^(deg - y) = g2^((deg - y) * 4^layer): from the table when there is one and the exponent lies in it
    const uint32_t deg = 2 * layer < 32 ? P.precision >> (2 * layer) : 0u;
    F29 x1inv;
    if (P.pw && deg > 0) {
        x1inv = load_limbs(P.pw + 8 * (size_t)((((deg - y) & (deg - 1)) << (2 * layer)) & (P.precision - 1)));
    } else {
        F29 rou = load_limbs(P.g2);
#pragma unroll 1
        for (uint32_t k = 0; k < 2 * layer; k++) rou = mul_t(rou, rou);
        x1inv = pow_t(rou, (deg - y) & (deg - 1), 16);
    }
    const F29 t = mul_t(special, x1inv);
    const F29 q = load_limbs(P.q4);
    const F29 A = add_t(r0, r2), B = sub_t(r0, r2), C = add_t(r1, r3), D = mul_t(sub_t(r1, r3), q);
    const F29 k0 = add_t(A, C), k2 = sub_t(A, C), k1 = sub_t(B, D), k3 = add_t(B, D);
    // 4 * P(special_x) = k0 + t*(k1 + t*(k2 + t*k3))
    F29 acc = add_t(mul_t(k3, t), k2);
    acc = add_t(mul_t(acc, t), k1);
    acc = add_t(mul_t(acc, t), k0);
    const F29 four_col = add_t(add_t(col, col), add_t(col, col));
    // src/lib.rs:85 compares the reduced evaluation with the column value as BigInt::from_bytes_be gives it,
    // unreduced: a column value encoded at or above p never compares equal there
    status[i] = (is_zero_t(sub_t(acc, four_col)) && be_is_canonical(L.col)) ? 0 : 5;
}

__global__ void __launch_bounds__(256) constraint_check_kernel(const ConstraintLane* __restrict__ lanes, size_t n,
                                                               CheckParams P, const uint8_t* __restrict__ kk,
                                                               const uint32_t* __restrict__ mini_poly,
                                                               uint8_t* __restrict__ status) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ConstraintLane& L = lanes[i];
    const F29 p = load_be(L.b1), d = load_be(L.b1 + 32), b = load_be(L.b1 + 64);
    const F29 pg = load_be(L.pg), l = load_be(L.l);
    const F29 one = f29_one();
    // x = g2^pos (:164), xs = x^num_steps, kx = constants_mini_polynomial(x^skips2) (:178)
    F29 x, xs, kx;
    if (P.pw && P.kext && P.precision == 65536u) {  // (the square-and-multiply below reads 16 bits of pos)
        const uint32_t pos = L.pos & (P.precision - 1);
        x = load_limbs(P.pw + 8 * (size_t)pos);
        xs = load_limbs(P.pw + 8 * (size_t)((pos << P.log_steps) & (P.precision - 1)));
        kx = load_limbs(P.kext + 8 * (size_t)(pos & ((P.precision >> P.log_skips2) - 1)));
    } else {
        x = pow_t(load_limbs(P.g2), L.pos, 16);
        // x^skips2 on the way to x^num_steps (both powers of two)
        F29 xk = x;
        xs = x;
#pragma unroll 1
        for (uint32_t k = 0; k < P.log_steps; k++) {
            xs = mul_t(xs, xs);
            if (k + 1 == P.log_skips2) xk = xs;
        }
        if (P.log_skips2 == 0) xk = x;
        kx = mini_poly_at(mini_poly, P.n_coeffs, xk);
    }
    const F29 last = load_limbs(P.last_step);
    const F29 x_last = sub_t(x, last), x_1 = sub_t(x, one);
    uint8_t st = 0;
    // transition (:181): p(g1 x) - p^3 - k - z*d == 0 with z = (x^steps - 1) / (x - last), cleared of the division
    {
        const F29 p3 = mul_t(mul_t(p, p), p);
        const F29 lhs = mul_t(sub_t(sub_t(pg, p3), kx), x_last);
        const F29 rhs = mul_t(sub_t(xs, one), d);
        if (!is_zero_t(sub_t(lhs, rhs))) st = 6;
    }
    // boundary (:184-187): p - b * (x-1)(x-last) - (inp + slope*(x-1)) == 0
    if (!st) {
        const F29 zero2 = mul_t(x_1, x_last);
        const uint32_t* io = reinterpret_cast<const uint32_t*>(kk + CHECK_PROOF_BYTES * (size_t)L.proof + 128);
        const F29 interp = add_t(load_limbs(io), mul_t(x_1, load_limbs(io + 8)));
        if (!is_zero_t(sub_t(sub_t(p, mul_t(zero2, b)), interp))) st = 7;
    }
    // linear combination (:190)
    if (!st) {
        const uint8_t* k = kk + CHECK_PROOF_BYTES * (size_t)L.proof;
        const F29 k1 = load_be(k), k2 = load_be(k + 32), k3 = load_be(k + 64), k4 = load_be(k + 96);
        F29 acc = sub_t(l, d);
        acc = sub_t(acc, mul_t(k1, p));
        acc = sub_t(acc, mul_t(mul_t(k2, p), xs));
        acc = sub_t(acc, mul_t(k3, b));
        acc = sub_t(acc, mul_t(mul_t(k4, b), xs));
        if (!is_zero_t(acc)) st = 8;
    }
    status[i] = st;
}

hipError_t launch_fri_check(const FriLane* lanes, size_t n, const CheckParams& P, uint8_t* status, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(fri_check_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, lanes, n, P, status);
    return hipGetLastError();
}

hipError_t launch_constraint_check(const ConstraintLane* lanes, size_t n, const CheckParams& P, const uint8_t* kk,
                                   const uint32_t* mini_poly, uint8_t* status, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(constraint_check_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, lanes, n, P, kk,
                       mini_poly, status);
    return hipGetLastError();
}

}  // namespace starkhip
