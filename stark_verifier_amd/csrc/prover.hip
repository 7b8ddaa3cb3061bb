// MiMC STARK proofs made on the device, many at a time: the counterpart of the external program the
// reference's proofs come from (reference README.md:9). Every value is what the reference's verifier
// recomputes (src/lib.rs:31-197), and the bytes are the wire format its parser reads
// (src/deserializer.rs:16-140), so verify_mimc_proof accepts the output; with input 3, 8192 steps and
// the constants of src/main.rs:209-212 the output is the reference's proof.bin, byte for byte.
//
// Per chunk of B proofs, all on the context's stream:
//   trace        one lane per proof walks x -> x^3 + k_i (the only sequential part)
//   P            inverse transform of the trace (order `steps`), zero-extended, forward transform (order 8*steps)
//   D, B         one lane per (proof, point): the quotients checked at src/lib.rs:181,187, written with P as
//                96-byte leaves in tree order (ProofBranch::permute_4_index, src/merkle_tree.rs:112-116)
//   trees        BLAKE2s pair hashes level by level; roots also land in a compact per-tree array
//   L            d + p*(k1 + k2*x^steps) + b*(k3 + k4*x^steps) with k_j = BLAKE2s(root || j) (src/lib.rs:131-146,190)
//   FRI          per layer: column[i] = the cubic through the four row values at x_i*{1,q,q^2,q^3} evaluated at the
//                layer's root read as a number (src/lib.rs:53-86), as a size-4 inverse DFT; tree; 40 sampled rows
//   indices      get_pseudorandom_indices (src/utils.rs:53-94) on the device, one lane per seed
//   serialize    one lane per 32-byte field of the output: a table built once per parameter set says which node of
//                which tree (or which sampled index) goes where
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <memory>
#include <new>
#include <vector>

#include "../../include/starkhip.h"
#include "blake2s.cuh"
#include "fieldio.cuh"
#include "host_field.hpp"
#include "internal.hpp"

namespace starkhip {
namespace prover {

using hostfe::U256;
using namespace fio;

constexpr uint32_t FRI_SAMPLES = 40;   // src/lib.rs:56
constexpr uint32_t SPOT_CHECKS = 80;   // src/lib.rs:148
constexpr uint32_t EXT = 8;            // EXTENSION_FACTOR, src/lib.rs:28
constexpr int MAX_TREES = 12;

// One 32-byte field (or one constant word) of the serialized proof.
struct Item {
    uint32_t out_off;   // where it goes inside the proof
    uint32_t add;       // added to the sampled index (or the index itself when set == 0xFF)
    uint32_t pre_word;  // non-zero: also written as a u32 just before out_off (value size / witness bytes)
    uint8_t src;        // 0..nl: tree of that FRI level; SRC_LEAF: 96-byte trace leaves; SRC_TRACE: trace tree nodes; SRC_WORD
    uint8_t level;      // bits 0..6: tree level above the leaves (or 32-byte part of a leaf); bit 7: take the sibling
    uint8_t set;        // which sample array: FRI layer k, or nl for the spot-check positions; 0xFF: none
    uint8_t sidx;       // entry of that array
};
constexpr uint8_t SRC_LEAF = 0xF0, SRC_TRACE = 0xF1, SRC_WORD = 0xFF;

struct Dev {  // by-value kernel argument: the buffers of one chunk
    uint32_t log_n, log_steps, nl, B;
    uint32_t ext_mask;            // (precision / skips2) - 1
    const uint32_t* xs;           // [N][8] G2^i
    const uint32_t* zfac;         // [N][8] (x - last) / (x^steps - 1), 0 where that divides by zero
    const uint32_t* invz2;        // [N][8] 1 / ((x-1)(x-last)), 0 where that divides by zero
    const uint32_t* kext;         // [N/skips2][8] round-constant polynomial over the extended domain
    uint32_t* P;                  // [B][N][8]
    uint32_t* io;                 // [B][16] input, slope of the boundary interpolant
    uint32_t* coef;               // [B][8][16] k1 + k2*w, k3 + k4*w for the 8 values w of x^steps
    uint8_t* leaves;              // [B][N][96]
    uint8_t* tnodes;              // [B][N][32] trace tree above the leaf pairs: node 1 = root, pair hashes at [N/2, N)
    uint8_t* tree[MAX_TREES];     // tree[k]: [B][2*N/4^k][32], leaves (tree order) in the upper half
    uint8_t* roots[MAX_TREES];    // roots[k]: [B][32]
    uint8_t* troot;               // [B][32]
    uint32_t* samples;            // set k < nl at [40*B*k + 40*proof], positions at [40*B*nl + 80*proof]
    uint32_t inv4[8], q4[8], inv_last_m1[8];
};

__device__ __forceinline__ uint32_t perm4(uint32_t i, uint32_t log_size) {  // src/merkle_tree.rs:112-116
    return (i >> (log_size - 2)) + 4u * (i & ((1u << (log_size - 2)) - 1u));
}

__device__ __forceinline__ F29 limbs_arg(const uint32_t (&l)[8]) {
    Fe x;
#pragma unroll
    for (int i = 0; i < 8; i++) x.l[i] = l[i];
    return f29_from_fe(x);
}

// ---- trace: lane = proof ----
__global__ void __launch_bounds__(64)
trace_kernel(const uint32_t* __restrict__ inputs, uint32_t n, uint32_t steps, const uint32_t* __restrict__ consts,
             uint32_t cmask, uint32_t* __restrict__ trace, uint32_t* __restrict__ outputs, uint32_t* __restrict__ io,
             Dev D) {
    const uint32_t pr = blockIdx.x * blockDim.x + threadIdx.x;
    if (pr >= n) return;
    // The value is carried both ways: canonical 8 x 32-bit for the store, nine tight limbs for the arithmetic (the limbs
    // the canonicalisation passes through anyway). This loop is a wavefront alone on its SIMD, which is served once in
    // four cycles whatever it issues, so its time is its instruction count: hence the squaring routine and no
    // re-conversion of the stored value.
    Fe xe = f29_to_fe(load_limbs(inputs + 8 * (size_t)pr));
    F29 x = f29_from_fe(xe);
    const F29 inp = x;
    uint32_t* row = trace + 8 * (size_t)pr * steps;
#pragma unroll 1
    for (uint32_t i = 0; i + 1 < steps; i++) {
        fe_store(row + 8 * (size_t)i, xe);
        const F29 c = load_limbs(consts + 8 * (i & cmask));
        F29 xt;
        xe = f29_to_fe_keep(f29_add(mul_t(f29_sqr(x), x), c), xt);
        x = xt;
    }
    fe_store(row + 8 * (size_t)(steps - 1), xe);
    fe_store(outputs + 8 * (size_t)pr, xe);
    store_limbs(io + 16 * (size_t)pr, inp);
    store_limbs(io + 16 * (size_t)pr + 8, mul_t(f29_sub(x, inp), limbs_arg(D.inv_last_m1)));
}

// ---- utils::mimc (reference src/utils.rs:8-16) for many inputs: steps - 1 rounds of x -> x^3 + k[i mod len]; lane = input ----
__global__ void __launch_bounds__(64)
mimc_kernel(const uint32_t* __restrict__ inputs, size_t n, uint64_t steps, const uint32_t* __restrict__ consts,
            uint32_t n_consts, uint32_t* __restrict__ outputs) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    F29 x = norm(load_limbs(inputs + 8 * t));
    uint32_t ci = 0;
#pragma unroll 1
    for (uint64_t i = 0; i + 1 < steps; i++) {
        const F29 c = norm(load_limbs(consts + 8 * (size_t)ci));
        x = norm(f29_add(mul_t(mul_t(x, x), x), c));
        ci = ci + 1 == n_consts ? 0 : ci + 1;
    }
    store_limbs(outputs + 8 * t, x);
}

// ---- D, B and the 96-byte leaves: lane = (proof, point) ----
__global__ void __launch_bounds__(256) columns_kernel(Dev D) {
    const uint32_t pr = blockIdx.y, N = 1u << D.log_n;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint32_t* Pp = D.P + 8 * ((size_t)pr << D.log_n);
    const F29 p = load_limbs(Pp + 8 * (size_t)i), pg = load_limbs(Pp + 8 * (size_t)((i + EXT) & (N - 1)));
    const F29 k = load_limbs(D.kext + 8 * (size_t)(i & D.ext_mask));
    const F29 x = load_limbs(D.xs + 8 * (size_t)i);
    const F29 inp = load_limbs(D.io + 16 * (size_t)pr), slope = load_limbs(D.io + 16 * (size_t)pr + 8);
    // d = (p(g1 x) - p^3 - k) * (x - last) / (x^steps - 1)
    const F29 p3 = mul_t(mul_t(p, p), p);
    const F29 d = mul_t(f29_sub(f29_sub(pg, p3), k), load_limbs(D.zfac + 8 * (size_t)i));
    // b = (p - interpolant) / ((x - 1)(x - last)), interpolant = inp + slope * (x - 1) (a sum of two tight values)
    const F29 interp = f29_add(inp, mul_t(f29_sub(x, one()), slope));
    const F29 b = mul_t(f29_sub2(p, interp), load_limbs(D.invz2 + 8 * (size_t)i));
    uint8_t* leaf = D.leaves + 96 * (((size_t)pr << D.log_n) + perm4(i, D.log_n));
    store_be(leaf, p);
    store_be(leaf + 32, d);
    store_be(leaf + 64, b);
}

// ---- D, B, the 96-byte leaves and the first level of the trace tree in one go: lane = (proof, pair of leaves) ----
// Leaves 2j and 2j + 1 of the tree order are points r + q N/4 and r + (q + 1) N/4 with r = j >> 1, q = 2 (j & 1)
// (perm4 above, inverted). The lane computes both points' (P, D, B), stores the two leaves (192 contiguous bytes) and
// hashes them from registers: the 192-byte message is six 32-byte big-endian values, word w of a value being the
// byte-swapped limb 7 - w, so the three message blocks are (P0, D0), (B0, P1), (D1, B1) and each is compressed as soon as
// its second value exists. Against columns_kernel + leaf_pairs_kernel this saves reading the leaves back (6 of the
// 14 MiB the two kernels move per proof) and one launch; STARKHIP_PROVER_FUSE=0 keeps the two kernels.
__device__ __forceinline__ void be_words(uint32_t* m, const Fe& x) {
#pragma unroll
    for (int w = 0; w < 8; w++) m[w] = __builtin_bswap32(x.l[7 - w]);
}

__global__ void __launch_bounds__(256) columns_pairs_kernel(Dev D) {
    const uint32_t pr = blockIdx.y, N = 1u << D.log_n, half = N >> 1, quarter = N >> 2;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= half) return;
    const uint32_t* Pp = D.P + 8 * ((size_t)pr << D.log_n);
    const F29 inp = load_limbs(D.io + 16 * (size_t)pr), slope = load_limbs(D.io + 16 * (size_t)pr + 8);
    uint4* leaf = reinterpret_cast<uint4*>(D.leaves + 96 * (((size_t)pr << D.log_n) + 2 * (size_t)j));
    uint32_t h[8], m[16];
    b2s_init(h);
    Fe b0;
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const uint32_t i = (j >> 1) + (2u * (j & 1u) + (uint32_t)s) * quarter;
        const F29 p = load_limbs(Pp + 8 * (size_t)i), pg = load_limbs(Pp + 8 * (size_t)((i + EXT) & (N - 1)));
        const F29 k = load_limbs(D.kext + 8 * (size_t)(i & D.ext_mask));
        const F29 x = load_limbs(D.xs + 8 * (size_t)i);
        const F29 p3 = mul_t(mul_t(p, p), p);
        const F29 d = mul_t(f29_sub(f29_sub(pg, p3), k), load_limbs(D.zfac + 8 * (size_t)i));
        const F29 interp = f29_add(inp, mul_t(f29_sub(x, one()), slope));
        const F29 b = mul_t(f29_sub2(p, interp), load_limbs(D.invz2 + 8 * (size_t)i));
        const Fe pe = f29_to_fe(p), de = f29_to_fe(d), be = f29_to_fe(b);
        if (s == 0) {
            be_words(m, pe);
            be_words(m + 8, de);
            leaf[0] = make_uint4(m[0], m[1], m[2], m[3]);
            leaf[1] = make_uint4(m[4], m[5], m[6], m[7]);
            leaf[2] = make_uint4(m[8], m[9], m[10], m[11]);
            leaf[3] = make_uint4(m[12], m[13], m[14], m[15]);
            b2s_compress(h, m, 64u, false);
            b0 = be;
        } else {
            be_words(m, b0);
            be_words(m + 8, pe);
            leaf[4] = make_uint4(m[0], m[1], m[2], m[3]);
            leaf[5] = make_uint4(m[4], m[5], m[6], m[7]);
            leaf[6] = make_uint4(m[8], m[9], m[10], m[11]);
            leaf[7] = make_uint4(m[12], m[13], m[14], m[15]);
            b2s_compress(h, m, 128u, false);
            be_words(m, de);
            be_words(m + 8, be);
            leaf[8] = make_uint4(m[0], m[1], m[2], m[3]);
            leaf[9] = make_uint4(m[4], m[5], m[6], m[7]);
            leaf[10] = make_uint4(m[8], m[9], m[10], m[11]);
            leaf[11] = make_uint4(m[12], m[13], m[14], m[15]);
            b2s_compress(h, m, 192u, true);
        }
    }
    uint4* dst = reinterpret_cast<uint4*>(D.tnodes + 32 * (((size_t)pr << D.log_n) + half + j));
    dst[0] = make_uint4(h[0], h[1], h[2], h[3]);
    dst[1] = make_uint4(h[4], h[5], h[6], h[7]);
}

// ---- first level of the trace tree: BLAKE2s over two 96-byte leaves; lane = (proof, pair) ----
__global__ void __launch_bounds__(256) leaf_pairs_kernel(Dev D) {
    const uint32_t pr = blockIdx.y, half = 1u << (D.log_n - 1);
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= half) return;
    const uint4* src = reinterpret_cast<const uint4*>(D.leaves + 96 * (((size_t)pr << D.log_n) + 2 * (size_t)j));
    uint32_t h[8];
    b2s_init(h);
#pragma unroll
    for (int blk = 0; blk < 3; blk++) {
        const uint4 a = src[4 * blk], b = src[4 * blk + 1], c = src[4 * blk + 2], d = src[4 * blk + 3];
        const uint32_t m[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
        b2s_compress(h, m, 64u * (blk + 1), blk == 2);
    }
    uint4* dst = reinterpret_cast<uint4*>(D.tnodes + 32 * (((size_t)pr << D.log_n) + half + j));
    dst[0] = make_uint4(h[0], h[1], h[2], h[3]);
    dst[1] = make_uint4(h[4], h[5], h[6], h[7]);
}

// ---- one tree level for every proof: nodes[i] = H(nodes[2i] || nodes[2i+1]), i in [count, 2*count) ----
__global__ void __launch_bounds__(256) level_kernel(uint4* __restrict__ nodes, uint64_t stride4, uint32_t count) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    uint4* base = nodes + stride4 * blockIdx.y;
    const size_t i = (size_t)count + t;
    const uint4* src = base + 4 * i;
    const uint4 a = src[0], b = src[1], c = src[2], d = src[3];
    const uint32_t m[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    uint32_t h[8];
    b2s_hash64(h, m);
    base[2 * i] = make_uint4(h[0], h[1], h[2], h[3]);
    base[2 * i + 1] = make_uint4(h[4], h[5], h[6], h[7]);
}

// the top of each tree (levels of at most 256 nodes): one workgroup per proof; copies the root out
__global__ void __launch_bounds__(256) top_kernel(uint4* nodes, uint64_t stride4, uint32_t count, uint4* __restrict__ roots) {
    uint4* base = nodes + stride4 * blockIdx.x;
    for (uint32_t c = count; c >= 1; c >>= 1) {
        if (threadIdx.x < c) {
            const size_t i = (size_t)c + threadIdx.x;
            // children written by other lanes of this workgroup one level ago: volatile loads, so that they are
            // not served from a stale vector-L1 line
            const volatile uint32_t* src = reinterpret_cast<const volatile uint32_t*>(base + 4 * i);
            uint32_t m[16];
#pragma unroll
            for (int k = 0; k < 16; k++) m[k] = src[k];
            uint32_t h[8];
            b2s_hash64(h, m);
            base[2 * i] = make_uint4(h[0], h[1], h[2], h[3]);
            base[2 * i + 1] = make_uint4(h[4], h[5], h[6], h[7]);
            if (c == 1) {
                roots[2 * (size_t)blockIdx.x] = make_uint4(h[0], h[1], h[2], h[3]);
                roots[2 * (size_t)blockIdx.x + 1] = make_uint4(h[4], h[5], h[6], h[7]);
            }
        }
        __threadfence_block();
        __syncthreads();
    }
}

// ---- coefficients of the linear combination: lane = (proof, r), r = i mod 8 selects x^steps ----
__global__ void __launch_bounds__(64) coef_kernel(Dev D) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 8 * D.B) return;
    const uint32_t pr = t >> 3, r = t & 7;
    const uint4* rt = reinterpret_cast<const uint4*>(D.troot + 32 * (size_t)pr);
    const uint4 r0 = rt[0], r1 = rt[1];
    F29 k[4];
#pragma unroll 1
    for (uint32_t j = 0; j < 4; j++) {  // BLAKE2s(root || byte j+1), read as a big-endian number (src/lib.rs:131-146)
        uint32_t m[16] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, j + 1, 0, 0, 0, 0, 0, 0, 0};
        uint32_t h[8];
        b2s_init(h);
        b2s_compress(h, m, 33u, true);
        k[j] = from_be_words(make_uint4(h[0], h[1], h[2], h[3]), make_uint4(h[4], h[5], h[6], h[7]));
    }
    // x^steps for a point i is G2^(steps * i), which depends on i mod 8 only
    const F29 w = load_limbs(D.xs + 8 * ((size_t)r << D.log_steps));
    uint32_t* o = D.coef + 16 * (size_t)t;
    store_limbs(o, f29_add(k[0], mul_t(k[1], w)));
    store_limbs(o + 8, f29_add(k[2], mul_t(k[3], w)));
}

// ---- L: lane = (proof, tree position) ----
__global__ void __launch_bounds__(256) lincomb_kernel(Dev D) {
    const uint32_t pr = blockIdx.y, N = 1u << D.log_n;
    const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= N) return;
    const uint32_t i = ((pos & 3u) << (D.log_n - 2)) + (pos >> 2);  // the point whose leaf sits at pos
    const uint8_t* leaf = D.leaves + 96 * (((size_t)pr << D.log_n) + pos);
    const F29 p = load_be(leaf), d = load_be(leaf + 32), b = load_be(leaf + 64);
    const uint32_t* c = D.coef + 16 * ((size_t)pr * 8 + (i & 7u));
    const F29 l = f29_add(d, f29_add(mul_t(p, load_limbs(c)), mul_t(b, load_limbs(c + 8))));
    store_be(D.tree[0] + 32 * (((size_t)pr << (D.log_n + 1)) + N + pos), l);
}

// ---- one FRI fold: lane = (proof, i), i < size/4 ----
__global__ void __launch_bounds__(256) fold_kernel(Dev D, uint32_t k) {
    const uint32_t pr = blockIdx.y, N = 1u << D.log_n;
    const uint32_t log_size = D.log_n - 2 * k, size = 1u << log_size, m = size >> 2;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    // rows i, i + m, i + 2m, i + 3m of this level are adjacent leaves 4i .. 4i+3 of its tree
    const uint8_t* src = D.tree[k] + 32 * (((size_t)pr << (log_size + 1)) + size + 4 * (size_t)i);
    const F29 r0 = load_be(src), r1 = load_be(src + 32), r2 = load_be(src + 64), r3 = load_be(src + 96);
    const F29 special = load_be(D.roots[k] + 32 * (size_t)pr);
    // 1 / x_i, x_i = (G2^(4^k))^i
    const F29 xinv = load_limbs(D.xs + 8 * (size_t)((N - (i << (2 * k))) & (N - 1)));
    const F29 t = mul_t(special, xinv);
    const F29 q = limbs_arg(D.q4);
    // 4 * c_j * x^j = sum_r row_r * q^(-r j)
    const F29 A = add_t(r0, r2), Bv = sub_t(r0, r2), C = add_t(r1, r3), Dq = mul_t(f29_sub(r1, r3), q);
    const F29 k0 = add_t(A, C), k2 = sub_t(A, C), k1 = sub_t(Bv, Dq), k3 = add_t(Bv, Dq);
    F29 acc = add_t(mul_t(k3, t), k2);
    acc = add_t(mul_t(acc, t), k1);
    acc = add_t(mul_t(acc, t), k0);
    const F29 col = mul_t(acc, limbs_arg(D.inv4));
    const uint32_t log_m = log_size - 2;
    store_be(D.tree[k + 1] + 32 * (((size_t)pr << (log_m + 1)) + m + perm4(i, log_m)), col);
}

// ---- query indices: get_pseudorandom_indices (src/utils.rs:53-94); lane = seed ----
__global__ void __launch_bounds__(64)
sample_kernel(const uint8_t* __restrict__ seeds, uint32_t n, uint32_t count, uint32_t real_modulus, uint32_t exclude,
              uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4* s = reinterpret_cast<const uint4*>(seeds + 32 * (size_t)i);
    const uint4 s0 = s[0], s1 = s[1];
    uint32_t h[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    uint32_t* o = out + (size_t)count * i;
    for (uint32_t j = 0; j < count; j += 8) {
        if (j) {
            uint32_t m[16];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                m[k] = h[k];
                m[8 + k] = 0;
            }
            b2s_init(h);
            b2s_compress(h, m, 32u, true);
        }
        for (uint32_t k = 0; k < 8 && j + k < count; k++) {
            uint32_t x = __builtin_bswap32(h[k]) % real_modulus;
            if (exclude) x = 1 + x + x / (exclude - 1);
            o[j + k] = x;
        }
    }
}

// ---- output: lane = (proof, item) ----
__global__ void __launch_bounds__(256)
serialize_kernel(Dev D, const Item* __restrict__ items, uint32_t n_items, uint8_t* __restrict__ out, uint64_t out_stride) {
    const uint32_t pr = blockIdx.y;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_items) return;
    const Item it = items[t];
    uint8_t* o = out + out_stride * pr + it.out_off;
    if (it.src == SRC_WORD) {
        *reinterpret_cast<uint32_t*>(o) = it.pre_word;
        return;
    }
    if (it.pre_word) *reinterpret_cast<uint32_t*>(o - 4) = it.pre_word;
    uint32_t idx = it.add;
    if (it.set != 0xFF) {
        const size_t base = it.set < D.nl ? (size_t)FRI_SAMPLES * ((size_t)D.B * it.set + pr)
                                           : (size_t)FRI_SAMPLES * D.B * D.nl + (size_t)SPOT_CHECKS * pr;
        idx += D.samples[base + it.sidx];
    }
    const uint32_t lvl = it.level & 0x7Fu, sib = it.level >> 7;
    const uint8_t* src;
    if (it.src == SRC_LEAF) {
        const uint32_t pos = perm4(idx & ((1u << D.log_n) - 1u), D.log_n) ^ sib;
        src = D.leaves + 96 * (((size_t)pr << D.log_n) + pos) + 32 * lvl;
    } else if (it.src == SRC_TRACE) {
        const uint32_t pos = perm4(idx & ((1u << D.log_n) - 1u), D.log_n);
        const uint32_t node = (((1u << (D.log_n - 1)) + (pos >> 1)) >> lvl) ^ sib;
        src = D.tnodes + 32 * (((size_t)pr << D.log_n) + node);
    } else {
        const uint32_t log_size = D.log_n - 2 * it.src;
        const uint32_t pos = perm4(idx & ((1u << log_size) - 1u), log_size);
        const uint32_t node = (((1u << log_size) + pos) >> lvl) ^ sib;
        src = D.tree[it.src] + 32 * (((size_t)pr << (log_size + 1)) + node);
    }
    const uint4 a = reinterpret_cast<const uint4*>(src)[0], b = reinterpret_cast<const uint4*>(src)[1];
    u32x4_a4 va = {a.x, a.y, a.z, a.w}, vb = {b.x, b.y, b.z, b.w};
    reinterpret_cast<u32x4_a4*>(o)[0] = va;
    reinterpret_cast<u32x4_a4*>(o)[1] = vb;
}

// ===================================== host side =====================================

static U256 h_add(const U256& a, const U256& b) {
    U256 r;
    hostfe::u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (hostfe::u128)a.w[i] + b.w[i];
        r.w[i] = (uint64_t)c;
        c >>= 64;
    }
    if (c) hostfe::add_small(r, hostfe::FOLD);
    return hostfe::canon(r);
}
static U256 h_neg(const U256& a) {
    if ((a.w[0] | a.w[1] | a.w[2] | a.w[3]) == 0) return a;
    U256 r;
    hostfe::u128 borrow = 0;
    for (int i = 0; i < 4; i++) {
        hostfe::u128 d = (hostfe::u128)hostfe::MODULUS.w[i] - a.w[i] - (uint64_t)borrow;
        r.w[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
    return r;
}
static U256 h_sub(const U256& a, const U256& b) { return h_add(a, h_neg(b)); }
static bool h_zero(const U256& a) { return (a.w[0] | a.w[1] | a.w[2] | a.w[3]) == 0; }

// Batched inverse with 0 -> 0 (what the verifier never reads: multiples of the extension factor are not sampled)
static void batch_inverse(std::vector<U256>& v) {
    const U256 one = hostfe::from_u64(1);
    std::vector<U256> partial(v.size() + 1);
    partial[0] = one;
    for (size_t i = 0; i < v.size(); i++) partial[i + 1] = h_zero(v[i]) ? partial[i] : hostfe::mul(partial[i], v[i]);
    U256 inv = hostfe::inverse(partial[v.size()]);
    for (size_t i = v.size(); i-- > 0;) {
        if (h_zero(v[i])) continue;
        const U256 x = v[i];
        v[i] = hostfe::mul(partial[i], inv);
        inv = hostfe::mul(inv, x);
    }
}

static unsigned ilog2(uint64_t x) {
    unsigned l = 0;
    while ((1ull << l) < x) l++;
    return l;
}

struct Layout {
    unsigned log_steps = 0, log_n = 0, nl = 0;
    std::vector<Item> items;
    size_t proof_size = 0;
};

// Number of FRI layers: the degree bound 2*steps is divided by 4 while it is above 16.
static unsigned fri_layers(uint64_t steps) {
    unsigned nl = 0;
    for (uint64_t d = 2 * steps; d > 16; d /= 4) nl++;
    return nl;
}

static bool params_ok(uint64_t steps, size_t n_constants) {
    auto pow2 = [](uint64_t x) { return x && !(x & (x - 1)); };
    // the prover's own limits: at least one FRI layer, at most 2^20 points per column, constants <= steps
    return pow2(steps) && pow2(n_constants) && n_constants <= steps && steps >= 16 && steps <= (1u << 17);
}

static void build_layout(Layout& L, uint64_t steps) {
    L.log_steps = ilog2(steps);
    L.log_n = L.log_steps + 3;
    L.nl = fri_layers(steps);
    const uint32_t N = 1u << L.log_n;
    uint32_t off = 0;
    auto word = [&](uint32_t v) {
        L.items.push_back(Item{off, 0, v, SRC_WORD, 0, 0xFF, 0});
        off += 4;
    };
    auto field = [&](uint8_t src, uint8_t level, uint8_t set, uint8_t sidx, uint32_t add, uint32_t pre) {
        L.items.push_back(Item{off, add, pre, src, level, set, sidx});
        off += 32;
    };
    auto branch = [&](uint8_t tree, uint8_t set, uint8_t sidx, uint32_t add) {  // 32-byte values
        const unsigned log_size = L.log_n - 2 * tree;
        off += 4;
        field(tree, 0, set, sidx, add, 32);
        field(tree, 0x80, set, sidx, add, 0);
        off += 4;
        for (unsigned l = 1; l < log_size; l++) field(tree, (uint8_t)(0x80 | l), set, sidx, add, l == 1 ? 32 * (log_size - 1) : 0);
    };
    auto trace_branch = [&](uint8_t set, uint8_t sidx, uint32_t add) {  // 96-byte values
        off += 4;
        for (uint8_t part = 0; part < 3; part++) field(SRC_LEAF, part, set, sidx, add, part == 0 ? 96 : 0);
        for (uint8_t part = 0; part < 3; part++) field(SRC_LEAF, (uint8_t)(0x80 | part), set, sidx, add, 0);
        off += 4;
        for (unsigned l = 0; l + 1 < L.log_n; l++)
            field(SRC_TRACE, (uint8_t)(0x80 | l), set, sidx, add, l == 0 ? 32 * (L.log_n - 1) : 0);
    };
    field(SRC_TRACE, (uint8_t)(L.log_n - 1), 0xFF, 0, 0, 0);  // merkle root
    field(0, (uint8_t)L.log_n, 0xFF, 0, 0, 0);                // root of the L tree
    for (unsigned k = 0; k < L.nl; k++) {
        const unsigned log_size = L.log_n - 2 * k;
        word(1);
        field((uint8_t)(k + 1), (uint8_t)(log_size - 2), 0xFF, 0, 0, 0);  // root2
        word(FRI_SAMPLES);
        for (uint32_t s = 0; s < FRI_SAMPLES; s++) branch((uint8_t)(k + 1), (uint8_t)k, (uint8_t)s, 0);
        word(4 * FRI_SAMPLES);
        for (uint32_t s = 0; s < FRI_SAMPLES; s++)
            for (uint32_t j = 0; j < 4; j++) branch((uint8_t)k, (uint8_t)k, (uint8_t)s, j << (log_size - 2));
    }
    const uint32_t fin = N >> (2 * L.nl);
    word(2);
    word(32 * fin);
    for (uint32_t i = 0; i < fin; i++) field((uint8_t)L.nl, 0, 0xFF, 0, i, 0);
    word(2 * SPOT_CHECKS);
    for (uint32_t s = 0; s < SPOT_CHECKS; s++) {
        trace_branch((uint8_t)L.nl, (uint8_t)s, 0);
        trace_branch((uint8_t)L.nl, (uint8_t)s, EXT);
    }
    word(SPOT_CHECKS);
    for (uint32_t s = 0; s < SPOT_CHECKS; s++) branch(0, (uint8_t)L.nl, (uint8_t)s, 0);
    L.proof_size = off;
}

// Everything that depends on (steps, round constants) only.
struct Plan : CtxAttachment {
    uint64_t steps = 0;
    std::vector<uint32_t> consts;  // canonical limbs, the cache key with steps
    Layout lay;
    U256 g2, g1;
    uint32_t *d_xs = nullptr, *d_zfac = nullptr, *d_invz2 = nullptr, *d_kext = nullptr, *d_consts = nullptr;
    Item* d_items = nullptr;
    uint32_t ext_mask = 0;
    uint32_t inv4[8], q4[8], inv_last_m1[8], g1l[8], g2l[8];
    // chunk buffers, grown on demand
    uint8_t* work = nullptr;
    size_t work_bytes = 0;
    // traces of a whole group of chunks ([group][steps][8]) followed by their [group][16] input / slope pairs
    uint8_t* traces = nullptr;
    size_t traces_bytes = 0;
    ~Plan() override {
        (void)hipFree(d_xs);
        (void)hipFree(d_zfac);
        (void)hipFree(d_invz2);
        (void)hipFree(d_kext);
        (void)hipFree(d_consts);
        (void)hipFree(d_items);
        (void)hipFree(work);
        (void)hipFree(traces);
    }
};

#define P_TRY(call)                                                                   \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) {                                                       \
            char b_[256];                                                             \
            snprintf(b_, sizeof b_, "%s failed: %s", #call, hipGetErrorString(e_));   \
            ctx_set_error(ctx, b_);                                                   \
            return e_ == hipErrorOutOfMemory ? STARK_ERR_NOMEM : STARK_ERR_HIP;       \
        }                                                                             \
    } while (0)

static int upload(stark_ctx* ctx, const std::vector<U256>& v, uint32_t** out) {
    std::vector<uint32_t> l(8 * v.size());
    for (size_t i = 0; i < v.size(); i++) hostfe::to_limbs(v[i], &l[8 * i]);
    P_TRY(hipMalloc((void**)out, l.size() * 4));
    P_TRY(hipMemcpy(*out, l.data(), l.size() * 4, hipMemcpyHostToDevice));
    return STARK_OK;
}

static int build_plan(stark_ctx* ctx, Plan& pl, uint64_t steps, const uint32_t* round_constants, size_t n_constants) {
    pl.steps = steps;
    build_layout(pl.lay, steps);
    const unsigned log_n = pl.lay.log_n;
    const uint64_t N = 1ull << log_n;
    const uint64_t skips2 = steps / n_constants;
    // G2 = 7^((p-1)/N) (src/lib.rs:114); p - 1 is a multiple of 2^32, so the division is a shift
    {
        U256 e = hostfe::MODULUS;
        e.w[0] -= 1;
        U256 q;
        for (int i = 0; i < 4; i++) q.w[i] = (e.w[i] >> log_n) | (i < 3 ? e.w[i + 1] << (64 - log_n) : 0);
        pl.g2 = hostfe::pow_u256(hostfe::from_u64(7), q);
    }
    pl.g1 = hostfe::pow_u64(pl.g2, EXT);
    const U256 one = hostfe::from_u64(1);
    std::vector<U256> xs(N);
    xs[0] = one;
    for (uint64_t i = 1; i < N; i++) xs[i] = hostfe::mul(xs[i - 1], pl.g2);
    const U256 last = xs[(steps - 1) * EXT];
    // 1 / (x^steps - 1): eight distinct values
    std::vector<U256> zni(EXT);
    for (uint32_t r = 0; r < EXT; r++) zni[r] = h_sub(xs[(size_t)r * steps], one);
    batch_inverse(zni);
    std::vector<U256> zfac(N), iz2(N);
    for (uint64_t i = 0; i < N; i++) {
        const U256 xl = h_sub(xs[i], last);
        zfac[i] = hostfe::mul(xl, zni[i % EXT]);
        iz2[i] = hostfe::mul(h_sub(xs[i], one), xl);
    }
    batch_inverse(iz2);
    int rc;
    if ((rc = upload(ctx, xs, &pl.d_xs)) != STARK_OK || (rc = upload(ctx, zfac, &pl.d_zfac)) != STARK_OK ||
        (rc = upload(ctx, iz2, &pl.d_invz2)) != STARK_OK)
        return rc;
    // round constants, canonical; their polynomial over the extended domain (N / skips2 points)
    pl.consts.resize(8 * n_constants);
    for (size_t i = 0; i < n_constants; i++)
        hostfe::to_limbs(hostfe::canon(hostfe::from_limbs(round_constants + 8 * i)), &pl.consts[8 * i]);
    P_TRY(hipMalloc((void**)&pl.d_consts, pl.consts.size() * 4));
    P_TRY(hipMemcpy(pl.d_consts, pl.consts.data(), pl.consts.size() * 4, hipMemcpyHostToDevice));
    {
        const size_t n_ext = (size_t)(N / skips2);
        uint32_t root[8], mod[8];
        hostfe::to_limbs(hostfe::MODULUS, mod);
        std::vector<uint32_t> buf(8 * n_ext, 0);
        hostfe::to_limbs(hostfe::pow_u64(pl.g1, skips2), root);
        if ((rc = stark_fft_inv(ctx, pl.consts.data(), n_constants, root, mod, buf.data())) != STARK_OK) return rc;
        hostfe::to_limbs(hostfe::pow_u64(pl.g2, skips2), root);
        if ((rc = stark_fft(ctx, buf.data(), n_ext, root, mod, buf.data())) != STARK_OK) return rc;
        P_TRY(hipMalloc((void**)&pl.d_kext, buf.size() * 4));
        P_TRY(hipMemcpy(pl.d_kext, buf.data(), buf.size() * 4, hipMemcpyHostToDevice));
        pl.ext_mask = (uint32_t)(n_ext - 1);
    }
    hostfe::to_limbs(hostfe::inverse(hostfe::from_u64(4)), pl.inv4);
    hostfe::to_limbs(xs[N / 4], pl.q4);
    hostfe::to_limbs(hostfe::inverse(h_sub(last, one)), pl.inv_last_m1);
    hostfe::to_limbs(pl.g1, pl.g1l);
    hostfe::to_limbs(pl.g2, pl.g2l);
    P_TRY(hipMalloc((void**)&pl.d_items, pl.lay.items.size() * sizeof(Item)));
    P_TRY(hipMemcpy(pl.d_items, pl.lay.items.data(), pl.lay.items.size() * sizeof(Item), hipMemcpyHostToDevice));
    return STARK_OK;
}

constexpr int PROVER_SLOT = 0;

static int get_plan(stark_ctx* ctx, uint64_t steps, const uint32_t* round_constants, size_t n_constants, Plan** out) {
    Plan* pl = static_cast<Plan*>(ctx_get_attachment(ctx, PROVER_SLOT));
    if (pl && pl->steps == steps && pl->consts.size() == 8 * n_constants) {
        bool same = true;
        for (size_t i = 0; i < n_constants && same; i++) {
            uint32_t c[8];
            hostfe::to_limbs(hostfe::canon(hostfe::from_limbs(round_constants + 8 * i)), c);
            same = memcmp(c, &pl->consts[8 * i], 32) == 0;
        }
        if (same) {
            *out = pl;
            return STARK_OK;
        }
    }
    if (hipStreamSynchronize((hipStream_t)stark_ctx_stream(ctx)) != hipSuccess) return STARK_ERR_HIP;
    ctx_set_attachment(ctx, PROVER_SLOT, nullptr);  // one parameter set is kept at a time
    std::unique_ptr<Plan> fresh(new Plan);
    int rc = build_plan(ctx, *fresh, steps, round_constants, n_constants);
    if (rc != STARK_OK) return rc;
    *out = fresh.get();
    ctx_set_attachment(ctx, PROVER_SLOT, fresh.release());
    return STARK_OK;
}

// Device bytes one proof of a chunk needs, apart from its output.
static size_t per_proof_bytes(const Layout& L) {
    const size_t N = (size_t)1 << L.log_n;
    size_t b = N * 32 /*P*/ + N * 96 /*leaves*/ + N * 32 /*trace tree*/ + 512 /*coef*/ + 32 /*troot*/;
    for (unsigned k = 0; k <= L.nl; k++) b += (N >> (2 * k)) * 64 + 32;
    b += 4 * (FRI_SAMPLES * L.nl + SPOT_CHECKS);
    return b;
}

static int build_trees_level(stark_ctx* ctx, uint8_t* nodes, size_t stride_bytes, uint32_t leaves, uint32_t B, uint8_t* roots,
                             hipStream_t st) {
    uint32_t count = leaves >> 1;
    for (; count > 256; count >>= 1)
        hipLaunchKernelGGL(level_kernel, dim3((count + 255) / 256, B), dim3(256), 0, st, (uint4*)nodes, (uint64_t)(stride_bytes / 16),
                           count);
    hipLaunchKernelGGL(top_kernel, dim3(B), dim3(256), 0, st, (uint4*)nodes, (uint64_t)(stride_bytes / 16), count, (uint4*)roots);
    P_TRY(hipGetLastError());
    return STARK_OK;
}

// The sequential part for a whole group of proofs at once: a lane per proof, so the time does not depend on the
// group's size until the chip is full. d_inputs [G][8] -> traces [G][steps][8], io [G][16], d_outputs [G][8].
static int trace_group(stark_ctx* ctx, Plan& pl, const uint32_t* d_inputs, size_t G, uint32_t* d_outputs, hipStream_t st) {
    const size_t steps = (size_t)pl.steps, need = G * steps * 32 + G * 64;
    if (need > pl.traces_bytes) {
        P_TRY(hipStreamSynchronize(st));
        if (pl.traces) P_TRY(hipFree(pl.traces));
        pl.traces = nullptr;
        pl.traces_bytes = 0;
        P_TRY(hipMalloc((void**)&pl.traces, need));
        pl.traces_bytes = need;
    }
    Dev D = {};
    memcpy(D.inv_last_m1, pl.inv_last_m1, 32);
    hipLaunchKernelGGL(trace_kernel, dim3((unsigned)((G + 63) / 64)), dim3(64), 0, st, d_inputs, (uint32_t)G, (uint32_t)steps,
                       pl.d_consts, (uint32_t)(pl.consts.size() / 8 - 1), (uint32_t*)pl.traces, d_outputs,
                       (uint32_t*)(pl.traces + G * steps * 32), D);
    P_TRY(hipGetLastError());
    return STARK_OK;
}

// One chunk of a traced group: d_trace [B][steps][8] (transformed in place), d_io [B][16] -> d_proofs (B x out_stride
// bytes); everything queued on st.
static int prove_chunk(stark_ctx* ctx, Plan& pl, uint32_t* d_trace, uint32_t* d_io, uint32_t B, uint8_t* d_proofs,
                       uint64_t out_stride, hipStream_t st) {
    const Layout& L = pl.lay;
    const uint32_t N = 1u << L.log_n, steps = (uint32_t)pl.steps;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    // carve the work buffer
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o += up(bytes);
        return at;
    };
    const size_t o_P = take((size_t)B * N * 32), o_leaves = take((size_t)B * N * 96), o_tn = take((size_t)B * N * 32),
                 o_coef = take((size_t)B * 512), o_troot = take((size_t)B * 32);
    size_t o_tree[MAX_TREES], o_root[MAX_TREES];
    for (unsigned k = 0; k <= L.nl; k++) {
        o_tree[k] = take((size_t)B * (N >> (2 * k)) * 64);
        o_root[k] = take((size_t)B * 32);
    }
    const size_t o_samples = take((size_t)B * 4 * (FRI_SAMPLES * L.nl + SPOT_CHECKS));
    if (o > pl.work_bytes) {
        P_TRY(hipStreamSynchronize(st));
        if (pl.work) P_TRY(hipFree(pl.work));
        pl.work = nullptr;
        pl.work_bytes = 0;
        P_TRY(hipMalloc((void**)&pl.work, o));
        pl.work_bytes = o;
    }
    Dev D = {};
    D.log_n = L.log_n;
    D.log_steps = L.log_steps;
    D.nl = L.nl;
    D.B = B;
    D.ext_mask = pl.ext_mask;
    D.xs = pl.d_xs;
    D.zfac = pl.d_zfac;
    D.invz2 = pl.d_invz2;
    D.kext = pl.d_kext;
    D.P = (uint32_t*)(pl.work + o_P);
    D.io = d_io;
    D.coef = (uint32_t*)(pl.work + o_coef);
    D.leaves = pl.work + o_leaves;
    D.tnodes = pl.work + o_tn;
    D.troot = pl.work + o_troot;
    for (unsigned k = 0; k <= L.nl; k++) {
        D.tree[k] = pl.work + o_tree[k];
        D.roots[k] = pl.work + o_root[k];
    }
    D.samples = (uint32_t*)(pl.work + o_samples);
    memcpy(D.inv4, pl.inv4, 32);
    memcpy(D.q4, pl.q4, 32);
    memcpy(D.inv_last_m1, pl.inv_last_m1, 32);

    // trace polynomial: inverse transform of the trace rows
    int rc = stark_ntt_batch_dev(ctx, d_trace, d_trace, L.log_steps, B, pl.g1l, 1, st);
    if (rc != STARK_OK) return rc;
    // zero-extend to N coefficients, evaluate over the extended domain
    P_TRY(hipMemsetAsync(D.P, 0, (size_t)B * N * 32, st));
    P_TRY(hipMemcpy2DAsync(D.P, (size_t)N * 32, d_trace, (size_t)steps * 32, (size_t)steps * 32, B, hipMemcpyDeviceToDevice, st));
    rc = stark_ntt_batch_dev(ctx, D.P, D.P, L.log_n, B, pl.g2l, 0, st);
    if (rc != STARK_OK) return rc;
    const dim3 gN((N + 255) / 256, B), gH((N / 2 + 255) / 256, B);
    static const bool fuse = !(getenv("STARKHIP_PROVER_FUSE") && atoi(getenv("STARKHIP_PROVER_FUSE")) == 0);
    if (fuse) {
        hipLaunchKernelGGL(columns_pairs_kernel, gH, dim3(256), 0, st, D);
    } else {
        hipLaunchKernelGGL(columns_kernel, gN, dim3(256), 0, st, D);
        hipLaunchKernelGGL(leaf_pairs_kernel, gH, dim3(256), 0, st, D);
    }
    P_TRY(hipGetLastError());
    if ((rc = build_trees_level(ctx, D.tnodes, (size_t)N * 32, N / 2, B, D.troot, st)) != STARK_OK) return rc;
    hipLaunchKernelGGL(coef_kernel, dim3((8 * B + 63) / 64), dim3(64), 0, st, D);
    hipLaunchKernelGGL(lincomb_kernel, gN, dim3(256), 0, st, D);
    P_TRY(hipGetLastError());
    if ((rc = build_trees_level(ctx, D.tree[0], (size_t)N * 64, N, B, D.roots[0], st)) != STARK_OK) return rc;
    for (unsigned k = 0; k < L.nl; k++) {
        const uint32_t m = N >> (2 * k + 2);
        hipLaunchKernelGGL(fold_kernel, dim3((m + 255) / 256, B), dim3(256), 0, st, D, k);
        P_TRY(hipGetLastError());
        if ((rc = build_trees_level(ctx, D.tree[k + 1], (size_t)m * 64, m, B, D.roots[k + 1], st)) != STARK_OK) return rc;
        hipLaunchKernelGGL(sample_kernel, dim3((B + 63) / 64), dim3(64), 0, st, D.roots[k + 1], B, FRI_SAMPLES,
                           m * (EXT - 1) / EXT, EXT, D.samples + (size_t)FRI_SAMPLES * B * k);
    }
    hipLaunchKernelGGL(sample_kernel, dim3((B + 63) / 64), dim3(64), 0, st, D.roots[0], B, SPOT_CHECKS,
                       (uint32_t)((uint64_t)N * (EXT - 1) / EXT), EXT, D.samples + (size_t)FRI_SAMPLES * B * L.nl);
    const uint32_t n_items = (uint32_t)L.items.size();
    hipLaunchKernelGGL(serialize_kernel, dim3((n_items + 255) / 256, B), dim3(256), 0, st, D, pl.d_items, n_items, d_proofs,
                       out_stride);
    P_TRY(hipGetLastError());
    return STARK_OK;
}

static size_t chunk_proofs(const Layout& L, size_t n) {
    size_t mb = 16384;
    if (const char* s = getenv("STARKHIP_PROVER_MB"))
        if (atoi(s) > 0) mb = (size_t)atoi(s);
    size_t B = (mb << 20) / (per_proof_bytes(L) + L.proof_size);
    if (B < 1) B = 1;
    if (B > 65535) B = 65535;  // grid.y
    return B < n ? B : n;
}

// Groups of proofs (at most 4 GiB of traces) are traced together, then finished in chunks of at most B.
// out_for(first, m) names the device buffer chunk [first, first + m) is written to; done(first, m) runs after the
// chunk has been queued.
template <class OutFn, class DoneFn>
static int prove_all(stark_ctx* ctx, Plan& pl, const uint32_t* d_inputs, size_t n, uint32_t* d_outputs, size_t B,
                     uint64_t out_stride, hipStream_t st, OutFn out_for, DoneFn done) {
    const size_t steps = (size_t)pl.steps;
    size_t G = ((size_t)4 << 30) / (steps * 32);
    if (G > n) G = n;
    for (size_t g0 = 0; g0 < n; g0 += G) {
        const size_t g = n - g0 < G ? n - g0 : G;
        int rc = trace_group(ctx, pl, d_inputs + 8 * g0, g, d_outputs + 8 * g0, st);
        if (rc != STARK_OK) return rc;
        uint32_t* io = (uint32_t*)(pl.traces + g * steps * 32);
        for (size_t c0 = 0; c0 < g; c0 += B) {
            const size_t m = g - c0 < B ? g - c0 : B;
            rc = prove_chunk(ctx, pl, (uint32_t*)pl.traces + c0 * steps * 8, io + 16 * c0, (uint32_t)m, out_for(g0 + c0, m),
                             out_stride, st);
            if (rc == STARK_OK) rc = done(g0 + c0, m);
            if (rc != STARK_OK) return rc;
        }
    }
    return STARK_OK;
}


}  // namespace prover
}  // namespace starkhip

using namespace starkhip;
using namespace starkhip::prover;

#define STARK_CATCH                                              \
    catch (const std::bad_alloc&) { return STARK_ERR_NOMEM; }    \
    catch (...) { return STARK_ERR_HIP; }

extern "C" int stark_mimc_batch_dev(stark_ctx* ctx, const void* d_inputs, size_t n, uint64_t steps, const void* d_round_constants,
                                    size_t n_constants, void* d_outputs, void* stream) try {
    if (!ctx) return STARK_ERR_INVALID;
    if (ctx_make_current(ctx) != hipSuccess) return STARK_ERR_HIP;
    // the reference's loop bound is steps - 1 on an unsigned value and it indexes i % len
    if (steps == 0 || (steps > 1 && n_constants == 0) || n_constants > 0xFFFFFFFFull) {
        ctx_set_error(ctx, "steps must be at least 1 and, above 1, there must be round constants");
        return STARK_ERR_INVALID;
    }
    if (n == 0) return STARK_OK;
    if (!d_inputs || !d_outputs || (steps > 1 && !d_round_constants)) return STARK_ERR_INVALID;
    hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)stark_ctx_stream(ctx);
    hipLaunchKernelGGL(mimc_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, (const uint32_t*)d_inputs, n, steps,
                       (const uint32_t*)d_round_constants, (uint32_t)n_constants, (uint32_t*)d_outputs);
    P_TRY(hipGetLastError());
    return STARK_OK;
} STARK_CATCH

extern "C" int stark_mimc_batch(stark_ctx* ctx, const uint32_t* inputs, size_t n, uint64_t steps, const uint32_t* round_constants,
                                size_t n_constants, uint32_t* outputs) try {
    if (!ctx) return STARK_ERR_INVALID;
    if (ctx_make_current(ctx) != hipSuccess) return STARK_ERR_HIP;
    if (steps == 0 || (steps > 1 && n_constants == 0) || n_constants > 0xFFFFFFFFull) {
        ctx_set_error(ctx, "steps must be at least 1 and, above 1, there must be round constants");
        return STARK_ERR_INVALID;
    }
    if (n == 0) return STARK_OK;
    if (!inputs || !outputs || (steps > 1 && !round_constants)) return STARK_ERR_INVALID;
    hipStream_t st = (hipStream_t)stark_ctx_stream(ctx);
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_out = up(32 * n), o_c = 2 * o_out;
    uint8_t* d = nullptr;
    P_TRY(hipMalloc((void**)&d, o_c + up(32 * n_constants + 32)));
    struct Free {
        uint8_t* p;
        ~Free() { (void)hipFree(p); }
    } guard{d};
    P_TRY(hipMemcpyAsync(d, inputs, 32 * n, hipMemcpyHostToDevice, st));
    if (n_constants) P_TRY(hipMemcpyAsync(d + o_c, round_constants, 32 * n_constants, hipMemcpyHostToDevice, st));
    int rc = stark_mimc_batch_dev(ctx, d, n, steps, d + o_c, n_constants, d + o_out, st);
    if (rc != STARK_OK) return rc;
    P_TRY(hipMemcpyAsync(outputs, d + o_out, 32 * n, hipMemcpyDeviceToHost, st));
    P_TRY(hipStreamSynchronize(st));
    return STARK_OK;
} STARK_CATCH

extern "C" size_t stark_mimc_proof_size(uint64_t num_steps, size_t n_constants) {
    try {
        if (!params_ok(num_steps, n_constants)) return 0;
        Layout L;
        build_layout(L, num_steps);
        return L.proof_size;
    } catch (...) {
        return 0;
    }
}

extern "C" int stark_mimc_prove_batch_dev(stark_ctx* ctx, const void* d_inputs, size_t n, uint64_t num_steps,
                                          const uint32_t* round_constants, size_t n_constants, void* d_proofs,
                                          uint64_t proof_stride, void* d_outputs, void* stream) try {
    if (!ctx) return STARK_ERR_INVALID;
    if (ctx_make_current(ctx) != hipSuccess) return STARK_ERR_HIP;
    if (!round_constants || (n && (!d_inputs || !d_proofs || !d_outputs))) return STARK_ERR_INVALID;
    if (!params_ok(num_steps, n_constants)) {
        ctx_set_error(ctx, "num_steps must be a power of two in [16, 2^17] and n_constants a power of two <= num_steps");
        return STARK_ERR_INVALID;
    }
    Plan* pl;
    int rc = get_plan(ctx, num_steps, round_constants, n_constants, &pl);
    if (rc != STARK_OK) return rc;
    if (proof_stride < pl->lay.proof_size || proof_stride % 4 != 0) {
        ctx_set_error(ctx, "proof_stride must be a multiple of 4 and at least stark_mimc_proof_size()");
        return STARK_ERR_INVALID;
    }
    hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)stark_ctx_stream(ctx);
    uint8_t* out = (uint8_t*)d_proofs;
    return prove_all(ctx, *pl, (const uint32_t*)d_inputs, n, (uint32_t*)d_outputs, chunk_proofs(pl->lay, n), proof_stride, st,
                     [&](size_t first, size_t) { return out + proof_stride * first; }, [](size_t, size_t) { return STARK_OK; });
} STARK_CATCH

extern "C" int stark_mimc_prove_batch(stark_ctx* ctx, const uint32_t* inputs, size_t n, uint64_t num_steps,
                                      const uint32_t* round_constants, size_t n_constants, uint8_t* proofs,
                                      uint32_t* outputs) try {
    if (!ctx) return STARK_ERR_INVALID;
    if (ctx_make_current(ctx) != hipSuccess) return STARK_ERR_HIP;
    if (!round_constants || (n && (!inputs || !proofs || !outputs))) return STARK_ERR_INVALID;
    if (!params_ok(num_steps, n_constants)) {
        ctx_set_error(ctx, "num_steps must be a power of two in [16, 2^17] and n_constants a power of two <= num_steps");
        return STARK_ERR_INVALID;
    }
    Plan* pl;
    int rc = get_plan(ctx, num_steps, round_constants, n_constants, &pl);
    if (rc != STARK_OK) return rc;
    if (n == 0) return STARK_OK;
    const size_t psz = pl->lay.proof_size;
    hipStream_t st = (hipStream_t)stark_ctx_stream(ctx);
    // small chunks, two output buffers: the copy of one chunk to the host runs while the next is computed
    size_t B = chunk_proofs(pl->lay, n);
    if (B > 256) B = 256;
    struct Res {
        uint8_t* d = nullptr;
        hipStream_t cs = nullptr;  // the download stream: auxiliary stream 1 of the context
        hipEvent_t ev[2] = {nullptr, nullptr};
        ~Res() {
            if (cs) (void)hipStreamSynchronize(cs);
            (void)hipFree(d);
            for (hipEvent_t e : ev)
                if (e) (void)hipEventDestroy(e);
        }
    } R;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_out = up(n * 32), o_b0 = 2 * o_out, o_b1 = o_b0 + up(B * psz);
    P_TRY(hipMalloc((void**)&R.d, o_b1 + up(B * psz)));
    P_TRY(ctx_aux_stream(ctx, 1, &R.cs));
    for (hipEvent_t& e : R.ev) P_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    P_TRY(hipMemcpyAsync(R.d, inputs, n * 32, hipMemcpyHostToDevice, st));
    uint8_t* bufs[2] = {R.d + o_b0, R.d + o_b1};
    struct Pending {
        size_t first = 0, m = 0;
        int buf = 0;
        bool valid = false;
    } prev;
    auto flush = [&](Pending& p) -> int {
        if (!p.valid) return STARK_OK;
        p.valid = false;
        P_TRY(hipStreamWaitEvent(R.cs, R.ev[p.buf], 0));
        P_TRY(copy_d2h(ctx, proofs + psz * p.first, bufs[p.buf], p.m * psz, R.cs));
        P_TRY(hipStreamSynchronize(R.cs));
        return STARK_OK;
    };
    size_t k = 0;
    rc = prove_all(ctx, *pl, (const uint32_t*)R.d, n, (uint32_t*)(R.d + o_out), B, psz, st,
                   [&](size_t, size_t) { return bufs[k & 1]; },
                   [&](size_t first, size_t m) -> int {
                       P_TRY(hipEventRecord(R.ev[k & 1], st));
                       Pending cur;
                       cur.first = first;
                       cur.m = m;
                       cur.buf = (int)(k & 1);
                       cur.valid = true;
                       k++;
                       const int r = flush(prev);  // the previous chunk goes out while this one is computed
                       prev = cur;
                       return r;
                   });
    if (rc == STARK_OK) rc = flush(prev);
    if (rc != STARK_OK) {
        (void)hipStreamSynchronize(st);
        return rc;
    }
    P_TRY(hipMemcpyAsync(outputs, R.d + o_out, n * 32, hipMemcpyDeviceToHost, st));
    P_TRY(hipStreamSynchronize(st));
    return STARK_OK;
} STARK_CATCH
