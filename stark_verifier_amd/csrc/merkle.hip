// Batched Merkle branch verification on gfx950: one lane walks one branch.
//
// Computes, per branch, what the reference's ProofBranch::verify does
// (reference src/merkle_tree.rs:119-171) and MultiProof::verify loops over
// (src/merkle_tree.rs:35-41):
//   idx   = permute_4_index(a, 2^(w+1))                          (:112-116, :120)
//   acc   = H(value || sibling) if idx even else H(sibling || value)   (:129-141)
//   per witness k: acc = H(acc || wit_k) if bit k+1 of idx is 0 else H(wit_k || acc)  (:145-163)
//   accept = (acc == root)                                        (:165-170)
// H = BLAKE2s-256. Inner levels hash exactly 64 bytes = one compression each.
//
// Device layout (chosen so a wavefront's loads are contiguous):
//   values, siblings : [n][value_size] bytes
//   witnesses        : [num_witnesses][n][32] bytes, level-major, leaf side first
//   indices          : [n] u32 (query index `a`, before permutation)
//   roots            : [num_roots][32] bytes; root_ids [n] u32 or null (all use roots[0])
//   accept           : [n] u8; digests (optional) : [n][32] recomputed roots
#include "blake2s.cuh"
#include "internal.hpp"

namespace starkhip {

// Word `q` (0-based, little-endian u32) of the message first||second, zero past its end.
// Word loads when value_size is a multiple of 4 (always true for 16-byte aligned rows of the
// sizes a proof carries: 32 and 96); byte loads otherwise.
__device__ __forceinline__ uint32_t leaf_word(const uint8_t* first, const uint8_t* second, uint32_t vs,
                                              uint32_t q, bool word_ok) {
    if (word_ok) {
        uint32_t vw = vs >> 2;
        if (q < vw) return reinterpret_cast<const uint32_t*>(first)[q];
        if (q < 2 * vw) return reinterpret_cast<const uint32_t*>(second)[q - vw];
        return 0u;
    }
    uint32_t r = 0;
    for (int k = 0; k < 4; k++) {
        uint32_t pos = 4 * q + k;
        uint32_t byte = pos < vs ? first[pos] : (pos < 2 * vs ? second[pos - vs] : 0u);
        r |= byte << (8 * k);
    }
    return r;
}

template <bool VS32>
__global__ void __launch_bounds__(256)
merkle_verify_kernel(const uint8_t* __restrict__ values, const uint8_t* __restrict__ siblings, uint32_t value_size,
                     const uint4* __restrict__ witnesses, uint32_t num_witnesses,
                     const uint32_t* __restrict__ indices, const uint32_t* __restrict__ roots,
                     const uint32_t* __restrict__ root_ids, size_t n, uint8_t* __restrict__ accept,
                     uint32_t* __restrict__ digests) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;

    // permute_4_index with L = 2^(w+1): ld4 = 2^(w-1)
    uint32_t a = indices[i];
    uint32_t ld4_log = num_witnesses - 1;
    uint32_t idx = (a >> ld4_log) + 4u * (a & ((1u << ld4_log) - 1u));

    uint32_t h[8];
    uint32_t m[16];

    // first witness load is issued before the leaf hash so its latency hides under it
    const uint4* wp = witnesses + 2 * i;
    uint4 w_lo = wp[0], w_hi = wp[1];

    // ---- level 0: hash of the two leaf values ----
    if (VS32) {
        const uint4* v = reinterpret_cast<const uint4*>(values) + 2 * i;
        const uint4* s = reinterpret_cast<const uint4*>(siblings) + 2 * i;
        uint4 v0 = v[0], v1 = v[1], s0 = s[0], s1 = s[1];
        bool odd = idx & 1u;
        uint4 f0 = odd ? s0 : v0, f1 = odd ? s1 : v1, g0 = odd ? v0 : s0, g1 = odd ? v1 : s1;
        m[0] = f0.x; m[1] = f0.y; m[2] = f0.z; m[3] = f0.w;
        m[4] = f1.x; m[5] = f1.y; m[6] = f1.z; m[7] = f1.w;
        m[8] = g0.x; m[9] = g0.y; m[10] = g0.z; m[11] = g0.w;
        m[12] = g1.x; m[13] = g1.y; m[14] = g1.z; m[15] = g1.w;
        b2s_hash64(h, m);
    } else {
        const uint8_t* v = values + (size_t)value_size * i;
        const uint8_t* s = siblings + (size_t)value_size * i;
        const uint8_t* first = (idx & 1u) ? s : v;
        const uint8_t* second = (idx & 1u) ? v : s;
        bool word_ok = (value_size & 3u) == 0;
        uint32_t total = 2 * value_size;
        uint32_t nblocks = (total + 63) / 64;
        b2s_init(h);
        for (uint32_t b = 0; b < nblocks; b++) {
#pragma unroll
            for (int j = 0; j < 16; j++) m[j] = leaf_word(first, second, value_size, 16 * b + j, word_ok);
            bool last = (b + 1 == nblocks);
            b2s_compress(h, m, last ? total : 64 * (b + 1), last);
        }
    }

    // ---- levels 1..w: one compression each ----
    uint32_t pos = idx >> 1;
    for (uint32_t l = 0; l < num_witnesses; l++) {
        uint4 c_lo = w_lo, c_hi = w_hi;
        if (l + 1 < num_witnesses) {  // prefetch the next level's witness
            const uint4* np = witnesses + 2 * ((size_t)(l + 1) * n + i);
            w_lo = np[0];
            w_hi = np[1];
        }
        bool odd = pos & 1u;
        m[0] = odd ? c_lo.x : h[0]; m[1] = odd ? c_lo.y : h[1]; m[2] = odd ? c_lo.z : h[2]; m[3] = odd ? c_lo.w : h[3];
        m[4] = odd ? c_hi.x : h[4]; m[5] = odd ? c_hi.y : h[5]; m[6] = odd ? c_hi.z : h[6]; m[7] = odd ? c_hi.w : h[7];
        m[8] = odd ? h[0] : c_lo.x; m[9] = odd ? h[1] : c_lo.y; m[10] = odd ? h[2] : c_lo.z; m[11] = odd ? h[3] : c_lo.w;
        m[12] = odd ? h[4] : c_hi.x; m[13] = odd ? h[5] : c_hi.y; m[14] = odd ? h[6] : c_hi.z; m[15] = odd ? h[7] : c_hi.w;
        b2s_hash64(h, m);
        pos >>= 1;
    }

    const uint32_t* rt = roots + 8 * (size_t)(root_ids ? root_ids[i] : 0u);
    uint32_t diff = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) diff |= h[j] ^ rt[j];
    accept[i] = diff == 0 ? 1 : 0;
    if (digests) {
        uint4* d = reinterpret_cast<uint4*>(digests) + 2 * i;
        d[0] = make_uint4(h[0], h[1], h[2], h[3]);
        d[1] = make_uint4(h[4], h[5], h[6], h[7]);
    }
}

hipError_t launch_merkle_verify(const uint8_t* values, const uint8_t* siblings, uint32_t value_size,
                                const uint8_t* witnesses, uint32_t num_witnesses, const uint32_t* indices,
                                const uint8_t* roots, const uint32_t* root_ids, size_t n, uint8_t* accept,
                                uint8_t* digests, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    dim3 block(256), grid((unsigned)((n + 255) / 256));
    if (value_size == 32)
        hipLaunchKernelGGL(merkle_verify_kernel<true>, grid, block, 0, stream, values, siblings, value_size,
                           reinterpret_cast<const uint4*>(witnesses), num_witnesses, indices,
                           reinterpret_cast<const uint32_t*>(roots), root_ids, n, accept,
                           reinterpret_cast<uint32_t*>(digests));
    else
        hipLaunchKernelGGL(merkle_verify_kernel<false>, grid, block, 0, stream, values, siblings, value_size,
                           reinterpret_cast<const uint4*>(witnesses), num_witnesses, indices,
                           reinterpret_cast<const uint32_t*>(roots), root_ids, n, accept,
                           reinterpret_cast<uint32_t*>(digests));
    return hipGetLastError();
}

}  // namespace starkhip
