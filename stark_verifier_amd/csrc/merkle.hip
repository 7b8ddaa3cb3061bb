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
#include <cstdlib>

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

// Lane-wise select through a 0 / 0xFFFFFFFF mask (one v_bitop3_b32).
__device__ __forceinline__ uint32_t pick(uint32_t mask, uint32_t if_set, uint32_t if_clear) {
    uint32_t r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xca" : "=v"(r) : "v"(mask), "v"(if_set), "v"(if_clear));
    return r;
}

#ifdef STARKHIP_STAMPS
// timing instrumentation (debug builds only): lane 0 of every workgroup records the 100 MHz wall clock when it
// starts, after the leaf hash and when it ends, with the compute unit it ran on
__device__ unsigned long long g_mstamps[16384][4];
#define MSTAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 16384) g_mstamps[blockIdx.x][k] = wall_clock64(); } while (0)
extern "C" int stark_debug_read_merkle_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mstamps), sizeof(g_mstamps));
}
#else
#define MSTAMP(k) do {} while (0)
#endif

template <bool VS32>
__global__ void __launch_bounds__(256)
merkle_verify_kernel(const uint8_t* __restrict__ values, const uint8_t* __restrict__ siblings, uint32_t value_size,
                     const uint4* __restrict__ witnesses, uint32_t num_witnesses,
                     const uint32_t* __restrict__ indices, const uint32_t* __restrict__ roots,
                     const uint32_t* __restrict__ root_ids, size_t n, uint8_t* __restrict__ accept,
                     uint32_t* __restrict__ digests) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    MSTAMP(0);

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
        const uint32_t vw[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        const uint32_t sw[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        const uint32_t odd = 0u - (idx & 1u);  // all ones: sibling first
#pragma unroll
        for (int j = 0; j < 8; j++) {
            m[j] = pick(odd, sw[j], vw[j]);
            m[8 + j] = pick(odd, vw[j], sw[j]);
        }
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

    MSTAMP(1);
    // ---- levels 1..w: one compression each ----
    uint32_t pos = idx >> 1;
    for (uint32_t l = 0; l < num_witnesses; l++) {
        uint4 c_lo = w_lo, c_hi = w_hi;
        if (l + 1 < num_witnesses) {  // prefetch the next level's witness
            const uint4* np = witnesses + 2 * ((size_t)(l + 1) * n + i);
            w_lo = np[0];
            w_hi = np[1];
        }
        const uint32_t ww[8] = {c_lo.x, c_lo.y, c_lo.z, c_lo.w, c_hi.x, c_hi.y, c_hi.z, c_hi.w};
        const uint32_t odd = 0u - (pos & 1u);  // all ones: witness first
#pragma unroll
        for (int j = 0; j < 8; j++) {
            m[j] = pick(odd, ww[j], h[j]);
            m[8 + j] = pick(odd, h[j], ww[j]);
        }
        b2s_hash64(h, m);
        pos >>= 1;
    }

    const uint32_t* rt = roots + 8 * (size_t)(root_ids ? root_ids[i] : 0u);
    uint32_t diff = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) diff |= h[j] ^ rt[j];
    accept[i] = diff == 0 ? 1 : 0;
    MSTAMP(2);
#ifdef STARKHIP_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 16384) {
        uint32_t hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        g_mstamps[blockIdx.x][3] = hw;
    }
#endif
    if (digests) {
        uint4* d = reinterpret_cast<uint4*>(digests) + 2 * i;
        d[0] = make_uint4(h[0], h[1], h[2], h[3]);
        d[1] = make_uint4(h[4], h[5], h[6], h[7]);
    }
}

// ---- gather variant: branches are read where they lie in a serialized proof ----
// A branch record in the wire format (reference src/deserializer.rs:104-125) is
//   u32 value_size | value | sibling_value | u32 witness_bytes | witness[0] | witness[1] | ...
// with every field at a multiple of 4 bytes when value_size is. Lane i reads its record at
// blob + rec_off[i] (the offset of `value`) and its root at blob + root_off[i], so the host
// uploads the proof bytes once and writes 20 bytes of descriptor per branch instead of copying
// ~0.5 KB of branch data into structure-of-arrays form. All records of one launch share
// (value_size, num_witnesses). 16-byte loads at 4-byte alignment.
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ uint4 load16_a4(const uint8_t* p) {
    const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// One branch: lane i of the launch's arrays (record, index, root, verdict).
template <bool VS32>
__device__ __forceinline__ void gather_branch(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ rec_off,
                                              uint32_t value_size, uint32_t num_witnesses,
                                              const uint32_t* __restrict__ indices, const uint64_t* __restrict__ root_off,
                                              size_t i, uint8_t* __restrict__ accept) {
    uint32_t a = indices[i];
    uint32_t ld4_log = num_witnesses - 1;
    uint32_t idx = (a >> ld4_log) + 4u * (a & ((1u << ld4_log) - 1u));
    const uint8_t* rec = blob + rec_off[i];
    const uint8_t* wit = rec + 2 * (size_t)value_size + 4;
    uint32_t h[8];
    uint32_t m[16];
    uint4 w_lo = load16_a4(wit), w_hi = load16_a4(wit + 16);
    if (VS32) {
        const uint4 v0 = load16_a4(rec), v1 = load16_a4(rec + 16);
        const uint4 s0 = load16_a4(rec + 32), s1 = load16_a4(rec + 48);
        const uint32_t vw[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        const uint32_t sw[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        const uint32_t odd = 0u - (idx & 1u);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            m[j] = pick(odd, sw[j], vw[j]);
            m[8 + j] = pick(odd, vw[j], sw[j]);
        }
        b2s_hash64(h, m);
    } else {
        const uint8_t* v = rec;
        const uint8_t* s = rec + value_size;
        const uint8_t* first = (idx & 1u) ? s : v;
        const uint8_t* second = (idx & 1u) ? v : s;
        uint32_t total = 2 * value_size;
        uint32_t nblocks = (total + 63) / 64;
        b2s_init(h);
        for (uint32_t b = 0; b < nblocks; b++) {
#pragma unroll
            for (int j = 0; j < 16; j++) m[j] = leaf_word(first, second, value_size, 16 * b + j, true);
            bool last = (b + 1 == nblocks);
            b2s_compress(h, m, last ? total : 64 * (b + 1), last);
        }
    }
    uint32_t pos = idx >> 1;
    for (uint32_t l = 0; l < num_witnesses; l++) {
        uint4 c_lo = w_lo, c_hi = w_hi;
        if (l + 1 < num_witnesses) {
            const uint8_t* np = wit + 32 * (size_t)(l + 1);
            w_lo = load16_a4(np);
            w_hi = load16_a4(np + 16);
        }
        const uint32_t ww[8] = {c_lo.x, c_lo.y, c_lo.z, c_lo.w, c_hi.x, c_hi.y, c_hi.z, c_hi.w};
        const uint32_t odd = 0u - (pos & 1u);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            m[j] = pick(odd, ww[j], h[j]);
            m[8 + j] = pick(odd, h[j], ww[j]);
        }
        b2s_hash64(h, m);
        pos >>= 1;
    }
    const uint8_t* rp = blob + root_off[i];
    const uint4 r0 = load16_a4(rp), r1 = load16_a4(rp + 16);
    const uint32_t rt[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
    uint32_t diff = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) diff |= h[j] ^ rt[j];
    accept[i] = diff == 0 ? 1 : 0;
}

template <bool VS32>
__global__ void __launch_bounds__(256)
merkle_gather_kernel(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ rec_off, uint32_t value_size,
                     uint32_t num_witnesses, const uint32_t* __restrict__ indices,
                     const uint64_t* __restrict__ root_off, size_t n, uint8_t* __restrict__ accept) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    gather_branch<VS32>(blob, rec_off, value_size, num_witnesses, indices, root_off, i, accept);
}

// Several groups of branches, each with its own (value_size, num_witnesses), in one launch: a launch per group pays
// that group's tail (its last wavefronts run their dependent compressions on an emptying chip) once per group; here the
// groups follow each other in the order given (deepest first leaves only the shallowest group's tail). Every group
// starts on a wavefront boundary of the launch's lane numbering, so a wavefront never mixes depths.
__global__ void __launch_bounds__(256)
merkle_gather_groups_kernel(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ rec_off,
                            const uint32_t* __restrict__ indices, const uint64_t* __restrict__ root_off,
                            uint8_t* __restrict__ accept, const GatherGroups G) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    // the group this lane falls into (compile-time indices into the argument block: selects, no indexed loads)
    uint32_t start = G.start[0], count = G.count[0], first = G.first[0], vs = G.value_size[0], w = G.num_witnesses[0];
#pragma unroll
    for (uint32_t k = 1; k < GatherGroups::MAX; k++)
        if (k < G.n && v >= G.start[k]) {
            start = G.start[k];
            count = G.count[k];
            first = G.first[k];
            vs = G.value_size[k];
            w = G.num_witnesses[k];
        }
    const uint32_t off = v - start;
    if (off >= count) return;
    const size_t i = (size_t)first + off;
    if (vs == 32) gather_branch<true>(blob, rec_off, 32u, w, indices, root_off, i, accept);
    else gather_branch<false>(blob, rec_off, vs, w, indices, root_off, i, accept);
}

// value_size must be a multiple of 4 and every offset a multiple of 4 (checked by the caller)
hipError_t launch_merkle_gather(const uint8_t* blob, const uint64_t* rec_off, uint32_t value_size, uint32_t num_witnesses,
                                const uint32_t* indices, const uint64_t* root_off, size_t n, uint8_t* accept,
                                hipStream_t stream) {
    if (n == 0) return hipSuccess;
    dim3 block(256), grid((unsigned)((n + 255) / 256));
    if (value_size == 32)
        hipLaunchKernelGGL(merkle_gather_kernel<true>, grid, block, 0, stream, blob, rec_off, value_size, num_witnesses,
                           indices, root_off, n, accept);
    else
        hipLaunchKernelGGL(merkle_gather_kernel<false>, grid, block, 0, stream, blob, rec_off, value_size, num_witnesses,
                           indices, root_off, n, accept);
    return hipGetLastError();
}

hipError_t launch_merkle_gather_groups(const uint8_t* blob, const uint64_t* rec_off, const uint32_t* indices,
                                       const uint64_t* root_off, uint8_t* accept, GatherGroups G, hipStream_t stream) {
    if (G.n == 0 || G.n > GatherGroups::MAX) return G.n == 0 ? hipSuccess : hipErrorInvalidValue;
    uint64_t at = 0;
    for (uint32_t g = 0; g < G.n; g++) {
        if (G.value_size[g] < 32 || (G.value_size[g] & 3u) || G.num_witnesses[g] == 0 || G.num_witnesses[g] > 31)
            return hipErrorInvalidValue;
        G.start[g] = (uint32_t)at;
        at += ((uint64_t)G.count[g] + 63) & ~(uint64_t)63;
        if (at >= (1ull << 32)) return hipErrorInvalidValue;
    }
    if (at == 0) return hipSuccess;
    // Unused LDS (52 KiB per workgroup of 256 lanes) caps the kernel at three wavefronts per SIMD. Two already saturate
    // the VALU with this kernel, so the cap costs it nothing, and the batch verifier's short check kernels, queued on
    // another stream, find free wavefront slots at once instead of waiting for a generation of these workgroups to drain
    // (resident verification of 1,250 proofs: 0.88 -> 0.84 ms). STARKHIP_GATHER_LDS=bytes changes it, 0 removes it.
    static const size_t lds_pad = getenv("STARKHIP_GATHER_LDS") ? (size_t)atol(getenv("STARKHIP_GATHER_LDS")) : 53248;
    if (lds_pad > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&merkle_gather_groups_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(merkle_gather_groups_kernel, dim3((unsigned)((at + 255) / 256)), dim3(256), lds_pad, stream, blob, rec_off,
                       indices, root_off, accept, G);
    return hipGetLastError();
}

// ---- query index sampler: get_pseudorandom_indices (reference src/utils.rs:53-94) ----
// One lane per seed. hashes = seed || H(seed) || H(H(seed)) || ... until 4*count bytes; index j is the
// big-endian u32 at byte 4j, reduced mod real_modulus = modulus*(exclude-1)/exclude (or modulus), and with
// `exclude` mapped past the multiples of it: x -> 1 + x + x/(exclude-1).
__global__ void __launch_bounds__(64)
sample_indices_kernel(const uint8_t* __restrict__ seeds, size_t n, uint32_t count, uint32_t real_modulus,
                      uint32_t exclude, uint32_t* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t h[8];
    for (int k = 0; k < 8; k++) {
        const uint8_t* b = seeds + 32 * i + 4 * k;
        h[k] = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
    }
    uint32_t* o = out + (size_t)count * i;
    for (uint32_t j = 0; j < count; j += 8) {
        if (j) {  // next 32 bytes of the chain: BLAKE2s of the previous 32
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

hipError_t launch_sample_indices(const uint8_t* seeds, size_t n, uint32_t count, uint32_t real_modulus, uint32_t exclude,
                                 uint32_t* out, hipStream_t stream) {
    if (n == 0 || count == 0) return hipSuccess;
    hipLaunchKernelGGL(sample_indices_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, seeds, n, count,
                       real_modulus, exclude, out);
    return hipGetLastError();
}

// ---- tree construction (prover side; the reference only verifies) ----
// nodes[1] is the root, nodes[i] = H(nodes[2i] || nodes[2i+1]), the 2^log_leaves leaves sit at
// nodes[L .. 2L) (already in the tree's leaf order, i.e. after permute_4_index). One lane per node of a
// level: 64 contiguous bytes in, 32 out (96 bytes and one compression per node).
__global__ void __launch_bounds__(256) merkle_level_kernel(uint4* __restrict__ nodes, size_t count) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const size_t i = count + t;
    const uint4* src = nodes + 4 * i;  // nodes[2i], nodes[2i+1]
    const uint4 a = src[0], b = src[1], c = src[2], d = src[3];
    const uint32_t m[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    uint32_t h[8];
    b2s_hash64(h, m);
    nodes[2 * i] = make_uint4(h[0], h[1], h[2], h[3]);
    nodes[2 * i + 1] = make_uint4(h[4], h[5], h[6], h[7]);
}

// the top of the tree (levels of at most 256 nodes) in one workgroup
__global__ void __launch_bounds__(256) merkle_top_kernel(uint4* nodes, uint32_t count) {
    for (uint32_t c = count; c >= 1; c >>= 1) {
        if (threadIdx.x < c) {
            const size_t i = (size_t)c + threadIdx.x;
            // the children were written by other lanes of this workgroup one level ago: volatile loads
            // so that they are not served from a stale vector-L1 line
            const volatile uint32_t* src = reinterpret_cast<const volatile uint32_t*>(nodes + 4 * i);
            uint32_t m[16];
#pragma unroll
            for (int k = 0; k < 16; k++) m[k] = src[k];
            uint32_t h[8];
            b2s_hash64(h, m);
            nodes[2 * i] = make_uint4(h[0], h[1], h[2], h[3]);
            nodes[2 * i + 1] = make_uint4(h[4], h[5], h[6], h[7]);
        }
        __threadfence_block();
        __syncthreads();
    }
}

hipError_t launch_merkle_build(uint8_t* nodes, unsigned log_leaves, hipStream_t stream) {
    size_t count = ((size_t)1 << log_leaves) >> 1;  // nodes on the level above the leaves
    uint4* n4 = reinterpret_cast<uint4*>(nodes);
    for (; count > 256; count >>= 1)
        hipLaunchKernelGGL(merkle_level_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, n4, count);
    if (count >= 1) hipLaunchKernelGGL(merkle_top_kernel, dim3(1), dim3(256), 0, stream, n4, (uint32_t)count);
    return hipGetLastError();
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
