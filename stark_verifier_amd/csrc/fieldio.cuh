// Field elements at the edges of a protocol kernel: the proof's 32-byte big-endian values
// (BigInt::from_bytes_be / to_bytes_be in the reference, src/lib.rs:54,78,169-174) and 8-limb
// little-endian table entries, to and from the nine-limb form the arithmetic runs in.
// Helpers ending in _t return canonical ("tight") values, so they can be fed to either operand
// of f29_mul and to f29_sub's subtrahend.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "field.cuh"
#include "field29.cuh"

namespace starkhip {
namespace fio {

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ F29 one() {
    F29 r;
    r.l[0] = 1;
#pragma unroll
    for (int i = 1; i < 9; i++) r.l[i] = 0;
    return r;
}

// canonical residue back in limb form: tight, and equal values have equal limbs
__device__ __forceinline__ F29 norm(const F29& x) { return f29_from_fe(f29_to_fe(x)); }

__device__ __forceinline__ F29 from_be_words(const uint4 a, const uint4 b) {
    Fe x;
    x.l[7] = __builtin_bswap32(a.x);
    x.l[6] = __builtin_bswap32(a.y);
    x.l[5] = __builtin_bswap32(a.z);
    x.l[4] = __builtin_bswap32(a.w);
    x.l[3] = __builtin_bswap32(b.x);
    x.l[2] = __builtin_bswap32(b.y);
    x.l[1] = __builtin_bswap32(b.z);
    x.l[0] = __builtin_bswap32(b.w);
    return f29_from_fe(x);  // tight limbs of the number as it stands (a hash read as a number may be p or more)
}

// 32 big-endian bytes at a 16-byte aligned address -> tight
__device__ __forceinline__ F29 load_be(const uint8_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    return from_be_words(q[0], q[1]);
}

// canonical -> 32 big-endian bytes at a 16-byte aligned address
__device__ __forceinline__ void store_be(uint8_t* p, const F29& v) {
    const Fe x = f29_to_fe(v);
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(__builtin_bswap32(x.l[7]), __builtin_bswap32(x.l[6]), __builtin_bswap32(x.l[5]),
                      __builtin_bswap32(x.l[4]));
    q[1] = make_uint4(__builtin_bswap32(x.l[3]), __builtin_bswap32(x.l[2]), __builtin_bswap32(x.l[1]),
                      __builtin_bswap32(x.l[0]));
}

// 8 little-endian limbs of a canonical value (table entries, transform outputs)
__device__ __forceinline__ F29 load_limbs(const uint32_t* l) {
    const uint4* q = reinterpret_cast<const uint4*>(l);
    const uint4 a = q[0], b = q[1];
    Fe x;
    x.l[0] = a.x;
    x.l[1] = a.y;
    x.l[2] = a.z;
    x.l[3] = a.w;
    x.l[4] = b.x;
    x.l[5] = b.y;
    x.l[6] = b.z;
    x.l[7] = b.w;
    return f29_from_fe(x);
}

__device__ __forceinline__ void store_limbs(uint32_t* l, const F29& v) {
    const Fe x = f29_to_fe(v);
    uint4* q = reinterpret_cast<uint4*>(l);
    q[0] = make_uint4(x.l[0], x.l[1], x.l[2], x.l[3]);
    q[1] = make_uint4(x.l[4], x.l[5], x.l[6], x.l[7]);
}

__device__ __forceinline__ F29 mul_t(const F29& a, const F29& b) { return f29_mul(a, b); }  // a lazy, b tight -> tight
__device__ __forceinline__ F29 add_t(const F29& a, const F29& b) { return norm(f29_add(a, b)); }
__device__ __forceinline__ F29 sub_t(const F29& a, const F29& b) { return norm(f29_sub(a, b)); }  // b tight

}  // namespace fio
}  // namespace starkhip
