// Small kernels beside the transform passes (ntt29.hip): the n = 1 and n = 2 transforms (the reference's simple_ft
// base case alone, reference src/fft.rs:18-35), twiddle-table construction, element-wise field operations for the
// arithmetic parity tests, and synthetic inputs. They run on the 8 x 32-bit arithmetic of field.cuh.
#include "field.cuh"
#include "field29.cuh"
#include "internal.hpp"

namespace starkhip {

// ---- n = 1, 2 ----
__global__ void ntt_tiny_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int log_n, size_t batch,
                                NttPassArgs sc) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch) return;
    Fe s;
#pragma unroll
    for (int l = 0; l < 8; l++) s.l[l] = sc.scale[l];
    if (log_n == 0) {
        Fe x = fe_canon(fe_load(in + 8 * i));
        if (sc.has_scale) x = fe_mul(x, s);
        fe_store(out + 8 * i, x);
    } else {
        Fe x = fe_canon(fe_load(in + 16 * i)), y = fe_canon(fe_load(in + 16 * i + 8));
        Fe u = fe_add(x, y), v = fe_sub(x, y);
        if (sc.has_scale) {
            u = fe_mul(u, s);
            v = fe_mul(v, s);
        }
        fe_store(out + 16 * i, u);
        fe_store(out + 16 * i + 8, v);
    }
}

hipError_t launch_ntt_tiny(const uint32_t* in, uint32_t* out, int log_n, size_t batch, const uint32_t scale[8],
                           int has_scale, hipStream_t stream) {
    if (batch == 0) return hipSuccess;
    NttPassArgs sc = {};
    sc.has_scale = (uint32_t)has_scale;
    for (int l = 0; l < 8; l++) sc.scale[l] = scale[l];
    hipLaunchKernelGGL(ntt_tiny_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, stream, in, out, log_n,
                       batch, sc);
    return hipGetLastError();
}

// ---- twiddle tables ----
struct ScaleArg {
    uint32_t l[8];
};

__global__ void fill_powers_kernel(uint32_t* __restrict__ out, size_t count, uint64_t row_len, uint64_t row0,
                                   const uint32_t* __restrict__ pw, ScaleArg scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint64_t e = row_len ? (row0 + (uint64_t)(i / row_len)) * (uint64_t)(i % row_len) : (uint64_t)i;
    Fe acc;
#pragma unroll
    for (int l = 0; l < 8; l++) acc.l[l] = scale.l[l];
#pragma unroll 1
    for (int k = 0; e != 0; k++, e >>= 1)
        if (e & 1) acc = fe_mul(acc, fe_load(pw + 8 * k));
    fe_store(out + 8 * i, acc);
}

hipError_t launch_fill_powers(uint32_t* out, size_t count, uint64_t row_len, uint64_t row0, const uint32_t* pw_dev,
                              const uint32_t scale[8], hipStream_t stream) {
    if (count == 0) return hipSuccess;
    ScaleArg s;
    for (int l = 0; l < 8; l++) s.l[l] = scale[l];
    hipLaunchKernelGGL(fill_powers_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, out, count,
                       row_len, row0, pw_dev, s);
    return hipGetLastError();
}

// ---- element-wise field ops (parity tests of the arithmetic itself) ----
__global__ void field_op_kernel(int op, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                uint32_t* __restrict__ r, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe x = fe_canon(fe_load(a + 8 * i));
    Fe y = op == 3 ? fe_zero() : fe_canon(fe_load(b + 8 * i));
    Fe z;
    if (op == 0) z = fe_add(x, y);
    else if (op == 1) z = fe_sub(x, y);
    else if (op == 2) z = fe_mul(x, y);
    else if (op == 3) z = x;
    else {
        // the 29-bit-limb arithmetic, entered with the raw (possibly non-canonical) a
        F29 xa = f29_from_fe(fe_load(a + 8 * i)), yb = f29_from_fe(y);
        if (op == 4) z = f29_to_fe(f29_mul(xa, yb));
        else if (op == 5) z = f29_to_fe(f29_add(xa, yb));
        else if (op == 6) z = f29_to_fe(f29_sub(xa, yb));
        else {  // op 7: (2a - 2b) * b with the multiplier's first operand at its lazy bound
            F29 t = f29_sub2(f29_add(xa, xa), f29_add(yb, yb));
            z = f29_to_fe(f29_mul(t, yb));
        }
    }
    fe_store(r + 8 * i, z);
}

hipError_t launch_field_op(int op, const uint32_t* a, const uint32_t* b, uint32_t* r, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(field_op_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, op, a, b, r, n);
    return hipGetLastError();
}

// ---- synthetic inputs: SplitMix64 stream, four 64-bit draws per element, reduced mod p ----
__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t index) {
    uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void fill_random_field_kernel(uint32_t* __restrict__ out, size_t n, uint64_t seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe x;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        uint64_t z = splitmix64_at(seed, 4 * (uint64_t)i + d);
        x.l[2 * d] = (uint32_t)z;
        x.l[2 * d + 1] = (uint32_t)(z >> 32);
    }
    fe_store(out + 8 * i, fe_canon(x));
}

hipError_t launch_fill_random_field(uint32_t* out, size_t n, uint64_t seed, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(fill_random_field_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, out, n, seed);
    return hipGetLastError();
}

}  // namespace starkhip
