// Local steps of a transform sharded over several GPUs (n = n1 * n2 split across ranks):
// the twiddle multiplication between the two transform stages and the block permutation that
// packs / transposes data either side of the all-to-all exchange. Both are streaming passes
// over 32-byte elements: one read and one write of every byte.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "field.cuh"
#include "field29.cuh"
#include "internal.hpp"

namespace starkhip {

// data[r][c] *= w^((row0 + r) * (col0 + c) mod 2^log_order). w^e is looked up as
// hi[e >> lo_bits] * lo[e & (2^lo_bits - 1)] from two tables of canonical elements that stay in L2.
__global__ void __launch_bounds__(256) twiddle_mul_kernel(uint32_t* __restrict__ data, uint64_t rows, uint64_t cols,
                                                          uint64_t row0, uint64_t col0, uint32_t log_order,
                                                          uint32_t lo_bits, const uint32_t* __restrict__ lo,
                                                          const uint32_t* __restrict__ hi) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const uint64_t r = i / cols, c = i - r * cols;
    const uint64_t e = ((row0 + r) * (col0 + c)) & (((uint64_t)1 << log_order) - 1);
    const uint64_t el = e & (((uint64_t)1 << lo_bits) - 1), eh = e >> lo_bits;
    F29 t = f29_mul(f29_from_fe(fe_load(lo + 8 * el)), f29_from_fe(fe_load(hi + 8 * eh)));
    F29 x = f29_mul(f29_from_fe(fe_load(data + 8 * i)), t);
    fe_store(data + 8 * i, f29_to_fe(x));
}

hipError_t launch_twiddle_mul(uint32_t* data, uint64_t rows, uint64_t cols, uint64_t row0, uint64_t col0,
                              uint32_t log_order, uint32_t lo_bits, const uint32_t* lo, const uint32_t* hi,
                              hipStream_t stream) {
    const uint64_t count = rows * cols;
    if (count == 0) return hipSuccess;
    if ((count + 255) / 256 >= (1ull << 31)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(twiddle_mul_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, data, rows, cols,
                       row0, col0, log_order, lo_bits, lo, hi);
    return hipGetLastError();
}

// out[b][a][k] = in[a][b][k], k < chunk 32-byte elements.
// chunk >= 4: every 16-byte half is copied by one lane, lanes consecutive along k (128 contiguous
// bytes or more per chunk on both sides).
__global__ void __launch_bounds__(256) permute_chunks_kernel(const uint4* __restrict__ in, uint4* __restrict__ out,
                                                             uint64_t A, uint64_t B, uint64_t chunk16) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A * B * chunk16) return;
    const uint64_t k = i % chunk16, ab = i / chunk16, b = ab % B, a = ab / B;
    out[(b * A + a) * chunk16 + k] = in[i];
}

// chunk == 1: a transpose of 32-byte elements through a 32 x 32 tile in LDS. Reads take 1 KiB
// contiguous per tile row (32 elements), writes 1 KiB contiguous per tile column.
__global__ void __launch_bounds__(256) transpose32_kernel(const uint4* __restrict__ in, uint4* __restrict__ out,
                                                          uint64_t A, uint64_t B) {
    __shared__ uint4 tile[32][65];  // [row][2 * col + half], pitch 65 keeps column reads off one bank group
    const uint64_t a0 = (uint64_t)blockIdx.y * 32, b0 = (uint64_t)blockIdx.x * 32;
    const uint32_t t = threadIdx.x, x = t & 63u, y = t >> 6;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t row = i * 4 + y;
        const uint64_t a = a0 + row, b = b0 + (x >> 1);
        if (a < A && b < B) tile[row][x] = in[(a * B + b) * 2 + (x & 1u)];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t col = i * 4 + y;
        const uint64_t b = b0 + col, a = a0 + (x >> 1);
        if (a < A && b < B) out[(b * A + a) * 2 + (x & 1u)] = tile[x >> 1][2 * col + (x & 1u)];
    }
}

hipError_t launch_permute3(const void* in, void* out, uint64_t A, uint64_t B, uint64_t chunk, hipStream_t stream) {
    if (A == 0 || B == 0 || chunk == 0) return hipSuccess;
    if (chunk == 1) {
        const uint64_t gx = (B + 31) / 32, gy = (A + 31) / 32;
        if (gx >= (1ull << 31) || gy > 65535) return hipErrorInvalidValue;
        hipLaunchKernelGGL(transpose32_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, stream, (const uint4*)in,
                           (uint4*)out, A, B);
    } else {
        const uint64_t count = A * B * chunk * 2;
        if ((count + 255) / 256 >= (1ull << 31)) return hipErrorInvalidValue;
        hipLaunchKernelGGL(permute_chunks_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream,
                           (const uint4*)in, (uint4*)out, A, B, chunk * 2);
    }
    return hipGetLastError();
}

}  // namespace starkhip
