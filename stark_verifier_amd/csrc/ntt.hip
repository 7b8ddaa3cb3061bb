// Number-theoretic transform kernels over p = 2^256 - 351*2^32 + 1 for gfx950.
//
// Computes what the reference's _fft does (reference src/fft.rs:37-61; base case simple_ft
// :18-35): natural-order in, natural-order out, out[i] = sum_j v[j] * w^(i*j) mod p, every
// output the canonical residue. The reference recurses on heap BigInts; here a transform of
// 2^log_n points is one to three passes, each pass a grid of workgroups that transform tiles
// of up to 4096 points held in LDS:
//
//   * inside a tile: Stockham autosort, radix-4 per lane (two butterfly levels per LDS round
//     trip, preceded by one radix-2 level when log2(tile) is odd). A lane keeps its four
//     elements (32 VGPRs) in registers; LDS holds eight limb planes per tile so that lanes
//     reading consecutive elements hit consecutive banks.
//   * across tiles (domains larger than LDS): the four-step split n = n1*m. Pass 1 transforms
//     columns (stride m) and multiplies output (k1, j) by w^(k1*j) from a table; the remaining
//     passes transform the rows and store transposed, out[k1 + n1*k]. Workgroups take several
//     adjacent columns / rows so every global access is a run of contiguous 32-byte elements.
//
// The inverse transform (reference fft_inv, src/fft.rs:64-86) runs the same kernels with the
// inverse root; its n^-1 factor is folded into the inter-pass twiddle table (or, for
// single-pass sizes, applied at the final store).
#include "field.cuh"
#include "field29.cuh"
#include "internal.hpp"

namespace starkhip {

__device__ __forceinline__ void lds_put(uint32_t* lds, uint32_t plane, uint32_t off, const Fe& v) {
#pragma unroll
    for (int l = 0; l < 8; l++) lds[l * plane + off] = v.l[l];
}

__device__ __forceinline__ Fe lds_get(const uint32_t* lds, uint32_t plane, uint32_t off) {
    Fe v;
#pragma unroll
    for (int l = 0; l < 8; l++) v.l[l] = lds[l * plane + off];
    return v;
}

__host__ __device__ inline uint32_t tile_pad(uint32_t log_c) {
    // Row padding (in elements) that spreads the 2^log_c tiles of a workgroup over the 32 LDS
    // banks when lanes run over the tile index first.
    if (log_c == 0) return 0;
    return log_c >= 5 ? 1u : (32u >> log_c);
}

size_t ntt_pass_lds_bytes(int log_t, int log_c) {
    size_t tp = ((size_t)1 << log_t) + tile_pad((uint32_t)log_c);
    return 32 * tp << log_c;
}

template <int LOGT>
__global__ void __launch_bounds__(1024) ntt_pass_kernel(const NttPassArgs a) {
    constexpr uint32_t T = 1u << LOGT, Q = T / 4, LOGQ = LOGT - 2;
    constexpr int ROUNDS = (LOGT + 1) / 2;
    extern __shared__ uint32_t lds[];

    const uint32_t log_c = a.log_c, C = 1u << log_c;
    const uint32_t Tp = T + tile_pad(log_c);
    const uint32_t plane = Tp << log_c;  // dwords per limb plane
    const uint32_t tid = threadIdx.x;
    const uint32_t tau = blockIdx.x, hi = tau >> a.dsh, lo = tau & a.dmask;
    const uint64_t poly = (uint64_t)blockIdx.y * a.poly_stride;

    uint32_t ns_log = 0;  // log2 of the sub-transform length already completed
#pragma unroll 1
    for (int round = 0; round < ROUNDS; round++) {
        const bool first = round == 0, last = round == ROUNDS - 1;
        const bool cfast = last ? a.store_cfast : a.load_cfast;
        uint32_t c, j;
        if (cfast) {
            c = tid & (C - 1);
            j = tid >> log_c;
        } else {
            c = tid >> LOGQ;
            j = tid & (Q - 1);
        }
        const bool valid = (uint64_t)lo * C + c < a.c_limit;
        const uint32_t row = c * Tp;

        Fe x0, x1, x2, x3;
        if (first) {
            if (valid) {
                const uint32_t* src = a.in + 8 * (poly + hi * a.inA + lo * a.inB + c * a.inC);
                x0 = fe_load(src + 8 * ((uint64_t)j * a.inS));
                x1 = fe_load(src + 8 * ((uint64_t)(j + Q) * a.inS));
                x2 = fe_load(src + 8 * ((uint64_t)(j + 2 * Q) * a.inS));
                x3 = fe_load(src + 8 * ((uint64_t)(j + 3 * Q) * a.inS));
                if (a.canon_in) {
                    x0 = fe_canon(x0);
                    x1 = fe_canon(x1);
                    x2 = fe_canon(x2);
                    x3 = fe_canon(x3);
                }
            } else {
                x0 = x1 = x2 = x3 = fe_zero();
            }
        } else {
            x0 = lds_get(lds, plane, row + j);
            x1 = lds_get(lds, plane, row + j + Q);
            x2 = lds_get(lds, plane, row + j + 2 * Q);
            x3 = lds_get(lds, plane, row + j + 3 * Q);
        }

        if ((LOGT & 1) && first) {
            // radix-2 level: butterflies (j, j+T/2) and (j+Q, j+Q+T/2), outputs at 2*jj, 2*jj+1
            Fe y0 = fe_add(x0, x2), y1 = fe_sub(x0, x2), y2 = fe_add(x1, x3), y3 = fe_sub(x1, x3);
            lds_put(lds, plane, row + 2 * j, y0);
            lds_put(lds, plane, row + 2 * j + 1, y1);
            lds_put(lds, plane, row + 2 * (j + Q), y2);
            lds_put(lds, plane, row + 2 * (j + Q) + 1, y3);
            __syncthreads();
            ns_log = 1;
            continue;
        }

        if (ns_log > 0) {
            // twiddles w_T^(r * (j mod Ns) * T/(4 Ns)), r = 1, 2, 3
            const uint32_t e = (j & ((1u << ns_log) - 1u)) << (LOGQ - ns_log);
            x1 = fe_mul(x1, fe_load(a.W + 8 * (size_t)e));
            x2 = fe_mul(x2, fe_load(a.W + 8 * (size_t)(2 * e)));
            x3 = fe_mul(x3, fe_load(a.W + 8 * (size_t)(3 * e)));
        }
        // radix-4 butterfly with the fourth root of unity I = w_T^(T/4)
        Fe y0, y1, y2, y3;
        {
            Fe a0 = fe_add(x0, x2), a1 = fe_sub(x0, x2), a2 = fe_add(x1, x3);
            Fe a3 = fe_mul(fe_sub(x1, x3), fe_load(a.W + 8 * (size_t)Q));
            y0 = fe_add(a0, a2);
            y1 = fe_add(a1, a3);
            y2 = fe_sub(a0, a2);
            y3 = fe_sub(a1, a3);
        }

        if (!last) {
            const uint32_t d = row + ((j >> ns_log) << (ns_log + 2)) + (j & ((1u << ns_log) - 1u));
            if (!first) __syncthreads();  // every lane has finished reading this round's inputs
            lds_put(lds, plane, d, y0);
            lds_put(lds, plane, d + (1u << ns_log), y1);
            lds_put(lds, plane, d + (2u << ns_log), y2);
            lds_put(lds, plane, d + (3u << ns_log), y3);
            __syncthreads();
        } else if (valid) {
            // lane holds outputs k = j + r*Q
            if (a.TW || a.has_scale) {
                Fe s;
#pragma unroll
                for (int l = 0; l < 8; l++) s.l[l] = a.scale[l];
                const uint32_t* tw = a.TW ? a.TW + 8 * (hi * a.twA + lo * a.twB + c * a.twC) : nullptr;
                y0 = fe_mul(y0, tw ? fe_load(tw + 8 * ((uint64_t)j * a.twS)) : s);
                y1 = fe_mul(y1, tw ? fe_load(tw + 8 * ((uint64_t)(j + Q) * a.twS)) : s);
                y2 = fe_mul(y2, tw ? fe_load(tw + 8 * ((uint64_t)(j + 2 * Q) * a.twS)) : s);
                y3 = fe_mul(y3, tw ? fe_load(tw + 8 * ((uint64_t)(j + 3 * Q) * a.twS)) : s);
            }
            uint32_t* dst = a.out + 8 * (poly + hi * a.outA + lo * a.outB + c * a.outC);
            fe_store(dst + 8 * ((uint64_t)j * a.outS), y0);
            fe_store(dst + 8 * ((uint64_t)(j + Q) * a.outS), y1);
            fe_store(dst + 8 * ((uint64_t)(j + 2 * Q) * a.outS), y2);
            fe_store(dst + 8 * ((uint64_t)(j + 3 * Q) * a.outS), y3);
        }
        ns_log += 2;
    }
}

template <int LOGT>
static hipError_t launch_one(const NttPassArgs& a, unsigned tiles, unsigned polys, hipStream_t stream) {
    size_t lds = ntt_pass_lds_bytes(LOGT, (int)a.log_c);
    unsigned threads = (1u << (LOGT - 2)) << a.log_c;
    if (threads > 1024 || lds > 160 * 1024) return hipErrorInvalidConfiguration;
    static bool attr_set[64] = {};  // per device: dynamic LDS above 64 KiB must be opted into
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 64 || !attr_set[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_pass_kernel<LOGT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if (dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(ntt_pass_kernel<LOGT>, dim3(tiles, polys), dim3(threads), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_ntt_pass(int log_t, const NttPassArgs& a, unsigned tiles, unsigned polys, hipStream_t stream) {
    switch (log_t) {
        case 2: return launch_one<2>(a, tiles, polys, stream);
        case 3: return launch_one<3>(a, tiles, polys, stream);
        case 4: return launch_one<4>(a, tiles, polys, stream);
        case 5: return launch_one<5>(a, tiles, polys, stream);
        case 6: return launch_one<6>(a, tiles, polys, stream);
        case 7: return launch_one<7>(a, tiles, polys, stream);
        case 8: return launch_one<8>(a, tiles, polys, stream);
        case 9: return launch_one<9>(a, tiles, polys, stream);
        case 10: return launch_one<10>(a, tiles, polys, stream);
        case 11: return launch_one<11>(a, tiles, polys, stream);
        case 12: return launch_one<12>(a, tiles, polys, stream);
        default: return hipErrorInvalidValue;
    }
}

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

__global__ void fill_powers_kernel(uint32_t* __restrict__ out, size_t count, uint64_t row_len,
                                   const uint32_t* __restrict__ pw, ScaleArg scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint64_t e = row_len ? (uint64_t)(i / row_len) * (uint64_t)(i % row_len) : (uint64_t)i;
    Fe acc;
#pragma unroll
    for (int l = 0; l < 8; l++) acc.l[l] = scale.l[l];
#pragma unroll 1
    for (int k = 0; e != 0; k++, e >>= 1)
        if (e & 1) acc = fe_mul(acc, fe_load(pw + 8 * k));
    fe_store(out + 8 * i, acc);
}

hipError_t launch_fill_powers(uint32_t* out, size_t count, uint64_t row_len, const uint32_t* pw_dev,
                              const uint32_t scale[8], hipStream_t stream) {
    if (count == 0) return hipSuccess;
    ScaleArg s;
    for (int l = 0; l < 8; l++) s.l[l] = scale[l];
    hipLaunchKernelGGL(fill_powers_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, out, count,
                       row_len, pw_dev, s);
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
