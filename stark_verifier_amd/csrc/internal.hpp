// Internal declarations shared by the HIP translation units of libstarkhip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "worker_pool.hpp"

struct stark_ctx;

namespace starkhip {

// ---- capi.hip: state other translation units keep on a context ----
// An attachment lives in a numbered slot of the context and is destroyed with it (after the stream has
// drained), or when the slot is set again.
struct CtxAttachment {
    virtual ~CtxAttachment() {}
};
constexpr int CTX_SLOTS = 4;
CtxAttachment* ctx_get_attachment(stark_ctx* ctx, int slot);
void ctx_set_attachment(stark_ctx* ctx, int slot, CtxAttachment* a);
void ctx_set_error(stark_ctx* ctx, const char* text);
// makes the context's device the calling thread's current device (every entry point does this first)
hipError_t ctx_make_current(stark_ctx* ctx);
// Every stream the library ever creates belongs to a context: its main stream (stark_ctx_stream) and at most
// CTX_AUX_STREAMS auxiliary ones, made on first use and shared by everything that overlaps work inside one call (the
// pipelined host paths of the transforms and of the batch verifier, the scalar checks beside the branch launch, the
// prover's download). With the caller's own stream that is four software streams per context at most, the number of
// hardware queues a process gets here (GPU_MAX_HW_QUEUES), so no two of them have to share a queue. All are plain
// streams of equal priority; see ctx_aux_stream in capi.hip for the rules their users keep.
constexpr int CTX_AUX_STREAMS = 2;
hipError_t ctx_aux_stream(stark_ctx* ctx, int which, hipStream_t* out);

// Copies between ordinary (pageable) host memory and the device through two pinned staging buffers of the context:
// host threads fill or drain one while the DMA engine moves the other (about twice the rate of a direct copy from
// pageable memory). copy_h2d returns once `src` has been read completely (the transfers are queued on `stream`);
// copy_d2h returns once `dst` is complete. Small copies and contexts without pinned memory use hipMemcpyAsync.
hipError_t copy_h2d(stark_ctx* ctx, void* d_dst, const void* src, size_t n, hipStream_t stream);
hipError_t copy_d2h(stark_ctx* ctx, void* dst, const void* d_src, size_t n, hipStream_t stream);

// ---- merkle.hip ----
hipError_t launch_merkle_verify(const uint8_t* values, const uint8_t* siblings, uint32_t value_size,
                                const uint8_t* witnesses, uint32_t num_witnesses, const uint32_t* indices,
                                const uint8_t* roots, const uint32_t* root_ids, size_t n, uint8_t* accept,
                                uint8_t* digests, hipStream_t stream);

// ---- ntt29.hip ----
// One pass of the multi-pass transform: every workgroup transforms 2^log_c tiles of 2^log_t
// elements. All offsets are in field elements (32 bytes each).
//   tile index tau = blockIdx.x, hi = tau >> dsh, lo = tau & dmask, c in [0, 2^log_c)
//   input  element j of tile (tau, c): in  + poly*poly_stride + hi*inA  + lo*inB  + c*inC  + j*inS
//   output element k of tile (tau, c): out + poly*poly_stride + hi*outA + lo*outB + c*outC + k*outS
//   if TW: output k is multiplied by TW[poly*twP + hi*twA + lo*twB + c*twC + k*twS] before the store
//   tiles with lo*2^log_c + c >= c_limit are skipped (ragged batch in the single-pass case)
//   the output's polynomial offset is poly*outP, and output k may be split in two: its k*outS term becomes
//   (k >> ksh)*outS2 + (k & kmask)*outS (the send-block layout of a transform sharded over several GPUs)
struct NttPassArgs {
    const uint32_t* in;
    uint32_t* out;
    const uint32_t* W;   // [2^log_t] powers of the tile's root of unity
    uint32_t iq[9];      // 29-bit limbs of W[2^log_t / 4], the primitive 4th root of unity every butterfly multiplies by
    uint32_t iq2[9];     // ... and of that root times 2^145 mod p (split multiplication)
    const uint32_t* TW;  // inter-pass twiddles (scaled by n^-1 for the inverse), or null
    uint64_t poly_stride;
    uint64_t inA, inB, inC, inS;
    uint64_t outA, outB, outC, outS;
    uint64_t twA, twB, twC, twS, twP;
    uint64_t outP;   // polynomial stride of the output (the input's is poly_stride)
    uint64_t outS2;  // see ksh
    uint32_t ksh, kmask;  // 31 and 0xFFFFFFFF: no split
    // input element j of a tile sits at (j >> jsh) * inS2 + (j & jmask) * inS when in_split is set (the launcher
    // fills jsh / jmask for "no split" otherwise): reads an intermediate that the previous pass stored tile by tile
    uint32_t in_split, jsh, jmask;
    uint64_t inS2;
    uint64_t c_limit;
    uint32_t dsh, dmask;
    uint32_t log_c;
    uint32_t load_cfast, store_cfast;  // lane order: tile index fastest (1) or element index fastest (0)
    uint32_t canon_in;   // inputs may be any 256-bit value: reduce on load (first pass only)
    uint32_t tiles, polys;  // 29-bit-limb kernels: grid = tiles * polys work items
    uint32_t grp_log;       // 0 = plain order, g + 1 = L2-grouped order with 2^g adjacent tiles per group
    uint32_t in_raw, out_raw;  // 29-bit-limb kernels: element is 12 unpacked dwords instead of 8 packed
    uint32_t has_scale;
    uint32_t scale[8];  // multiplies every output when has_scale (single-pass inverse)
    // Single-pass inverse with at least one twiddled round: the factor rides on the last round instead of four more
    // multiplications per lane: Wlast = W with every entry times the factor (the three twiddled inputs; entry 0, the
    // factor itself, multiplies the fourth input).
    const uint32_t* Wlast;
    // 1: the streaming form of a two-pass plan at 1024-point tiles (ntt29.hip, kinds 7 and 8): the first pass leaves the
    // inter-pass factors to the last pass, whose TW / tw* fields then address them by input row and element
    uint32_t stream;
    // streaming form: eight work counters (one per XCD) in device memory, owned by the caller and used by one launch at
    // a time; the launcher zeroes them on the launch's stream
    uint32_t* tickets;
};

// W entries are 12 unpacked dwords
size_t ntt29_pass_lds_bytes(int log_t, int log_c);
// bytes per element of the multi-pass intermediate (nine unpacked limbs)
size_t ntt29_mid_bytes();
hipError_t launch_ntt29_pass(int log_t, const NttPassArgs& a, unsigned tiles, unsigned polys, hipStream_t stream);
// workgroups the streaming form of a pass keeps resident on a device of n_cu compute units
unsigned ntt29_stream_resident(int n_cu);
hipError_t launch_unpack_table(const uint32_t* in, uint32_t* out, size_t count, hipStream_t stream);
// ---- tables.hip ----
// n = 1 or 2 (no radix-4 round exists): out = transform(in), one lane per polynomial
hipError_t launch_ntt_tiny(const uint32_t* in, uint32_t* out, int log_n, size_t batch, const uint32_t scale[8],
                           int has_scale, hipStream_t stream);
// out[i] = scale * base^e(i), e(i) = i (row_len == 0) or (row0 + i / row_len) * (i % row_len);
// pw[k] = base^(2^k), k < 64
hipError_t launch_fill_powers(uint32_t* out, size_t count, uint64_t row_len, uint64_t row0, const uint32_t* pw_dev,
                              const uint32_t scale[8], hipStream_t stream);
// element-wise field op for tests: 0 add, 1 sub, 2 mul, 3 canon(a)
hipError_t launch_field_op(int op, const uint32_t* a, const uint32_t* b, uint32_t* r, size_t n, hipStream_t stream);
// synthetic canonical field elements from SplitMix64(seed), 4 draws per element
hipError_t launch_fill_random_field(uint32_t* out, size_t n, uint64_t seed, hipStream_t stream);

// Merkle branches read in place from serialized proofs (merkle.hip)
hipError_t launch_merkle_gather(const uint8_t* blob, const uint64_t* rec_off, uint32_t value_size, uint32_t num_witnesses,
                                const uint32_t* indices, const uint64_t* root_off, size_t n, uint8_t* accept,
                                hipStream_t stream);

// The same for up to eight groups of lanes in one launch: group g covers lanes first[g] .. first[g] + count[g] of the
// arrays and has its own value size and witness count; groups run in the order given (start[] is filled in here).
struct GatherGroups {
    static constexpr uint32_t MAX = 8;
    uint32_t n;
    uint32_t first[MAX], count[MAX], value_size[MAX], num_witnesses[MAX], start[MAX];
};
hipError_t launch_merkle_gather_groups(const uint8_t* blob, const uint64_t* rec_off, const uint32_t* indices,
                                       const uint64_t* root_off, uint8_t* accept, GatherGroups G, hipStream_t stream);

hipError_t launch_merkle_build(uint8_t* nodes, unsigned log_leaves, hipStream_t stream);
hipError_t launch_sample_indices(const uint8_t* seeds, size_t n, uint32_t count, uint32_t real_modulus, uint32_t exclude,
                                 uint32_t* out, hipStream_t stream);

// ---- protocol-layer field checks of the batch driver (checks.hip) ----
// One sampled position of one FRI layer (reference src/lib.rs:73-86); byte fields are the proof's own
// big-endian values, copied as they are.
struct alignas(16) FriLane {
    uint8_t rows[4][32];  // poly_values[i*4 + j]
    uint8_t col[32];      // column_values[i]
    uint8_t special[32];  // the previous layer's root (special_x)
    uint32_t y, layer;
    uint32_t pad[2];
};
// One spot-check position of verify_mimc_proof (reference src/lib.rs:163-192)
struct alignas(16) ConstraintLane {
    uint8_t b1[96];  // P(x) | D(x) | B(x)
    uint8_t pg[32];  // P(g1 x)
    uint8_t l[32];   // L(x)
    uint32_t pos, proof;
    uint32_t pad[2];
};
struct CheckParams {
    uint32_t g2[8], q4[8], last_step[8];  // canonical limbs
    uint32_t precision, log_steps, log_skips2, n_coeffs;
    // Optional tables in device memory (launch_check_tables fills them; null: every lane computes its own powers):
    // pw[i] = g2^i for i < precision, kext[j] = constants_mini_polynomial(g2^(skips2 * j)) for j < precision / skips2,
    // 8 canonical limbs each. A sampled position's x, x^steps, x^skips2 and the inverse of a FRI point are all powers of
    // g2 with exponents below precision, so a lane looks up what it would otherwise reach by square-and-multiply
    // (24 multiplications) and by a Horner walk over the n_coeffs coefficients (63 multiplications and reductions).
    const uint32_t* pw = nullptr;
    const uint32_t* kext = nullptr;
};
// pw: precision x 8 dwords, kext: (precision >> log_skips2) x 8 dwords; P.pw / P.kext are ignored
hipError_t launch_check_tables(const CheckParams& P, const uint32_t* mini_poly, uint32_t* pw, uint32_t* kext, hipStream_t stream);
hipError_t launch_fri_check(const FriLane* lanes, size_t n, const CheckParams& P, uint8_t* status, hipStream_t stream);
// kk: per proof CHECK_PROOF_BYTES: 4 x 32 bytes BLAKE2s(merkle_root || {1,2,3,4}), then 8 + 8 canonical limbs: the
// instance's input and the slope (output - input) / (last_step - 1) of its boundary interpolant;
// mini_poly: n_coeffs x 8 canonical limbs
constexpr size_t CHECK_PROOF_BYTES = 192;
hipError_t launch_constraint_check(const ConstraintLane* lanes, size_t n, const CheckParams& P, const uint8_t* kk,
                                   const uint32_t* mini_poly, uint8_t* status, hipStream_t stream);

// ---- batch driver, fast path for proofs with the fixed instance's structure (batch_dev.hip) ----
namespace batch {
size_t fast_proof_size();
bool fast_is_canonical(const uint8_t* proof, size_t len);
// chunk: proof bytes (host); bases[i]: offset of canonical proof i inside it (a multiple of 4); io: per proof 8 + 8
// canonical limbs (input, slope of the boundary interpolant); status_out[i]: 0 accepted, else the code of a failing check
int fast_verify(stark_ctx* ctx, const uint8_t* chunk, size_t chunk_len, const uint64_t* bases, const uint32_t* io, size_t mc,
                const CheckParams& CP, const uint32_t* d_mini, uint8_t* status_out);
// the same with the proofs, offsets and (input, slope) pairs already in device memory
int fast_verify_resident(stark_ctx* ctx, const uint8_t* d_blob, const uint64_t* d_base, const uint32_t* d_io, size_t mc,
                         const CheckParams& CP, const uint32_t* d_mini, uint8_t* status_out);
// The same for a chunk in host memory taken in pieces of whole proofs (cuts[p] .. cuts[p + 1] = byte range of piece p,
// n_pieces + 1 entries): a helper thread stages and queues the pieces back to back on the context's stream while the
// caller prepares piece after piece and queues its kernels on a second stream, so the upload, the kernels and the
// host's structure check overlap. begin: device buffers for the chunk and at most max_proofs canonical proofs
// (max_piece_proofs in any one piece), uploads started; piece: queue the checks of the `count` canonical proofs of
// the next piece, one call per piece in order (bases = offsets inside the chunk, io as above); end: wait, verdicts in
// the order the proofs were queued. end must be called after begin succeeded, also after a failed piece (status_out
// may then be null). The context must not be used by anything else between begin and end.
struct FastSession;
int fast_session_begin(stark_ctx* ctx, const uint8_t* chunk, const size_t* cuts, size_t n_pieces, size_t max_proofs,
                       size_t max_piece_proofs, FastSession** out);
int fast_session_piece(stark_ctx* ctx, FastSession* s, const uint64_t* bases, const uint32_t* io, size_t count,
                       const CheckParams& CP, const uint32_t* d_mini);
int fast_session_end(stark_ctx* ctx, FastSession* s, uint8_t* status_out);
// structure check of device-resident proofs at d_proofs + i * stride; canonical_out is a host array
int fast_resident_scratch(stark_ctx* ctx, size_t bytes, uint8_t** out);
int fast_verify_resident_run(stark_ctx* ctx, const uint8_t* d_proofs, uint64_t stride, size_t first, size_t mc,
                             const uint32_t* d_inputs, const uint32_t* d_outputs, const uint32_t inv_last_m1[8],
                             const CheckParams& CP, const uint32_t* d_mini, uint8_t* canonical_out, uint8_t* status_out);
int fast_classify_resident(stark_ctx* ctx, const uint8_t* d_proofs, uint64_t stride, size_t n, uint8_t* canonical_out);
// d_io[k] = (canonical input, slope) of proof d_who[k] from device-resident n x 8-limb inputs / outputs
int fast_make_io(stark_ctx* ctx, const uint32_t* d_inputs, const uint32_t* d_outputs, const uint32_t* d_who, size_t mc,
                 const uint32_t inv_last_m1[8], uint32_t* d_io);
}  // namespace batch

// sharded-transform helpers (shard.hip)
hipError_t launch_twiddle_mul(uint32_t* data, uint64_t rows, uint64_t cols, uint64_t row0, uint64_t col0,
                              uint32_t log_order, uint32_t lo_bits, const uint32_t* lo, const uint32_t* hi,
                              hipStream_t stream);
hipError_t launch_permute3(const void* in, void* out, uint64_t A, uint64_t B, uint64_t chunk, hipStream_t stream);

}  // namespace starkhip
