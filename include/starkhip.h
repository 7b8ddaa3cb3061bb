/*
 * libstarkhip — C ABI of the MI355X (gfx950) hot path of a MiMC STARK verifier:
 * the radix-2 NTT / inverse NTT over p = 2^256 - 351*2^32 + 1 and batched BLAKE2s
 * Merkle-branch verification.
 *
 * The entry points are what a host-language binding of the reference crate would bind in
 * place of the bodies of `fft::fft_inv` (reference src/fft.rs:64), `ProofBranch::verify`
 * (src/merkle_tree.rs:119) and `MultiProof::verify` (src/merkle_tree.rs:25); their callers
 * are src/lib.rs:58,68,125,158,159.
 *
 * Conventions
 *   - A field element is 8 little-endian uint32 limbs (32 bytes). Inputs may be any 256-bit
 *     value; outputs are always the canonical residue in [0, p), which is what the reference's
 *     BigInt arithmetic (`%`, utils::negative_to_positive) yields.
 *   - Digests, Merkle values and witnesses are opaque bytes, exactly as in the proof file.
 *   - Every function returns STARK_OK or a negative status; nothing unwinds across the ABI.
 *     Where the reference panics (bad lengths, a root of unity of the wrong order, a branch
 *     with no witnesses) the call returns STARK_ERR_INVALID and writes nothing. A Merkle
 *     mismatch is not an error here: it is reported per branch in `accept[]`; the binding
 *     re-creates the reference's panic / None from that.
 *   - Host-pointer entry points copy in, run on the context's stream, copy out, and return
 *     after the stream has drained. Buffers of 8 MiB and more in ordinary (pageable) memory are
 *     copied through pinned staging buffers by host threads; a buffer that is pinned already
 *     (hipHostMalloc, hipHostRegister) is handed to the copy engine as it is. The `_dev` variants take device pointers (any allocator:
 *     hipMalloc, a torch tensor's data_ptr) and a hipStream_t passed as void*; they only
 *     enqueue work and do not synchronise.
 *   - A context is bound to one device and is not thread-safe; use one per host thread.
 */
#ifndef STARKHIP_H
#define STARKHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STARK_OK 0
#define STARK_ERR_INVALID (-1)     /* argument rejected (shapes on which the reference panics) */
#define STARK_ERR_HIP (-2)         /* a HIP runtime call failed; see stark_last_error */
#define STARK_ERR_NOMEM (-3)       /* device or host allocation failed */
#define STARK_ERR_UNSUPPORTED (-4) /* valid request outside what this build implements */

typedef struct stark_ctx stark_ctx;

/* Library version: major*10000 + minor*100 + patch. */
int stark_version(void);

/* Create / destroy a context on HIP device `device`. */
int stark_ctx_create(int device, stark_ctx** out);
void stark_ctx_destroy(stark_ctx* ctx);

/* Text of the last failure on this context (never NULL). */
const char* stark_last_error(const stark_ctx* ctx);

/* The context's own stream (a hipStream_t), for callers that queue work beside `_dev` calls. */
void* stark_ctx_stream(stark_ctx* ctx);

/* Block until everything queued on the context's stream has finished. */
int stark_ctx_synchronize(stark_ctx* ctx);

/* ---------------------------------------------------------------------------------------
 * NTT — replaces the body of fft::fft_inv (reference src/fft.rs:64-86) and its private
 * helpers _fft (:37-61) and simple_ft (:18-35).
 * ------------------------------------------------------------------------------------- */

/*
 * fft_inv(v, root_of_unity, modulus) -> Vec<BigInt>     (reference src/fft.rs:64)
 *   out[i] = n^-1 * sum_j v[j] * root^(-i*j) mod p,  n = number of elements.
 * `modulus` must be p (the only modulus the reference is ever called with, src/lib.rs:29,125).
 * Requires n to be a power of two and `root_of_unity` to have multiplicative order exactly n:
 * for any other input the reference indexes out of bounds and panics -> STARK_ERR_INVALID.
 * `v` and `out` are host arrays of n*8 limbs and may alias.
 */
int stark_fft_inv(stark_ctx* ctx, const uint32_t* v, size_t n, const uint32_t root_of_unity[8],
                  const uint32_t modulus[8], uint32_t* out);

/*
 * Forward transform out[i] = sum_j v[j] * root^(i*j) mod p. The reference has no public
 * wrapper for it; it is _fft (src/fft.rs:37) applied to the root table [root^0 .. root^(n-1)].
 * Same argument rules as stark_fft_inv.
 */
int stark_fft(stark_ctx* ctx, const uint32_t* v, size_t n, const uint32_t root_of_unity[8],
              const uint32_t modulus[8], uint32_t* out);

/*
 * Batched transform of `batch` polynomials of 2^log_n elements each, stored back to back
 * (polynomial b occupies limbs [b*2^log_n*8, (b+1)*2^log_n*8)). inverse != 0 selects fft_inv
 * semantics (inverse root and n^-1 scaling), inverse == 0 the forward transform.
 * Host version: in place on `limbs`.
 */
int stark_ntt_batch(stark_ctx* ctx, uint32_t* limbs, unsigned log_n, size_t batch,
                    const uint32_t root_of_unity[8], int inverse);

/*
 * Device version: d_in and d_out are device pointers to batch*2^log_n*32 bytes, 16-byte
 * aligned; they may be equal (in place). Work is queued on `stream` (NULL = the context's
 * stream). Multi-pass sizes use a scratch buffer owned by the context, so calls on one
 * context must not overlap on different streams. Twiddle tables are built on first use of a
 * (log_n, root, direction) triple and cached in the context.
 */
int stark_ntt_batch_dev(stark_ctx* ctx, const void* d_in, void* d_out, unsigned log_n, size_t batch,
                        const uint32_t root_of_unity[8], int inverse, void* stream);

/*
 * stark_ntt_batch_dev for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc). It has no counterpart in the reference. Same arguments and results as
 * stark_ntt_batch_dev, but it calls no hipMalloc / hipFree: the multi-pass intermediate goes to a fixed
 * STARK_STATIC_SCRATCH_BYTES array that is part of the library's device image (one copy per GPU, held until the
 * process ends), and the twiddle tables must already be in the context: call stark_ntt_batch_dev once with the same
 * (log_n, root_of_unity, inverse) beforehand (batch 0 is enough), otherwise STARK_ERR_INVALID. Transforms above
 * STARK_STATIC_MAX_LOG_N points do not fit the array: STARK_ERR_UNSUPPORTED. All calls on one GPU, from any
 * context, share that GPU's array: calls that overlap in time on one GPU race (queue them on one stream or order
 * them with events). Nothing in the array has to be cleared between calls: every word a pass reads was written by
 * the pass before it in the same call.
 */
#define STARK_STATIC_SCRATCH_BYTES ((size_t)36 << 20) /* 2^20 points x 36 bytes of intermediate */
#define STARK_STATIC_MAX_LOG_N 20
int stark_ntt_batch_dev_static(stark_ctx* ctx, const void* d_in, void* d_out, unsigned log_n, size_t batch,
                               const uint32_t root_of_unity[8], int inverse, void* stream);

/*
 * Local steps of one transform sharded over several GPUs. A transform of n = n1 * n2 points
 * (input index j = j1 + n1*j2, output index k = k1*n2 + k2) is
 *   (A) for every j1: an n2-point transform over j2        -> stark_ntt_batch_dev, root^n1
 *   (B) multiply element (j1, k2) by root^(j1*k2)          -> stark_twiddle_mul_dev
 *   (C) for every k2: an n1-point transform over j1        -> stark_ntt_batch_dev, root^n2
 * which is what the recursion of _fft (reference src/fft.rs:37-61) computes, regrouped; a rank
 * that owns a range of j1 runs (A) and (B) locally, the ranks exchange blocks (all-to-all), and
 * the owner of a range of k2 runs (C). The exchange itself is the host layer's (RCCL through
 * torch.distributed); these two entry points are the device work either side of it. They have
 * no counterpart in the reference, which is single-threaded.
 */

/* d_data[r][c] *= w^((row0 + r) * (col0 + c)) for a row-major rows x cols block of canonical or
 * non-canonical elements (outputs canonical); w = root_of_unity, or its inverse if inverse != 0;
 * root_of_unity must have order exactly 2^log_order (else STARK_ERR_INVALID). In place. */
int stark_twiddle_mul_dev(stark_ctx* ctx, void* d_data, size_t rows, size_t cols, uint64_t row0, uint64_t col0,
                          unsigned log_order, const uint32_t root_of_unity[8], int inverse, void* stream);

/* d_out[b][a][k] = d_in[a][b][k] for a x b blocks of `chunk` consecutive 32-byte elements
 * (chunk == 1 is a plain transpose). Packs the send blocks of an all-to-all and transposes the
 * received ones. d_in and d_out must not overlap; a <= 2097120 when chunk == 1. */
int stark_permute3_dev(stark_ctx* ctx, const void* d_in, void* d_out, size_t a, size_t b, size_t chunk, void* stream);

/*
 * The same transform with the three steps fused into two device calls, for hosts that drive the exchange themselves
 * (RCCL grouped send/recv or ncclAllToAll on their own communicator) or hand it in as a callback. No counterpart in
 * the reference. n = 2^log_n = n1 * n2 with n1 = 2^(log_n / 2); `world` ranks (a power of two that divides n1 and
 * n2), r1 = n1 / world, r2 = n2 / world. Per rank, all buffers hold n / world elements:
 *   d_in   "rows"    [r1][n2]         element (a, j2)  = x[(rank * r1 + a) + n1 * j2]
 *   d_send           [world][r1][r2]  block d goes to rank d; d_recv [world][r1][r2], block s came from rank s
 *   d_out  "columns" [n1][r2]         element (k1, b)  = X[k1 * n2 + rank * r2 + b]
 * Stage A: the r1 row transforms (n2 points, root^n1), each output (a, k2) multiplied by root^(+-(rank * r1 + a) * k2)
 * in the last pass's store and written straight into the send block of the rank that owns k2. Stage C: the r2 column
 * transforms (n1 points, root^n2) of the received [n1][r2] matrix, columns adjacent in memory. X = the transform of x
 * with fft_inv semantics when inverse != 0 (reference src/fft.rs:64-86), else the forward transform (_fft, :37-61).
 * STARK_ERR_UNSUPPORTED below 2^4 points or when world is too large for the fused store (r2 below the last pass's
 * tile count): use the three-step form above.
 */
int stark_ntt_sharded_stage_a_dev(stark_ctx* ctx, const void* d_in, void* d_send, unsigned log_n, unsigned world,
                                  unsigned rank, const uint32_t root_of_unity[8], int inverse, void* stream);
int stark_ntt_sharded_stage_c_dev(stark_ctx* ctx, const void* d_recv, void* d_out, unsigned log_n, unsigned world,
                                  const uint32_t root_of_unity[8], int inverse, void* stream);

/* Equal-block all-to-all on device memory, queued on `stream`: block d of d_send (block_bytes each, `world` blocks)
 * goes to rank d, block s of d_recv comes from rank s. Returns 0 on success. */
typedef int (*stark_all_to_all_fn)(void* user, const void* d_send, void* d_recv, size_t block_bytes, void* stream);

/* Stage A, exchange(user, ...), stage C in one call. d_x holds the "rows" input and is overwritten; d_y is the other
 * n / world element buffer. `exchange` is called whenever it is not NULL (also with world == 1, where a one-rank
 * communicator moves the single block to itself) and must be given when world > 1. The "columns" result is left in
 * d_y when an exchange ran and in d_x otherwise. Queued on `stream` (NULL = the context's stream); does not synchronise. */
int stark_ntt_sharded_dev(stark_ctx* ctx, void* d_x, void* d_y, unsigned log_n, unsigned world, unsigned rank,
                          const uint32_t root_of_unity[8], int inverse, stark_all_to_all_fn exchange, void* user,
                          void* stream);

/* Transforms of the `cols` columns of a row-major [2^log_n][cols] matrix of elements (column c = elements c,
 * c + cols, c + 2 cols, ...): `cols` interleaved polynomials, the layout stage C works on. cols a power of two,
 * 4 <= 2^log_n <= 2^24; d_in and d_out may be equal. Same semantics per column as stark_ntt_batch_dev. */
int stark_ntt_columns_dev(stark_ctx* ctx, const void* d_in, void* d_out, unsigned log_n, size_t cols,
                          const uint32_t root_of_unity[8], int inverse, void* stream);

/* ---------------------------------------------------------------------------------------
 * Merkle — replaces the bodies of ProofBranch::verify (reference src/merkle_tree.rs:119-171)
 * and the loop of MultiProof::verify (src/merkle_tree.rs:25-44).
 * ------------------------------------------------------------------------------------- */

/*
 * Verify n branches that share one shape (value_size bytes per value, num_witnesses
 * witnesses per branch); the binding groups a MultiProof's branches by shape.
 *   values, siblings : [n][value_size]      ProofBranch.value / .sibling_value
 *   witnesses        : [num_witnesses][n][32]   level-major; level 0 is ProofBranch.witnesses[0]
 *   indices          : [n]                  the `a` passed to ProofBranch::verify (unpermuted)
 *   roots            : [num_roots][32]; root_ids : [n] indices into roots, or NULL = roots[0]
 *   accept           : [n] out, 1 = recomputed root equals the branch's root, 0 = mismatch
 *   digests          : [n][32] out, the recomputed roots; may be NULL
 * STARK_ERR_INVALID where the reference panics before hashing: num_witnesses == 0 (division
 * by zero at :113-114), value_size < 32 (slice at :133), and for num_witnesses > 30 (the
 * reference's 2u32.pow overflows), or a root id >= num_roots.
 */
int stark_merkle_verify_batch(stark_ctx* ctx, const uint8_t* values, const uint8_t* siblings,
                              uint32_t value_size, const uint8_t* witnesses, uint32_t num_witnesses,
                              const uint32_t* indices, const uint8_t* roots, uint32_t num_roots,
                              const uint32_t* root_ids, size_t n, uint8_t* accept, uint8_t* digests);

/* Device version: same layout, device pointers (values/siblings/witnesses/roots/digests
 * 16-byte aligned), queued on `stream` (NULL = the context's stream). root ids are not range
 * checked here. */
int stark_merkle_verify_batch_dev(stark_ctx* ctx, const void* d_values, const void* d_siblings,
                                  uint32_t value_size, const void* d_witnesses, uint32_t num_witnesses,
                                  const void* d_indices, const void* d_roots, const void* d_root_ids,
                                  size_t n, void* d_accept, void* d_digests, void* stream);

/*
 * Tree construction — no counterpart in the reference (its prover is external, README.md:9); builds the
 * tree ProofBranch::verify walks: nodes[1] is the root, nodes[i] = BLAKE2s(nodes[2i] || nodes[2i+1]), the
 * 2^log_leaves 32-byte leaves at nodes[L .. 2L) in the tree's own leaf order (after permute_4_index,
 * src/merkle_tree.rs:112-116); nodes[0] is unused. Used to make synthetic trees on the device.
 * Host version: leaves [L][32] in, nodes [2L][32] out. Device version: in place on d_nodes (2L x 32 bytes,
 * 16-byte aligned, leaves already in the upper half).
 */
int stark_merkle_build(stark_ctx* ctx, const uint8_t* leaves, unsigned log_leaves, uint8_t* nodes);
int stark_merkle_build_dev(stark_ctx* ctx, void* d_nodes, unsigned log_leaves, void* stream);

/*
 * get_pseudorandom_indices(seed, count, modulus, excludeMultiplesOf) (reference src/utils.rs:53-94) for n seeds at
 * once: out[i][0..count) for seeds[i][32]. exclude_multiples_of = 0 is the reference's `None`. STARK_ERR_INVALID
 * where the reference divides by zero (exclude_multiples_of == 1, or a modulus that reduces to 0).
 * Callers in the reference: src/lib.rs:56 (per FRI layer) and :149.
 */
int stark_pseudorandom_indices(stark_ctx* ctx, const uint8_t* seeds, size_t n, uint32_t count, uint32_t modulus,
                               uint32_t exclude_multiples_of, uint32_t* out);
int stark_pseudorandom_indices_dev(stark_ctx* ctx, const void* d_seeds, size_t n, uint32_t count, uint32_t modulus,
                                   uint32_t exclude_multiples_of, void* d_out, void* stream);

/* ---------------------------------------------------------------------------------------
 * Batch driver — verify_mimc_proof (reference src/lib.rs:99-197, with verify_low_degree_proof
 * :31-97) over many serialized proofs (deserializer::from_bytes, src/deserializer.rs:142).
 * ------------------------------------------------------------------------------------- */

/*
 * proofs: concatenated proof files; proof i is bytes [offsets[i], offsets[i+1]).
 * inp, output: 8-limb field elements; round_constants: n_constants x 8 limbs; the modulus is p.
 * accept[i] = 1 exactly when the reference returns true for proof i. status (may be NULL):
 *   0 accepted                         1 parameter checks, reference returns false (:101-111)
 *   2 malformed proof bytes            3 branch count / shape on which the reference panics
 *   4 a recomputed Merkle root differs 5 low-degree check (:85)   6 transition (:181)
 *   7 boundary (:187)                  8 linear combination (:190)
 *   9 instance-size asserts (:50-51, :120-123: num_steps 8192, 64 round constants)
 * A proof whose structure words (branch counts, value sizes, witness counts, element tags) are those of an
 * 8192-step proof - every proof the reference accepts is one - is not parsed on the host: its bytes are
 * uploaded and the device samples the indices (src/utils.rs:53-94), hashes the 1,240 branches where they lie
 * and runs the scalar field checks. Any other proof is parsed by host threads (STARKHIP_HOST_THREADS, default
 * min(cores, 16)), its branches are hashed by the device in one call per (value size, witness count), and
 * the same check kernels decide. As in the reference, the final FRI "points" component is read and not
 * verified (src/deserializer.rs:47-59, src/lib.rs:94).
 */
int stark_mimc_verify_batch(stark_ctx* ctx, const uint8_t* proofs, const uint64_t* offsets, size_t n_proofs,
                            const uint32_t inp[8], uint64_t num_steps, const uint32_t* round_constants,
                            size_t n_constants, const uint32_t output[8], uint8_t* accept, uint8_t* status);

/* The same with one (input, output) pair per proof: inputs and outputs are n_proofs x 8 limbs, proof i is checked
 * as verify_mimc_proof(inputs[i], num_steps, round_constants, outputs[i], proof_i, p). */
int stark_mimc_verify_batch_io(stark_ctx* ctx, const uint8_t* proofs, const uint64_t* offsets, size_t n_proofs,
                               const uint32_t* inputs, uint64_t num_steps, const uint32_t* round_constants,
                               size_t n_constants, const uint32_t* outputs, uint8_t* accept, uint8_t* status);

/* The same for proofs already in device memory (what stark_mimc_prove_batch_dev leaves there): proof i is the
 * proof_stride bytes at d_proofs + i * proof_stride (proof_stride a multiple of 4; bytes after a complete proof are
 * ignored, as the reference's parser ignores them); d_inputs and d_outputs are n_proofs x 8 limbs in device memory
 * (16-byte aligned; d_proofs 4-byte aligned); round_constants, accept and status are host pointers. Returns after the
 * verdicts have been written. The device buffers are read on the context's own stream (stark_ctx_stream), which does
 * not wait for any other stream: work that produces them on another stream (a caller's stream passed to
 * stark_mimc_prove_batch_dev, torch's current stream) must have completed, or that stream must have been
 * synchronised, before this call. */
int stark_mimc_verify_batch_dev(stark_ctx* ctx, const void* d_proofs, uint64_t proof_stride, size_t n_proofs,
                                const void* d_inputs, uint64_t num_steps, const uint32_t* round_constants,
                                size_t n_constants, const void* d_outputs, uint8_t* accept, uint8_t* status);

/* ---------------------------------------------------------------------------------------
 * Proof generation — no counterpart in the reference: its proofs come from an external program
 * (README.md:9). Makes the proofs verify_mimc_proof (src/lib.rs:99-197) accepts, in the wire format
 * deserializer::from_bytes reads (src/deserializer.rs:16-140), many at a time and entirely on the device:
 * MiMC trace, the P / D / B columns over the 8x extended domain, their Merkle tree, the linear
 * combination L with its tree, the FRI layers with theirs, the sampled branches. With input 3,
 * num_steps 8192 and the 64 constants of src/main.rs:209-212 the output equals the reference's proof.bin.
 * ------------------------------------------------------------------------------------- */

/*
 * utils::mimc(input, steps, round_constants, modulus) (reference src/utils.rs:8-16; called at src/main.rs:216 to get the
 * public output the verifier is given) for n inputs at once: steps - 1 rounds of x -> x^3 + round_constants[i mod
 * n_constants] mod p. inputs / round_constants may be any 256-bit values; outputs are canonical. STARK_ERR_INVALID where
 * the reference's loop bound underflows (steps == 0) or its index divides by zero (no constants and steps > 1).
 * Host version: host pointers, returns when outputs are written. Device version: device pointers (16-byte aligned),
 * queued on `stream` (NULL = the context's stream).
 */
int stark_mimc_batch(stark_ctx* ctx, const uint32_t* inputs, size_t n, uint64_t steps, const uint32_t* round_constants,
                     size_t n_constants, uint32_t* outputs);
int stark_mimc_batch_dev(stark_ctx* ctx, const void* d_inputs, size_t n, uint64_t steps, const void* d_round_constants,
                         size_t n_constants, void* d_outputs, void* stream);

/* Bytes of one serialized proof for these parameters (566,508 for 8192 steps); 0 if they are rejected:
 * num_steps a power of two in [16, 2^17], n_constants a power of two <= num_steps. */
size_t stark_mimc_proof_size(uint64_t num_steps, size_t n_constants);

/*
 * inputs: n x 8 limbs (any 256-bit values); round_constants: n_constants x 8 limbs (host memory).
 * proofs: n x stark_mimc_proof_size() bytes out, proof i at proofs + i * size; outputs: n x 8 limbs, the
 * last trace value mimc(input) (src/utils.rs:8-16) the verifier is called with.
 * Tables that depend on (num_steps, round_constants) are built on first use and kept in the context (one
 * parameter set at a time). Proofs are made in chunks sized by STARKHIP_PROVER_MB (default 16384 MiB of
 * device memory, about 17 MiB per proof at 8192 steps).
 */
int stark_mimc_prove_batch(stark_ctx* ctx, const uint32_t* inputs, size_t n, uint64_t num_steps,
                           const uint32_t* round_constants, size_t n_constants, uint8_t* proofs, uint32_t* outputs);

/* Device version: d_inputs, d_proofs (proof i at d_proofs + i * proof_stride; proof_stride a multiple of 4, at
 * least the proof size), d_outputs are device pointers, 16-byte aligned; round_constants stays a host pointer.
 * Queued on `stream` (NULL = the context's stream); does not synchronise. */
int stark_mimc_prove_batch_dev(stark_ctx* ctx, const void* d_inputs, size_t n, uint64_t num_steps,
                               const uint32_t* round_constants, size_t n_constants, void* d_proofs,
                               uint64_t proof_stride, void* d_outputs, void* stream);

/* ---------------------------------------------------------------------------------------
 * Encoding helpers (host only, no context): field elements in proofs and hashes are 32 big-endian bytes
 * (BigInt::from_bytes_be at reference src/lib.rs:54,78,134-146,169-174; to_bytes_be on the prover's side), across
 * this ABI they are 8 little-endian uint32 limbs. n elements each way; no reduction (a 32-byte value at or above p
 * stays as it is: the transform entry points accept any 256-bit value). The buffers must not overlap.
 * ------------------------------------------------------------------------------------- */
void stark_be32_to_limbs(const uint8_t* be, size_t n, uint32_t* limbs);
void stark_limbs_to_be32(const uint32_t* limbs, size_t n, uint8_t* be);

/* ---------------------------------------------------------------------------------------
 * Test and benchmark support (no counterpart in the reference).
 * ------------------------------------------------------------------------------------- */

/* Element-wise field arithmetic on host arrays of n*8 limbs: op 0 = a+b, 1 = a-b, 2 = a*b,
 * 3 = canonical form of a (b ignored, may be NULL); 4..7 run the transform kernels' 29-bit-limb
 * arithmetic: 4 = a*b, 5 = a+b, 6 = a-b, 7 = (2a-2b)*b. Exercises the device arithmetic alone. */
int stark_field_op(stark_ctx* ctx, int op, const uint32_t* a, const uint32_t* b, uint32_t* r, size_t n);

/* stark_mimc_verify_batch_io with diagnostic switches, for tests that compare the verifier's paths and its scalar
 * checks against a restatement of src/lib.rs:31-197 one check at a time. flags is a sum of:
 *   STARK_DIAG_IGNORE_MERKLE      a Merkle mismatch (status 4) is not recorded, so the scalar checks decide alone:
 *                                 proofs with damaged values reach statuses 5..8. Disables a soundness check, so it
 *                                 exists only in the test build of the library (compiled with -DSTARKHIP_DIAG into
 *                                 libstarkhip_diag.so, `make diag`); libstarkhip.so returns STARK_ERR_UNSUPPORTED
 *                                 and verifies nothing when the flag is set.
 *   STARK_DIAG_HOST_FIELD_CHECKS  the scalar checks run on host threads in the reference's own formulation
 *                                 (Lagrange interpolation, modular division) instead of the device kernels
 *   STARK_DIAG_GENERAL_PATH       every proof is parsed on the host (no device-side fast path) */
#define STARK_DIAG_IGNORE_MERKLE 1u
#define STARK_DIAG_HOST_FIELD_CHECKS 2u
#define STARK_DIAG_GENERAL_PATH 4u
int stark_mimc_verify_batch_diag(stark_ctx* ctx, const uint8_t* proofs, const uint64_t* offsets, size_t n_proofs,
                                 const uint32_t* inputs, uint64_t num_steps, const uint32_t* round_constants,
                                 size_t n_constants, const uint32_t* outputs, unsigned flags, uint8_t* accept,
                                 uint8_t* status);

/* Fill a device buffer with n canonical field elements drawn from SplitMix64(seed): element i
 * takes draws 4i..4i+3 as its four little-endian 64-bit words, reduced mod p. */
int stark_fill_random_field_dev(stark_ctx* ctx, void* d_out, size_t n, uint64_t seed, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STARKHIP_H */
