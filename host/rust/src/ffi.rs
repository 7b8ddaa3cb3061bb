//! extern "C" declarations of include/starkhip.h and the conversions between the crate's BigInt values and the
//! library's field elements (8 little-endian u32 limbs). Source only: not compiled in the repository it ships in.
use num_bigint::{BigInt, Sign};
use std::os::raw::{c_char, c_int, c_uint, c_void};

#[repr(C)]
pub struct StarkCtx {
    _p: [u8; 0],
}

pub const STARK_OK: c_int = 0;

/// int (*stark_all_to_all_fn)(void* user, const void* d_send, void* d_recv, size_t block_bytes, void* stream)
pub type AllToAllFn =
    unsafe extern "C" fn(user: *mut c_void, d_send: *const c_void, d_recv: *mut c_void, block_bytes: usize, stream: *mut c_void) -> c_int;

extern "C" {
    pub fn stark_ctx_create(device: c_int, out: *mut *mut StarkCtx) -> c_int;
    pub fn stark_ctx_destroy(ctx: *mut StarkCtx);
    pub fn stark_last_error(ctx: *const StarkCtx) -> *const c_char;
    pub fn stark_ctx_stream(ctx: *mut StarkCtx) -> *mut c_void;
    pub fn stark_ctx_synchronize(ctx: *mut StarkCtx) -> c_int;

    // fft.rs
    pub fn stark_fft_inv(ctx: *mut StarkCtx, v: *const u32, n: usize, root: *const u32, modulus: *const u32, out: *mut u32) -> c_int;
    pub fn stark_fft(ctx: *mut StarkCtx, v: *const u32, n: usize, root: *const u32, modulus: *const u32, out: *mut u32) -> c_int;
    pub fn stark_ntt_batch(ctx: *mut StarkCtx, limbs: *mut u32, log_n: c_uint, batch: usize, root: *const u32, inverse: c_int) -> c_int;
    pub fn stark_ntt_batch_dev(ctx: *mut StarkCtx, d_in: *const c_void, d_out: *mut c_void, log_n: c_uint, batch: usize,
                               root: *const u32, inverse: c_int, stream: *mut c_void) -> c_int;
    pub fn stark_ntt_batch_dev_static(ctx: *mut StarkCtx, d_in: *const c_void, d_out: *mut c_void, log_n: c_uint, batch: usize,
                                      root: *const u32, inverse: c_int, stream: *mut c_void) -> c_int;
    pub fn stark_ntt_sharded_stage_a_dev(ctx: *mut StarkCtx, d_in: *const c_void, d_send: *mut c_void, log_n: c_uint, world: c_uint,
                                         rank: c_uint, root: *const u32, inverse: c_int, stream: *mut c_void) -> c_int;
    pub fn stark_ntt_sharded_stage_c_dev(ctx: *mut StarkCtx, d_recv: *const c_void, d_out: *mut c_void, log_n: c_uint, world: c_uint,
                                         root: *const u32, inverse: c_int, stream: *mut c_void) -> c_int;
    pub fn stark_ntt_sharded_dev(ctx: *mut StarkCtx, d_x: *mut c_void, d_y: *mut c_void, log_n: c_uint, world: c_uint, rank: c_uint,
                                 root: *const u32, inverse: c_int, exchange: Option<AllToAllFn>, user: *mut c_void,
                                 stream: *mut c_void) -> c_int;

    // merkle_tree.rs, utils.rs
    pub fn stark_merkle_verify_batch(ctx: *mut StarkCtx, values: *const u8, siblings: *const u8, value_size: u32,
                                     witnesses: *const u8, num_witnesses: u32, indices: *const u32, roots: *const u8,
                                     num_roots: u32, root_ids: *const u32, n: usize, accept: *mut u8, digests: *mut u8) -> c_int;
    pub fn stark_pseudorandom_indices(ctx: *mut StarkCtx, seeds: *const u8, n: usize, count: u32, modulus: u32,
                                      exclude_multiples_of: u32, out: *mut u32) -> c_int;

    // lib.rs
    pub fn stark_mimc_verify_batch_io(ctx: *mut StarkCtx, proofs: *const u8, offsets: *const u64, n_proofs: usize,
                                      inputs: *const u32, num_steps: u64, round_constants: *const u32, n_constants: usize,
                                      outputs: *const u32, accept: *mut u8, status: *mut u8) -> c_int;
    pub fn stark_mimc_proof_size(num_steps: u64, n_constants: usize) -> usize;
    pub fn stark_mimc_prove_batch(ctx: *mut StarkCtx, inputs: *const u32, n: usize, num_steps: u64, round_constants: *const u32,
                                  n_constants: usize, proofs: *mut u8, outputs: *mut u32) -> c_int;
    pub fn stark_mimc_batch(ctx: *mut StarkCtx, inputs: *const u32, n: usize, steps: u64, round_constants: *const u32,
                            n_constants: usize, outputs: *mut u32) -> c_int;

    // 32 big-endian bytes (BigInt::from_bytes_be / to_bytes_be) <-> limbs
    pub fn stark_be32_to_limbs(be: *const u8, n: usize, limbs: *mut u32);
    pub fn stark_limbs_to_be32(limbs: *const u32, n: usize, be: *mut u8);
}

/// The GPU this process works on: STARKHIP_DEVICE if set, else LOCAL_RANK (one process per GPU under a launcher that
/// shows every GPU to every rank), else 0 (also right when each rank is given one GPU through HIP_VISIBLE_DEVICES).
pub fn device_index() -> c_int {
    for var in ["STARKHIP_DEVICE", "LOCAL_RANK"] {
        if let Ok(v) = std::env::var(var) {
            if let Ok(d) = v.trim().parse::<c_int>() {
                return d;
            }
        }
    }
    0
}

thread_local! {
    /// One context per host thread (a context is not thread-safe), on device_index().
    pub static CTX: *mut StarkCtx = unsafe {
        let mut c = std::ptr::null_mut();
        assert_eq!(stark_ctx_create(device_index(), &mut c), STARK_OK, "no HIP device / libstarkhip failed to start");
        c
    };
}

/// BigInt (any sign) -> canonical residue as 8 little-endian u32 limbs
pub fn to_limbs(x: &BigInt, p: &BigInt) -> [u32; 8] {
    let r = crate::utils::negative_to_positive(x, p); // src/utils.rs:25
    let (_, bytes) = r.to_bytes_le();
    let mut b = [0u8; 32];
    b[..bytes.len()].copy_from_slice(&bytes);
    let mut l = [0u32; 8];
    for i in 0..8 {
        l[i] = u32::from_le_bytes([b[4 * i], b[4 * i + 1], b[4 * i + 2], b[4 * i + 3]]);
    }
    l
}

pub fn from_limbs(l: &[u32]) -> BigInt {
    let mut b = [0u8; 32];
    for i in 0..8 {
        b[4 * i..4 * i + 4].copy_from_slice(&l[i].to_le_bytes());
    }
    BigInt::from_bytes_le(Sign::Plus, &b)
}

/// The modulus itself as limbs (to_limbs would reduce it to zero)
pub fn modulus_limbs(p: &BigInt) -> [u32; 8] {
    to_limbs(p, &(p + 1))
}
