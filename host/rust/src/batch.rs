//! Batch functions beside verify_mimc_proof (reference src/lib.rs:99): many serialized proofs per call.
//! Source only: not compiled in the repository it ships in.
use crate::ffi;
use num_bigint::BigInt;

fn limbs(v: &[BigInt], modulus: &BigInt) -> Vec<u32> {
    v.iter().flat_map(|x| ffi::to_limbs(x, modulus)).collect()
}

/// verify_mimc_proof over many serialized proofs (the bytes deserializer::from_bytes reads, src/deserializer.rs:142),
/// one (input, output) pair per proof. true exactly where verify_mimc_proof returns true; where it returns false or
/// panics: false.
pub fn verify_mimc_proofs(inputs: &[BigInt], num_steps: usize, round_constants: &[BigInt], outputs: &[BigInt],
                          proofs: &[Vec<u8>], modulus: &BigInt) -> Vec<bool> {
    assert!(inputs.len() == proofs.len() && outputs.len() == proofs.len());
    let mut offsets = vec![0u64];
    for p in proofs {
        offsets.push(offsets.last().unwrap() + p.len() as u64);
    }
    let blob: Vec<u8> = proofs.concat();
    let (i, o, c) = (limbs(inputs, modulus), limbs(outputs, modulus), limbs(round_constants, modulus));
    let mut accept = vec![0u8; proofs.len()];
    let rc = ffi::CTX.with(|ctx| unsafe {
        ffi::stark_mimc_verify_batch_io(*ctx, blob.as_ptr(), offsets.as_ptr(), proofs.len(), i.as_ptr(), num_steps as u64, c.as_ptr(),
                                        round_constants.len(), o.as_ptr(), accept.as_mut_ptr(), std::ptr::null_mut())
    });
    assert!(rc == ffi::STARK_OK, "stark_mimc_verify_batch_io failed (status {})", rc);
    accept.iter().map(|&a| a == 1).collect()
}

/// One proof per input, in the format from_bytes reads; also returns mimc(input) per proof (utils::mimc, src/utils.rs:8).
pub fn mk_mimc_proofs(inputs: &[BigInt], num_steps: usize, round_constants: &[BigInt], modulus: &BigInt) -> (Vec<Vec<u8>>, Vec<BigInt>) {
    let size = unsafe { ffi::stark_mimc_proof_size(num_steps as u64, round_constants.len()) };
    assert!(size != 0, "num_steps / round_constants rejected");
    let (i, c) = (limbs(inputs, modulus), limbs(round_constants, modulus));
    let mut blob = vec![0u8; size * inputs.len()];
    let mut out = vec![0u32; 8 * inputs.len()];
    let rc = ffi::CTX.with(|ctx| unsafe {
        ffi::stark_mimc_prove_batch(*ctx, i.as_ptr(), inputs.len(), num_steps as u64, c.as_ptr(), round_constants.len(),
                                    blob.as_mut_ptr(), out.as_mut_ptr())
    });
    assert!(rc == ffi::STARK_OK, "stark_mimc_prove_batch failed (status {})", rc);
    (blob.chunks(size).map(|p| p.to_vec()).collect(), out.chunks(8).map(ffi::from_limbs).collect())
}
