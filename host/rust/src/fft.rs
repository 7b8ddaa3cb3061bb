//! New body of fft::fft_inv (reference src/fft.rs:64-86; signature unchanged) and the forward transform the crate
//! only has as the private _fft (src/fft.rs:37-61). Source only: not compiled in the repository it ships in.
use crate::ffi;
use num_bigint::BigInt;

fn transform(v: &Vec<BigInt>, root_of_unity: &BigInt, modulus: &BigInt, inverse: bool) -> Vec<BigInt> {
    let flat: Vec<u32> = v.iter().flat_map(|x| ffi::to_limbs(x, modulus)).collect();
    let (r, m) = (ffi::to_limbs(root_of_unity, modulus), ffi::modulus_limbs(modulus));
    let mut out = vec![0u32; flat.len()];
    let rc = ffi::CTX.with(|c| unsafe {
        if inverse {
            ffi::stark_fft_inv(*c, flat.as_ptr(), v.len(), r.as_ptr(), m.as_ptr(), out.as_mut_ptr())
        } else {
            ffi::stark_fft(*c, flat.as_ptr(), v.len(), r.as_ptr(), m.as_ptr(), out.as_mut_ptr())
        }
    });
    // the original indexes out of bounds (panics) for a length that is not the order of the root (src/fft.rs:19-21,74-77)
    assert!(rc == ffi::STARK_OK, "fft: invalid length / root of unity (status {})", rc);
    out.chunks(8).map(ffi::from_limbs).collect()
}

pub fn fft_inv(v: &Vec<BigInt>, root_of_unity: &BigInt, modulus: &BigInt) -> Vec<BigInt> {
    transform(v, root_of_unity, modulus, true)
}

pub fn fft(v: &Vec<BigInt>, root_of_unity: &BigInt, modulus: &BigInt) -> Vec<BigInt> {
    transform(v, root_of_unity, modulus, false)
}
