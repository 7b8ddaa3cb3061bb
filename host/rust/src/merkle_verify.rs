//! New bodies of MultiProof::verify (reference src/merkle_tree.rs:25-44) and ProofBranch::verify (:119-171);
//! signatures unchanged. Source only: not compiled in the repository it ships in.
use crate::ffi;
use crate::merkle_tree::{MerkleDigest, MultiProof, ProofBranch, Value};
use std::collections::BTreeMap;

impl MultiProof {
    pub fn verify(&self, indices: &[u32], rt: Option<MerkleDigest>) -> Option<Vec<Value>> {
        assert!(self.branches.len() == indices.len(), "branches len {} != indices len {}", self.branches.len(), indices.len());
        let root = match rt {
            Some(r) => r,
            None => self.root,
        };
        // one device call per (value size, witness count); witnesses level-major: [level][branch][32]
        let mut groups: BTreeMap<(usize, usize), Vec<usize>> = BTreeMap::new();
        for (i, b) in self.branches.iter().enumerate() {
            groups.entry((b.value.len(), b.witnesses.len())).or_default().push(i);
        }
        for ((vs, w), ids) in groups {
            let n = ids.len();
            let (mut values, mut sibs, mut wit) = (vec![0u8; n * vs], vec![0u8; n * vs], vec![0u8; w * n * 32]);
            let idx: Vec<u32> = ids.iter().map(|&i| indices[i]).collect();
            for (k, &i) in ids.iter().enumerate() {
                let b = &self.branches[i];
                values[k * vs..(k + 1) * vs].copy_from_slice(&b.value);
                // the reference hashes whatever length the sibling has (:131); the wire format gives both the same
                sibs[k * vs..(k + 1) * vs].copy_from_slice(&b.sibling_value);
                for l in 0..w {
                    wit[(l * n + k) * 32..(l * n + k + 1) * 32].copy_from_slice(&b.witnesses[l]);
                }
            }
            let mut accept = vec![0u8; n];
            let rc = ffi::CTX.with(|c| unsafe {
                ffi::stark_merkle_verify_batch(*c, values.as_ptr(), sibs.as_ptr(), vs as u32, wit.as_ptr(), w as u32, idx.as_ptr(),
                                               root.as_ptr(), 1, std::ptr::null(), n, accept.as_mut_ptr(), std::ptr::null_mut())
            });
            // the reference panics on these shapes before hashing (:113 division by zero, :133 slice)
            assert!(rc == ffi::STARK_OK, "merkle verify: invalid branch shape (status {})", rc);
            // reference: assert!(&res[0..32] == root, ...) at :165
            assert!(accept.iter().all(|&a| a == 1), "values didn't match up");
        }
        Some(self.branches.iter().map(|b| b.value.clone()).collect())
    }
}

impl ProofBranch {
    pub fn verify(&self, root: &MerkleDigest, a: u32) -> Option<Value> {
        let (vs, w) = (self.value.len(), self.witnesses.len());
        let wit: Vec<u8> = self.witnesses.iter().flat_map(|d| d.iter().copied()).collect();
        let mut accept = [0u8; 1];
        let rc = ffi::CTX.with(|c| unsafe {
            ffi::stark_merkle_verify_batch(*c, self.value.as_ptr(), self.sibling_value.as_ptr(), vs as u32, wit.as_ptr(), w as u32, &a,
                                           root.as_ptr(), 1, std::ptr::null(), 1, accept.as_mut_ptr(), std::ptr::null_mut())
        });
        assert!(rc == ffi::STARK_OK, "merkle verify: invalid branch shape (status {})", rc);
        assert!(accept[0] == 1, "values didn't match up");
        Some(self.value.clone())
    }
}
