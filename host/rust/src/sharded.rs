//! One transform spread over the GPUs of a node (one process per GPU): stark_ntt_sharded_dev with an RCCL all-to-all
//! as the exchange callback. The RCCL declarations are the usual nccl.h ones (librccl.so). Source only: not compiled
//! in the repository it ships in.
use crate::ffi;
use std::os::raw::{c_int, c_void};

#[allow(non_camel_case_types)]
pub type ncclComm_t = *mut c_void;
const NCCL_UINT8: c_int = 1; // ncclUint8

extern "C" {
    fn ncclGroupStart() -> c_int;
    fn ncclGroupEnd() -> c_int;
    fn ncclSend(sendbuff: *const c_void, count: usize, datatype: c_int, peer: c_int, comm: ncclComm_t, stream: *mut c_void) -> c_int;
    fn ncclRecv(recvbuff: *mut c_void, count: usize, datatype: c_int, peer: c_int, comm: ncclComm_t, stream: *mut c_void) -> c_int;
}

pub struct Exchange {
    pub comm: ncclComm_t,
    pub world: usize,
}

/// Equal-block all-to-all over xGMI: one send and one receive per peer in a single RCCL group, so all seven links of
/// a GPU carry one block each at the same time.
unsafe extern "C" fn all_to_all(user: *mut c_void, d_send: *const c_void, d_recv: *mut c_void, block_bytes: usize, stream: *mut c_void) -> c_int {
    let ex = &*(user as *const Exchange);
    let mut rc = ncclGroupStart();
    for peer in 0..ex.world {
        if rc == 0 {
            rc = ncclSend((d_send as *const u8).add(peer * block_bytes) as *const c_void, block_bytes, NCCL_UINT8, peer as c_int, ex.comm, stream);
        }
        if rc == 0 {
            rc = ncclRecv((d_recv as *mut u8).add(peer * block_bytes) as *mut c_void, block_bytes, NCCL_UINT8, peer as c_int, ex.comm, stream);
        }
    }
    let end = ncclGroupEnd();
    if rc != 0 { rc } else { end }
}

/// d_x: this rank's "rows" slice [n1 / world][n2] of x (element (a, j2) = x[(rank * n1 / world + a) + n1 * j2]), n / world
/// elements of 32 bytes, overwritten; d_y: a second buffer of the same size. The "columns" slice of the result,
/// [n1][n2 / world] with element (k1, b) = X[k1 * n2 + rank * n2 / world + b], is left in d_y (the exchange runs on a one-rank communicator too).
pub fn ntt_sharded(d_x: *mut c_void, d_y: *mut c_void, log_n: u32, rank: usize, ex: &Exchange, root: &[u32; 8], inverse: bool) {
    let rc = ffi::CTX.with(|c| unsafe {
        ffi::stark_ntt_sharded_dev(*c, d_x, d_y, log_n, ex.world as u32, rank as u32, root.as_ptr(), inverse as c_int,
                                   Some(all_to_all), ex as *const Exchange as *mut c_void, std::ptr::null_mut())
    });
    assert!(rc == ffi::STARK_OK, "stark_ntt_sharded_dev failed (status {})", rc);
}
