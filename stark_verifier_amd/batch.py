"""Host mirror of the batch driver: `verify_mimc_proof` (reference src/lib.rs:99) over many
serialized proofs, Merkle branches hashed on the GPU."""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np

from . import _lib
from .fft import MODULUS, _limbs, _ptr

STATUS = {0: "accepted", 1: "parameters rejected", 2: "malformed proof", 3: "branch shape", 4: "merkle mismatch",
          5: "low-degree check", 6: "transition constraint", 7: "boundary constraint", 8: "linear combination",
          9: "instance-size assert"}


def verify_mimc_batch(proofs: Sequence[bytes], inp: int, num_steps: int, round_constants: Sequence[int], output: int,
                      ctx: _lib.Context | None = None) -> Tuple[np.ndarray, np.ndarray]:
    """Returns (accept, status): accept[i] == 1 exactly when the reference's verify_mimc_proof(inp,
    num_steps, round_constants, output, proof_i, p) returns true; status[i] is a key of STATUS."""
    ctx = ctx or _lib.default_context()
    n = len(proofs)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(p) for p in proofs], dtype=np.uint64)
    blob = np.frombuffer(b"".join(proofs), dtype=np.uint8) if n else np.zeros(0, dtype=np.uint8)
    rc = np.ascontiguousarray(np.concatenate([_limbs(c % MODULUS) for c in round_constants]) if len(round_constants)
                              else np.zeros(0, dtype=np.uint32))
    accept = np.zeros(n, dtype=np.uint8)
    status = np.zeros(n, dtype=np.uint8)
    i, o = _limbs(inp % MODULUS), _limbs(output % MODULUS)
    ctx.check(ctx.lib.stark_mimc_verify_batch(ctx.handle, blob.ctypes.data if n else None, offsets.ctypes.data, n, _ptr(i),
                                              num_steps, rc.ctypes.data if rc.size else None, len(round_constants), _ptr(o),
                                              accept.ctypes.data if n else None, status.ctypes.data if n else None))
    return accept, status


def verify_mimc_batch_io(proofs: Sequence[bytes], inputs: Sequence[int], num_steps: int, round_constants: Sequence[int],
                         outputs: Sequence[int], ctx: _lib.Context | None = None) -> Tuple[np.ndarray, np.ndarray]:
    """verify_mimc_batch with one (input, output) pair per proof: proof i is checked as the reference's
    verify_mimc_proof(inputs[i], num_steps, round_constants, outputs[i], proof_i, p)."""
    ctx = ctx or _lib.default_context()
    n = len(proofs)
    if len(inputs) != n or len(outputs) != n:
        raise _lib.InvalidArgument(_lib.STARK_ERR_INVALID, "one input and one output per proof")
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(p) for p in proofs], dtype=np.uint64)
    blob = np.frombuffer(b"".join(proofs), dtype=np.uint8) if n else np.zeros(0, dtype=np.uint8)
    rc = np.ascontiguousarray(np.concatenate([_limbs(c % MODULUS) for c in round_constants]) if len(round_constants)
                              else np.zeros(0, dtype=np.uint32))
    i = np.ascontiguousarray(np.concatenate([_limbs(int(x) % MODULUS) for x in inputs])) if n else np.zeros(8, dtype=np.uint32)
    o = np.ascontiguousarray(np.concatenate([_limbs(int(x) % MODULUS) for x in outputs])) if n else np.zeros(8, dtype=np.uint32)
    accept = np.zeros(n, dtype=np.uint8)
    status = np.zeros(n, dtype=np.uint8)
    ctx.check(ctx.lib.stark_mimc_verify_batch_io(ctx.handle, blob.ctypes.data if n else None, offsets.ctypes.data, n,
                                                 i.ctypes.data, num_steps, rc.ctypes.data if rc.size else None,
                                                 len(round_constants), o.ctypes.data, accept.ctypes.data if n else None,
                                                 status.ctypes.data if n else None))
    return accept, status


DIAG_IGNORE_MERKLE, DIAG_HOST_FIELD_CHECKS, DIAG_GENERAL_PATH = 1, 2, 4


def verify_mimc_batch_diag(proofs: Sequence[bytes], inputs: Sequence[int], num_steps: int, round_constants: Sequence[int],
                           outputs: Sequence[int], flags: int, ctx: _lib.Context | None = None) -> Tuple[np.ndarray, np.ndarray]:
    """Test support: verify_mimc_batch_io with the library's diagnostic switches (a sum of DIAG_*).
    DIAG_IGNORE_MERKLE lets the scalar checks decide even when a branch mismatched; never use it to verify."""
    ctx = ctx or _lib.default_context()
    n = len(proofs)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(p) for p in proofs], dtype=np.uint64)
    blob = np.frombuffer(b"".join(proofs), dtype=np.uint8) if n else np.zeros(0, dtype=np.uint8)
    rc = np.ascontiguousarray(np.concatenate([_limbs(c % MODULUS) for c in round_constants]) if len(round_constants)
                              else np.zeros(0, dtype=np.uint32))
    i = np.ascontiguousarray(np.concatenate([_limbs(int(x) % MODULUS) for x in inputs])) if n else np.zeros(8, dtype=np.uint32)
    o = np.ascontiguousarray(np.concatenate([_limbs(int(x) % MODULUS) for x in outputs])) if n else np.zeros(8, dtype=np.uint32)
    accept = np.zeros(n, dtype=np.uint8)
    status = np.zeros(n, dtype=np.uint8)
    ctx.check(ctx.lib.stark_mimc_verify_batch_diag(ctx.handle, blob.ctypes.data if n else None, offsets.ctypes.data, n,
                                                   i.ctypes.data, num_steps, rc.ctypes.data if rc.size else None,
                                                   len(round_constants), o.ctypes.data, flags,
                                                   accept.ctypes.data if n else None, status.ctypes.data if n else None))
    return accept, status


def verify_mimc_batch_dev(d_proofs: int, proof_stride: int, n: int, d_inputs: int, num_steps: int,
                          round_constants: Sequence[int], d_outputs: int,
                          ctx: _lib.Context | None = None) -> Tuple[np.ndarray, np.ndarray]:
    """verify_mimc_batch_io for proofs, inputs and outputs already in device memory (raw device pointers, e.g. what
    prover.mk_mimc_proofs_dev / stark_mimc_prove_batch_dev left there): proof i is the proof_stride bytes at
    d_proofs + i * proof_stride. Returns host arrays (accept, status)."""
    ctx = ctx or _lib.default_context()
    rc = np.ascontiguousarray(np.concatenate([_limbs(c % MODULUS) for c in round_constants]) if len(round_constants)
                              else np.zeros(0, dtype=np.uint32))
    accept = np.zeros(n, dtype=np.uint8)
    status = np.zeros(n, dtype=np.uint8)
    ctx.check(ctx.lib.stark_mimc_verify_batch_dev(ctx.handle, d_proofs, proof_stride, n, d_inputs, num_steps,
                                                  rc.ctypes.data if rc.size else None, len(round_constants), d_outputs,
                                                  accept.ctypes.data if n else None, status.ctypes.data if n else None))
    return accept, status


def verify_mimc_batch_sharded(proofs: Sequence[bytes], inp: int, num_steps: int, round_constants: Sequence[int], output: int,
                              ctx: _lib.Context | None = None, verify=None):
    """The same batch split over the ranks of the default process group (one process per GPU): proofs are
    independent (the reference's verify_mimc_proof keeps no state between calls, src/lib.rs:99), so rank r
    verifies the contiguous slice `shard_range(len(proofs), r, world)` and the per-proof flags are gathered
    (the only exchange). Every rank gets the full (accept, status) vectors. `verify` replaces the device
    call (CPU tests of the partitioning)."""
    from .sharding import gather_accept_bits, shard_range
    import torch.distributed as dist

    on = dist.is_available() and dist.is_initialized()
    world, rank = (dist.get_world_size(), dist.get_rank()) if on else (1, 0)
    lo, hi = shard_range(len(proofs), rank, world)
    fn = verify or (lambda blobs: verify_mimc_batch(blobs, inp, num_steps, round_constants, output, ctx))
    accept, status = fn(list(proofs[lo:hi]))
    return gather_accept_bits(accept, len(proofs)), gather_accept_bits(status, len(proofs))


def verify_mimc_batch_io_sharded(proofs: Sequence[bytes], inputs: Sequence[int], num_steps: int, round_constants: Sequence[int],
                                 outputs: Sequence[int], ctx: _lib.Context | None = None, verify=None):
    """verify_mimc_batch_io split over the ranks of the default process group: rank r verifies the contiguous slice
    `shard_range(len(proofs), r, world)` with its own inputs and outputs, and the per-proof flags are gathered (the only
    exchange). `verify(blobs, inputs, outputs)` replaces the device call (CPU tests of the partitioning)."""
    from .sharding import gather_accept_bits, shard_range
    import torch.distributed as dist

    on = dist.is_available() and dist.is_initialized()
    world, rank = (dist.get_world_size(), dist.get_rank()) if on else (1, 0)
    lo, hi = shard_range(len(proofs), rank, world)
    fn = verify or (lambda b, i, o: verify_mimc_batch_io(b, i, num_steps, round_constants, o, ctx))
    accept, status = fn(list(proofs[lo:hi]), list(inputs[lo:hi]), list(outputs[lo:hi]))
    return gather_accept_bits(accept, len(proofs)), gather_accept_bits(status, len(proofs))
