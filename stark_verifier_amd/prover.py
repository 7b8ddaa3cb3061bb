"""Host mirror of the proof generator: MiMC STARK proofs in the reference's wire format
(src/deserializer.rs:16-140), made on the GPU many at a time.  The reference itself only verifies;
its proofs come from an external program (README.md:9)."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

from . import _lib
from .fft import MODULUS, _limbs


def mimc(inputs: Sequence[int], steps: int, round_constants: Sequence[int], ctx: _lib.Context | None = None) -> List[int]:
    """utils::mimc (reference src/utils.rs:8-16) for many inputs: steps - 1 rounds of x -> x^3 + k[i mod len] mod p."""
    ctx = ctx or _lib.default_context()
    n = len(inputs)
    inp = np.ascontiguousarray(np.concatenate([_limbs(int(x) % (1 << 256)) for x in inputs])) if n else np.zeros(8, dtype=np.uint32)
    rc = (np.ascontiguousarray(np.concatenate([_limbs(int(c) % (1 << 256)) for c in round_constants]))
          if len(round_constants) else np.zeros(8, dtype=np.uint32))
    out = np.empty(8 * max(n, 1), dtype=np.uint32)
    ctx.check(ctx.lib.stark_mimc_batch(ctx.handle, inp.ctypes.data, n, steps, rc.ctypes.data if len(round_constants) else None,
                                       len(round_constants), out.ctypes.data))
    return [int.from_bytes(out[8 * i:8 * i + 8].astype("<u4").tobytes(), "little") for i in range(n)]


def mimc_proof_size(num_steps: int, n_constants: int) -> int:
    """Bytes of one serialized proof, 0 when the parameters are rejected."""
    return int(_lib.load().stark_mimc_proof_size(num_steps, n_constants))


def mk_mimc_proofs(inputs: Sequence[int], num_steps: int, round_constants: Sequence[int],
                   ctx: _lib.Context | None = None) -> Tuple[List[bytes], List[int]]:
    """One proof per input: returns (proofs, outputs) with outputs[i] = mimc(inputs[i], num_steps,
    round_constants) (reference src/utils.rs:8-16), the value verify_mimc_proof is called with."""
    ctx = ctx or _lib.default_context()
    n = len(inputs)
    size = mimc_proof_size(num_steps, len(round_constants))
    if size == 0:
        raise _lib.InvalidArgument(_lib.STARK_ERR_INVALID, "num_steps / round_constants rejected")
    if n == 0:
        return [], []
    inp = np.ascontiguousarray(np.concatenate([_limbs(int(x) % (1 << 256)) for x in inputs]))
    rc = np.ascontiguousarray(np.concatenate([_limbs(int(c) % MODULUS) for c in round_constants]))
    proofs = np.empty(n * size, dtype=np.uint8)
    outputs = np.empty(n * 8, dtype=np.uint32)
    ctx.check(ctx.lib.stark_mimc_prove_batch(ctx.handle, inp.ctypes.data, n, num_steps, rc.ctypes.data,
                                             len(round_constants), proofs.ctypes.data, outputs.ctypes.data))
    blob = proofs.tobytes()
    outs = [int.from_bytes(outputs[8 * i:8 * i + 8].astype("<u4").tobytes(), "little") for i in range(n)]
    return [blob[i * size:(i + 1) * size] for i in range(n)], outs


def mk_mimc_proofs_dev(d_inputs: int, n: int, num_steps: int, round_constants: Sequence[int], d_proofs: int, proof_stride: int,
                       d_outputs: int, ctx: _lib.Context | None = None, stream: int | None = None) -> None:
    """Device-resident version: d_inputs / d_outputs are n x 8 limbs, proof i is written at d_proofs + i * proof_stride
    (raw device pointers, 16-byte aligned). Queued on `stream` (default: the context's); does not synchronise."""
    ctx = ctx or _lib.default_context()
    rc = np.ascontiguousarray(np.concatenate([_limbs(int(c) % MODULUS) for c in round_constants]))
    ctx.check(ctx.lib.stark_mimc_prove_batch_dev(ctx.handle, d_inputs, n, num_steps, rc.ctypes.data, len(round_constants),
                                                 d_proofs, proof_stride, d_outputs, stream))


def mk_mimc_proofs_sharded(inputs: Sequence[int], num_steps: int, round_constants: Sequence[int],
                           ctx: _lib.Context | None = None, prove=None) -> Tuple[List[bytes], List[int], Tuple[int, int]]:
    """Proof generation split over the ranks of the default process group (one process per GPU): proofs are independent,
    so rank r makes the proofs of the contiguous slice `shard_range(len(inputs), r, world)` and nothing is exchanged (a
    proof is 0.5 MB; the caller gathers them if it wants them in one place). Returns (proofs, outputs, (lo, hi)) for this
    rank's slice. `prove(inputs)` replaces the device call (CPU tests of the partitioning)."""
    from .sharding import shard_range
    import torch.distributed as dist

    on = dist.is_available() and dist.is_initialized()
    world, rank = (dist.get_world_size(), dist.get_rank()) if on else (1, 0)
    lo, hi = shard_range(len(inputs), rank, world)
    fn = prove or (lambda xs: mk_mimc_proofs(xs, num_steps, round_constants, ctx))
    proofs, outs = fn(list(inputs[lo:hi]))
    return proofs, outs, (lo, hi)
