"""Batch sharding across the GPUs of one node (one process per GPU).

The hot path partitions: polynomials, branches and proofs are independent units (the
reference verifies branches one by one in a loop, src/merkle_tree.rs:35-41, and has no state
shared between proofs, src/lib.rs:99), so every rank works on a contiguous slice with no
data-path collective.  The only exchange is the final gather of per-unit accept flags.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of `total` units owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# Tests set this to run the collectives even in a one-rank process group (the RCCL code path on a one-GPU box).
ALWAYS_COLLECTIVE = False


def gather_accept_bits(local_accept, total: int, device=None):
    """All ranks contribute their slice of accept flags; every rank gets the full [total] uint8 vector.

    Uses torch.distributed (RCCL when the process group is "nccl", gloo on CPU).  With a single
    process (no initialised process group) it returns the input unchanged. RCCL only moves device
    memory: under "nccl" the flags go through the GPU of this process (`device`, default
    _lib.current_device()).
    """
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not ALWAYS_COLLECTIVE):
        return np.asarray(local_accept, dtype=np.uint8)
    world, rank = dist.get_world_size(), dist.get_rank()
    if isinstance(local_accept, torch.Tensor):
        t = local_accept.to(torch.uint8)
    else:
        t = torch.from_numpy(np.ascontiguousarray(local_accept, dtype=np.uint8))
    if device is None and dist.get_backend() == "nccl" and not t.is_cuda:
        from . import _lib
        device = torch.device("cuda", _lib.current_device())
    if device is not None:
        t = t.to(device)
    # all_gather wants equal sizes: pad every slice to the largest one
    width = (total + world - 1) // world
    padded = torch.zeros(width, dtype=torch.uint8, device=t.device)
    padded[: t.numel()] = t
    # one collective into one buffer and one copy to the host (a list of per-rank tensors costs a copy and a
    # synchronisation per rank)
    flat = torch.empty(world * width, dtype=torch.uint8, device=t.device)
    dist.all_gather_into_tensor(flat, padded)
    host = flat.cpu().numpy().reshape(world, width)
    out = np.empty(total, dtype=np.uint8)
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        out[lo:hi] = host[r, : hi - lo]
    return out


# ---- one transform sharded over several ranks -------------------------------------------------
#
# n = n1 * n2, input index j = j1 + n1*j2, output index k = k1*n2 + k2:
#   X[k1*n2 + k2] = sum_j1 w_n1^(j1*k1) * [ w^(j1*k2) * sum_j2 x[j1 + n1*j2] * w_n2^(j2*k2) ]
# which is the recursion of the reference's _fft (src/fft.rs:37-61) regrouped into two stages of
# independent transforms. Rank g owns a range of j1 for stage A and the twiddle step, a range of
# k2 for stage C; one all-to-all moves the data between the two ownerships. Contiguous ("blocked")
# input or output costs one more all-to-all each, because a contiguous slice of x is a range of
# j2 (not j1) and a contiguous slice of X is a range of k1 (not k2).

class HipLocalOps:
    """Device work of the sharded transform through the C ABI, on torch CUDA tensors of shape
    [elements, 8] int32. The library's launches, torch's allocations / copies and the collectives
    all go to the context's own stream, which torch sees as an ExternalStream; `stream_context`
    makes it current and orders it after, and the caller's stream after, the work around it."""

    def __init__(self, ctx=None):
        import torch
        from . import _lib
        self.ctx = ctx or _lib.default_context()
        self.tstream = torch.cuda.ExternalStream(self.ctx.stream)

    def stream_context(self):
        import contextlib
        import torch

        @contextlib.contextmanager
        def cm():
            outer = torch.cuda.current_stream()
            self.tstream.wait_stream(outer)
            with torch.cuda.stream(self.tstream):
                yield
            outer.wait_stream(self.tstream)
        return cm()

    def _stream(self):
        return self.tstream.cuda_stream

    def ntt_batch(self, t, log_m: int, batch: int, root: int, inverse: bool):
        from . import fft
        fft.ntt_batch_dev(t.data_ptr(), t.data_ptr(), log_m, batch, root, inverse, self.ctx, self._stream())

    def twiddle_mul(self, t, rows: int, cols: int, row0: int, col0: int, log_order: int, root: int, inverse: bool):
        from . import fft
        r = fft._limbs(root % fft.MODULUS)
        self.ctx.check(self.ctx.lib.stark_twiddle_mul_dev(self.ctx.handle, t.data_ptr(), rows, cols, row0, col0,
                                                          log_order, fft._ptr(r), int(bool(inverse)), self._stream() or None))

    def permute3(self, t, a: int, b: int, chunk: int):
        import torch
        out = torch.empty_like(t)
        self.ctx.check(self.ctx.lib.stark_permute3_dev(self.ctx.handle, t.data_ptr(), out.data_ptr(), a, b, chunk,
                                                       self._stream() or None))
        return out

    def stage_a(self, t, log_n: int, world: int, rank: int, root: int, inverse: bool):
        """Rows [r1][n2] -> send blocks [world][r1][r2]: row transforms, twiddles and packing in one pass sequence."""
        import torch
        from . import fft
        out = torch.empty_like(t)
        r = fft._limbs(root % fft.MODULUS)
        self.ctx.check(self.ctx.lib.stark_ntt_sharded_stage_a_dev(self.ctx.handle, t.data_ptr(), out.data_ptr(), log_n, world,
                                                                  rank, fft._ptr(r), int(bool(inverse)), self._stream() or None))
        return out

    def stage_c(self, t, log_n: int, world: int, root: int, inverse: bool):
        """Received blocks [n1][r2] -> "columns" result [n1][r2]: column transforms where the data lies."""
        import torch
        from . import fft
        out = torch.empty_like(t)
        r = fft._limbs(root % fft.MODULUS)
        self.ctx.check(self.ctx.lib.stark_ntt_sharded_stage_c_dev(self.ctx.handle, t.data_ptr(), out.data_ptr(), log_n, world,
                                                                  fft._ptr(r), int(bool(inverse)), self._stream() or None))
        return out


def _all_to_all(t, group):
    """Equal-split all-to-all of a [world * m, 8] tensor along dim 0. Device tensors go through RCCL
    when the group's backend is "nccl"; any other backend (gloo) is staged through host memory."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (ALWAYS_COLLECTIVE and dist.is_initialized()):
        return t
    if t.is_cuda and dist.get_backend(group) != "nccl":
        src = t.cpu()
        dst = torch.empty_like(src)
        dist.all_to_all_single(dst, src, group=group)
        return dst.to(t.device)
    out = torch.empty_like(t)
    dist.all_to_all_single(out, t.contiguous(), group=group)
    return out


def _redistribute(t, rows: int, cols: int, world: int, group, ops):
    """t is this rank's [rows][cols] slab of a [world*rows][cols] matrix split by rows. Returns this
    rank's [cols/world][world*rows] slab of the transposed matrix (split by rows again)."""
    packed = ops.permute3(t, rows, world, cols // world) if world > 1 else t   # [dest][rows][cols/world]
    got = _all_to_all(packed, group)                                              # [src][rows][cols/world]
    return ops.permute3(got, world * rows, cols // world, 1)                      # [cols/world][world*rows]


def sharded_split(log_n: int):
    """(log n1, log n2) used by ntt_sharded."""
    return log_n // 2, log_n - log_n // 2


def ntt_sharded(x_local, log_n: int, root_of_unity: int, inverse: bool = False, group=None,
                in_layout: str = "blocked", out_layout: str = "blocked", ops=None, fused=None):
    """One 2^log_n-point transform whose data is spread over the ranks of `group`.

    x_local: this rank's [n/world, 8] int32 tensor of limbs.
      in_layout "blocked": the contiguous slice x[rank*n/world : (rank+1)*n/world].
      in_layout "strided": row-major [n1/world][n2] with element (a, j2) = x[(rank*n1/world + a) + n1*j2].
    Returns this rank's [n/world, 8] tensor of the result X (fft_inv semantics if `inverse`):
      out_layout "blocked": X[rank*n/world : (rank+1)*n/world].
      out_layout "strided": row-major [n2/world][n1] with element (b, k1) = X[k1*n2 + rank*n2/world + b].
      out_layout "columns": row-major [n1][n2/world] with element (k1, b) = X[k1*n2 + rank*n2/world + b]
                            (the transpose of "strided"; what the fused path produces without a further step).
    "strided" in and "strided" / "columns" out need a single all-to-all; each "blocked" side adds one.
    world must divide both n1 = 2^(log_n // 2) and n2 = n / n1. With no process group it runs the
    same steps on one rank.

    fused (default: whenever `ops` offers stage_a / stage_c and the size allows): stage A multiplies by the twiddles
    in its last pass's store and writes the send blocks directly, stage C transforms the received matrix's columns
    where they lie (the library's stark_ntt_sharded_stage_{a,c}_dev). Otherwise the three-step form: row
    transforms, stark_twiddle_mul_dev, pack / transpose kernels around the all-to-all, row transforms.
    """
    import contextlib
    import torch.distributed as dist

    if in_layout not in ("blocked", "strided") or out_layout not in ("blocked", "strided", "columns"):
        raise ValueError("in_layout is 'blocked' or 'strided', out_layout 'blocked', 'strided' or 'columns'")
    ops = ops or HipLocalOps()
    on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    l1, l2 = sharded_split(log_n)
    n1, n2, n = 1 << l1, 1 << l2, 1 << log_n
    if n1 % world or n2 % world:
        raise ValueError(f"world size {world} must divide n1={n1} and n2={n2}")
    if tuple(x_local.shape) != (n // world, 8):
        raise ValueError(f"x_local must have shape ({n // world}, 8), got {tuple(x_local.shape)}")
    from .fft import MODULUS
    w = root_of_unity % MODULUS
    r1, r2 = n1 // world, n2 // world
    if fused is None:
        fused = hasattr(ops, "stage_a") and l1 >= 2 and world & (world - 1) == 0

    with getattr(ops, "stream_context", contextlib.nullcontext)():
        if in_layout == "blocked":
            # local slab [j2 in my range][j1] -> [j1 in my range][j2]
            s = _redistribute(x_local, r2, n1, world, group, ops)
        else:
            s = x_local if fused else x_local.clone()
        u = None
        if fused:
            try:
                send = ops.stage_a(s, log_n, world, rank, w, inverse)               # [dest][j1 in my range][k2 in dest's]
            except Exception as e:  # sizes the fused store does not take: the three-step form below
                if getattr(e, "code", None) != -4:
                    raise
                fused = False
                s = s.clone() if s is x_local else s
            else:
                got = _all_to_all(send, group)                                      # [n1][k2 in my range]
                u = ops.stage_c(got, log_n, world, w, inverse)                      # "columns"
                if out_layout == "strided":
                    u = ops.permute3(u, n1, r2, 1)                                  # [k2 in my range][k1]
                elif out_layout == "blocked":
                    # rows k1 of rank d are the contiguous block d of the columns result: exchange, then interleave
                    u = ops.permute3(_all_to_all(u, group), world, r1, r2) if world > 1 else u
        if not fused:
            ops.ntt_batch(s, l2, r1, pow(w, n1, MODULUS), inverse)                  # stage A over j2
            ops.twiddle_mul(s, r1, n2, rank * r1, 0, log_n, w, inverse)             # w^(+-j1*k2)
            u = _redistribute(s, r1, n2, world, group, ops)                         # [k2 in my range][j1]
            ops.ntt_batch(u, l1, r2, pow(w, n2, MODULUS), inverse)                  # stage C over j1
            if out_layout == "blocked":
                u = _redistribute(u, r2, n1, world, group, ops)                     # [k1 in my range][k2]
            elif out_layout == "columns":
                u = ops.permute3(u, r2, n1, 1)
        if hasattr(u, "record_stream") and u.is_cuda:
            import torch
            u.record_stream(torch.cuda.current_stream())
    return u
