"""The allocation-free transform entry point (stark_ntt_batch_dev_static) against stark_ntt_batch_dev, through the
C ABI: identical bytes on two calls in a row, the second with a smaller input; its size limit and its requirement
that the twiddle tables exist are errors, not faults."""
import ctypes

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
P = oracle.P


def _fill(ctx, count, seed):
    t = torch.empty((count, 8), dtype=torch.int32, device="cuda")
    ctx.check(ctx.lib.stark_fill_random_field_dev(ctx.handle, t.data_ptr(), count, seed, None))
    ctx.synchronize()
    return t


def _root(log_n):
    from stark_verifier_amd import fft
    w = pow(7, (P - 1) >> log_n, P)
    return w, fft._limbs(w)


def _static(ctx, x, y, log_n, batch, limbs, inverse):
    from stark_verifier_amd import fft
    return ctx.lib.stark_ntt_batch_dev_static(ctx.handle, x.data_ptr(), y.data_ptr(), log_n, batch, fft._ptr(limbs),
                                              int(inverse), None)


def test_static_scratch_entry_point_matches_the_allocating_one():
    import stark_verifier_amd as sv
    from stark_verifier_amd import fft, _lib
    ctx = sv.Context(0)
    try:
        # (log_n, batch, inverse): a two-pass size that fills several groups of the static scratch, then smaller
        # inputs (fewer polynomials, fewer points, a single-pass size), then the largest size the scratch takes
        for step, (log_n, batch, inverse) in enumerate([(16, 40, False), (14, 3, True), (16, 1, True), (10, 5, False),
                                                        (20, 2, False)]):
            n = 1 << log_n
            w, limbs = _root(log_n)
            x = _fill(ctx, batch * n, 0xA110C + step)
            ref = torch.empty_like(x)
            fft.ntt_batch_dev(x.data_ptr(), ref.data_ptr(), log_n, batch, w, inverse, ctx)   # builds the tables too
            ctx.synchronize()
            got = torch.full_like(x, -1)
            torch.cuda.synchronize()          # torch fills on its own stream, the library launches on the context's
            assert _static(ctx, x, got, log_n, batch, limbs, inverse) == _lib.STARK_OK, ctx.lib.stark_last_error(ctx.handle)
            ctx.synchronize()
            assert torch.equal(got, ref), (log_n, batch, inverse)
            if log_n <= 14:   # and against the oracle directly
                h = x[:n].cpu().numpy().view(np.uint32)
                exp = oracle.fft_inv(h, w) if inverse else oracle.fft_fwd(h, w)
                assert np.array_equal(got[:n].cpu().numpy().view(np.uint32), exp)
            # in place
            t = x.clone()
            torch.cuda.synchronize()
            assert _static(ctx, t, t, log_n, batch, limbs, inverse) == _lib.STARK_OK
            ctx.synchronize()
            assert torch.equal(t, ref)
        # no tables for this direction yet: rejected, output untouched
        log_n = 13
        w, limbs = _root(log_n)
        x = _fill(ctx, 1 << log_n, 5)
        y = torch.full_like(x, 7)
        torch.cuda.synchronize()
        assert _static(ctx, x, y, log_n, 1, limbs, False) == _lib.STARK_ERR_INVALID
        assert b"stark_ntt_batch_dev" in ctx.lib.stark_last_error(ctx.handle)
        fft.ntt_batch_dev(x.data_ptr(), y.data_ptr(), log_n, 0, w, False, ctx)             # batch 0 builds the tables
        assert _static(ctx, x, y, log_n, 1, limbs, False) == _lib.STARK_OK
        ctx.synchronize()
        assert np.array_equal(y.cpu().numpy().view(np.uint32), oracle.fft_fwd(x.cpu().numpy().view(np.uint32), w))
        # above the scratch's size: a clear error, nothing launched
        log_n = 21
        w, limbs = _root(log_n)
        x = _fill(ctx, 1 << log_n, 6)
        y = torch.full_like(x, 7)
        torch.cuda.synchronize()
        fft.ntt_batch_dev(x.data_ptr(), y.data_ptr(), log_n, 0, w, False, ctx)
        assert _static(ctx, x, y, log_n, 1, limbs, False) == _lib.STARK_ERR_UNSUPPORTED
        assert b"static scratch" in ctx.lib.stark_last_error(ctx.handle)
        ctx.synchronize()
        assert bool((y == 7).all())
        # a root of the wrong order is rejected as in stark_ntt_batch_dev
        assert _static(ctx, x, y, 20, 1, limbs, False) == _lib.STARK_ERR_INVALID
    finally:
        ctx.close()


def test_static_entry_point_allocates_nothing_and_serves_two_contexts_in_turn():
    """What the entry point is for: the device's free memory is the same after the calls as before them (a hipMalloc
    of any size would take at least one allocation granule), also at the 2^20-point size, whose allocating twin takes
    another kernel path; and two contexts on one GPU, which share the GPU's one scratch array, give the right bytes when
    their calls come one after the other."""
    import stark_verifier_amd as sv
    from stark_verifier_amd import fft, _lib
    a, b = sv.Context(0), sv.Context(0)
    try:
        cases = [(16, 24, False), (20, 9, True), (14, 3, False)]
        prepared = []
        for step, (log_n, batch, inverse) in enumerate(cases):
            w, limbs = _root(log_n)
            x = _fill(a, batch * (1 << log_n), 0xBEE + step)
            ref = torch.empty_like(x)
            fft.ntt_batch_dev(x.data_ptr(), ref.data_ptr(), log_n, batch, w, inverse, a)      # tables of context a
            fft.ntt_batch_dev(x.data_ptr(), ref.data_ptr(), log_n, 0, w, inverse, b)          # tables of context b
            a.synchronize()
            b.synchronize()
            prepared.append((log_n, batch, inverse, limbs, x, ref, torch.full_like(x, -1), torch.full_like(x, -1)))
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        for log_n, batch, inverse, limbs, x, ref, ya, yb in prepared:
            assert _static(a, x, ya, log_n, batch, limbs, inverse) == _lib.STARK_OK, a.lib.stark_last_error(a.handle)
            a.synchronize()
            assert _static(b, x, yb, log_n, batch, limbs, inverse) == _lib.STARK_OK, b.lib.stark_last_error(b.handle)
            b.synchronize()
        free_after = torch.cuda.mem_get_info()[0]
        assert free_after == free_before, (free_before, free_after)
        for log_n, batch, inverse, limbs, x, ref, ya, yb in prepared:
            assert torch.equal(ya, ref) and torch.equal(yb, ref), (log_n, batch, inverse)
    finally:
        a.close()
        b.close()
