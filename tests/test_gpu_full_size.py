"""Full-size configurations on the GPU: 1024 x 2^16 forward + inverse, one 2^24 transform, and the
sharded code path on the same 2^24 vector. Bit-exact against the oracle where the oracle is run
(whole 2^24 transform, sampled 2^16 polynomials), size-independent identities elsewhere."""
import ctypes

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
P = oracle.P
SEED = 0x5354524B


def _fill(ctx, count, seed):
    t = torch.empty((count, 8), dtype=torch.int32, device="cuda")
    ctx.check(ctx.lib.stark_fill_random_field_dev(ctx.handle, t.data_ptr(), count, seed, None))
    ctx.synchronize()
    return t


def test_cfg2_1024_polys_of_2_16_forward_inverse():
    import stark_verifier_amd as sv
    from stark_verifier_amd import fft
    ctx = sv.default_context()
    log_n, B = 16, 1024
    n = 1 << log_n
    w = pow(7, (P - 1) >> log_n, P)
    assert w == 0x5caa52206ecc9d72853f7c616e06672484967c1e2e5a87d6922259c0982b76cc  # G2, src/lib.rs:121
    x = _fill(ctx, B * n, SEED + 2)
    y = torch.empty_like(x)
    fft.ntt_batch_dev(x.data_ptr(), y.data_ptr(), log_n, B, w, False, ctx)
    ctx.synchronize()
    for b in (0, 1, 511, 1023):
        exp = oracle.fft_fwd(x[b * n:(b + 1) * n].cpu().numpy().view(np.uint32), w)
        assert np.array_equal(y[b * n:(b + 1) * n].cpu().numpy().view(np.uint32), exp), b
    z = torch.empty_like(x)
    fft.ntt_batch_dev(y.data_ptr(), z.data_ptr(), log_n, B, w, True, ctx)
    ctx.synchronize()
    assert torch.equal(x, z)
    # the inverse alone against the oracle's fft_inv on one polynomial
    b = 77
    exp = oracle.fft_inv(y[b * n:(b + 1) * n].cpu().numpy().view(np.uint32), w)
    assert np.array_equal(z[b * n:(b + 1) * n].cpu().numpy().view(np.uint32), exp)
    # in place gives the same bytes
    fft.ntt_batch_dev(x.data_ptr(), x.data_ptr(), log_n, B, w, False, ctx)
    ctx.synchronize()
    assert torch.equal(x, y)


def test_cfg3_single_2_24_transform_and_sharded_path():
    import stark_verifier_amd as sv
    from stark_verifier_amd import fft
    from stark_verifier_amd.sharding import ntt_sharded, HipLocalOps
    ctx = sv.default_context()
    log_n = 24
    n = 1 << log_n
    w = pow(7, (P - 1) >> log_n, P)
    assert w == 0x6b7f42cee2b0355e3c9637dea849609e83b5488d1eb2644e1729a9dc31f6259e
    x = _fill(ctx, n, SEED + 3)
    y = torch.empty_like(x)
    fft.ntt_batch_dev(x.data_ptr(), y.data_ptr(), log_n, 1, w, False, ctx)
    ctx.synchronize()
    exp = oracle.fft_fwd(x.cpu().numpy().view(np.uint32), w)  # ~10 s on one core
    assert np.array_equal(y.cpu().numpy().view(np.uint32), exp)
    del exp
    # the multi-rank code path (one rank here): byte-identical to the single call
    ops = HipLocalOps(ctx)
    ys = ntt_sharded(x, log_n, w, False, ops=ops)
    torch.cuda.synchronize()
    assert torch.equal(ys, y)
    xs = ntt_sharded(ys, log_n, w, True, ops=ops)
    torch.cuda.synchronize()
    assert torch.equal(xs, x)


def test_headline_shape_64_polys_of_2_20_against_the_oracle_in_full():
    """The benchmark's own shape (64 polynomials of 2^20 points in one call, the 2 GiB chunk of the multi-pass path):
    six polynomials spread over the batch are compared element by element with the oracle's restatement of
    fft.rs:37-86, forward and inverse (fft_inv semantics, src/fft.rs:64-86), out of place and in place."""
    import stark_verifier_amd as sv
    from stark_verifier_amd import fft
    ctx = sv.default_context()
    log_n, B = 20, 64
    n = 1 << log_n
    w = pow(7, (P - 1) >> log_n, P)
    assert w == 0xd3cd75fbef8d5e5ff367cbc0866ba9b493e153ed47f2c01c2b996aa041d2363b
    x = _fill(ctx, B * n, SEED + 20)
    y = torch.empty_like(x)
    fft.ntt_batch_dev(x.data_ptr(), y.data_ptr(), log_n, B, w, False, ctx)
    ctx.synchronize()
    picks = (0, 1, 17, 31, 32, 63)
    host = lambda t, b: t[b * n:(b + 1) * n].cpu().numpy().view(np.uint32)
    for b in picks:
        assert np.array_equal(host(y, b), oracle.fft_fwd(host(x, b), w)), ("forward", b)
    # inverse of fresh data (not of the forward output, so an error the two directions share cannot cancel)
    z = torch.empty_like(x)
    fft.ntt_batch_dev(x.data_ptr(), z.data_ptr(), log_n, B, w, True, ctx)
    ctx.synchronize()
    for b in picks:
        assert np.array_equal(host(z, b), oracle.fft_inv(host(x, b), w)), ("inverse", b)
    # in place: same bytes as out of place, both directions
    t = x.clone()
    torch.cuda.synchronize()              # torch copies on its own stream, the library launches on the context's
    fft.ntt_batch_dev(t.data_ptr(), t.data_ptr(), log_n, B, w, False, ctx)
    ctx.synchronize()
    assert torch.equal(t, y)
    fft.ntt_batch_dev(t.data_ptr(), t.data_ptr(), log_n, B, w, True, ctx)
    ctx.synchronize()
    assert torch.equal(t, x)
    t.copy_(x)
    torch.cuda.synchronize()
    fft.ntt_batch_dev(t.data_ptr(), t.data_ptr(), log_n, B, w, True, ctx)
    ctx.synchronize()
    assert torch.equal(t, z)


@pytest.mark.parametrize("log_n", [15, 17, 18, 19, 21, 22, 23])
def test_every_multi_pass_size_against_the_oracle_in_full(log_n):
    """The sizes between the ones above: two passes with tiles of unequal or odd size (2^17 = 2^8 x 2^9, 2^19, 2^21),
    three passes (2^22, 2^23). One polynomial each way, every element compared with the oracle's restatement of
    fft.rs:37-86, plus the second polynomial of a batch of two (polynomial stride)."""
    import stark_verifier_amd as sv
    from stark_verifier_amd import fft
    ctx = sv.default_context()
    n = 1 << log_n
    w = pow(7, (P - 1) >> log_n, P)
    x = _fill(ctx, 2 * n, SEED + 40 + log_n)
    y = torch.empty_like(x)
    host = lambda t, b: t[b * n:(b + 1) * n].cpu().numpy().view(np.uint32)
    fft.ntt_batch_dev(x.data_ptr(), y.data_ptr(), log_n, 2, w, False, ctx)
    ctx.synchronize()
    assert np.array_equal(host(y, 1), oracle.fft_fwd(host(x, 1), w))
    fft.ntt_batch_dev(x.data_ptr(), y.data_ptr(), log_n, 2, w, True, ctx)
    ctx.synchronize()
    assert np.array_equal(host(y, 0), oracle.fft_inv(host(x, 0), w))
