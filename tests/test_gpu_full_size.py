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
