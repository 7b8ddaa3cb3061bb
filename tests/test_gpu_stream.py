"""The streaming form of the two-pass transforms at 2^20 and 2^16 points (resident workgroups that fetch the next work item
straight into LDS while the current one finishes, work handed out by per-XCD counters) against the plain kernels and
the oracle: same bytes for every batch size, both directions, in and out of place, on the context's own stream and on
a stream of the caller's, and twice in a row in one context (the counters are zeroed per launch).

What is transformed follows the reference's _fft / fft_inv (src/fft.rs:37-86): out[i] = sum_j v[j] w^(ij) mod p,
canonical residues, scaled by n^-1 for the inverse."""
import os

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
P = oracle.P


def _root(log_n):
    return pow(7, (P - 1) >> log_n, P)


def _ctx(monkeypatch, stream_on, **env):
    """A fresh context whose plans are built with the streaming form on or off (the planner reads the switches when it
    builds a plan; monkeypatch puts the environment back after the test)."""
    import stark_verifier_amd as sv
    monkeypatch.setenv("STARKHIP_STREAM", "1" if stream_on else "0")
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    return sv.Context(0)


def _fill(ctx, count, seed):
    t = torch.empty((count, 8), dtype=torch.int32, device="cuda")
    ctx.check(ctx.lib.stark_fill_random_field_dev(ctx.handle, t.data_ptr(), count, seed, None))
    ctx.synchronize()
    return t


def _run(ctx, x, batch, inverse, in_place=False, stream=None, log_n=20):
    from stark_verifier_amd import fft
    LOG_N, W = log_n, _root(log_n)
    y = x.clone() if in_place else torch.empty_like(x)
    torch.cuda.synchronize()  # the copy runs on torch's stream, the transform on the library's
    src = y if in_place else x
    if stream is None:
        fft.ntt_batch_dev(src.data_ptr(), y.data_ptr(), LOG_N, batch, W, inverse, ctx)
        ctx.synchronize()
    else:
        fft.ntt_batch_dev(src.data_ptr(), y.data_ptr(), LOG_N, batch, W, inverse, ctx, stream=stream.cuda_stream)
        stream.synchronize()
    return y


@pytest.mark.parametrize("log_n,batch", [(20, 1), (20, 7), (20, 8), (20, 11), (20, 24), (16, 100), (16, 128), (16, 333), (16, 1024)])
def test_streaming_and_plain_kernels_give_the_same_bytes(log_n, batch, monkeypatch):
    # with the threshold lowered to 4 items per resident workgroup, 1 and 7 polynomials of 2^20 points (100 of 2^16) stay
    # below it (plain kernels in both contexts) and 8 (128) is the first size above it; the default threshold (16) is met
    # by the 64 polynomials of 2^20 and the 1024 of 2^16 points of tests/test_gpu_full_size.py
    N, W = 1 << log_n, _root(log_n)
    plain = _ctx(monkeypatch, False)
    x = _fill(plain, batch * N, 100 + batch)
    ref_f = _run(plain, x, batch, False, log_n=log_n)
    ref_i = _run(plain, x, batch, True, log_n=log_n)
    plain.close()
    fast = _ctx(monkeypatch, True, STARKHIP_STREAM_MIN=4)
    try:
        for _ in range(2):  # twice: the work counters of the first run must not leak into the second
            assert torch.equal(_run(fast, x, batch, False, log_n=log_n), ref_f)
            assert torch.equal(_run(fast, x, batch, True, log_n=log_n), ref_i)
        assert torch.equal(_run(fast, x, batch, False, in_place=True, log_n=log_n), ref_f)
        assert torch.equal(_run(fast, x, batch, True, in_place=True, log_n=log_n), ref_i)
        # a stream of the caller's (its own counters)
        s = torch.cuda.Stream()
        torch.cuda.synchronize()
        assert torch.equal(_run(fast, x, batch, False, stream=s, log_n=log_n), ref_f)
    finally:
        fast.close()
    # the oracle on the first and the last polynomial
    for b in (0, batch - 1):
        xin = x[b * N:(b + 1) * N].cpu().numpy().view(np.uint32)
        assert np.array_equal(ref_f[b * N:(b + 1) * N].cpu().numpy().view(np.uint32), oracle.fft_fwd(xin, W)), b
    assert np.array_equal(ref_i[:N].cpu().numpy().view(np.uint32), oracle.fft_inv(x[:N].cpu().numpy().view(np.uint32), W))


def test_streaming_with_every_item_count_per_workgroup(monkeypatch):
    # threshold 1: also the sizes where a resident workgroup gets one or two items and the counters run out at once
    # (512 work items per polynomial of 2^20 points, 32 per polynomial of 2^16)
    for log_n, batch in ((20, 1), (20, 2), (20, 3), (16, 16), (16, 17), (16, 40)):
        plain = _ctx(monkeypatch, False)
        x = _fill(plain, batch << log_n, 7 + batch)
        ref = _run(plain, x, batch, False, log_n=log_n)
        ref_i = _run(plain, x, batch, True, log_n=log_n)
        plain.close()
        fast = _ctx(monkeypatch, True, STARKHIP_STREAM_MIN=1)
        try:
            assert torch.equal(_run(fast, x, batch, False, log_n=log_n), ref), (log_n, batch)
            assert torch.equal(_run(fast, x, batch, True, in_place=True, log_n=log_n), ref_i), (log_n, batch, "inverse in place")
        finally:
            fast.close()


def test_streaming_with_fewer_resident_workgroups_than_the_chip_holds(monkeypatch):
    # one workgroup per compute unit: every workgroup walks twice as many items
    plain = _ctx(monkeypatch, False)
    N, W = 1 << 20, _root(20)
    x = _fill(plain, 8 * N, 99)
    ref = _run(plain, x, 8, True)  # (8 polynomials = 16 items for each of the 256 workgroups the child keeps resident)
    plain.close()
    import subprocess, sys
    # STARKHIP_STREAM_WGS is read once per process, so this case runs in a child
    code = (
        "import os, sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "import stark_verifier_amd as sv\n"
        "from stark_verifier_amd import fft\n"
        "ctx = sv.Context(0)\n"
        "x = torch.empty((8 << 20, 8), dtype=torch.int32, device='cuda')\n"
        "ctx.check(ctx.lib.stark_fill_random_field_dev(ctx.handle, x.data_ptr(), 8 << 20, 99, None))\n"
        "y = torch.empty_like(x)\n"
        "fft.ntt_batch_dev(x.data_ptr(), y.data_ptr(), 20, 8, %d, True, ctx)\n"
        "ctx.synchronize()\n"
        "torch.save(y.cpu(), sys.argv[1])\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), W))
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "y.pt")
        env = dict(os.environ, STARKHIP_STREAM="1", STARKHIP_STREAM_WGS="1")
        subprocess.run([sys.executable, "-c", code, out], check=True, env=env, timeout=300)
        got = torch.load(out, weights_only=True)
    assert torch.equal(got, ref.cpu())


def test_streaming_on_more_caller_streams_than_the_context_has_counter_blocks_ready(monkeypatch):
    # the context holds sixteen counter blocks from its creation; the seventeenth and eighteenth stream get their own
    log_n, batch = 16, 20
    plain = _ctx(monkeypatch, False)
    x = _fill(plain, batch << log_n, 5)
    ref = _run(plain, x, batch, False, log_n=log_n)
    plain.close()
    fast = _ctx(monkeypatch, True, STARKHIP_STREAM_MIN=1)
    try:
        streams = [torch.cuda.Stream() for _ in range(18)]
        torch.cuda.synchronize()
        for k, s in enumerate(streams):   # one after the other: calls on one context share its intermediate buffer
            assert torch.equal(_run(fast, x, batch, False, stream=s, log_n=log_n), ref), k
        for k in (17, 0, 16):             # and again on streams that already have their block
            assert torch.equal(_run(fast, x, batch, True, in_place=False, stream=streams[k], log_n=log_n),
                               _run(fast, x, batch, True, log_n=log_n)), k
    finally:
        fast.close()
