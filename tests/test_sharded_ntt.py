"""One transform sharded over ranks (stark_verifier_amd.sharding.ntt_sharded): every layout
combination must reassemble to the oracle's transform of the whole vector, bit for bit.

CPU: 1 rank and 2 ranks over gloo with the oracle standing in for the device steps (checks the
orchestration). GPU: the same through the HIP library, 1 rank in-process and 2 ranks (gloo,
staged exchange) sharing the one card."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle
import synth
from shard_ops import OracleLocalOps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = oracle.P


def _expected_slices(X, log_n, world, layout):
    """Per-rank [n/world, 8] arrays of the full result X in the given output layout."""
    n = 1 << log_n
    l1 = log_n // 2
    n1, n2 = 1 << l1, n >> l1
    if layout == "blocked":
        return [X[r * n // world:(r + 1) * n // world] for r in range(world)]
    M = X.reshape(n1, n2, 8)  # X[k1*n2 + k2]
    r2 = n2 // world
    if layout == "columns":
        return [np.ascontiguousarray(M[:, r * r2:(r + 1) * r2]).reshape(-1, 8) for r in range(world)]
    return [np.ascontiguousarray(M[:, r * r2:(r + 1) * r2].transpose(1, 0, 2)).reshape(-1, 8) for r in range(world)]


def _input_slices(x, log_n, world, layout):
    n = 1 << log_n
    l1 = log_n // 2
    n1, n2 = 1 << l1, n >> l1
    if layout == "blocked":
        return [x[r * n // world:(r + 1) * n // world] for r in range(world)]
    M = x.reshape(n2, n1, 8)  # x[j1 + n1*j2] at [j2][j1]
    r1 = n1 // world
    return [np.ascontiguousarray(M[:, r * r1:(r + 1) * r1].transpose(1, 0, 2)).reshape(-1, 8) for r in range(world)]


@pytest.mark.parametrize("log_n", [4, 7, 10])
@pytest.mark.parametrize("inverse", [False, True])
def test_single_rank_all_layouts_cpu(log_n, inverse):
    from stark_verifier_amd.sharding import ntt_sharded
    n = 1 << log_n
    w = pow(7, (P - 1) >> log_n, P)
    x = synth.field_elements(77 + log_n, n)
    X = oracle.fft_inv(x, w) if inverse else oracle.fft_fwd(x, w)
    for li in ("blocked", "strided"):
        for lo in ("blocked", "strided", "columns"):
            for fused in (True, False):
                xin = torch.from_numpy(_input_slices(x, log_n, 1, li)[0].view(np.int32).copy())
                out = ntt_sharded(xin, log_n, w, inverse, in_layout=li, out_layout=lo, ops=OracleLocalOps(), fused=fused)
                assert np.array_equal(out.numpy().view(np.uint32), _expected_slices(X, log_n, 1, lo)[0]), (li, lo, fused)


def test_rejects_bad_world_or_shape():
    from stark_verifier_amd.sharding import ntt_sharded
    w = pow(7, (P - 1) >> 6, P)
    with pytest.raises(ValueError):
        ntt_sharded(torch.zeros((63, 8), dtype=torch.int32), 6, w, ops=OracleLocalOps())
    with pytest.raises(ValueError):
        ntt_sharded(torch.zeros((64, 8), dtype=torch.int32), 6, w, in_layout="cyclic", ops=OracleLocalOps())


WORKER = r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist
root_dir, mode, log_n = sys.argv[1], sys.argv[2], int(sys.argv[3])
sys.path.insert(0, root_dir); sys.path.insert(0, os.path.join(root_dir, "tests"))
import oracle, synth
from test_sharded_ntt import _expected_slices, _input_slices
from stark_verifier_amd.sharding import ntt_sharded
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
P = oracle.P
n = 1 << log_n
w = pow(7, (P - 1) >> log_n, P)
x = synth.field_elements(4242, n)
if mode == "hip":
    import stark_verifier_amd as sv
    from stark_verifier_amd.sharding import HipLocalOps
    torch.cuda.set_device(0)
    ops = HipLocalOps(sv.Context(0)); dev = "cuda"
else:
    from shard_ops import OracleLocalOps
    ops = OracleLocalOps(); dev = "cpu"
for inverse in (False, True):
    X = oracle.fft_inv(x, w) if inverse else oracle.fft_fwd(x, w)
    for li in ("blocked", "strided"):
        for lo in ("blocked", "strided", "columns"):
            for fused in (True, False):
                xin = torch.from_numpy(_input_slices(x, log_n, world, li)[rank].view(np.int32).copy()).to(dev)
                out = ntt_sharded(xin, log_n, w, inverse, in_layout=li, out_layout=lo, ops=ops, fused=fused)
                got = out.cpu().numpy().view(np.uint32)
                assert np.array_equal(got, _expected_slices(X, log_n, world, lo)[rank]), (rank, inverse, li, lo, fused)
if mode == "hip":
    # the one-call C entry point with the exchange handed in as a callback (here: gloo, staged through host memory)
    from stark_verifier_amd import _lib, fft
    ctx = ops.ctx
    def exchange(user, d_send, d_recv, block_bytes, stream):
        try:
            n_el = block_bytes * world // 32
            assert stream == ctx.stream
            # raw device pointers <-> torch tensors through the library's block copy (permute3 with a = b = 1)
            send_t = torch.empty((n_el, 8), dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            assert ctx.lib.stark_permute3_dev(ctx.handle, d_send, send_t.data_ptr(), 1, 1, n_el, None) == 0
            ctx.synchronize()
            src = send_t.cpu()
            dst = torch.empty_like(src)
            dist.all_to_all_single(dst, src)
            back = dst.cuda()
            torch.cuda.synchronize()
            assert ctx.lib.stark_permute3_dev(ctx.handle, back.data_ptr(), d_recv, 1, 1, n_el, None) == 0
            ctx.synchronize()
            return 0
        except Exception as e:   # never unwind into C
            print("exchange failed:", repr(e), flush=True)
            return 1
    cb = _lib.ALL_TO_ALL_FN(exchange)
    for inverse in (False, True):
        X = oracle.fft_inv(x, w) if inverse else oracle.fft_fwd(x, w)
        xin = torch.from_numpy(_input_slices(x, log_n, world, "strided")[rank].view(np.int32).copy()).cuda()
        y = torch.empty_like(xin)
        torch.cuda.synchronize()
        r = fft._limbs(w)
        rc = ctx.lib.stark_ntt_sharded_dev(ctx.handle, xin.data_ptr(), y.data_ptr(), log_n, world, rank, fft._ptr(r),
                                           int(inverse), cb, None, None)
        assert rc == 0, ctx.lib.stark_last_error(ctx.handle)
        ctx.synchronize()
        assert np.array_equal(y.cpu().numpy().view(np.uint32), _expected_slices(X, log_n, world, "columns")[rank]), inverse
dist.barrier()
if rank == 0:
    print("ok")
dist.destroy_process_group()
"""


def _run_ranks(tmp_path, world, mode, log_n):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT, mode, str(log_n)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert "ok" in r.stdout


def test_two_rank_gloo_sharded_transform_cpu(tmp_path):
    _run_ranks(tmp_path, 2, "oracle", 8)


def test_four_rank_gloo_sharded_transform_cpu(tmp_path):
    _run_ranks(tmp_path, 4, "oracle", 6)


# ---------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [4, 9, 14, 16])
def test_single_rank_hip_all_layouts(log_n):
    from stark_verifier_amd.sharding import ntt_sharded
    n = 1 << log_n
    w = pow(7, (P - 1) >> log_n, P)
    x = synth.field_elements(99 + log_n, n)
    for inverse in (False, True):
        X = oracle.fft_inv(x, w) if inverse else oracle.fft_fwd(x, w)
        for li in ("blocked", "strided"):
            for lo in ("blocked", "strided", "columns"):
                for fused in (True, False):
                    xin = torch.from_numpy(_input_slices(x, log_n, 1, li)[0].view(np.int32).copy()).cuda()
                    out = ntt_sharded(xin, log_n, w, inverse, in_layout=li, out_layout=lo, fused=fused)
                    assert np.array_equal(out.cpu().numpy().view(np.uint32), _expected_slices(X, log_n, 1, lo)[0]), \
                        (inverse, li, lo, fused)


@pytest.mark.gpu
def test_twiddle_and_permute_kernels_against_numpy():
    import stark_verifier_amd as sv
    from stark_verifier_amd.sharding import HipLocalOps
    from shard_ops import OracleLocalOps
    ops, ref = HipLocalOps(), OracleLocalOps()
    log_order = 11
    w = pow(7, (P - 1) >> log_order, P)
    for rows, cols, row0, col0 in ((3, 5, 0, 0), (16, 128, 48, 7), (1, 2048, 2047, 0), (37, 33, 1000, 1500)):
        a = synth.field_elements(rows * cols + 1, rows * cols)
        a[0] = 0xFFFFFFFF  # a non-canonical input must come out canonical
        for inverse in (False, True):
            t = torch.from_numpy(a.view(np.int32).copy()).cuda()
            with ops.stream_context():
                ops.twiddle_mul(t, rows, cols, row0, col0, log_order, w, inverse)
            e = torch.from_numpy(a.view(np.int32).copy())
            ref.twiddle_mul(e, rows, cols, row0, col0, log_order, w, inverse)
            assert torch.equal(t.cpu(), e)
    for A, B, chunk in ((1, 1, 1), (33, 65, 1), (64, 32, 1), (100, 7, 1), (3, 4, 5), (8, 2, 64), (5, 9, 2)):
        t = torch.arange(A * B * chunk * 8, dtype=torch.int32).reshape(-1, 8)
        with ops.stream_context():
            got = ops.permute3(t.cuda(), A, B, chunk).cpu()
        assert torch.equal(got, ref.permute3(t, A, B, chunk)), (A, B, chunk)
    with pytest.raises(sv.InvalidArgument), ops.stream_context():
        ops.twiddle_mul(torch.zeros((4, 8), dtype=torch.int32).cuda(), 2, 2, 0, 0, 10, w, False)  # wrong order


@pytest.mark.gpu
def test_two_rank_gloo_sharded_transform_hip(tmp_path):
    _run_ranks(tmp_path, 2, "hip", 14)


NCCL_WORKER = r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist
root_dir = sys.argv[1]
sys.path.insert(0, root_dir); sys.path.insert(0, os.path.join(root_dir, "tests"))
import oracle, synth
from stark_verifier_amd import sharding
from stark_verifier_amd.sharding import ntt_sharded, gather_accept_bits
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
sharding.ALWAYS_COLLECTIVE = True          # run all_gather / all_to_all_single over RCCL although there is one rank
flags = np.array([1, 0, 1, 1, 0], dtype=np.uint8)
got = gather_accept_bits(flags, 5)          # host flags in: staged through this process's GPU
assert isinstance(got, np.ndarray) and list(got) == [1, 0, 1, 1, 0]
got = gather_accept_bits(torch.from_numpy(flags).cuda(), 5)
assert list(got) == [1, 0, 1, 1, 0]
P = oracle.P
log_n = 12
w = pow(7, (P - 1) >> log_n, P)
x = synth.field_elements(31337, 1 << log_n)
for inverse in (False, True):
    X = oracle.fft_inv(x, w) if inverse else oracle.fft_fwd(x, w)
    out = ntt_sharded(torch.from_numpy(x.view(np.int32).copy()).cuda(), log_n, w, inverse)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), X), inverse
# the sharded batch helper: default context = this process's GPU, flags gathered over RCCL
import ref_verifier as rv
from stark_verifier_amd import batch
good = open(os.path.join(root_dir, "tests", "golden", "proof.bin"), "rb").read()
bad = bytearray(good); bad[len(good) // 3] ^= 2
consts = rv.mimc_constants()
acc, st = batch.verify_mimc_batch_sharded([good, bytes(bad), good], 3, 8192, consts, rv.mimc(3, 8192, consts))
assert list(acc) == [1, 0, 1] and list(st) == [0, 4, 0]
dist.barrier()
print("ok")
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_rccl_collectives_one_rank(tmp_path):
    """The "nccl" (RCCL) branches of the accept-flag gather and of the transform's all-to-all, forced to run in a
    one-rank process group (a one-GPU box cannot host two RCCL ranks)."""
    script = tmp_path / "nccl_worker.py"
    script.write_text(NCCL_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert "ok" in r.stdout


@pytest.mark.gpu
def test_column_transforms_and_one_rank_stages_against_the_oracle():
    """stark_ntt_columns_dev on [rows][cols] matrices (one pass up to 2^6 rows, two above; in and out of place) and the
    fused stage A / stage C pair on one rank (stark_ntt_sharded_dev without an exchange), against the oracle's transform
    of every column / of the whole vector."""
    import stark_verifier_amd as sv
    from stark_verifier_amd import _lib, fft
    ctx = sv.default_context()
    for log_rows, cols in ((2, 8), (5, 1), (6, 64), (7, 4), (10, 16), (12, 2), (9, 256)):
        rows = 1 << log_rows
        w = pow(7, (P - 1) >> log_rows, P)
        x = synth.field_elements(500 + log_rows, rows * cols)
        x[0] = 0xFFFFFFFF                       # a non-canonical input
        M = x.reshape(rows, cols, 8)
        for inverse in (False, True):
            exp = np.empty_like(M)
            for c in range(cols):
                col = np.ascontiguousarray(M[:, c])
                exp[:, c] = oracle.fft_inv(col, w) if inverse else oracle.fft_fwd(col, w)
            t = torch.from_numpy(x.view(np.int32).copy()).cuda()
            y = torch.empty_like(t)
            torch.cuda.synchronize()
            r = fft._limbs(w)
            ctx.check(ctx.lib.stark_ntt_columns_dev(ctx.handle, t.data_ptr(), y.data_ptr(), log_rows, cols, fft._ptr(r), int(inverse), None))
            ctx.check(ctx.lib.stark_ntt_columns_dev(ctx.handle, t.data_ptr(), t.data_ptr(), log_rows, cols, fft._ptr(r), int(inverse), None))
            ctx.synchronize()
            assert np.array_equal(y.cpu().numpy().view(np.uint32).reshape(rows, cols, 8), exp), (log_rows, cols, inverse)
            assert torch.equal(t, y)
    r = fft._limbs(pow(7, (P - 1) >> 10, P))
    t = torch.zeros((1 << 10) * 3, 8, dtype=torch.int32).cuda()
    assert ctx.lib.stark_ntt_columns_dev(ctx.handle, t.data_ptr(), t.data_ptr(), 10, 3, fft._ptr(r), 0, None) == _lib.STARK_ERR_UNSUPPORTED
    assert ctx.lib.stark_ntt_columns_dev(ctx.handle, t.data_ptr(), t.data_ptr(), 9, 4, fft._ptr(r), 0, None) == _lib.STARK_ERR_INVALID
    # one rank: rows in, columns out (= the natural order when world == 1), result left in the first buffer
    for log_n in (4, 5, 11, 16, 18):
        n = 1 << log_n
        w = pow(7, (P - 1) >> log_n, P)
        x = synth.field_elements(900 + log_n, n)
        for inverse in (False, True):
            X = oracle.fft_inv(x, w) if inverse else oracle.fft_fwd(x, w)
            a = torch.from_numpy(_input_slices(x, log_n, 1, "strided")[0].view(np.int32).copy()).cuda()
            b = torch.empty_like(a)
            torch.cuda.synchronize()
            r = fft._limbs(w)
            ctx.check(ctx.lib.stark_ntt_sharded_dev(ctx.handle, a.data_ptr(), b.data_ptr(), log_n, 1, 0, fft._ptr(r), int(inverse),
                                                    _lib.ALL_TO_ALL_FN(0), None, None))
            ctx.synchronize()
            assert np.array_equal(a.cpu().numpy().view(np.uint32), X), (log_n, inverse)
    assert ctx.lib.stark_ntt_sharded_dev(ctx.handle, a.data_ptr(), b.data_ptr(), 18, 3, 0, fft._ptr(r), 0, _lib.ALL_TO_ALL_FN(0), None,
                                         None) == _lib.STARK_ERR_INVALID


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,world,split", [(10, 4, None), (14, 8, None), (13, 2, None), (18, 4, "4,5"), (18, 16, "5,4")])
def test_all_ranks_in_one_process_stage_by_stage(log_n, world, split, monkeypatch):
    """Every rank's stage A and stage C run one after the other on the one GPU, the exchange done in numpy between
    them: the send blocks of every rank equal the definition (row transforms, twiddles, packing), and the reassembled
    result equals the oracle's transform of the whole vector. With STARKHIP_SPLIT the row transforms of stage A are a
    two-pass plan, so the fused store rides on a transposing last pass (what sizes above 2^24 use)."""
    import stark_verifier_amd as sv
    from stark_verifier_amd.sharding import HipLocalOps
    if split:
        monkeypatch.setenv("STARKHIP_SPLIT", split)
    ctx = sv.Context(0)          # a fresh context: plans are cached per context and the split is read when they are built
    try:
        ops, ref = HipLocalOps(ctx), OracleLocalOps()
        n = 1 << log_n
        l1 = log_n // 2
        n1, n2 = 1 << l1, 1 << (log_n - l1)
        r1, r2 = n1 // world, n2 // world
        w = pow(7, (P - 1) >> log_n, P)
        x = synth.field_elements(1234 + log_n, n)
        for inverse in (False, True):
            X = oracle.fft_inv(x, w) if inverse else oracle.fft_fwd(x, w)
            sends = []
            for rank in range(world):
                rows = torch.from_numpy(_input_slices(x, log_n, world, "strided")[rank].view(np.int32).copy())
                with ops.stream_context():
                    got = ops.stage_a(rows.cuda(), log_n, world, rank, w, inverse).cpu()
                assert torch.equal(got, ref.stage_a(rows, log_n, world, rank, w, inverse)), (rank, inverse)
                sends.append(got.numpy().reshape(world, r1 * r2, 8))
            for rank in range(world):
                recv = np.concatenate([sends[src][rank] for src in range(world)])           # block src came from rank src
                with ops.stream_context():
                    cols = ops.stage_c(torch.from_numpy(recv.copy()).cuda(), log_n, world, w, inverse).cpu()
                assert np.array_equal(cols.numpy().view(np.uint32), _expected_slices(X, log_n, world, "columns")[rank]), (rank, inverse)
        if split == "5,4":
            # 32 ranks would cut the send blocks below the last pass's tile count: refused, the three-step form remains
            from stark_verifier_amd import _lib, fft
            t = torch.zeros(((n1 // 32) * n2, 8), dtype=torch.int32).cuda()
            u = torch.zeros_like(t)
            torch.cuda.synchronize()
            rc = ctx.lib.stark_ntt_sharded_stage_a_dev(ctx.handle, t.data_ptr(), u.data_ptr(), log_n, 32, 0, fft._ptr(fft._limbs(w)), 0, None)
            assert rc == _lib.STARK_ERR_UNSUPPORTED
    finally:
        ctx.close()


@pytest.mark.gpu
def test_c_host_drives_the_exchange_with_rccl_send_recv():
    """host/test_rccl_exchange.c: a plain C host builds a one-rank RCCL communicator and passes the grouped
    ncclSend / ncclRecv all-to-all of the binding example to stark_ntt_sharded_dev (where it runs although world is 1)
    and issues it itself between the two stage calls; 2^16 and 2^24 points, forward and inverse, byte-identical to
    stark_ntt_batch_dev."""
    import subprocess
    exe = os.path.join(ROOT, "host", "test_rccl_exchange")
    if not os.path.exists(exe):
        pytest.fail("host/test_rccl_exchange is not built (make -C stark_verifier_amd/csrc host)")
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("byte-identical") == 4
