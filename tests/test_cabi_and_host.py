"""CPU-side checks: the C ABI library loads and exports every symbol the header declares,
the host mirrors reject bad shapes before touching a device, and the batch-sharding helper
used for multi-GPU runs partitions work correctly (2 ranks, gloo)."""
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "starkhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(stark_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from stark_verifier_amd import _lib
    lib = _lib.load()
    declared = _declared_functions()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/starkhip.h but not exported"
    assert sorted(_lib.EXPORTS) == declared
    assert lib.stark_version() >= 100
    # exported with C linkage (no mangled twins)
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    syms = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(declared) <= syms


def test_big_endian_limb_helpers_round_trip():
    """32 big-endian bytes (BigInt::from_bytes_be, reference src/lib.rs:54,169-174) <-> 8 little-endian u32 limbs:
    host-only helpers, no context and no device needed."""
    import numpy as np
    from stark_verifier_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    vals = [0, 1, 2**256 - 1, 2**256 - 351 * 2**32 + 1, int.from_bytes(rng.bytes(32), "big"), 0x0102030405060708 << 130]
    be = np.frombuffer(b"".join(v.to_bytes(32, "big") for v in vals), dtype=np.uint8).copy()
    limbs = np.zeros(8 * len(vals), dtype=np.uint32)
    lib.stark_be32_to_limbs(be.ctypes.data, len(vals), limbs.ctypes.data)
    got = [int.from_bytes(limbs[8 * i:8 * i + 8].astype("<u4").tobytes(), "little") for i in range(len(vals))]
    assert got == vals
    back = np.zeros_like(be)
    lib.stark_limbs_to_be32(limbs.ctypes.data, len(vals), back.ctypes.data)
    assert np.array_equal(back, be)


def test_context_creation_fails_loudly_without_gpu():
    import torch
    from stark_verifier_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.StarkError):
        _lib.Context(0)


def test_null_context_is_rejected_not_crashed():
    from stark_verifier_amd import _lib
    lib = _lib.load()
    assert lib.stark_ntt_batch(None, None, 4, 1, None, 0) == _lib.STARK_ERR_INVALID
    assert lib.stark_merkle_verify_batch(None, None, None, 32, None, 3, None, None, 1, None, 4, None, None) == _lib.STARK_ERR_INVALID
    assert lib.stark_last_error(None) == b"null context"


def test_host_mirror_shape_checks_happen_before_any_device_call():
    from stark_verifier_amd import merkle_tree as mt
    with pytest.raises(AssertionError):  # src/merkle_tree.rs:28
        mt.MultiProof([mt.ProofBranch([bytes(32)], bytes(32), bytes(32))], bytes(32)).verify([1, 2])


def test_worker_pool_stress(tmp_path):
    """The host thread pool behind the staging copies and the batch driver's loops (csrc/worker_pool.hpp, pure C++):
    every task index runs exactly once for task counts around the worker count over many generations, closures do not
    outlive their run, two pools work side by side. Built with ThreadSanitizer where the toolchain has it."""
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    src = os.path.join(root, "tests", "worker_pool_test.cpp")
    exe = str(tmp_path / "worker_pool_test")
    base = ["g++", "-std=c++17", "-O2", "-pthread", "-o", exe, src]
    if subprocess.run(base[:4] + ["-fsanitize=thread"] + base[4:], capture_output=True).returncode != 0:
        subprocess.run(base, check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "worker pool ok" in out.stdout, out.stdout + out.stderr


def test_shard_ranges_cover_and_balance():
    from stark_verifier_amd.sharding import shard_range
    for total in (0, 1, 7, 8, 1000, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


WORKER = r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from stark_verifier_amd.sharding import shard_range, gather_accept_bits
import oracle, synth
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
case = synth.merkle_case(11, 6, 1001)
wit = case["witnesses"].copy(); wit[2, 5::50, 3] ^= 1
lo, hi = shard_range(1001, rank, world)
# each rank checks only its slice (the CPU oracle stands in for the device call in this CPU test)
acc = oracle.merkle_verify_batch(case["values"][lo:hi], case["siblings"][lo:hi], np.ascontiguousarray(wit[:, lo:hi]),
                                 case["indices"][lo:hi], case["root"])
full = gather_accept_bits(acc, 1001)
if rank == 0:
    exp = oracle.merkle_verify_batch(case["values"], case["siblings"], wit, case["indices"], case["root"])
    assert np.array_equal(full, exp), "gathered accept vector differs"
    assert int((full == 0).sum()) == len(range(5, 1001, 50))
    print("ok")
dist.destroy_process_group()
"""


def test_two_rank_gloo_accept_bit_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ok" in r.stdout


BATCH_WORKER = r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from stark_verifier_amd import batch
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
proofs = [bytes([i]) * (i + 1) for i in range(11)]
seen = []
def fake(blobs):   # stands in for the device call: accepts even-valued proofs, status = first byte
    seen.extend(blobs)
    return (np.array([b[0] % 2 == 0 for b in blobs], dtype=np.uint8), np.array([b[0] for b in blobs], dtype=np.uint8))
acc, st = batch.verify_mimc_batch_sharded(proofs, 3, 8192, [1], 5, verify=fake)
assert list(acc) == [1, 0] * 5 + [1] and list(st) == list(range(11))
assert len(seen) in (5, 6) and all(p in proofs for p in seen)   # each rank saw only its slice
# one (input, output) pair per proof, and proof generation: the same contiguous slices, nothing else exchanged
from stark_verifier_amd import prover
ins, outs = list(range(100, 111)), list(range(200, 211))
def fake_io(blobs, i, o):
    assert len(blobs) == len(i) == len(o) and all(b[0] + 100 == x and b[0] + 200 == y for b, x, y in zip(blobs, i, o))
    return fake(blobs)
acc, st = batch.verify_mimc_batch_io_sharded(proofs, ins, 8192, [1], outs, verify=fake_io)
assert list(acc) == [1, 0] * 5 + [1] and list(st) == list(range(11))
made, mouts, (lo, hi) = prover.mk_mimc_proofs_sharded(ins, 8192, [1], prove=lambda xs: ([bytes([x - 100]) for x in xs], [x + 1 for x in xs]))
assert made == [bytes([k]) for k in range(lo, hi)] and mouts == [101 + k for k in range(lo, hi)]
sizes = [None, None]
dist.all_gather_object(sizes, (lo, hi))
assert sizes[0][0] == 0 and sizes[0][1] == sizes[1][0] and sizes[1][1] == 11
if rank == 0:
    print("ok")
dist.destroy_process_group()
"""


def test_two_rank_gloo_batch_partitioning(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(BATCH_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT],
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ok" in r.stdout


def test_default_device_follows_the_caller_before_local_rank(monkeypatch):
    """Which GPU the default context opens: a caller that has already chosen a device through torch keeps it, even
    device 0 under torchrun; only a process that has not touched torch.cuda falls back to its LOCAL_RANK."""
    from stark_verifier_amd import _lib

    class FakeCuda:
        def __init__(self, initialised, current, count):
            self._i, self._c, self._n = initialised, current, count

        def is_available(self):
            return True

        def is_initialized(self):
            return self._i

        def current_device(self):
            return self._c

        def device_count(self):
            return self._n

    class FakeTorch:
        def __init__(self, cuda):
            self.cuda = cuda

    monkeypatch.setenv("LOCAL_RANK", "3")
    monkeypatch.setattr(_lib, "torch", FakeTorch(FakeCuda(True, 0, 8)))
    assert _lib.current_device() == 0      # the caller set device 0 on purpose
    monkeypatch.setattr(_lib, "torch", FakeTorch(FakeCuda(True, 5, 8)))
    assert _lib.current_device() == 5
    monkeypatch.setattr(_lib, "torch", FakeTorch(FakeCuda(False, 0, 8)))
    assert _lib.current_device() == 3      # nothing chosen yet: one process per GPU
    monkeypatch.setattr(_lib, "torch", FakeTorch(FakeCuda(False, 0, 1)))
    assert _lib.current_device() == 0      # per-rank HIP_VISIBLE_DEVICES: only one GPU to see
    monkeypatch.delenv("LOCAL_RANK")
    monkeypatch.setattr(_lib, "torch", FakeTorch(FakeCuda(False, 0, 8)))
    assert _lib.current_device() == 0
