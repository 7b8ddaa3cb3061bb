"""Batch verification of proofs that already lie in device memory: python tools/bench_resident.py [N] [calls].
N distinct proofs are made on the device (1 % of them damaged afterwards), then stark_mimc_verify_batch_dev is timed
over `calls` calls (host wall clock around each call: the call ends with the verdicts on the host)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import stark_verifier_amd as sv
from stark_verifier_amd import prover
from stark_verifier_amd.fft import _limbs, MODULUS
import ref_verifier as rv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 30
consts = rv.mimc_constants()
steps = 8192
size = prover.mimc_proof_size(steps, len(consts))
rng = np.random.default_rng(11)
inputs = [3] + [int.from_bytes(rng.bytes(32), "little") % MODULUS for _ in range(N - 1)]
ctx = sv.default_context()
lib = ctx.lib
inp = np.ascontiguousarray(np.concatenate([_limbs(x) for x in inputs]))
rcs = np.ascontiguousarray(np.concatenate([_limbs(c % MODULUS) for c in consts]))
d_in = torch.from_numpy(inp.view(np.int32)).cuda()
d_out = torch.empty(8 * N, dtype=torch.int32, device="cuda")
d_proofs = torch.empty(N * size, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
ctx.check(lib.stark_mimc_prove_batch_dev(ctx.handle, d_in.data_ptr(), N, steps, rcs.ctypes.data, len(consts),
                                         d_proofs.data_ptr(), size, d_out.data_ptr(), None))
ctx.synchronize()
expect = np.ones(N, dtype=np.uint8)
for i in range(7, N, 100):  # one bit of a witness or value somewhere in the proof's branches
    d_proofs[i * size + 200 + (i * 7919) % (size - 4000)] ^= 1
    expect[i] = 0
torch.cuda.synchronize()
acc = np.zeros(N, dtype=np.uint8); st = np.zeros(N, dtype=np.uint8)


def call():
    ctx.check(lib.stark_mimc_verify_batch_dev(ctx.handle, d_proofs.data_ptr(), size, N, d_in.data_ptr(), steps, rcs.ctypes.data,
                                              len(consts), d_out.data_ptr(), acc.ctypes.data, st.ctypes.data))


t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.1:  # buffers at their size, clocks up
    call()
ts = []
for _ in range(calls):
    t0 = time.perf_counter()
    call()
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print(json.dumps({"proofs": N, "calls": calls, "ms_per_call_median": float(np.median(ts)), "ms_per_call_mean": float(ts.mean()),
                  "proofs_per_s": N / float(np.median(ts)) * 1e3, "accept_vector_as_constructed": bool(np.array_equal(acc, expect)),
                  "rejected": int(N - acc.sum())}))
