"""cfg 5: the reference's bundled proof.bin x N with 1 % corrupted copies through the batch driver.
python tools/bench_batch.py [N]   -> one JSON line (proofs/s, whole call, host buffers in and out)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import stark_verifier_amd as sv
from stark_verifier_amd import batch
import ref_verifier as rv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
good = open(os.path.join(ROOT, "tests", "golden", "proof.bin"), "rb").read()
consts = rv.mimc_constants()
output = rv.mimc(3, 8192, consts)
rng = np.random.default_rng(5)
blobs, expect = [], []
for i in range(N):
    if i % 100 == 7:
        b = bytearray(good)
        b[int(rng.integers(200, len(good) - 3000))] ^= 1 << int(rng.integers(0, 8))   # lands in a value, a witness or a length
        blobs.append(bytes(b)); expect.append(0)
    else:
        blobs.append(good); expect.append(1)
ctx = sv.default_context()
batch.verify_mimc_batch(blobs[:8], 3, 8192, consts, output)   # warm-up: library load, device buffers
t0 = time.perf_counter()
accept, status = batch.verify_mimc_batch(blobs, 3, 8192, consts, output)
dt_py = time.perf_counter() - t0
# the C call alone, on an already concatenated buffer (what a compiled host passes)
import ctypes
from stark_verifier_amd.fft import _limbs, _ptr, MODULUS
offsets = np.zeros(N + 1, dtype=np.uint64); offsets[1:] = np.cumsum([len(b) for b in blobs], dtype=np.uint64)
blob = np.frombuffer(b"".join(blobs), dtype=np.uint8)
rcs = np.ascontiguousarray(np.concatenate([_limbs(c % MODULUS) for c in consts]))
acc2 = np.zeros(N, dtype=np.uint8); st2 = np.zeros(N, dtype=np.uint8)
i_, o_ = _limbs(3), _limbs(output % MODULUS)
t0 = time.perf_counter()
ctx.check(ctx.lib.stark_mimc_verify_batch(ctx.handle, blob.ctypes.data, offsets.ctypes.data, N, _ptr(i_), 8192, rcs.ctypes.data, 64,
                                          _ptr(o_), acc2.ctypes.data, st2.ctypes.data))
dt = time.perf_counter() - t0
ok = [int(a) for a in accept] == expect and np.array_equal(acc2, accept)
print(json.dumps({"proofs": N, "seconds": dt, "proofs_per_s": N / dt, "branches_per_s": 1240 * N / dt,
                  "MB_per_s": sum(map(len, blobs)) / dt * 1e-6, "accept_vector_as_constructed": ok,
                  "rejected": int(N - accept.sum()), "seconds_through_python_wrapper": dt_py, "host_threads": os.environ.get("STARKHIP_HOST_THREADS", "default(<=16)")}))
