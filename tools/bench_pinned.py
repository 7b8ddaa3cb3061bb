"""The batch verifier from host buffers, pageable against pinned: python tools/bench_pinned.py [N].
proof.bin x N with 1 % damaged; stark_mimc_verify_batch timed over 5 calls each way (median)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import stark_verifier_amd as sv
from stark_verifier_amd.fft import _limbs, _ptr, MODULUS
import ref_verifier as rv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
good = open(os.path.join(ROOT, "tests", "golden", "proof.bin"), "rb").read()
consts = rv.mimc_constants()
output = rv.mimc(3, 8192, consts)
rng = np.random.default_rng(5)
size = len(good)
host = np.empty(N * size, dtype=np.uint8)
expect = np.ones(N, dtype=np.uint8)
g = np.frombuffer(good, dtype=np.uint8)
for i in range(N):
    host[i * size:(i + 1) * size] = g
    if i % 100 == 7:
        host[i * size + int(rng.integers(200, size - 3000))] ^= 1 << int(rng.integers(0, 8))
        expect[i] = 0
pinned = torch.empty(N * size, dtype=torch.uint8, pin_memory=True)
pinned.numpy()[:] = host
offsets = (np.arange(N + 1, dtype=np.uint64) * np.uint64(size))
rcs = np.ascontiguousarray(np.concatenate([_limbs(c % MODULUS) for c in consts]))
i_, o_ = _limbs(3), _limbs(output % MODULUS)
ctx = sv.default_context()
res = {"proofs": N}
for name, ptr in (("pageable", host.ctypes.data), ("pinned", pinned.data_ptr()), ("pageable_again", host.ctypes.data), ("pinned_again", pinned.data_ptr())):
    acc = np.zeros(N, dtype=np.uint8); st = np.zeros(N, dtype=np.uint8)
    ts = []
    for _ in range(6):
        t0 = time.perf_counter()
        ctx.check(ctx.lib.stark_mimc_verify_batch(ctx.handle, ptr, offsets.ctypes.data, N, _ptr(i_), 8192, rcs.ctypes.data, 64,
                                                  _ptr(o_), acc.ctypes.data, st.ctypes.data))
        ts.append(time.perf_counter() - t0)
    dt = float(np.median(ts[1:]))
    res[name] = {"ms": dt * 1e3, "proofs_per_s": N / dt, "GB_per_s": N * size / dt * 1e-9, "ok": bool(np.array_equal(acc, expect))}
print(json.dumps(res))
