"""PCIe-inclusive rate of the host-pointer entry points (stark_ntt_batch, stark_merkle_verify_batch):
buffers start and end in ordinary host memory."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import stark_verifier_amd as sv
from stark_verifier_amd import fft
import synth
P = 2**256 - 351 * 2**32 + 1
ctx = sv.default_context()
out = {}
for log_n, B in ((20, 8), (16, 256), (10, 16384)):
    n = 1 << log_n
    w = pow(7, (P - 1) >> log_n, P)
    x = np.tile(synth.field_elements(3, n), (B, 1)).reshape(B, n, 8)
    fft.ntt_batch(x, w, ctx=ctx)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        ctx.check(ctx.lib.stark_ntt_batch(ctx.handle, x.ctypes.data, log_n, B, fft._ptr(fft._limbs(w)), 0))
    dt = (time.perf_counter() - t0) / reps
    out["ntt_2^%d_x%d" % (log_n, B)] = {"ms": dt * 1e3, "transforms_per_s": B / dt, "host_GBps_each_way": 32 * n * B / dt * 1e-9}
print(json.dumps(out, indent=1))
