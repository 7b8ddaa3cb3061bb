"""Device time of the sharded-transform helper kernels at the 2^24 sizes: the 32-byte-element transpose and
block permutation (HBM-bound: one read + one write of every byte) and the twiddle multiplication."""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stark_verifier_amd as sv
from stark_verifier_amd.sharding import HipLocalOps

P = 2**256 - 351 * 2**32 + 1
ctx = sv.Context(0)
ops = HipLocalOps(ctx)
n = 1 << 24
x = torch.empty((n, 8), dtype=torch.int32, device="cuda")
ctx.check(ctx.lib.stark_fill_random_field_dev(ctx.handle, x.data_ptr(), n, 3, None))
ctx.synchronize()
w = pow(7, (P - 1) >> 24, P)


def timed(fn, reps=10):
    with ops.stream_context():
        fn(); fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


out = {}
for name, A, B, chunk in (("transpose_4096x4096", 4096, 4096, 1), ("transpose_512x32768", 512, 32768, 1),
                          ("pack_512x8x512", 512, 8, 512), ("pack_4096x8x512", 4096, 8, 512)):
    ms = timed(lambda: ops.permute3(x[:A * B * chunk], A, B, chunk))
    out[name] = {"ms": ms, "GBps_read_plus_write": 2 * 32 * A * B * chunk / ms * 1e-6}
ms = timed(lambda: ops.twiddle_mul(x, 4096, 4096, 0, 0, 24, w, False))
out["twiddle_mul_4096x4096"] = {"ms": ms, "Gelements_per_s": n / ms * 1e-6, "GBps_read_plus_write": 2 * 32 * n / ms * 1e-6}
print(json.dumps(out, indent=1))
