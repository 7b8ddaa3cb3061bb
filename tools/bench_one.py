"""Time one batched transform size: python tools/bench_one.py LOG_N BATCH [reps] [inv]. Prints ms and transforms/s."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stark_verifier_amd as sv
from stark_verifier_amd import fft
P = 2**256 - 351 * 2**32 + 1
log_n, B = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
inverse = len(sys.argv) > 4 and sys.argv[4] == "inv"
ctx = sv.Context(0)
ts = torch.cuda.ExternalStream(ctx.stream)
n = 1 << log_n
w = pow(7, (P - 1) >> log_n, P)
x = torch.empty((B * n, 8), dtype=torch.int32, device="cuda")
ctx.check(ctx.lib.stark_fill_random_field_dev(ctx.handle, x.data_ptr(), B * n, 9, None))
y = torch.empty_like(x)
for _ in range(2):
    fft.ntt_batch_dev(x.data_ptr(), y.data_ptr(), log_n, B, w, inverse, ctx)
ctx.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(ts):
    a.record()
    for _ in range(reps):
        fft.ntt_batch_dev(x.data_ptr(), y.data_ptr(), log_n, B, w, inverse, ctx)
    b.record()
ctx.synchronize()
ms = a.elapsed_time(b) / reps
print(("inverse " if inverse else "") + "split=%s log_n=%d batch=%d  %.3f ms  %.1f transforms/s  %.1f Gbf/s" % (os.environ.get("STARKHIP_SPLIT", "default"), log_n, B, ms, B / ms * 1e3, (n // 2) * log_n * B / ms * 1e-6))
