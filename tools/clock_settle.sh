# How the measured rate depends on how long the chip has been busy: the same transform timed over 5, 10, 50 and 200
# back-to-back repetitions after 2 untimed ones, and in one process per-step times of 60 consecutive steps.
set -e
mkdir -p gpurun_out
out=gpurun_out/clock_settle.txt
: > $out
for r in 5 10 50 200; do
  timeout -k 10 120 python tools/bench_one.py 20 64 $r 2>/dev/null >> $out
done
timeout -k 10 120 python - >> $out 2>/dev/null <<'P'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import stark_verifier_amd as sv
from stark_verifier_amd import fft
P = 2**256 - 351 * 2**32 + 1
ctx = sv.Context(0)
ts = torch.cuda.ExternalStream(ctx.stream)
n, B = 1 << 20, 64
w = pow(7, (P - 1) >> 20, P)
x = torch.empty((B * n, 8), dtype=torch.int32, device="cuda")
ctx.check(ctx.lib.stark_fill_random_field_dev(ctx.handle, x.data_ptr(), B * n, 9, None))
y = torch.empty_like(x)
fft.ntt_batch_dev(x.data_ptr(), y.data_ptr(), 20, B, w, False, ctx)
ctx.synchronize()
time.sleep(2.0)  # idle: the clocks fall back
ev = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
with torch.cuda.stream(ts):
    ev[0].record()
    for k in range(60):
        fft.ntt_batch_dev(x.data_ptr(), y.data_ptr(), 20, B, w, False, ctx)
        ev[k + 1].record()
ctx.synchronize()
print("per-step ms after 2 s idle:", " ".join("%.2f" % ev[k].elapsed_time(ev[k + 1]) for k in range(60)))
P
cat $out
