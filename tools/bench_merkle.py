"""Device time of the Merkle branch kernel: python tools/bench_merkle.py [n_branches] [depth] [value_size]."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stark_verifier_amd as sv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 20
vs = int(sys.argv[3]) if len(sys.argv) > 3 else 32
w = depth - 1
ctx = sv.Context(0)
ts = torch.cuda.ExternalStream(ctx.stream)
g = torch.Generator(device="cuda").manual_seed(1)
vals = torch.randint(0, 256, (n, vs), dtype=torch.uint8, device="cuda", generator=g)
sibs = torch.randint(0, 256, (n, vs), dtype=torch.uint8, device="cuda", generator=g)
wit = torch.randint(0, 256, (w, n, 32), dtype=torch.uint8, device="cuda", generator=g)
idx = torch.randint(0, 1 << depth, (n,), dtype=torch.int32, device="cuda", generator=g)
root = torch.zeros(32, dtype=torch.uint8, device="cuda")
acc = torch.zeros(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
def step():
    ctx.check(ctx.lib.stark_merkle_verify_batch_dev(ctx.handle, vals.data_ptr(), sibs.data_ptr(), vs, wit.data_ptr(), w, idx.data_ptr(),
                                                    root.data_ptr(), None, n, acc.data_ptr(), None, None))
WARM = int(os.environ.get("BENCH_WARMUP", "3"))
STEPS = int(os.environ.get("BENCH_STEPS", "10"))
for _ in range(WARM): step()
ctx.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(ts):
    a.record()
    for _ in range(STEPS): step()
    b.record()
ctx.synchronize()
ms = a.elapsed_time(b) / STEPS
comp = n * (w + (2 * vs + 63) // 64)
print("n=%d depth=%d vs=%d  %.3f ms  %.3f G branches/s  %.2f G compressions/s" % (n, depth, vs, ms, n / ms * 1e-6, comp / ms * 1e-6))

# tree build: 2^24 leaves (one compression and 96 bytes per node)
logL = 24
L = 1 << logL
nodes = torch.randint(0, 256, (2 * L, 32), dtype=torch.uint8, device="cuda", generator=g)
def build():
    ctx.check(ctx.lib.stark_merkle_build_dev(ctx.handle, nodes.data_ptr(), logL, None))
build(); ctx.synchronize()
with torch.cuda.stream(ts):
    a.record()
    for _ in range(5): build()
    b.record()
ctx.synchronize()
ms = a.elapsed_time(b) / 5
print("tree build 2^%d leaves  %.3f ms  %.2f G compressions/s  %.0f GB/s (96 B per node)" % (logL, ms, (L - 1) / ms * 1e-6, 96 * (L - 1) / ms * 1e-6))
