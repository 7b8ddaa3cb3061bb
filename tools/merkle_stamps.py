"""When each workgroup of one Merkle launch starts and ends (debug build with STARKHIP_STAMPS:
make -C stark_verifier_amd/csrc dbg).  python tools/merkle_stamps.py [n_branches]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
os.environ["STARKHIP_LIB"] = os.path.join(ROOT, "stark_verifier_amd", "libstarkhip_dbg.so")
sys.path.insert(0, ROOT)
import stark_verifier_amd as sv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
w = 19
ctx = sv.Context(0)
g = torch.Generator(device="cuda").manual_seed(1)
vals = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
sibs = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
wit = torch.randint(0, 256, (w, n, 32), dtype=torch.uint8, device="cuda", generator=g)
idx = torch.randint(0, 1 << 20, (n,), dtype=torch.int32, device="cuda", generator=g)
root = torch.zeros(32, dtype=torch.uint8, device="cuda")
acc = torch.zeros(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for _ in range(4):
    ctx.check(ctx.lib.stark_merkle_verify_batch_dev(ctx.handle, vals.data_ptr(), sibs.data_ptr(), 32, wit.data_ptr(), w, idx.data_ptr(),
                                                    root.data_ptr(), None, n, acc.data_ptr(), None, None))
ctx.synchronize()
buf = np.zeros((16384, 4), dtype=np.uint64)
ctx.lib.stark_debug_read_merkle_stamps.argtypes = [ctypes.c_void_p]
assert ctx.lib.stark_debug_read_merkle_stamps(buf.ctypes.data) == 0
wgs = min(16384, (n + 255) // 256)
s = buf[:wgs].astype(np.int64)
t0 = s[:, 0].min()
start, leaf, end = (s[:, 0] - t0) / 100.0, (s[:, 1] - t0) / 100.0, (s[:, 2] - t0) / 100.0   # microseconds
print("workgroups %d; kernel span %.1f us" % (wgs, end.max()))
print("start  percentiles 0/10/50/90/100: " + " ".join("%.1f" % np.percentile(start, p) for p in (0, 10, 50, 90, 100)))
print("end    percentiles 0/10/50/90/100: " + " ".join("%.1f" % np.percentile(end, p) for p in (0, 10, 50, 90, 100)))
dur = end - start
print("duration percentiles 0/10/50/90/100: " + " ".join("%.1f" % np.percentile(dur, p) for p in (0, 10, 50, 90, 100)))
print("leaf hash percentiles 10/50/90: " + " ".join("%.1f" % np.percentile(leaf - start, p) for p in (10, 50, 90)))
first = start < np.percentile(start, 50)
print("first half of the workgroups: duration median %.1f; second half: %.1f" % (np.median(dur[first]), np.median(dur[~first])))
# how many workgroups are running over time
ts = np.linspace(0, end.max(), 21)
print("resident workgroups at 5%% steps: " + " ".join(str(int(((start <= t) & (end > t)).sum())) for t in ts))
hw = s[:, 3]
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
print("HW_ID fields seen: cu %s sh %s se %s" % (sorted(set(cu.tolist())), sorted(set(sh.tolist())), sorted(set(se.tolist()))))
