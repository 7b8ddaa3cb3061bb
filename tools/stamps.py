"""Per-phase wall-clock profile of the NTT pass kernels (needs the -DSTARKHIP_STAMPS debug build,
stark_verifier_amd/libstarkhip_dbg.so: `make -C stark_verifier_amd/csrc dbg`).

Lane 0 of every wavefront records the 100 MHz wall clock at entry and three times per round (inputs in registers,
outputs handed on, exchange synchronised).  Printed per pass: the median length of every interval over all
wavefronts, the share of a wavefront's life it takes, and the timeline of the workgroups that shared one compute
unit, so that what overlaps with what can be read off.

  python tools/stamps.py [log_n] [batch]
"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import stark_verifier_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libstarkhip_dbg.so")
from stark_verifier_amd import fft
from stark_verifier_amd._lib import Context

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
P = 2**256 - 351 * 2**32 + 1
root = pow(7, (P - 1) >> log_n, P)
ctx = Context(0)
lib = ctx.lib
x = torch.empty((batch << log_n, 8), dtype=torch.int32, device="cuda")
lib.stark_fill_random_field_dev(ctx.handle, ctypes.c_void_p(x.data_ptr()), ctypes.c_size_t(batch << log_n), ctypes.c_uint64(1), None)
y = torch.empty_like(x)
for _ in range(3):
    fft.ntt_batch_dev(x.data_ptr(), y.data_ptr(), log_n, batch, root, False, ctx=ctx)
    ctx.synchronize()
WAVES, SLOTS = 8, 24
buf = np.zeros((2, 8192, WAVES, SLOTS), dtype=np.uint64)
lib.stark_debug_read_stamps.argtypes = [ctypes.c_void_p]
rc = lib.stark_debug_read_stamps(buf.ctypes.data)
assert rc == 0, rc
rounds = 5
names = ["entry -> inputs of round 0 in registers (global loads)"]
for r in range(rounds):
    last = r == rounds - 1
    names.append(f"round {r}: compute, outputs to " + ("memory (stores issued)" if last else "LDS"))
    names.append(f"round {r}: " + ("stores drained" if last else "exchange synchronised (barrier or wave-local)"))
    if not last:
        names.append(f"round {r + 1}: inputs in registers (LDS reads, first twiddle)")
for p in range(2):
    s = buf[p, :, :, : 3 * rounds + 1].astype(np.int64) * 10  # ns, [wg][wave][stamp]
    used = s[:, 0, 0] > 0
    s = s[used]
    hw = buf[p, used, 0, 16].astype(np.int64)
    xcc = buf[p, used, 0, 17].astype(np.int64) & 15
    t0 = s[:, :, 0].min()
    span = s.max() - t0
    d = np.diff(s, axis=2)
    life = s[:, :, -1] - s[:, :, 0]
    print(f"pass {p}: span {span/1e3:.1f} us, workgroups {s.shape[0]}, wavefront lifetime median {np.median(life)/1e3:.2f} us, "
          f"average resident workgroups {life.mean(axis=1).sum()/span:.1f}")
    for i, nm in enumerate(names):
        col = d[:, :, i].ravel()
        print(f"  {nm:75s} median {np.median(col)/1e3:6.2f} us  p10 {np.percentile(col,10)/1e3:6.2f}  p90 {np.percentile(col,90)/1e3:6.2f}  "
              f"share of life {col.sum()/life.sum()*100:5.1f} %")
    # spread of the wavefronts of one workgroup at the barriers: how long the first to arrive waits for the last
    for r in range(rounds - 1):
        arrive = s[:, :, 2 + 3 * r]
        print(f"  round {r}: spread of arrival at the exchange within a workgroup: median {np.median(arrive.max(axis=1)-arrive.min(axis=1))/1e3:.2f} us")
    # timeline of one compute unit: workgroups by placement (XCC, SE/SH/CU bits of HW_ID)
    key = (xcc << 16) | ((hw >> 8) & 0xFF)
    keys, counts = np.unique(key, return_counts=True)
    k = keys[np.argmax(counts)]
    idx = np.nonzero(key == k)[0]
    idx = idx[np.argsort(s[idx, 0, 0])][:10]
    print(f"  timeline of one compute unit (placement key {k:#x}, {counts.max()} workgroups); times in us from the launch's first stamp;")
    print("  per workgroup: entry | round-0 inputs | end of rounds 0..4 (wave 0 / latest wave)")
    for w in idx:
        t = (s[w] - t0) / 1e3
        ends = " ".join(f"{t[0, 3 + 3 * r]:7.2f}/{t[:, 3 + 3 * r].max():7.2f}" for r in range(rounds))
        print(f"   wg {w:5d}: {t[:, 0].min():7.2f} | {t[0, 1]:7.2f}/{t[:, 1].max():7.2f} | {ends}")
