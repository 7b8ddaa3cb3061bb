"""Per-phase wall-clock profile of the NTT pass kernels (needs the -DSTARKHIP_STAMPS debug build,
stark_verifier_amd/libstarkhip_dbg.so). Prints, per pass, the median duration of every phase of a
workgroup and how many workgroups were in each phase on average."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import stark_verifier_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libstarkhip_dbg.so")
from stark_verifier_amd import fft
from stark_verifier_amd._lib import Context

log_n, batch = 20, 16
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
buf = np.zeros((2, 8192, 16), dtype=np.uint64)
lib.stark_debug_read_stamps.argtypes = [ctypes.c_void_p]
rc = lib.stark_debug_read_stamps(buf.ctypes.data)
assert rc == 0, rc
rounds = 5
for p in range(2):
    s = buf[p, :, : rounds + 2].astype(np.int64) * 10  # ns
    t0 = s[:, 0].min()
    span = s.max() - t0
    d = np.diff(s, axis=1)
    print(f"pass {p}: span {span/1e3:.1f} us, workgroups {s.shape[0]}")
    names = [f"round{r}" for r in range(rounds)] + ["drain"]
    names[0] = "start+load+round0"
    for i, nm in enumerate(names):
        col = d[:, i]
        print(f"  {nm:20s} median {np.median(col)/1e3:7.2f} us  p10 {np.percentile(col,10)/1e3:7.2f}  p90 {np.percentile(col,90)/1e3:7.2f}   avg workgroups in phase {col.sum()/span:7.1f}")
    life = s[:, -1] - s[:, 0]
    print(f"  lifetime median {np.median(life)/1e3:.2f} us; avg resident workgroups {life.sum()/span:.1f}")
