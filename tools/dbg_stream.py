"""Where the streaming two-pass kernels differ from the plain ones (debugging aid): python tools/dbg_stream.py [batch]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stark_verifier_amd as sv
from stark_verifier_amd import fft
P = 2**256 - 351 * 2**32 + 1
log_n, B = 20, int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = 1 << log_n
w = pow(7, (P - 1) >> log_n, P)
outs = []
for mode in ("0", "1"):
    os.environ["STARKHIP_STREAM"] = mode
    ctx = sv.Context(0)
    x = torch.empty((B * n, 8), dtype=torch.int32, device="cuda")
    ctx.check(ctx.lib.stark_fill_random_field_dev(ctx.handle, x.data_ptr(), B * n, 9, None))
    y = torch.empty_like(x)
    fft.ntt_batch_dev(x.data_ptr(), y.data_ptr(), log_n, B, w, False, ctx)
    ctx.synchronize()
    outs.append(y.cpu().numpy().view(np.uint32).reshape(B, n, 8))
    ctx.close()
a, b = outs
bad = (a != b).any(axis=2)
print("mismatching elements:", int(bad.sum()), "of", bad.size)
for p in range(B):
    idx = np.nonzero(bad[p])[0]
    if len(idx) == 0:
        print(f"poly {p}: ok")
        continue
    k1, k2 = idx % 1024, idx // 1024   # output index = k1 + 1024 * k2
    print(f"poly {p}: {len(idx)} bad; k1 values {np.unique(k1)[:40]} ({len(np.unique(k1))} distinct); k2 values {np.unique(k2)[:40]} ({len(np.unique(k2))} distinct)")
