"""CPU stand-in for the device steps of the sharded transform, built on the oracle: lets the
orchestration in stark_verifier_amd.sharding (index maps, packing, all-to-all) be checked with
gloo on machines without a GPU. Test infrastructure only."""
import numpy as np
import torch

import oracle

P = oracle.P


class OracleLocalOps:
    def ntt_batch(self, t, log_m, batch, root, inverse):
        m = 1 << log_m
        a = t.numpy().view(np.uint32).reshape(batch, m, 8)
        for b in range(batch):
            a[b] = oracle.fft_inv(a[b], root) if inverse else oracle.fft_fwd(a[b], root)

    def twiddle_mul(self, t, rows, cols, row0, col0, log_order, root, inverse):
        w = pow(root, P - 2, P) if inverse else root
        a = t.numpy().view(np.uint32).reshape(rows * cols, 8)
        vals = oracle.limbs_to_ints(a)
        out = []
        for r in range(rows):
            base = pow(w, row0 + r, P)
            tw = pow(base, col0, P)
            for c in range(cols):
                out.append(vals[r * cols + c] * tw % P)
                tw = tw * base % P
        a[:] = oracle.ints_to_limbs(out)

    def permute3(self, t, a, b, chunk):
        return t.reshape(a, b, chunk, 8).permute(1, 0, 2, 3).contiguous().reshape(-1, 8)

    # the fused stages (the library's stark_ntt_sharded_stage_{a,c}_dev), from their definition
    def stage_a(self, t, log_n, world, rank, root, inverse):
        l1 = log_n // 2
        n1, n2 = 1 << l1, 1 << (log_n - l1)
        r1, r2 = n1 // world, n2 // world
        s = t.clone()
        self.ntt_batch(s, log_n - l1, r1, pow(root, n1, P), inverse)
        self.twiddle_mul(s, r1, n2, rank * r1, 0, log_n, root, inverse)
        return s.reshape(r1, world, r2, 8).permute(1, 0, 2, 3).contiguous().reshape(-1, 8)   # [dest][a][b]

    def stage_c(self, t, log_n, world, root, inverse):
        l1 = log_n // 2
        n1, n2 = 1 << l1, 1 << (log_n - l1)
        r2 = n2 // world
        cols = t.reshape(n1, r2, 8).permute(1, 0, 2).contiguous().reshape(-1, 8)               # [b][j1]
        self.ntt_batch(cols, l1, r2, pow(root, n2, P), inverse)
        return cols.reshape(r2, n1, 8).permute(1, 0, 2).contiguous().reshape(-1, 8)            # [k1][b]
