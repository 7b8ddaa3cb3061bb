"""Randomized soak: transforms and Merkle batches of random shapes against the CPU oracle for a given number of
seconds. python tools/soak.py [seconds] [seed]. Test infrastructure (uses the oracle); prints a summary line.
One case in ten is a batch of 2^16- or 2^20-point polynomials through the device-resident entry point with the
streaming kernels' batch threshold lowered to one work item per workgroup, so that their landing, ticket and tail paths
see every item count."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle, synth
import stark_verifier_amd as sv

P = oracle.P
os.environ.setdefault("STARKHIP_STREAM_MIN", "1")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
n_ntt = n_mk = n_stream = 0
t_say = time.time() + 30
while time.time() < t_end:
    if time.time() > t_say:   # a line every half minute: a silent run is taken to be hung
        print("... %d s left" % int(t_end - time.time()), flush=True)
        t_say = time.time() + 30
    r_case = rng.random()
    if r_case < 0.1:
        import torch
        log_n = int(rng.choice([16, 16, 20]))
        n = 1 << log_n
        batch = int(rng.integers(1, 7)) if log_n == 20 else int(rng.integers(1, 97))
        w = pow(7, (P - 1) >> log_n, P)
        ctx = sv.default_context()
        x = torch.empty((batch * n, 8), dtype=torch.int32, device="cuda")
        ctx.check(ctx.lib.stark_fill_random_field_dev(ctx.handle, x.data_ptr(), batch * n, int(rng.integers(1 << 30)), None))
        ctx.synchronize()  # the fill runs on the library's stream, the copy below on torch's
        inverse = bool(rng.integers(2))
        in_place = bool(rng.integers(2))
        y = x.clone() if in_place else torch.empty_like(x)
        torch.cuda.synchronize()
        sv.fft.ntt_batch_dev((y if in_place else x).data_ptr(), y.data_ptr(), log_n, batch, w, inverse, ctx)
        ctx.synchronize()
        for b in {0, batch - 1}:
            xin = x[b * n:(b + 1) * n].cpu().numpy().view(np.uint32)
            exp = oracle.fft_inv(xin, w) if inverse else oracle.fft_fwd(xin, w)
            assert np.array_equal(y[b * n:(b + 1) * n].cpu().numpy().view(np.uint32), exp), ("stream", log_n, batch, inverse, in_place, b)
        n_stream += 1
    elif r_case < 0.64:
        log_n = int(rng.integers(0, 17))
        n = 1 << log_n
        batch = int(rng.integers(1, max(2, min(300, (1 << 18) // n))))
        w = pow(7, (P - 1) >> log_n, P)
        x = synth.field_elements(int(rng.integers(1 << 30)), batch * n).reshape(batch, n, 8)
        if rng.random() < 0.3:   # non-canonical 256-bit inputs
            x = rng.integers(0, 2**32, size=x.shape, dtype=np.uint64).astype(np.uint32)
        inverse = bool(rng.integers(2))
        got = sv.fft.ntt_batch(x, w, inverse=inverse)
        for b in {0, batch - 1, int(rng.integers(batch))}:
            exp = oracle.fft_inv(x[b], w) if inverse else oracle.fft_fwd(x[b], w)
            assert np.array_equal(got[b], exp), ("ntt", log_n, batch, inverse, b)
        n_ntt += 1
    else:
        w = int(rng.integers(1, 21))
        vs = int(rng.choice([32, 32, 32, 36, 64, 96, 100, 33]))
        n = int(rng.integers(1, 3000))
        w = min(w, 14)   # the tree itself is built on the CPU
        case = synth.merkle_case(int(rng.integers(1 << 30)), w, n, vs)
        wit = case["witnesses"].copy()
        vals = case["values"].copy()
        for _ in range(int(rng.integers(0, 5))):
            if rng.random() < 0.5:
                wit[int(rng.integers(w)), int(rng.integers(n)), int(rng.integers(32))] ^= 1 << int(rng.integers(8))
            else:
                vals[int(rng.integers(n)), int(rng.integers(vs))] ^= 1 << int(rng.integers(8))
        idx = case["indices"].copy()
        if rng.random() < 0.3:
            idx[int(rng.integers(n))] ^= 1
        acc, dig = sv.merkle_tree.verify_batch(vals, case["siblings"], wit, idx, case["root"], want_digests=True)
        exp = oracle.merkle_verify_batch(vals, case["siblings"], wit, idx, case["root"])
        assert np.array_equal(acc, exp), ("merkle", w, vs, n)
        n_mk += 1
print("soak ok: %d transform cases, %d streaming-kernel batches, %d merkle cases in %.0f s" % (n_ntt, n_stream, n_mk, budget))
