"""Randomized soak of whole proofs for a given number of seconds: python tools/soak_proofs.py [seconds] [seed].
  * the device prover against the CPU oracle prover on random instance sizes (16..2048 steps), constants and inputs;
  * for the reference's instance (8192 steps): proofs of random inputs, randomly damaged, through the batch verifier's
    fast path, its general path and the device-resident entry point: all three must agree on every verdict, valid proofs
    must be accepted and any damage outside the unverified `points` component and the ignored tail rejected.
Test infrastructure (uses the oracle); prints a summary line."""
import os, struct, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import prover as oprover
import proofio
import stark_verifier_amd as sv
from stark_verifier_amd import batch, prover
from stark_verifier_amd.fft import _limbs

P = oprover.P
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
consts8192 = [(i ** 7) ^ 42 for i in range(64)]
size = prover.mimc_proof_size(8192, 64)
n_small = n_verify = n_damaged = 0
t_say = time.time() + 30
while time.time() < t_end:
    if time.time() > t_say:   # a line every half minute: a silent run is taken to be hung
        print("... %d s left" % int(t_end - time.time()), flush=True)
        t_say = time.time() + 30
    if rng.random() < 0.5:
        steps = 1 << int(rng.integers(4, 12))
        n_const = 1 << int(rng.integers(0, int(np.log2(steps)) + 1))
        consts = [int.from_bytes(rng.bytes(32), "little") % P for _ in range(n_const)]
        inputs = [int.from_bytes(rng.bytes(32), "little") >> int(rng.integers(0, 250)) for _ in range(int(rng.integers(1, 4)))]
        proofs, outs = prover.mk_mimc_proofs(inputs, steps, consts)
        for x, pf, o in zip(inputs, proofs, outs):
            want, wout = oprover.mk_mimc_proof(x % P, steps, consts)
            assert o == wout and pf == want, ("prover", steps, n_const, x)
        n_small += len(inputs)
    else:
        n = int(rng.integers(2, 24))
        inputs = [int.from_bytes(rng.bytes(32), "little") % P for _ in range(n)]
        proofs, outs = prover.mk_mimc_proofs(inputs, 8192, consts8192)
        expect = []
        blobs = []
        for pf in proofs:
            b = bytearray(pf)
            kind = int(rng.integers(4))
            ok = True
            if kind == 1:      # one bit anywhere
                o = int(rng.integers(len(b)))
                b[o] ^= 1 << int(rng.integers(8))
                lo = bytes(pf).find(proofio.parse_proof(pf)[0].points)
                ok = lo <= o < lo + 2048
            elif kind == 2:    # a structure word
                o = 4 * int(rng.integers(16, len(b) // 4 - 1))
                old = bytes(b[o:o + 4])
                b[o:o + 4] = struct.pack("<I", int(rng.choice([0, 31, 32, 96, 480, 2**31])))
                ok = None if bytes(b[o:o + 4]) != old else True   # verdict left to the path comparison
            elif kind == 3:    # ignored tail
                b += rng.bytes(4 * int(rng.integers(1, 8)))
            blobs.append(bytes(b))
            expect.append(ok)
        a_fast, s_fast = batch.verify_mimc_batch_io(blobs, inputs, 8192, consts8192, outs)
        os.environ["STARKHIP_BATCH_PATH"] = "general"
        a_gen, s_gen = batch.verify_mimc_batch_io(blobs, inputs, 8192, consts8192, outs)
        del os.environ["STARKHIP_BATCH_PATH"]
        assert np.array_equal(a_fast, a_gen) and np.array_equal(s_fast, s_gen), ("paths differ", s_fast, s_gen)
        for i, ok in enumerate(expect):
            if ok is not None:
                assert bool(a_fast[i]) == ok, ("verdict", i, ok, int(s_fast[i]))
        # device-resident entry point on the same bytes (equal stride: pad to the longest)
        stride = (max(len(b) for b in blobs) + 3) & ~3
        host = np.zeros((n, stride), dtype=np.uint8)
        for i, b in enumerate(blobs):
            host[i, :len(b)] = np.frombuffer(b, dtype=np.uint8)
        d_proofs = torch.from_numpy(host).cuda()
        d_in = torch.from_numpy(np.concatenate([_limbs(x) for x in inputs]).view(np.int32)).cuda()
        d_out = torch.from_numpy(np.concatenate([_limbs(x) for x in outs]).view(np.int32)).cuda()
        torch.cuda.synchronize()
        a_dev, s_dev = batch.verify_mimc_batch_dev(d_proofs.data_ptr(), stride, n, d_in.data_ptr(), 8192, consts8192, d_out.data_ptr())
        assert np.array_equal(a_dev, a_fast), ("resident differs", s_dev, s_fast)
        n_verify += n
        n_damaged += sum(1 for ok in expect if ok is not True)
print("soak ok: %d small proofs equal to the oracle's; %d proofs of the reference's instance (%d damaged) with equal verdicts on all three paths, in %.0f s"
      % (n_small, n_verify, n_damaged, budget))
