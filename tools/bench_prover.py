"""Proof generation on the device, then the batch verifier on the N distinct proofs it made.
python tools/bench_prover.py [N] [--verify]   -> one JSON line.
Times: stark_mimc_prove_batch_dev with inputs, proofs and outputs resident in device memory (HIP events are not
needed: the call is queued on the context's stream and the stream is drained), the host-buffer entry point
(device -> pageable host copy of 566,508 bytes per proof included), and stark_mimc_verify_batch_io on the result."""
import json, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import stark_verifier_amd as sv
from stark_verifier_amd import batch, prover
from stark_verifier_amd.fft import _limbs, MODULUS
import ref_verifier as rv

N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1000
do_verify = "--verify" in sys.argv
consts = rv.mimc_constants()
steps = 8192
size = prover.mimc_proof_size(steps, len(consts))
rng = np.random.default_rng(11)
inputs = [3] + [int.from_bytes(rng.bytes(32), "little") % MODULUS for _ in range(N - 1)]
ctx = sv.default_context()
lib = ctx.lib
inp = np.ascontiguousarray(np.concatenate([_limbs(x) for x in inputs]))
rcs = np.ascontiguousarray(np.concatenate([_limbs(c % MODULUS) for c in consts]))
d_in = torch.from_numpy(inp.view(np.int32)).cuda()
d_out = torch.empty(8 * N, dtype=torch.int32, device="cuda")
d_proofs = torch.empty(N * size, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()


def run_dev(n):
    ctx.check(lib.stark_mimc_prove_batch_dev(ctx.handle, d_in.data_ptr(), n, steps, rcs.ctypes.data, len(consts),
                                             d_proofs.data_ptr(), size, d_out.data_ptr(), None))
    ctx.synchronize()


run_dev(min(N, 64))  # tables, work buffers, first-use costs
run_dev(N)           # work buffer at its final size
t0 = time.perf_counter()
run_dev(N)
dt_dev = time.perf_counter() - t0
first = bytes(d_proofs[:size].cpu().numpy().tobytes())
gold = open(os.path.join(ROOT, "tests", "golden", "proof.bin"), "rb").read()
res = {"proofs": N, "steps": steps, "proof_bytes": size, "device_seconds": dt_dev, "device_proofs_per_s": N / dt_dev,
       "first_proof_is_reference_fixture": first == gold}
t0 = time.perf_counter()
proofs = np.empty(N * size, dtype=np.uint8)
outs = np.empty(8 * N, dtype=np.uint32)
ctx.check(lib.stark_mimc_prove_batch(ctx.handle, inp.ctypes.data, N, steps, rcs.ctypes.data, len(consts), proofs.ctypes.data,
                                     outs.ctypes.data))
dt_host = time.perf_counter() - t0
res.update({"host_seconds": dt_host, "host_proofs_per_s": N / dt_host})
if do_verify:
    offsets = (np.arange(N + 1, dtype=np.uint64) * np.uint64(size))
    acc = np.zeros(N, dtype=np.uint8); st = np.zeros(N, dtype=np.uint8)
    t0 = time.perf_counter()
    ctx.check(lib.stark_mimc_verify_batch_io(ctx.handle, proofs.ctypes.data, offsets.ctypes.data, N, inp.ctypes.data, steps,
                                             rcs.ctypes.data, len(consts), outs.ctypes.data, acc.ctypes.data, st.ctypes.data))
    dt_v = time.perf_counter() - t0
    res.update({"verify_seconds": dt_v, "verify_proofs_per_s": N / dt_v, "all_accepted": bool(acc.all()),
                "distinct_proofs": len({proofs[i * size:i * size + 64].tobytes() for i in range(N)})})
    acc = np.zeros(N, dtype=np.uint8); st = np.zeros(N, dtype=np.uint8)
    run_dev(N)
    t0 = time.perf_counter()
    ctx.check(lib.stark_mimc_verify_batch_dev(ctx.handle, d_proofs.data_ptr(), size, N, d_in.data_ptr(), steps, rcs.ctypes.data,
                                              len(consts), d_out.data_ptr(), acc.ctypes.data, st.ctypes.data))
    dt_vd = time.perf_counter() - t0
    res.update({"verify_resident_seconds": dt_vd, "verify_resident_proofs_per_s": N / dt_vd, "resident_all_accepted": bool(acc.all())})
print(json.dumps(res))
