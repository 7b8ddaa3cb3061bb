"""Proof generation on the device against the CPU oracle prover and the reference's own fixture.

The oracle prover (oracle/prover.py) is pinned by reproducing tests/golden/proof.bin (the proof file the
reference ships) byte for byte; the device prover must equal the oracle on every input, and the batch
verifier must accept what it makes.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))

import ref_verifier as rv  # noqa: E402
from oracle import prover as oprover  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "proof.bin")
P = rv.P


def test_oracle_prover_reproduces_the_reference_fixture():
    gold = open(GOLD, "rb").read()
    proof, out = oprover.mk_mimc_proof(3, 8192, rv.mimc_constants())
    assert out == rv.mimc(3, 8192, rv.mimc_constants())
    assert proof == gold


def test_oracle_prover_small_instance_shapes():
    # a 64-step instance: 2 FRI layers, final 32 values; the parser reads it back
    import proofio
    proof, _ = oprover.mk_mimc_proof(5, 64, [(i ** 7) ^ 42 for i in range(16)])
    pr, used = proofio.parse_proof(proof)
    assert used == len(proof)
    assert len(pr.fri_layers) == 2 and len(pr.points) == 32 * 32
    assert len(pr.merkle_branches.branches) == 160 and len(pr.linear_comb_branches.branches) == 80


def test_proof_size_of_the_fixed_instance_matches_the_fixture():
    """Pure host arithmetic over the wire format: no device needed."""
    from stark_verifier_amd import prover
    assert prover.mimc_proof_size(8192, 64) == os.path.getsize(GOLD)
    assert prover.mimc_proof_size(64, 16) == len(oprover.mk_mimc_proof(5, 64, [(i ** 7) ^ 42 for i in range(16)])[0])
    assert prover.mimc_proof_size(8, 4) == 0 and prover.mimc_proof_size(8192, 3) == 0


@pytest.mark.gpu
def test_device_prover_reproduces_the_reference_fixture():
    from stark_verifier_amd import prover
    gold = open(GOLD, "rb").read()
    assert prover.mimc_proof_size(8192, 64) == len(gold)
    proofs, outs = prover.mk_mimc_proofs([3], 8192, rv.mimc_constants())
    assert outs[0] == rv.mimc(3, 8192, rv.mimc_constants())
    assert proofs[0] == gold


@pytest.mark.gpu
@pytest.mark.parametrize("steps,n_const", [(16, 4), (64, 16), (256, 64), (1024, 1024), (2048, 2), (32768, 64)])
def test_device_prover_matches_oracle_small(steps, n_const):
    from stark_verifier_amd import prover
    rng = np.random.default_rng(steps)
    consts = [int.from_bytes(rng.bytes(32), "little") % P for _ in range(n_const)]
    inputs = [0, 1, P - 1, P + 5, (1 << 256) - 1] + [int.from_bytes(rng.bytes(32), "little") for _ in range(3)]
    if steps > 8192:
        inputs = inputs[3:5]          # the oracle needs seconds per proof at this size (7 FRI layers, 2^18 points)
    proofs, outs = prover.mk_mimc_proofs(inputs, steps, consts)
    for x, pf, o in zip(inputs, proofs, outs):
        want, wout = oprover.mk_mimc_proof(x % P, steps, consts)
        assert o == wout
        assert pf == want


@pytest.mark.gpu
def test_device_prover_batch_accepted_by_verifier():
    from stark_verifier_amd import batch, prover
    consts = rv.mimc_constants()
    rng = np.random.default_rng(7)
    inputs = [3] + [int.from_bytes(rng.bytes(32), "little") % P for _ in range(40)]
    proofs, outs = prover.mk_mimc_proofs(inputs, 8192, consts)
    assert proofs[0] == open(GOLD, "rb").read()
    # oracle agreement on two of the random inputs
    for i in (1, 17):
        want, wout = oprover.mk_mimc_proof(inputs[i], 8192, consts)
        assert outs[i] == wout and proofs[i] == want
    accept, status = batch.verify_mimc_batch_io(proofs, inputs, 8192, consts, outs)
    assert accept.tolist() == [1] * len(inputs), status.tolist()
    # a proof checked against another proof's input/output pair fails the boundary constraint
    swapped_in = inputs[1:] + inputs[:1]
    accept, status = batch.verify_mimc_batch_io(proofs, swapped_in, 8192, consts, outs)
    assert accept.tolist() == [0] * len(inputs)
    assert set(status.tolist()) == {7}


@pytest.mark.gpu
def test_device_prover_chunking(monkeypatch):
    from stark_verifier_amd import _lib, prover
    consts = rv.mimc_constants()
    inputs = list(range(10, 17))
    whole, outs = prover.mk_mimc_proofs(inputs, 8192, consts)
    monkeypatch.setenv("STARKHIP_PROVER_MB", "40")  # two proofs per chunk
    ctx = _lib.Context(0)
    parts, outs2 = prover.mk_mimc_proofs(inputs, 8192, consts, ctx=ctx)
    ctx.close()
    assert parts == whole and outs == outs2


@pytest.mark.gpu
def test_device_prover_rejects_bad_parameters():
    from stark_verifier_amd import _lib, prover
    assert prover.mimc_proof_size(8, 4) == 0
    assert prover.mimc_proof_size(8192, 3) == 0
    assert prover.mimc_proof_size(64, 128) == 0
    with pytest.raises(_lib.InvalidArgument):
        prover.mk_mimc_proofs([1], 100, [1, 2])
    assert prover.mk_mimc_proofs([], 64, [1, 2]) == ([], [])


@pytest.mark.gpu
def test_prove_and_verify_without_leaving_the_device(monkeypatch):
    """stark_mimc_prove_batch_dev -> stark_mimc_verify_batch_dev on the same device buffers; then damage some proofs
    in place (values, witnesses, structure words, an input) and compare with the host-buffer verifier on a copy."""
    import torch
    from stark_verifier_amd import _lib, batch, prover
    from stark_verifier_amd.fft import _limbs
    consts = rv.mimc_constants()
    n = 24
    rng = np.random.default_rng(5)
    inputs = [3] + [int.from_bytes(rng.bytes(32), "little") % P for _ in range(n - 1)]
    size = prover.mimc_proof_size(8192, 64)
    stride = size + 12          # a few ignored bytes after each proof
    ctx = _lib.default_context()
    d_in = torch.from_numpy(np.concatenate([_limbs(x) for x in inputs]).view(np.int32)).cuda()
    d_out = torch.zeros(8 * n, dtype=torch.int32, device="cuda")
    d_proofs = torch.zeros(n * stride, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    prover.mk_mimc_proofs_dev(d_in.data_ptr(), n, 8192, consts, d_proofs.data_ptr(), stride, d_out.data_ptr())
    ctx.synchronize()
    assert bytes(d_proofs[:size].cpu().numpy().tobytes()) == open(GOLD, "rb").read()
    a, s = batch.verify_mimc_batch_dev(d_proofs.data_ptr(), stride, n, d_in.data_ptr(), 8192, consts, d_out.data_ptr())
    assert a.tolist() == [1] * n and s.tolist() == [0] * n
    # damage in place
    view = d_proofs.view(n, stride)
    view[1, 5000] ^= 1          # some value or witness
    view[2, 70] ^= 4            # root2 of the first layer
    view[3, 100] = 77           # the first branch count: structure word -> general path
    view[4, size - 40] ^= 8     # last witness of the last branch
    view[5, 64] = 3             # element tag: parser error
    d_in2 = d_in.clone()
    d_in2[8 * 6] += 1           # proof 6 checked against another input
    torch.cuda.synchronize()
    a, s = batch.verify_mimc_batch_dev(d_proofs.data_ptr(), stride, n, d_in2.data_ptr(), 8192, consts, d_out.data_ptr())
    host = d_proofs.cpu().numpy()
    blobs = [host[i * stride:(i + 1) * stride].tobytes() for i in range(n)]
    ins = list(inputs)
    ins[6] += 1
    outs = [int.from_bytes(d_out[8 * i:8 * i + 8].cpu().numpy().astype("<u4").tobytes(), "little") for i in range(n)]
    a_h, s_h = batch.verify_mimc_batch_io(blobs, ins, 8192, consts, outs)
    assert np.array_equal(a, a_h) and np.array_equal(s, s_h)
    assert a.tolist() == [1, 0, 0, 0, 0, 0, 0] + [1] * (n - 7)
    assert s[1] == 4 and s[2] == 4 and s[4] == 4 and s[5] == 2 and s[6] == 7 and s[3] != 0
    monkeypatch.setenv("STARKHIP_BATCH_PATH", "general")
    a_g, s_g = batch.verify_mimc_batch_dev(d_proofs.data_ptr(), stride, n, d_in2.data_ptr(), 8192, consts, d_out.data_ptr())
    assert np.array_equal(a, a_g) and np.array_equal(s, s_g)
    # asking for the structure first and checking the matching proofs only (the default queues both at once)
    monkeypatch.delenv("STARKHIP_BATCH_PATH")
    monkeypatch.setenv("STARKHIP_BATCH_RESIDENT", "twostep")
    a_t, s_t = batch.verify_mimc_batch_dev(d_proofs.data_ptr(), stride, n, d_in2.data_ptr(), 8192, consts, d_out.data_ptr())
    assert np.array_equal(a, a_t) and np.array_equal(s, s_t)


@pytest.mark.gpu
def test_mimc_batch_matches_the_reference_function():
    """utils::mimc for many inputs: any step count >= 1 and any number of constants (the reference indexes i % len)."""
    from stark_verifier_amd import _lib, prover
    rng = np.random.default_rng(12)
    inputs = [0, 1, 3, P - 1, P, (1 << 256) - 1] + [int.from_bytes(rng.bytes(32), "little") for _ in range(70)]
    for steps, n_const in ((1, 0), (1, 3), (2, 1), (3, 2), (100, 3), (257, 64), (8192, 64)):
        consts = [int.from_bytes(rng.bytes(32), "little") for _ in range(n_const)] if steps != 8192 else rv.mimc_constants()
        got = prover.mimc(inputs, steps, consts)
        for x, g in zip(inputs, got):
            assert g == rv.mimc(x % P, steps, [c % P for c in consts]) % P, (steps, n_const, x)
    assert prover.mimc([], 10, [1]) == []
    with pytest.raises(_lib.InvalidArgument):
        prover.mimc([1], 0, [1])
    with pytest.raises(_lib.InvalidArgument):
        prover.mimc([1], 5, [])


@pytest.mark.gpu
def test_device_resident_verifier_odd_strides_and_empty_batch():
    """A stride that is not a multiple of 4 (or shorter than a proof) cannot take the in-place path: every proof is
    copied out and parsed, with the same verdicts."""
    import torch
    from stark_verifier_amd import batch, prover
    from stark_verifier_amd.fft import _limbs
    consts = rv.mimc_constants()
    inputs = [3, 9, 27]
    proofs, outs = prover.mk_mimc_proofs(inputs, 8192, consts)
    size = len(proofs[0])
    d_in = torch.from_numpy(np.concatenate([_limbs(x) for x in inputs]).view(np.int32)).cuda()
    d_out = torch.from_numpy(np.concatenate([_limbs(x) for x in outs]).view(np.int32)).cuda()
    for stride in (size + 1, size + 2, size - 4):
        host = np.zeros((3, stride), dtype=np.uint8)
        for i, p in enumerate(proofs):
            host[i, :min(size, stride)] = np.frombuffer(p, dtype=np.uint8)[:stride]
        host[1, 3000] ^= 2
        d = torch.from_numpy(host).cuda()
        torch.cuda.synchronize()
        a, s = batch.verify_mimc_batch_dev(d.data_ptr(), stride, 3, d_in.data_ptr(), 8192, consts, d_out.data_ptr())
        if stride >= size:
            assert a.tolist() == [1, 0, 1] and s[1] == 4, (stride, s.tolist())
        else:
            assert a.tolist() == [0, 0, 0] and set(s.tolist()) == {2}, (stride, s.tolist())   # truncated: parser error
    a, s = batch.verify_mimc_batch_dev(0, size, 0, 0, 8192, consts, 0)
    assert a.size == 0


@pytest.mark.gpu
def test_device_prover_dev_entry_point_checks_the_stride():
    import torch
    from stark_verifier_amd import _lib, prover
    from stark_verifier_amd.fft import _limbs
    consts = rv.mimc_constants()
    size = prover.mimc_proof_size(8192, 64)
    d_in = torch.from_numpy(_limbs(3).view(np.int32)).cuda()
    d_out = torch.zeros(8, dtype=torch.int32, device="cuda")
    d_proofs = torch.zeros(size + 64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for stride in (size - 4, size + 2):
        with pytest.raises(_lib.InvalidArgument):
            prover.mk_mimc_proofs_dev(d_in.data_ptr(), 1, 8192, consts, d_proofs.data_ptr(), stride, d_out.data_ptr())
    prover.mk_mimc_proofs_dev(d_in.data_ptr(), 1, 8192, consts, d_proofs.data_ptr(), size + 64, d_out.data_ptr())
    _lib.default_context().synchronize()
    assert bytes(d_proofs[:size].cpu().numpy().tobytes()) == open(GOLD, "rb").read()
    assert int(d_proofs[size:].sum()) == 0          # nothing written past the proof
