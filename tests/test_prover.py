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


@pytest.mark.gpu
def test_device_prover_reproduces_the_reference_fixture():
    from stark_verifier_amd import prover
    gold = open(GOLD, "rb").read()
    assert prover.mimc_proof_size(8192, 64) == len(gold)
    proofs, outs = prover.mk_mimc_proofs([3], 8192, rv.mimc_constants())
    assert outs[0] == rv.mimc(3, 8192, rv.mimc_constants())
    assert proofs[0] == gold


@pytest.mark.gpu
@pytest.mark.parametrize("steps,n_const", [(16, 4), (64, 16), (256, 64), (1024, 1024), (2048, 2)])
def test_device_prover_matches_oracle_small(steps, n_const):
    from stark_verifier_amd import prover
    rng = np.random.default_rng(steps)
    consts = [int.from_bytes(rng.bytes(32), "little") % P for _ in range(n_const)]
    inputs = [0, 1, P - 1, P + 5, (1 << 256) - 1] + [int.from_bytes(rng.bytes(32), "little") for _ in range(3)]
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
