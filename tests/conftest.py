import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def proof_bytes():
    with open(os.path.join(ROOT, "tests", "golden", "proof.bin"), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def diag_ctx():
    """A context of the test build of the library (tests/libstarkhip_diag.so, compiled with -DSTARKHIP_DIAG), the only
    build in which stark_mimc_verify_batch_diag honours STARK_DIAG_IGNORE_MERKLE."""
    from stark_verifier_amd import _lib
    ctx = _lib.Context(0, lib=_lib.load_from(os.path.join(ROOT, "tests", "libstarkhip_diag.so")))
    yield ctx
    ctx.close()


# (fault, domain index that carries it): see oracle.prover.mk_mimc_proof_faulty. First, middle and last entries of the
# sampled index sets, every FRI layer, every scalar check the reference reaches after the branches.
MALICIOUS_FAULTS = [
    (("fold", 0, 0), 12345), (("fold", 1, 39), 777), (("fold", 2, 20), 4242), (("fold", 3, 0), 99), (("fold", 4, 39), 5),
    (("D", 0), 31337), (("D", 41), 2), (("D", 79), 65535),
    (("B", 0), 50001), (("B", 40), 11), (("B", 79), 40000),
    (("L", 0), 23456), (("L", 39), 65533), (("L", 79), 9),
]


@pytest.fixture(scope="session")
def malicious_proofs():
    """[(name, proof bytes, fault, domain index of the fault)] made by the oracle's malicious prover for the reference's instance (input 3, 8192
    steps, the constants of src/main.rs:209-212): every Merkle branch verifies, exactly one scalar check fails."""
    from oracle import prover as oprover
    consts = [(i ** 7) ^ 42 for i in range(64)]
    base = oprover.columns(3, 8192, consts)
    out = []
    for fault, where in MALICIOUS_FAULTS:
        blob, _, index = oprover.mk_mimc_proof_faulty(3, 8192, consts, fault, where=where, base=base)
        out.append(("-".join(str(x) for x in fault), blob, fault, index))
    return out
