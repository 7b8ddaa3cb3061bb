"""Proofs from a malicious prover: every Merkle branch verifies, exactly one of the scalar asserts the reference reaches
after the branches (src/lib.rs:85,181,187,190) fails, at the first, a middle or the last entry of a sampled index set.

These are the proofs a verifier exists for: the trees are consistent, so nothing is caught as a Merkle mismatch
(status 4) and the verdict rests on the lanes of the scalar check kernels alone. Expected verdicts and statuses come from
the Python restatement of src/lib.rs:31-197 (tests/ref_verifier.py) with the oracle's real MultiProof::verify; the device
must give exactly those on its fast path (host buffers, no host parsing), on the device-resident entry point, on the
general (parsing) path and with the scalar checks on host threads."""
import numpy as np
import pytest

import oracle
import proofio
import ref_verifier as rv
from test_oracle import _oracle_hooks

EXPECT = {"fold": 5, "D": 6, "B": 7, "L": 8}


def _all_branches_verify(blob):
    """Every ProofBranch::verify of the proof (src/merkle_tree.rs:119-171) with the indices the verifier derives."""
    proof, _ = proofio.parse_proof(blob)
    n, root, deg = 0, proof.l_merkle_root, 65536
    for layer in proof.fri_layers:
        ys = [int(y) for y in oracle.get_pseudorandom_indices(layer.root2, 40, deg // 4, 8)]
        for br, a in zip(layer.column_branches.branches, ys):
            assert oracle.branch_verify(br.value, br.sibling_value, br.witnesses, layer.root2, a)[0]
        pos = [y + (deg // 4) * j for y in ys for j in range(4)]
        for br, a in zip(layer.poly_branches.branches, pos):
            assert oracle.branch_verify(br.value, br.sibling_value, br.witnesses, root, a)[0]
        n += 200
        root, deg = layer.root2, deg // 4
    positions = [int(x) for x in oracle.get_pseudorandom_indices(proof.l_merkle_root, 80, 65536, 8)]
    aug = [q for x in positions for q in (x, (x + 8) % 65536)]
    for br, a in zip(proof.merkle_branches.branches, aug):
        assert oracle.branch_verify(br.value, br.sibling_value, br.witnesses, proof.merkle_root, a)[0]
    for br, a in zip(proof.linear_comb_branches.branches, positions):
        assert oracle.branch_verify(br.value, br.sibling_value, br.witnesses, proof.l_merkle_root, a)[0]
    return n + 240


def _ref_status(blob, inp, output, consts):
    proof, used = proofio.parse_proof(blob)
    assert used == len(blob)
    return rv.verify_mimc_proof_status(inp, 8192, consts, output, proof, _oracle_hooks())


def test_malicious_proofs_are_what_they_claim(malicious_proofs, proof_bytes):
    """CPU: the fixtures have the fixed instance's structure (so they take the device's fast path), all 1,240 branches
    verify under the oracle, and the restatement of the reference stops at exactly the intended assert."""
    consts = rv.mimc_constants()
    output = rv.mimc(3, 8192, consts)
    seen = set()
    for name, blob, fault, index in malicious_proofs:
        assert len(blob) == len(proof_bytes) and blob != proof_bytes
        assert _all_branches_verify(blob) == 1240, name
        st = _ref_status(blob, 3, output, consts)
        assert st == EXPECT[fault[0]], (name, st)
        seen.add(st)
    assert seen == {5, 6, 7, 8}


def test_fault_sits_at_the_intended_entry(malicious_proofs):
    """The wrong value is the `slot`-th sampled entry and no other: for a fold fault the lanes of that layer are
    recomputed here and exactly lane `slot` disagrees; for the others the faulty domain index is drawn exactly once, as
    entry `slot` of the spot-check positions."""
    consts = rv.mimc_constants()
    for name, blob, fault, index in malicious_proofs:
        proof, _ = proofio.parse_proof(blob)
        if fault[0] == "fold":
            layer, slot = fault[1], fault[2]
            root, g, deg = proof.l_merkle_root, pow(7, (rv.P - 1) // 65536, rv.P), 65536
            for k in range(layer):
                root, g, deg = proof.fri_layers[k].root2, pow(g, 4, rv.P), deg // 4
            L = proof.fri_layers[layer]
            ys = [int(y) for y in oracle.get_pseudorandom_indices(L.root2, 40, deg // 4, 8)]
            q = [pow(g, deg // 4 * j, rv.P) for j in range(4)]
            bad = []
            for i, y in enumerate(ys):
                x1 = pow(g, y, rv.P)
                rows = [int.from_bytes(L.poly_branches.branches[4 * i + j].value, "big") for j in range(4)]
                poly = rv.interp_4([q[j] * x1 % rv.P for j in range(4)], rows)
                if rv.eval_poly_at(poly, int.from_bytes(root, "big")) != int.from_bytes(L.column_branches.branches[i].value, "big"):
                    bad.append(i)
            assert bad == [slot], (name, bad)
        else:
            positions = [int(x) for x in oracle.get_pseudorandom_indices(proof.l_merkle_root, 80, 65536, 8)]
            assert positions.count(index) == 1 and positions.index(index) == fault[1], name
            # the value that is wrong sits in the branch opened for that entry, and only there
            honest, _ = proofio.parse_proof(open(proofio.__file__.replace("proofio.py", "golden/proof.bin"), "rb").read())
            assert proof.merkle_root != honest.merkle_root or fault[0] == "L"


# ---- GPU ------------------------------------------------------------------------------------------------------------

def _upload(blobs, inputs, outputs):
    import torch
    from stark_verifier_amd.fft import _limbs
    size = len(blobs[0])
    host = np.zeros((len(blobs), size), dtype=np.uint8)
    for i, b in enumerate(blobs):
        assert len(b) == size
        host[i] = np.frombuffer(b, dtype=np.uint8)
    d = torch.from_numpy(host).cuda()
    d_in = torch.from_numpy(np.concatenate([_limbs(x) for x in inputs]).view(np.int32)).cuda()
    d_out = torch.from_numpy(np.concatenate([_limbs(x) for x in outputs]).view(np.int32)).cuda()
    torch.cuda.synchronize()
    return d, d_in, d_out, size


def _all_paths(blobs, expect, consts, output, ctx, monkeypatch, capfd, label):
    """The same proofs through every path of the verifier; returns nothing, asserts identical statuses."""
    from stark_verifier_amd import batch
    n = len(blobs)
    ins, outs = [3] * n, [output] * n
    exp = np.array(expect, dtype=np.uint8)

    def check(which, a, s):
        bad = [(i, int(x), int(y)) for i, (x, y) in enumerate(zip(s, exp)) if x != y]
        assert not bad, (label, which, bad)
        assert np.array_equal(a, (exp == 0).astype(np.uint8)), (label, which)

    # fast path from host buffers: nothing may be parsed on the host
    monkeypatch.setenv("STARKHIP_BATCH_TIMING", "1")
    capfd.readouterr()
    a, s = batch.verify_mimc_batch_io(blobs, ins, 8192, consts, outs, ctx=ctx)
    assert "(0 proofs parsed on the host)" in capfd.readouterr().err
    monkeypatch.delenv("STARKHIP_BATCH_TIMING")
    check("fast", a, s)
    a, s = batch.verify_mimc_batch(blobs, 3, 8192, consts, output, ctx=ctx)
    check("fast, one input", a, s)
    # device-resident
    d, d_in, d_out, size = _upload(blobs, ins, outs)
    a, s = batch.verify_mimc_batch_dev(d.data_ptr(), size, n, d_in.data_ptr(), 8192, consts, d_out.data_ptr(), ctx=ctx)
    check("resident", a, s)
    monkeypatch.setenv("STARKHIP_BATCH_RESIDENT", "twostep")
    a, s = batch.verify_mimc_batch_dev(d.data_ptr(), size, n, d_in.data_ptr(), 8192, consts, d_out.data_ptr(), ctx=ctx)
    check("resident, two steps", a, s)
    monkeypatch.delenv("STARKHIP_BATCH_RESIDENT")
    # the launch structures of the fast path
    monkeypatch.setenv("STARKHIP_BATCH_OVERLAP", "0")
    a, s = batch.verify_mimc_batch_io(blobs, ins, 8192, consts, outs, ctx=ctx)
    check("fast, checks behind the branch launch", a, s)
    monkeypatch.delenv("STARKHIP_BATCH_OVERLAP")
    monkeypatch.setenv("STARKHIP_BATCH_ORDER", "checks")
    a, s = batch.verify_mimc_batch_io(blobs, ins, 8192, consts, outs, ctx=ctx)
    check("fast, checks queued ahead of the branch launch", a, s)
    a, s = batch.verify_mimc_batch_dev(d.data_ptr(), size, n, d_in.data_ptr(), 8192, consts, d_out.data_ptr(), ctx=ctx)
    check("resident, checks queued ahead of the branch launch", a, s)
    monkeypatch.delenv("STARKHIP_BATCH_ORDER")
    monkeypatch.setenv("STARKHIP_PIECE_MB", "1")
    a, s = batch.verify_mimc_batch_io(blobs, ins, 8192, consts, outs, ctx=ctx)
    check("fast, one proof per piece", a, s)
    monkeypatch.delenv("STARKHIP_PIECE_MB")
    # general path (host parsing, per-bucket launches, lanes packed on the host), then the scalar checks on host threads
    a, s = batch.verify_mimc_batch_diag(blobs, ins, 8192, consts, outs, batch.DIAG_GENERAL_PATH, ctx=ctx)
    check("general", a, s)
    a, s = batch.verify_mimc_batch_diag(blobs, ins, 8192, consts, outs, batch.DIAG_HOST_FIELD_CHECKS, ctx=ctx)
    check("host field checks", a, s)


@pytest.mark.gpu
def test_malicious_proofs_get_the_reference_status_on_every_path(malicious_proofs, proof_bytes, monkeypatch, capfd):
    from stark_verifier_amd import _lib
    consts = rv.mimc_constants()
    output = rv.mimc(3, 8192, consts)
    blobs = [proof_bytes] + [m[1] for m in malicious_proofs] + [proof_bytes]
    expect = [0] + [_ref_status(b, 3, output, consts) for _, b, _, _ in malicious_proofs] + [0]
    assert expect[1:-1] == [EXPECT[f[0]] for _, _, f, _ in malicious_proofs]
    _all_paths(blobs, expect, consts, output, _lib.default_context(), monkeypatch, capfd, "mixed batch")
    # each malicious proof alone (a batch of one: every lane of the launch belongs to the failing proof) ...
    for (name, blob, fault, index), st in zip(malicious_proofs, expect[1:-1]):
        _all_paths([blob], [st], consts, output, _lib.default_context(), monkeypatch, capfd, name)


@pytest.mark.gpu
def test_malicious_proof_among_many_valid_ones_and_after_a_larger_batch(malicious_proofs, proof_bytes, monkeypatch, capfd):
    """The lane and verdict buffers of the fast path are kept between calls and only ever grow. A large all-valid batch
    leaves them full of "pass" bytes; the small failing batches that follow in the same context must not read any of
    them. Also: malicious proofs at the first, a middle and the last place of a batch of 300."""
    from stark_verifier_amd import _lib, batch
    consts = rv.mimc_constants()
    output = rv.mimc(3, 8192, consts)
    ctx = _lib.Context(0)
    try:
        n = 300
        blobs = [proof_bytes] * n
        a, s = batch.verify_mimc_batch(blobs, 3, 8192, consts, output, ctx=ctx)
        assert a.all() and not s.any()
        d, d_in, d_out, size = _upload(blobs, [3] * n, [output] * n)
        a, s = batch.verify_mimc_batch_dev(d.data_ptr(), size, n, d_in.data_ptr(), 8192, consts, d_out.data_ptr(), ctx=ctx)
        assert a.all() and not s.any()
        for name, blob, fault, index in malicious_proofs:
            st = EXPECT[fault[0]]
            for small in ([blob], [proof_bytes, blob], [blob, proof_bytes, proof_bytes]):
                exp = [st if b is blob else 0 for b in small]
                a, s = batch.verify_mimc_batch(small, 3, 8192, consts, output, ctx=ctx)
                assert list(s) == exp and list(a) == [int(x == 0) for x in exp], (name, "host buffers", list(s))
                dd, di, do, _ = _upload(small, [3] * len(small), [output] * len(small))
                a, s = batch.verify_mimc_batch_dev(dd.data_ptr(), size, len(small), di.data_ptr(), 8192, consts, do.data_ptr(), ctx=ctx)
                assert list(s) == exp and list(a) == [int(x == 0) for x in exp], (name, "resident", list(s))
        # malicious proofs inside a large batch, then the large valid batch once more
        mixed = list(blobs)
        expect = [0] * n
        for k, (name, blob, fault, index) in zip([0, 1, 63, 64, 65, 127, 128, 150, 191, 192, 255, 256, 298, 299], malicious_proofs):
            mixed[k] = blob
            expect[k] = EXPECT[fault[0]]
        _all_paths(mixed, expect, consts, output, ctx, monkeypatch, capfd, "300 proofs")
        a, s = batch.verify_mimc_batch(blobs, 3, 8192, consts, output, ctx=ctx)
        assert a.all() and not s.any()
    finally:
        ctx.close()


@pytest.mark.gpu
def test_scalar_checks_without_their_tables_give_the_same_statuses(malicious_proofs, proof_bytes, monkeypatch, capfd):
    """The check kernels look up g2^pos, x^steps and the constants' polynomial in tables that a context builds with its
    first call for a parameter set; STARKHIP_CHECK_TABLES=0 makes every lane compute them itself (square-and-multiply and
    a Horner walk, the form that follows src/lib.rs:164-178 literally). Both must give the reference's statuses."""
    from stark_verifier_amd import _lib
    consts = rv.mimc_constants()
    output = rv.mimc(3, 8192, consts)
    blobs = [proof_bytes] + [m[1] for m in malicious_proofs] + [proof_bytes]
    expect = [0] + [EXPECT[f[0]] for _, _, f, _ in malicious_proofs] + [0]
    monkeypatch.setenv("STARKHIP_CHECK_TABLES", "0")
    ctx = _lib.Context(0)
    try:
        _all_paths(blobs, expect, consts, output, ctx, monkeypatch, capfd, "per-lane powers")
    finally:
        ctx.close()
