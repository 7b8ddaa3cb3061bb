"""Batch driver (stark_mimc_verify_batch): accept bits against the Python restatement of the
reference's verify_mimc_proof (tests/ref_verifier.py with the oracle plugged in, itself pinned by
proof.bin) on the reference's bundled proof and corrupted copies of it."""
import struct

import numpy as np
import pytest

import oracle
import proofio
import synth
import ref_verifier as rv
from test_oracle import _oracle_hooks

pytestmark = pytest.mark.gpu


def _ref_accepts(blob, inp, steps, consts, output):
    """What the reference does with these bytes: True, or False for both `return false` and a panic."""
    try:
        proof, _ = proofio.parse_proof(blob)
        return bool(rv.verify_mimc_proof(inp, steps, consts, output, proof, _oracle_hooks()))
    except (AssertionError, ValueError, IndexError, struct.error, ZeroDivisionError):
        return False


def _variants(good: bytes):
    proof, _ = proofio.parse_proof(good)
    v = {"good": good}
    # offsets of interesting bytes, found by searching the serialized fields
    def off_of(field: bytes, start=0):
        o = good.find(field, start)
        assert o >= 0
        return o
    def flip(o, mask=0x01):
        b = bytearray(good)
        b[o] ^= mask
        return bytes(b)
    v["witness_bit"] = flip(off_of(proof.fri_layers[2].poly_branches.branches[17].witnesses[3]) + 5, 0x10)
    v["column_value_bit"] = flip(off_of(proof.fri_layers[0].column_branches.branches[0].value) + 31)
    v["trace_value_bit"] = flip(off_of(proof.merkle_branches.branches[3].value) + 40)
    v["lincomb_sibling_bit"] = flip(off_of(proof.linear_comb_branches.branches[79].sibling_value), 0x80)
    v["root2_bit"] = flip(off_of(proof.fri_layers[4].root2) + 1)
    v["merkle_root_bit"] = flip(0)
    v["points_bit"] = flip(off_of(proof.points) + 100)          # never verified by the reference: still accepted
    v["truncated"] = good[:len(good) // 2]
    v["bad_element_type"] = good[:64] + struct.pack("<I", 7) + good[68:]
    v["trailing_bytes"] = good + b"\x00" * 13                    # tolerated by the reference's parser
    v["empty"] = b""
    # a branch that loses its last witness: shape change -> different tree depth -> mismatch
    return v


def test_batch_accept_bits_match_reference_restatement(proof_bytes):
    from stark_verifier_amd import batch
    consts = rv.mimc_constants()
    output = rv.mimc(3, 8192, consts)
    variants = _variants(proof_bytes)
    names = list(variants)
    expect = {k: _ref_accepts(variants[k], 3, 8192, consts, output) for k in names}
    assert expect["good"] and expect["points_bit"] and expect["trailing_bytes"]
    assert not any(expect[k] for k in names if k not in ("good", "points_bit", "trailing_bytes"))
    order = [names[(7 * i) % len(names)] for i in range(60)]
    accept, status = batch.verify_mimc_batch([variants[k] for k in order], 3, 8192, consts, output)
    assert [bool(a) for a in accept] == [expect[k] for k in order]
    by = {k: int(status[order.index(k)]) for k in names}
    assert by["good"] == 0 and by["points_bit"] == 0 and by["trailing_bytes"] == 0
    assert by["truncated"] == 2 and by["bad_element_type"] == 2 and by["empty"] == 2
    for k in ("witness_bit", "column_value_bit", "trace_value_bit", "lincomb_sibling_bit", "root2_bit", "merkle_root_bit"):
        assert by[k] == 4, (k, by[k])


def test_batch_public_parameters(proof_bytes):
    from stark_verifier_amd import batch
    consts = rv.mimc_constants()
    output = rv.mimc(3, 8192, consts)
    good = [proof_bytes, proof_bytes]
    # wrong output / input: boundary constraint (src/lib.rs:187)
    for inp, out in ((3, output + 1), (4, output)):
        a, s = batch.verify_mimc_batch(good, inp, 8192, consts, out)
        assert list(a) == [0, 0] and list(s) == [7, 7]
        assert not _ref_accepts(proof_bytes, inp, 8192, consts, out)
    # wrong round constant: transition constraint (src/lib.rs:181)
    bad = list(consts)
    bad[5] += 1
    a, s = batch.verify_mimc_batch(good, 3, 8192, bad, output)
    assert list(a) == [0, 0] and list(s) == [6, 6]
    assert not _ref_accepts(proof_bytes, 3, 8192, bad, output)
    # (the context keeps what it derived from the last parameter set: the right constants again, and once more)
    for _ in range(2):
        a, s = batch.verify_mimc_batch(good, 3, 8192, consts, output)
        assert list(a) == [1, 1] and list(s) == [0, 0]
    a, s = batch.verify_mimc_batch(good, 3, 8192, bad, output)
    assert list(a) == [0, 0] and list(s) == [6, 6]
    # parameter checks that make the reference return false (src/lib.rs:101-111)
    for steps, cs in ((8191, consts), (8192, consts[:63]), (2**30, consts), (32, consts)):
        a, s = batch.verify_mimc_batch(good, 3, steps, cs, output)
        assert list(a) == [0, 0] and list(s) == [1, 1], (steps, len(cs))
    # other instance sizes trip the reference's asserts (src/lib.rs:120-123)
    a, s = batch.verify_mimc_batch(good, 3, 4096, consts, output)
    assert list(a) == [0, 0] and list(s) == [9, 9]
    a, s = batch.verify_mimc_batch(good, 3, 8192, consts[:32], output)
    assert list(a) == [0, 0] and list(s) == [9, 9]
    # empty batch
    a, s = batch.verify_mimc_batch([], 3, 8192, consts, output)
    assert a.size == 0


def test_batch_chunking_and_threads_give_the_same_answer(proof_bytes, monkeypatch):
    from stark_verifier_amd import batch
    consts = rv.mimc_constants()
    output = rv.mimc(3, 8192, consts)
    v = _variants(proof_bytes)
    blobs = [v["good"], v["witness_bit"], v["good"], v["truncated"], v["good"], v["trace_value_bit"], v["points_bit"]] * 3
    ref, _ = batch.verify_mimc_batch(blobs, 3, 8192, consts, output)
    assert list(ref) == [1, 0, 1, 0, 1, 0, 1] * 3
    monkeypatch.setenv("STARKHIP_BATCH_MB", "1")   # one or two proofs per chunk
    monkeypatch.setenv("STARKHIP_HOST_THREADS", "1")
    a, _ = batch.verify_mimc_batch(blobs, 3, 8192, consts, output)
    assert np.array_equal(a, ref)


@pytest.mark.parametrize("piece_mb", ["1", "3", "9"])
def test_pipelined_upload_pieces_give_the_same_answer(proof_bytes, monkeypatch, piece_mb):
    """Proofs that all start on 4-byte boundaries go up in pieces whose upload, structure check and kernels overlap:
    one proof per piece, five per piece and pieces large enough for the pinned staging buffers give the verdicts and
    status codes of a single piece, with proofs of another structure (general path) in between."""
    from stark_verifier_amd import batch
    consts = rv.mimc_constants()
    output = rv.mimc(3, 8192, consts)
    v = _variants(proof_bytes)
    names = ["good", "witness_bit", "good", "bad_element_type", "trace_value_bit", "points_bit", "root2_bit", "good",
             "lincomb_sibling_bit", "column_value_bit", "merkle_root_bit"]
    blobs = [v[k] for k in names] * 3
    assert all(len(b) % 4 == 0 for b in blobs)
    ref_a, ref_s = batch.verify_mimc_batch(blobs, 3, 8192, consts, output)
    assert list(ref_a) == [1, 0, 1, 0, 0, 1, 0, 1, 0, 0, 0] * 3
    monkeypatch.setenv("STARKHIP_PIECE_MB", piece_mb)
    a, s = batch.verify_mimc_batch(blobs, 3, 8192, consts, output)
    assert np.array_equal(a, ref_a) and np.array_equal(s, ref_s)
    monkeypatch.setenv("STARKHIP_HOST_THREADS", "1")
    a, s = batch.verify_mimc_batch(blobs, 3, 8192, consts, output)
    assert np.array_equal(a, ref_a) and np.array_equal(s, ref_s)
    # the launch structure of a piece: a branch launch per bucket instead of one for all buckets, and the scalar checks
    # behind the branch launch instead of beside it on a second stream
    monkeypatch.setenv("STARKHIP_BATCH_GATHER", "split")
    a, s = batch.verify_mimc_batch(blobs, 3, 8192, consts, output)
    assert np.array_equal(a, ref_a) and np.array_equal(s, ref_s)
    monkeypatch.setenv("STARKHIP_BATCH_OVERLAP", "0")
    a, s = batch.verify_mimc_batch(blobs, 3, 8192, consts, output)
    assert np.array_equal(a, ref_a) and np.array_equal(s, ref_s)
    monkeypatch.delenv("STARKHIP_BATCH_GATHER")
    a, s = batch.verify_mimc_batch(blobs, 3, 8192, consts, output)
    assert np.array_equal(a, ref_a) and np.array_equal(s, ref_s)


def _merkle_blind_hooks():
    """The oracle's hooks with MultiProof::verify replaced by one that returns the values unchecked (the reference's
    length assert at src/merkle_tree.rs:28 kept): what is left decides on the scalar checks alone."""
    hooks = dict(_oracle_hooks())

    def multiproof_verify(branches, indices, root):
        assert len(branches) == len(indices)
        return [br.value for br in branches]

    hooks["multiproof_verify"] = multiproof_verify
    return hooks


def _ref_status_blind(blob, inp, steps, consts, output):
    proof, _ = proofio.parse_proof(blob)
    return rv.verify_mimc_proof_status(inp, steps, consts, output, proof, _merkle_blind_hooks())


def test_scalar_checks_status_by_status_against_the_restatement(proof_bytes, diag_ctx):
    """With Merkle mismatches ignored (stark_mimc_verify_batch_diag of the test build of the library; the shipped
    library refuses the flag and no environment variable reaches it) the scalar
    checks alone decide, so damaged values exercise every one of them. Expected status = the first failing step of the
    Python restatement of src/lib.rs:31-197 run with a MultiProof::verify that checks nothing; the device kernels
    (fast path and general path) and the host formulation must all report exactly that."""
    from stark_verifier_amd import batch
    consts = rv.mimc_constants()
    output = rv.mimc(3, 8192, consts)
    proof, _ = proofio.parse_proof(proof_bytes)
    rng = np.random.default_rng(3)
    blobs = [proof_bytes]
    fields = []
    for l in proof.fri_layers:
        fields += [b.value for b in l.column_branches.branches] + [b.value for b in l.poly_branches.branches]
    fields += [b.value for b in proof.merkle_branches.branches] + [b.value for b in proof.linear_comb_branches.branches]
    for _ in range(90):
        f = fields[int(rng.integers(len(fields)))]
        o = proof_bytes.find(f)
        b = bytearray(proof_bytes)
        b[o + int(rng.integers(len(f)))] ^= 1 << int(rng.integers(8))
        blobs.append(bytes(b))
    # damage aimed at each check: the first sampled position's L value (status 8 only), its B value (7), its D value (6)
    offs = proofio.value_offsets(proof_bytes)
    mb0 = offs["merkle", 0]
    for off in (offs["lincomb", 0] + 3, mb0 + 64 + 9, mb0 + 32 + 30):
        b = bytearray(proof_bytes)
        b[off] ^= 4
        blobs.append(bytes(b))
    expect = np.array([_ref_status_blind(b, 3, 8192, consts, output) for b in blobs], dtype=np.uint8)
    assert expect[0] == 0 and list(expect[-3:]) == [8, 7, 6] and set(expect.tolist()) >= {0, 5, 6, 7, 8}
    n = len(blobs)
    ins, outs = [3] * n, [output] * n
    blind = batch.DIAG_IGNORE_MERKLE
    for flags in (blind | batch.DIAG_GENERAL_PATH, blind | batch.DIAG_HOST_FIELD_CHECKS):
        a, s = batch.verify_mimc_batch_diag(blobs, ins, 8192, consts, outs, flags, ctx=diag_ctx)
        assert np.array_equal(s, expect), (flags, [(i, int(x), int(y)) for i, (x, y) in enumerate(zip(s, expect)) if x != y])
        assert np.array_equal(a, (expect == 0).astype(np.uint8))
    # the same proofs through the production entry point: every damaged one is a Merkle mismatch first
    a, s = batch.verify_mimc_batch(blobs, 3, 8192, consts, output)
    assert a[0] == 1 and not a[1:].any() and set(s[1:].tolist()) == {4}


def test_shipped_library_refuses_to_overlook_merkle_mismatches(proof_bytes, diag_ctx):
    """STARK_DIAG_IGNORE_MERKLE switches a soundness check off: libstarkhip.so answers STARK_ERR_UNSUPPORTED and writes
    no verdict; only tests/libstarkhip_diag.so honours it. The other two switches choose between equivalent paths and
    work in both builds."""
    from stark_verifier_amd import _lib, batch
    consts = rv.mimc_constants()
    output = rv.mimc(3, 8192, consts)
    bad = bytearray(proof_bytes)
    bad[proofio.value_offsets(proof_bytes)["lincomb", 0] + 3] ^= 4
    blobs = [proof_bytes, bytes(bad)]
    for flags in (batch.DIAG_IGNORE_MERKLE, batch.DIAG_IGNORE_MERKLE | batch.DIAG_GENERAL_PATH, 7):
        with pytest.raises(_lib.StarkError) as e:
            batch.verify_mimc_batch_diag(blobs, [3, 3], 8192, consts, [output] * 2, flags)
        assert e.value.code == _lib.STARK_ERR_UNSUPPORTED
    with pytest.raises(_lib.InvalidArgument):
        batch.verify_mimc_batch_diag(blobs, [3, 3], 8192, consts, [output] * 2, 8)
    for flags in (0, batch.DIAG_GENERAL_PATH, batch.DIAG_HOST_FIELD_CHECKS):
        a, s = batch.verify_mimc_batch_diag(blobs, [3, 3], 8192, consts, [output] * 2, flags)
        assert list(a) == [1, 0] and list(s) == [0, 4]
    a, s = batch.verify_mimc_batch_diag(blobs, [3, 3], 8192, consts, [output] * 2, batch.DIAG_IGNORE_MERKLE, ctx=diag_ctx)
    assert list(a) == [1, 0] and list(s) == [0, 8]


def _put_at(blob, off, new):
    return blob[:off] + new + blob[off + len(new):]


def test_column_value_at_or_above_p_is_rejected_like_the_reference(proof_bytes, diag_ctx):
    """src/lib.rs:85 compares eval_quartic(...) (reduced) with BigInt::from_bytes_be(column value) (not reduced): a
    column value encoded as v + p never matches, although it is the same field element. Four equal rows make the
    interpolated cubic the constant v, so the lane's evaluation is v; the column value is then v (accepted by the
    scalar checks) or v + p (rejected with status 5)."""
    from stark_verifier_amd import batch
    consts = rv.mimc_constants()
    output = rv.mimc(3, 8192, consts)
    proof, _ = proofio.parse_proof(proof_bytes)
    offs = proofio.value_offsets(proof_bytes)
    blobs = []
    for layer_i, sample, v in ((0, 0, 5), (2, 17, 351 * 2 ** 32 - 2), (4, 39, 0)):
        assert v + rv.P < 2 ** 256
        base = proof_bytes
        for j in range(4):
            base = _put_at(base, offs["poly", layer_i, 4 * sample + j], v.to_bytes(32, "big"))
        blobs.append(_put_at(base, offs["col", layer_i, sample], v.to_bytes(32, "big")))
        blobs.append(_put_at(base, offs["col", layer_i, sample], (v + rv.P).to_bytes(32, "big")))
    blobs.append(_put_at(proof_bytes, offs["col", 1, 3], b"\xff" * 32))
    expect = [_ref_status_blind(b, 3, 8192, consts, output) for b in blobs]
    assert expect == [0, 5, 0, 5, 0, 5, 5]
    n = len(blobs)
    blind = batch.DIAG_IGNORE_MERKLE
    for flags in (blind | batch.DIAG_GENERAL_PATH, blind | batch.DIAG_HOST_FIELD_CHECKS):
        a, s = batch.verify_mimc_batch_diag(blobs, [3] * n, 8192, consts, [output] * n, flags, ctx=diag_ctx)
        assert list(s) == expect, (flags, list(s))
    a, s = batch.verify_mimc_batch(blobs, 3, 8192, consts, output)      # production: Merkle mismatch either way
    assert not a.any() and set(s.tolist()) == {4}


def _mp(branches):
    out = struct.pack("<I", len(branches))
    for value, sib, wit in branches:
        out += struct.pack("<I", len(value)) + value + sib + struct.pack("<I", 32 * len(wit)) + b"".join(wit)
    return out


def test_proofs_of_other_shapes_are_rejected_without_a_crash(proof_bytes):
    """Layer counts other than five, empty multiproofs, branches without witnesses and values under 32 bytes. Where
    the reference panics (src/merkle_tree.rs:28,113,133; src/utils.rs:85 `% 0` once the domain has been divided by
    four eight times) the proof must come back as accept = 0 with status 2 or 3, never a crash, in a batch whose
    other proofs are still decided; every verdict equals the restatement's."""
    from stark_verifier_amd import batch
    consts = rv.mimc_constants()
    output = rv.mimc(3, 8192, consts)
    proof, _ = proofio.parse_proof(proof_bytes)
    head = proof_bytes[:64]
    points = struct.pack("<II", 2, 32) + bytes(32)
    empty = _mp([])

    def layer(root2, col, poly):
        return struct.pack("<I", 1) + root2 + col + poly

    def ser(mp):
        return _mp([(b.value, b.sibling_value, b.witnesses) for b in mp.branches])

    real_layers = [layer(l.root2, ser(l.column_branches), ser(l.poly_branches)) for l in proof.fri_layers]
    tail = ser(proof.merkle_branches) + ser(proof.linear_comb_branches)
    assert head + b"".join(real_layers) + proof_bytes[proof_bytes.find(struct.pack("<II", 2, 2048)):][:8 + 2048] + tail == proof_bytes
    pts = proof_bytes[proof_bytes.find(struct.pack("<II", 2, 2048)):][:8 + 2048]
    variants = {}
    for nl in (0, 1, 6, 7, 8, 9, 20):
        variants[f"{nl}_empty_layers"] = head + layer(bytes(32), empty, empty) * nl + points + empty + empty
    variants["8_layers_then_real_tail"] = head + layer(bytes(32), empty, empty) * 8 + pts + tail
    variants["0_layers_real_tail"] = head + pts + tail                       # FRI loop does not run; branches still checked
    variants["4_real_layers"] = head + b"".join(real_layers[:4]) + pts + tail
    variants["6_layers_last_repeated"] = head + b"".join(real_layers + real_layers[-1:]) + pts + tail
    b0 = proof.fri_layers[0].column_branches.branches[0]
    no_wit = [(b.value, b.sibling_value, b.witnesses if i else []) for i, b in enumerate(proof.fri_layers[0].column_branches.branches)]
    variants["branch_without_witnesses"] = head + layer(proof.fri_layers[0].root2, _mp(no_wit), ser(proof.fri_layers[0].poly_branches)) + b"".join(real_layers[1:]) + pts + tail
    short = [(b.value[:31] if i == 7 else b.value, b.sibling_value[:31] if i == 7 else b.sibling_value, b.witnesses)
             for i, b in enumerate(proof.linear_comb_branches.branches)]
    variants["value_of_31_bytes"] = head + b"".join(real_layers) + pts + ser(proof.merkle_branches) + _mp(short)
    short96 = [(b.value[:64] if i == 0 else b.value, b.sibling_value[:64] if i == 0 else b.sibling_value, b.witnesses)
               for i, b in enumerate(proof.merkle_branches.branches)]
    variants["trace_leaf_of_64_bytes"] = head + b"".join(real_layers) + pts + _mp(short96) + ser(proof.linear_comb_branches)
    names = list(variants)
    blobs = [proof_bytes] + [variants[k] for k in names] + [proof_bytes]
    accept, status = batch.verify_mimc_batch(blobs, 3, 8192, consts, output)
    assert accept[0] == 1 and accept[-1] == 1 and status[0] == 0 and status[-1] == 0
    for k, a, s in zip(names, accept[1:-1], status[1:-1]):
        assert bool(a) == _ref_accepts(variants[k], 3, 8192, consts, output), k
        assert (s == 0) == bool(a), (k, s)
        if k in ("0_layers_real_tail", "4_real_layers"):
            # the reference never looks at the number of FRI layers (src/lib.rs:53,94): a proof whose remaining
            # layers and branches verify is accepted, and so it is here
            assert a == 1, k
        elif k == "6_layers_last_repeated":
            assert s == 4, (k, s)          # well-formed, but the repeated layer's branches do not hash to its root
        else:
            assert s in (2, 3), (k, s)
    # one at a time as well (chunking and bucketing with a single odd proof)
    for k in ("8_empty_layers", "20_empty_layers", "8_layers_then_real_tail"):
        a, s = batch.verify_mimc_batch([variants[k]], 3, 8192, consts, output)
        assert list(a) == [0] and s[0] in (2, 3)


WORKER = r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist
root_dir = sys.argv[1]
sys.path.insert(0, root_dir); sys.path.insert(0, os.path.join(root_dir, "tests"))
import ref_verifier as rv
from stark_verifier_amd import batch
dist.init_process_group("gloo")
rank = dist.get_rank()
good = open(os.path.join(root_dir, "tests", "golden", "proof.bin"), "rb").read()
bad = bytearray(good); bad[len(good) // 3] ^= 2; bad = bytes(bad)
blobs = [good, bad, good, good, bad, good, good]
consts = rv.mimc_constants()
acc, st = batch.verify_mimc_batch_sharded(blobs, 3, 8192, consts, rv.mimc(3, 8192, consts))
assert list(acc) == [1, 0, 1, 1, 0, 1, 1], list(acc)
assert list(st) == [0, 4, 0, 0, 4, 0, 0], list(st)
dist.barrier()
if rank == 0:
    print("ok")
dist.destroy_process_group()
"""


def test_two_rank_sharded_batch_verify(tmp_path):
    import os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), str(script), root], capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert "ok" in r.stdout


def test_in_place_and_flattened_merkle_paths_agree(proof_bytes, monkeypatch):
    """By default the device reads branch records straight from the uploaded proof bytes (needs 4-byte aligned
    proof offsets); STARKHIP_BATCH_MERKLE=flatten copies them into structure-of-arrays buffers first. A proof
    whose length is not a multiple of 4 pushes the rest of its chunk onto the flatten path automatically."""
    from stark_verifier_amd import batch
    consts = rv.mimc_constants()
    output = rv.mimc(3, 8192, consts)
    v = _variants(proof_bytes)
    aligned = [v["good"], v["witness_bit"], v["trace_value_bit"], v["good"], v["root2_bit"], v["lincomb_sibling_bit"]] * 2
    a1, s1 = batch.verify_mimc_batch(aligned, 3, 8192, consts, output)
    assert list(a1) == [1, 0, 0, 1, 0, 0] * 2 and list(s1) == [0, 4, 4, 0, 4, 4] * 2
    misaligned = [v["trailing_bytes"]] + aligned          # 13 extra bytes: every later proof starts off a 4-byte boundary
    a2, s2 = batch.verify_mimc_batch(misaligned, 3, 8192, consts, output)
    assert list(a2) == [1] + list(a1) and list(s2) == [0] + list(s1)
    monkeypatch.setenv("STARKHIP_BATCH_MERKLE", "flatten")
    a3, s3 = batch.verify_mimc_batch(aligned, 3, 8192, consts, output)
    assert np.array_equal(a3, a1) and np.array_equal(s3, s1)


def test_fuzzed_proof_bytes_never_crash_and_match_the_restatement(proof_bytes):
    """Random damage to header and length fields, truncations and splices: the driver must return (never
    crash or read out of bounds) and agree with the reference restatement on every proof."""
    from stark_verifier_amd import batch
    consts = rv.mimc_constants()
    output = rv.mimc(3, 8192, consts)
    rng = np.random.default_rng(77)
    n = len(proof_bytes)
    blobs = []
    for t in range(120):
        b = bytearray(proof_bytes)
        kind = t % 6
        if kind == 0:      # flip bits near the start (roots, element types, first branch headers)
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, 400))] ^= 1 << int(rng.integers(8))
        elif kind == 1:    # overwrite a 4-byte aligned word somewhere with a huge or tiny length
            o = 4 * int(rng.integers(16, n // 4 - 1))
            b[o:o + 4] = struct.pack("<I", int(rng.choice([0, 1, 31, 33, 96, 2**31, 2**32 - 1, 4096])))
        elif kind == 2:    # truncate
            b = b[:int(rng.integers(0, n))]
        elif kind == 3:    # drop a slice (shifts everything after it)
            o = int(rng.integers(64, n - 200)); b = b[:o] + b[o + 4 * int(rng.integers(1, 20)):]
        elif kind == 4:    # damage inside the unverified points or anywhere random
            b[int(rng.integers(0, n))] ^= 0xFF
        else:              # duplicate a tail
            b = b + b[-int(rng.integers(1, 5000)):]
        blobs.append(bytes(b))
    accept, status = batch.verify_mimc_batch(blobs, 3, 8192, consts, output)
    expect = [_ref_accepts(b, 3, 8192, consts, output) for b in blobs]
    assert [bool(a) for a in accept] == expect
    assert all((s == 0) == bool(a) for a, s in zip(accept, status))


def test_fast_and_general_paths_agree(proof_bytes, monkeypatch):
    """Proofs whose structure words are the fixed instance's are decided on the device without host parsing;
    STARKHIP_BATCH_PATH=general sends every proof through the parsing path. Both must give the same accept bit
    and the same status for every proof: damaged values, witnesses and roots (fast path), damaged lengths and
    truncations (general path either way), and valid proofs of other inputs."""
    from stark_verifier_amd import batch, prover
    consts = rv.mimc_constants()
    rng = np.random.default_rng(21)
    inputs = [3, 5, int.from_bytes(rng.bytes(32), "little") % rv.P]
    made, outs = prover.mk_mimc_proofs(inputs, 8192, consts)
    assert made[0] == proof_bytes
    v = _variants(proof_bytes)
    blobs, ins, outs_ = [], [], []
    for k in v:
        blobs.append(v[k]); ins.append(3); outs_.append(outs[0])
    n = len(proof_bytes)
    for t in range(200):
        src = t % 3
        b = bytearray(made[src])
        if t % 4 == 3:
            o = 4 * int(rng.integers(16, n // 4 - 1))            # a whole word: often a structure word
            b[o:o + 4] = struct.pack("<I", int(rng.choice([0, 32, 96, 480, 2**31])))
        else:
            b[int(rng.integers(0, n))] ^= 1 << int(rng.integers(8))
        blobs.append(bytes(b)); ins.append(inputs[src]); outs_.append(outs[src])
    for src in range(3):
        blobs.append(made[src]); ins.append(inputs[src]); outs_.append(outs[src])
        blobs.append(made[src]); ins.append(inputs[src]); outs_.append(outs[(src + 1) % 3])   # wrong output
    a_fast, s_fast = batch.verify_mimc_batch_io(blobs, ins, 8192, consts, outs_)
    monkeypatch.setenv("STARKHIP_BATCH_PATH", "general")
    a_gen, s_gen = batch.verify_mimc_batch_io(blobs, ins, 8192, consts, outs_)
    assert np.array_equal(a_fast, a_gen)
    assert np.array_equal(s_fast, s_gen), [(i, int(x), int(y)) for i, (x, y) in enumerate(zip(s_fast, s_gen)) if x != y]
    assert list(a_fast[-6:]) == [1, 0, 1, 0, 1, 0] and list(s_fast[-6:]) == [0, 7, 0, 7, 0, 7]
    assert set(s_fast.tolist()) >= {0, 2, 4, 7}


def test_proofs_at_odd_offsets_take_the_fast_path_through_a_packed_copy(proof_bytes, monkeypatch, capfd):
    """Trailing bytes of odd length (tolerated by the reference's parser) push every later proof off a 4-byte boundary;
    such proofs are packed into an aligned buffer and still decided on the device without host parsing."""
    from stark_verifier_amd import batch
    consts = rv.mimc_constants()
    output = rv.mimc(3, 8192, consts)
    v = _variants(proof_bytes)
    blobs = []
    for i, k in enumerate(["good", "witness_bit", "good", "trace_value_bit", "good", "root2_bit", "points_bit", "good"] * 3):
        blobs.append(v[k] + b"\x07" * (1 + i % 5))
    monkeypatch.setenv("STARKHIP_BATCH_TIMING", "1")
    a, s = batch.verify_mimc_batch(blobs, 3, 8192, consts, output)
    err = capfd.readouterr().err
    assert list(a) == [1, 0, 1, 0, 1, 0, 1, 1] * 3 and list(s) == [0, 4, 0, 4, 0, 4, 0, 0] * 3
    assert "(0 proofs parsed on the host)" in err, err  # nothing went through the parsing path
    monkeypatch.setenv("STARKHIP_BATCH_PATH", "general")
    a2, s2 = batch.verify_mimc_batch(blobs, 3, 8192, consts, output)
    assert np.array_equal(a, a2) and np.array_equal(s, s2)


@pytest.mark.gpu
def test_pinned_caller_buffers_skip_the_staging_copy_and_give_the_same_answer(proof_bytes):
    """A caller's buffer that is pinned already (hipHostMalloc, a pinned tensor) is handed to the copy engine as it is,
    not copied through the library's staging buffers: same verdicts from the batch verifier, same transform from the
    host-buffer NTT entry point (both above the 8 MiB from which pageable buffers are staged)."""
    import torch
    from stark_verifier_amd import _lib, batch, fft
    from stark_verifier_amd.fft import _limbs, _ptr, MODULUS
    consts = rv.mimc_constants()
    output = rv.mimc(3, 8192, consts)
    v = _variants(proof_bytes)
    names = ["good", "witness_bit", "good", "trace_value_bit", "points_bit", "root2_bit", "good", "column_value_bit"]
    blobs = [v[k] for k in names] * 4   # 32 proofs, 18 MB
    ref_a, ref_s = batch.verify_mimc_batch(blobs, 3, 8192, consts, output)
    ctx = _lib.default_context()
    total = sum(map(len, blobs))
    pinned = torch.empty(total, dtype=torch.uint8, pin_memory=True)
    pinned.numpy()[:] = np.frombuffer(b"".join(blobs), dtype=np.uint8)
    offsets = np.zeros(len(blobs) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(b) for b in blobs], dtype=np.uint64)
    rcs = np.ascontiguousarray(np.concatenate([_limbs(c % MODULUS) for c in consts]))
    acc = np.zeros(len(blobs), dtype=np.uint8)
    st = np.zeros(len(blobs), dtype=np.uint8)
    i_, o_ = _limbs(3), _limbs(output % MODULUS)
    for _ in range(2):
        ctx.check(ctx.lib.stark_mimc_verify_batch(ctx.handle, pinned.data_ptr(), offsets.ctypes.data, len(blobs), _ptr(i_), 8192,
                                                  rcs.ctypes.data, 64, _ptr(o_), acc.ctypes.data, st.ctypes.data))
        assert np.array_equal(acc, ref_a) and np.array_equal(st, ref_s)
    # host-buffer transform: 8 polynomials of 2^16 points = 16 MiB each way, pinned in and out
    log_n, B = 16, 8
    w = pow(7, (oracle.P - 1) >> log_n, oracle.P)
    x = synth.field_elements(77, B << log_n).reshape(B, 1 << log_n, 8)
    ref = fft.ntt_batch(x, w)
    buf = torch.empty(x.size, dtype=torch.int32, pin_memory=True)   # the host entry point works in place
    buf.numpy()[:] = x.reshape(-1).view(np.int32)
    w_ = _limbs(w)
    ctx.check(ctx.lib.stark_ntt_batch(ctx.handle, buf.data_ptr(), log_n, B, _ptr(w_), 0))
    assert np.array_equal(buf.numpy().view(np.uint32).reshape(x.shape), ref)
