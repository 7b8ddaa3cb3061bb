"""GPU parity tests: the HIP library (through its C ABI) against the CPU oracle.

Everything here is integer / byte work, so the bar is bit-exact equality.
"""
import hashlib
import os
import random

import numpy as np
import pytest

import oracle
import proofio
import ref_verifier as rv
import synth

pytestmark = pytest.mark.gpu

P = oracle.P


@pytest.fixture(scope="module")
def sv():
    import stark_verifier_amd as sv
    sv.default_context()  # fails loudly without a GPU / without the built library
    return sv


def root_of_unity(log_n):
    return pow(7, (P - 1) >> log_n, P)


# ---- field arithmetic -------------------------------------------------------------------

EDGE = [0, 1, 2, P - 1, P - 2, 2**255, 2**256 - 1, P, P + 1, 351 * 2**32 - 1, 2**32 - 1, 2**32,
        (1 << 224) - 1, 0x15EFFFFFFFF, 0x1E140FFFFFD4200000001, (1 << 256) - (1 << 32), P - (1 << 32)]


def test_field_ops_edges_and_random(sv):
    rng = random.Random(11)
    xs = EDGE + [rng.getrandbits(256) for _ in range(64)]
    a = [x for x in xs for _ in xs]
    b = [y for _ in xs for y in xs]
    a += [rng.getrandbits(256) for _ in range(50000)]
    b += [rng.getrandbits(256) for _ in range(50000)]
    A, B = oracle.ints_to_limbs(a), oracle.ints_to_limbs(b)
    for op in ("add", "sub", "mul"):
        got = sv.fft.field_op(op, A, B)
        exp = oracle.fe_batch(op, A, B)
        assert np.array_equal(got, exp), op
    # and the oracle agrees with Python integers on the edge block
    m = len(xs) ** 2
    got = oracle.limbs_to_ints(sv.fft.field_op("mul", A[:m], B[:m]))
    assert got == [(x % P) * (y % P) % P for x, y in zip(a[:m], b[:m])]
    got = oracle.limbs_to_ints(sv.fft.field_op("canon", A[:m]))
    assert got == [x % P for x in a[:m]]


def test_field29_ops_edges_and_random(sv):
    """The carry-free 29-bit-limb arithmetic the transform kernels use, against Python integers."""
    rng = random.Random(12)
    xs = EDGE + [rng.getrandbits(256) for _ in range(48)]
    a = [x for x in xs for _ in xs]
    b = [y for _ in xs for y in xs]
    a += [rng.getrandbits(256) for _ in range(60000)]
    b += [rng.getrandbits(256) for _ in range(60000)]
    # operands built from all-ones / all-zero 29-bit limbs stress the carry paths
    lim = [0, (1 << 29) - 1, 1 << 28, 1, (1 << 29) - 2]
    for _ in range(4000):
        a.append(sum(rng.choice(lim) << (29 * i) for i in range(9)) % (1 << 256))
        b.append(sum(rng.choice(lim) << (29 * i) for i in range(9)) % (1 << 256))
    A, B = oracle.ints_to_limbs(a), oracle.ints_to_limbs(b)
    for op, f in (("mul29", lambda x, y: x * (y % P) % P), ("add29", lambda x, y: (x + y) % P),
                  ("sub29", lambda x, y: (x - y) % P), ("lazymul29", lambda x, y: (2 * x - 2 * y) * y % P)):
        got = oracle.limbs_to_ints(sv.fft.field_op(op, A, B))
        exp = [f(x, y) for x, y in zip(a, b)]
        bad = [i for i in range(len(a)) if got[i] != exp[i]]
        assert not bad, (op, len(bad), hex(a[bad[0]]), hex(b[bad[0]]))


# ---- NTT ---------------------------------------------------------------------------------

@pytest.mark.parametrize("log_n", list(range(0, 15)))
def test_ntt_forward_inverse_vs_oracle(sv, log_n):
    n = 1 << log_n
    batch = 3 if log_n < 12 else 2
    w = root_of_unity(log_n)
    x = synth.field_elements(synth.SEED + log_n, batch * n).reshape(batch, n, 8)
    fwd = sv.fft.ntt_batch(x, w, inverse=False)
    inv = sv.fft.ntt_batch(x, w, inverse=True)
    for b in range(batch):
        assert np.array_equal(fwd[b], oracle.fft_fwd(x[b], w)), ("fwd", b)
        assert np.array_equal(inv[b], oracle.fft_inv(x[b], w)), ("inv", b)
    assert np.array_equal(sv.fft.ntt_batch(fwd, w, inverse=True), x)


def test_ntt_2p16_full_oracle_compare(sv):
    log_n, n = 16, 1 << 16
    w = root_of_unity(log_n)
    assert w == 0x5caa52206ecc9d72853f7c616e06672484967c1e2e5a87d6922259c0982b76cc  # G2, src/lib.rs:121
    x = synth.field_elements(synth.SEED + 2, 2 * n).reshape(2, n, 8)
    fwd = sv.fft.ntt_batch(x, w, inverse=False)
    assert np.array_equal(fwd[0], oracle.fft_fwd(x[0], w))
    assert np.array_equal(fwd[1], oracle.fft_fwd(x[1], w))
    inv = sv.fft.ntt_batch(x, w, inverse=True)
    assert np.array_equal(inv[1], oracle.fft_inv(x[1], w))
    assert np.array_equal(sv.fft.ntt_batch(fwd, w, inverse=True), x)


def test_ntt_noncanonical_inputs(sv):
    # inputs >= p are legal 256-bit values; results must equal the transform of their residues
    log_n, n = 6, 64
    w = root_of_unity(log_n)
    rng = random.Random(5)
    vals = [rng.choice([P, P + 1, 2**256 - 1, P + rng.getrandbits(40), rng.getrandbits(256)]) for _ in range(n)]
    x = oracle.ints_to_limbs(vals)
    got = sv.fft.ntt_batch(x, w, inverse=False)
    assert np.array_equal(got, oracle.fft_fwd(oracle.ints_to_limbs([v % P for v in vals]), w))


def test_fft_inv_reference_call_site(sv):
    # fft_inv(round_constants, val, modulus) exactly as src/lib.rs:125 calls it
    consts = rv.mimc_constants()
    val = 56670364103764250102176604807203318908867195832872336813161821519223575486477
    out = sv.fft.fft_inv(consts, val, P)
    assert out == oracle.limbs_to_ints(oracle.fft_inv(oracle.ints_to_limbs(consts), val))
    assert out[0] == 0x7825542ac5
    assert out[63] == 0xfd0228d8bad580fd57d1b765459647de9d36dd4494d4c94f69da1bed69fc475e
    h = hashlib.blake2s(b"".join(x.to_bytes(32, "big") for x in out)).hexdigest()
    assert h == "978477fb3b799ecb5ed286b82d8e088ea4ce48fe2b46dc5de7bf8c4d24c13c06"
    assert sv.fft.fft(out, val, P) == consts


def test_fft_rejects_what_the_reference_panics_on(sv):
    w8 = root_of_unity(3)
    with pytest.raises(ValueError):
        sv.fft.fft_inv(list(range(16)), w8, P)  # order 8, 16 values
    with pytest.raises(ValueError):
        sv.fft.fft_inv(list(range(4)), w8, P)  # order 8, 4 values
    with pytest.raises(ValueError):
        sv.fft.fft_inv(list(range(6)), w8, P)  # not a power of two
    with pytest.raises(ValueError):
        sv.fft.fft_inv([], w8, P)
    with pytest.raises(sv.StarkError):
        sv.fft.fft_inv(list(range(8)), w8, P - 2)  # another modulus
    assert sv.fft.fft_inv([5], 1, P) == [5]
    assert sv.fft.fft_inv([5 + P], 1, P) == [5]


@pytest.mark.parametrize("log_n,split", [(20, None), (18, "6,6,6"), (21, "7,7,7"), (17, None), (13, "4,9")])
def test_ntt_large_sampled_and_roundtrip(sv, log_n, split, monkeypatch):
    """Sizes the oracle cannot afford in full: direct-sum evaluation at sampled indices
    (out[k] = sum_j v[j] w^(jk)) plus inverse(forward(x)) == x over the whole vector."""
    if split:
        monkeypatch.setenv("STARKHIP_SPLIT", split)
    n = 1 << log_n
    w = root_of_unity(log_n)
    ctx = sv.Context(0)  # fresh context so the split override is honoured
    x = synth.field_elements(synth.SEED + 100 + log_n, n)
    fwd = sv.fft.ntt_batch(x, w, inverse=False, ctx=ctx)
    rng = random.Random(log_n)
    ks = [0, 1, n - 1, n // 2, n // 2 + 1] + [rng.randrange(n) for _ in range(3)]
    for k in ks:
        assert oracle.limbs_to_int(fwd[k]) == oracle.dft_at(x, w, k), k
    back = sv.fft.ntt_batch(fwd, w, inverse=True, ctx=ctx)
    assert np.array_equal(back, x)
    ctx.close()


@pytest.mark.parametrize("log_n,split,logc", [(13, None, None), (14, "4,10", None), (16, None, "0"), (17, "9,8", None), (20, None, None)])
def test_ntt_two_pass_intermediate_layouts_agree(sv, log_n, split, logc, monkeypatch):
    """Two-pass plans keep their intermediate tile by tile (default) or in place at the row stride
    (STARKHIP_TILED_MID=0): same outputs, forward and inverse, and equal to the oracle where it is affordable."""
    if split:
        monkeypatch.setenv("STARKHIP_SPLIT", split)
    if logc:
        monkeypatch.setenv("STARKHIP_LOGC", logc)
    n, batch = 1 << log_n, 3 if log_n <= 17 else 1
    w = root_of_unity(log_n)
    x = synth.field_elements(synth.SEED + 300 + log_n, batch * n).reshape(batch, n, 8)
    outs = []
    for mode in ("1", "0"):
        monkeypatch.setenv("STARKHIP_TILED_MID", mode)
        ctx = sv.Context(0)  # the layout is chosen when a context builds the plan
        fwd = sv.fft.ntt_batch(x, w, inverse=False, ctx=ctx)
        inv = sv.fft.ntt_batch(x, w, inverse=True, ctx=ctx)
        ctx.close()
        outs.append((fwd, inv))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])
    if log_n <= 14:
        for b in range(batch):
            assert np.array_equal(outs[0][0][b], oracle.fft_fwd(x[b], w))
            assert np.array_equal(outs[0][1][b], oracle.fft_inv(x[b], w))


@pytest.mark.parametrize("log_n,batch", [(16, 70), (21, 3), (10, 5000), (13, 300)])
def test_ntt_host_buffers_pipelined_equals_sequential(sv, log_n, batch, monkeypatch):
    """Host-buffer batches larger than one staging buffer are uploaded, transformed and downloaded group by group on
    three streams; STARKHIP_HOST_PIPELINE=0 keeps the three steps in sequence. Same outputs, both directions, and equal
    to the oracle on the first, a middle and the last polynomial."""
    n = 1 << log_n
    w = root_of_unity(log_n)
    x = synth.field_elements(synth.SEED + 400 + log_n, batch * n).reshape(batch, n, 8)
    for inverse in (False, True):
        monkeypatch.setenv("STARKHIP_HOST_PIPELINE", "1")
        a = sv.fft.ntt_batch(x, w, inverse=inverse)
        monkeypatch.setenv("STARKHIP_HOST_PIPELINE", "0")
        b = sv.fft.ntt_batch(x, w, inverse=inverse)
        assert np.array_equal(a, b)
        if log_n <= 16:
            for i in (0, batch // 2, batch - 1):
                want = oracle.fft_inv(x[i], w) if inverse else oracle.fft_fwd(x[i], w)
                assert np.array_equal(a[i], want), (inverse, i)


@pytest.mark.parametrize("log_n", [2, 3, 4, 5, 8, 11, 12])
def test_ntt_single_pass_inverse_scale_on_the_last_round_or_after_it(sv, log_n, monkeypatch):
    """Single-pass inverse transforms carry n^-1 on the last round's twiddles (default, from 8 points on) or apply it to
    the outputs (STARKHIP_FOLD_SCALE=0): both equal the oracle's fft_inv, non-canonical inputs included."""
    n, batch = 1 << log_n, 5
    w = root_of_unity(log_n)
    x = synth.field_elements(synth.SEED + 500 + log_n, batch * n).reshape(batch, n, 8)
    x[1] = 0xFFFFFFFF  # every element 2^256 - 1
    want = np.stack([oracle.fft_inv(oracle.ints_to_limbs([v % P for v in oracle.limbs_to_ints(x[b])]), w) for b in range(batch)])
    for mode in ("1", "0"):
        monkeypatch.setenv("STARKHIP_FOLD_SCALE", mode)
        ctx = sv.Context(0)  # the choice is made when a context builds the plan
        got = sv.fft.ntt_batch(x, w, inverse=True, ctx=ctx)
        ctx.close()
        assert np.array_equal(got, want), mode


def test_ntt_linearity_2p20(sv):
    log_n, n = 20, 1 << 20
    w = root_of_unity(log_n)
    x = synth.field_elements(7, n)
    y = synth.field_elements(8, n)
    s = oracle.fe_batch("add", x, y)
    fx, fy, fs = (sv.fft.ntt_batch(v, w) for v in (x, y, s))
    assert np.array_equal(fs, oracle.fe_batch("add", fx, fy))


# ---- Merkle ------------------------------------------------------------------------------

def test_merkle_kat_reference_unit_test_vector(sv):
    from test_oracle import KAT_DIGESTS, KAT_ROOT, KAT_VALUE
    digs = [bytes.fromhex(d) for d in KAT_DIGESTS]
    br = sv.merkle_tree.ProofBranch(witnesses=digs[1:], sibling_value=digs[0], value=bytes.fromhex(KAT_VALUE))
    assert br.verify(bytes.fromhex(KAT_ROOT), 5845) == bytes.fromhex(KAT_VALUE)
    with pytest.raises(sv.merkle_tree.MerkleMismatch):
        br.verify(bytes.fromhex(KAT_ROOT), 5846)
    assert br.verify(bytes.fromhex(KAT_ROOT), 5846, strict=False) is None


def _gpu_hooks(sv, counter):
    def fft_inv(vals, root):
        return sv.fft.fft_inv(vals, root, P)

    def multiproof_verify(branches, indices, root):
        mp = sv.merkle_tree.MultiProof(
            [sv.merkle_tree.ProofBranch(b.witnesses, b.sibling_value, b.value) for b in branches], root)
        counter[0] += len(branches)
        return mp.verify([int(i) for i in indices], None, strict=False)

    def get_indices(seed, count, modulus, exclude):
        return [int(x) for x in oracle.get_pseudorandom_indices(seed, count, modulus, exclude)]

    return {"fft_inv": fft_inv, "multiproof_verify": multiproof_verify, "get_indices": get_indices}


def test_proof_bin_verifies_through_gpu_primitives(sv, proof_bytes):
    """The reference's integration check with fft_inv and MultiProof::verify on the GPU."""
    proof, _ = proofio.parse_proof(proof_bytes)
    consts = rv.mimc_constants()
    output = rv.mimc(3, 8192, consts)
    counter = [0]
    assert rv.verify_mimc_proof(3, 8192, consts, output, proof, _gpu_hooks(sv, counter))
    assert counter[0] == 1240
    # corrupt one witness, one value, one sibling: each must flip the decision
    for mutate in ("witness", "value", "sibling"):
        proof, _ = proofio.parse_proof(proof_bytes)
        br = proof.merkle_branches.branches[33]
        if mutate == "witness":
            w = bytearray(br.witnesses[14]); w[31] ^= 1; br.witnesses[14] = bytes(w)
        elif mutate == "value":
            v = bytearray(br.value); v[95] ^= 0x80; br.value = bytes(v)
        else:
            v = bytearray(br.sibling_value); v[40] ^= 2; br.sibling_value = bytes(v)
        assert not rv.verify_mimc_proof(3, 8192, consts, output, proof, _gpu_hooks(sv, [0])), mutate


def test_proof_bin_digests_match_oracle_per_branch(sv, proof_bytes):
    proof, _ = proofio.parse_proof(proof_bytes)
    layer = proof.fri_layers[1]
    ys = oracle.get_pseudorandom_indices(layer.root2, 40, 4096, 8)
    brs = layer.column_branches.branches
    w = len(brs[0].witnesses)
    values = np.frombuffer(b"".join(b.value for b in brs), dtype=np.uint8).reshape(len(brs), 32)
    sibs = np.frombuffer(b"".join(b.sibling_value for b in brs), dtype=np.uint8).reshape(len(brs), 32)
    wit = np.frombuffer(b"".join(b.witnesses[l] for l in range(w) for b in brs), dtype=np.uint8).reshape(w, len(brs), 32)
    root = np.frombuffer(layer.root2, dtype=np.uint8)
    acc, dig = sv.merkle_tree.verify_batch(values, sibs, wit, ys, root, want_digests=True)
    assert acc.all()
    for i, b in enumerate(brs):
        ok, d = oracle.branch_verify(b.value, b.sibling_value, b.witnesses, layer.root2, int(ys[i]))
        assert ok and d == dig[i].tobytes()
    # wrong indices: digests must still equal the oracle's recomputation, accept must be 0
    bad = (ys + 1).astype(np.uint32)
    acc, dig = sv.merkle_tree.verify_batch(values, sibs, wit, bad, root, want_digests=True)
    for i, b in enumerate(brs):
        ok, d = oracle.branch_verify(b.value, b.sibling_value, b.witnesses, layer.root2, int(bad[i]))
        assert bool(acc[i]) == ok and d == dig[i].tobytes()


@pytest.mark.parametrize("w,vs,n", [(1, 32, 64), (2, 32, 100), (5, 32, 1000), (12, 32, 5000), (19, 32, 20000),
                                    (7, 96, 3000), (4, 64, 500), (3, 33, 300), (3, 100, 300), (6, 160, 700)])
def test_merkle_synthetic_trees_with_corruption(sv, w, vs, n):
    case = synth.merkle_case(1000 + w * 7 + vs, w, n, vs)
    rng = np.random.default_rng(w + vs)
    values, sibs, wit, idx = (case[k].copy() for k in ("values", "siblings", "witnesses", "indices"))
    # 5% of branches get one flipped bit somewhere
    bad = rng.choice(n, size=max(1, n // 20), replace=False)
    for i in bad:
        which = rng.integers(0, 3)
        if which == 0:
            values[i, rng.integers(0, vs)] ^= 1 << rng.integers(0, 8)
        elif which == 1:
            sibs[i, rng.integers(0, vs)] ^= 1 << rng.integers(0, 8)
        else:
            wit[rng.integers(0, w), i, rng.integers(0, 32)] ^= 1 << rng.integers(0, 8)
    acc, dig = sv.merkle_tree.verify_batch(values, sibs, wit, idx, case["root"], want_digests=True)
    exp = oracle.merkle_verify_batch(values, sibs, wit, idx, case["root"])
    assert np.array_equal(acc, exp)
    expect_bad = np.zeros(n, dtype=bool)
    expect_bad[bad] = True
    assert np.array_equal(acc == 0, expect_bad)


def test_merkle_multiple_roots_and_root_ids(sv):
    a = synth.merkle_case(1, 6, 300)
    b = synth.merkle_case(2, 6, 300)
    values = np.concatenate([a["values"], b["values"]])
    sibs = np.concatenate([a["siblings"], b["siblings"]])
    wit = np.concatenate([a["witnesses"], b["witnesses"]], axis=1)
    idx = np.concatenate([a["indices"], b["indices"]])
    roots = np.stack([a["root"], b["root"]])
    rid = np.array([0] * 300 + [1] * 300, dtype=np.uint32)
    assert sv.merkle_tree.verify_batch(values, sibs, wit, idx, roots, rid).all()
    rid[5], rid[400] = 1, 0
    acc = sv.merkle_tree.verify_batch(values, sibs, wit, idx, roots, rid)
    assert list(np.flatnonzero(acc == 0)) == [5, 400]
    assert np.array_equal(acc, oracle.merkle_verify_batch(values, sibs, wit, idx, roots, rid))
    with pytest.raises(ValueError):
        rid[0] = 2
        sv.merkle_tree.verify_batch(values, sibs, wit, idx, roots, rid)


def test_merkle_rejects_what_the_reference_panics_on(sv):
    z32 = np.zeros((4, 32), dtype=np.uint8)
    idx = np.zeros(4, dtype=np.uint32)
    with pytest.raises(ValueError):  # no witnesses: permute_4_index divides by zero
        sv.merkle_tree.verify_batch(z32, z32, np.zeros((0, 4, 32), dtype=np.uint8), idx, z32[:1])
    with pytest.raises(ValueError):  # value shorter than a digest
        sv.merkle_tree.verify_batch(z32[:, :16], z32[:, :16], np.zeros((2, 4, 32), dtype=np.uint8), idx, z32[:1])
    with pytest.raises(AssertionError):  # branches len != indices len (src/merkle_tree.rs:28)
        sv.merkle_tree.MultiProof([sv.merkle_tree.ProofBranch([bytes(32)], bytes(32), bytes(32))]).verify([1, 2])
    assert sv.merkle_tree.MultiProof([], bytes(32)).verify([]) == []


def test_merkle_index_out_of_range_matches_oracle(sv):
    case = synth.merkle_case(9, 4, 64)
    idx = case["indices"].astype(np.uint64)
    big = (idx + np.arange(64, dtype=np.uint64) * 32 + (1 << 31)).astype(np.uint32)
    acc = sv.merkle_tree.verify_batch(case["values"], case["siblings"], case["witnesses"], big, case["root"])
    exp = oracle.merkle_verify_batch(case["values"], case["siblings"], case["witnesses"], big, case["root"])
    assert np.array_equal(acc, exp)


def test_merkle_1m_depth20_full_size_properties(sv):
    """Full benchmark size (1M branches, depth 20): accept vector known by construction."""
    n, w = 1 << 20, 19
    case = synth.merkle_case(77, w, n)
    rng = np.random.default_rng(3)
    wit = case["witnesses"]
    bad = rng.choice(n, size=n // 100, replace=False)
    wit[rng.integers(0, w, size=bad.size), bad, rng.integers(0, 32, size=bad.size)] ^= 0x10
    acc = sv.merkle_tree.verify_batch(case["values"], case["siblings"], wit, case["indices"], case["root"])
    exp = np.ones(n, dtype=np.uint8)
    exp[bad] = 0
    assert np.array_equal(acc, exp)
    sample = rng.choice(n, size=2000, replace=False)
    exp_s = oracle.merkle_verify_batch(case["values"][sample], case["siblings"][sample],
                                       np.ascontiguousarray(wit[:, sample]), case["indices"][sample], case["root"])
    assert np.array_equal(acc[sample], exp_s)


def test_cpp_host_mirror_binary(sv):
    """host/stark_host.hpp (the C++ mirror of fft_inv / ProofBranch::verify / MultiProof::verify)
    driven by host/test_host.cpp: the reference's unit-test vector and its fft_inv call site."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "host", "test_host")
    if not os.path.exists(exe):
        pytest.fail("host/test_host is not built (make -C stark_verifier_amd/csrc host)")
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, os.path.join(root, "tests", "golden", "proof.bin")], capture_output=True, text=True,
                       timeout=120, env=env)
    assert r.returncode == 0 and "host mirror ok" in r.stdout, r.stdout + r.stderr


def test_twiddle_table_cache_eviction_keeps_results_exact(sv, monkeypatch):
    """With a 1 MiB table budget every new size evicts the previous plans; results must not change."""
    monkeypatch.setenv("STARKHIP_TABLE_MB", "1")
    ctx = sv.Context(0)
    try:
        for rep in range(2):
            for log_n in (13, 14, 15, 13):
                w = root_of_unity(log_n)
                x = synth.field_elements(500 + log_n, 1 << log_n)
                got = sv.fft.ntt_batch(x, w, inverse=bool(rep), ctx=ctx)
                exp = oracle.fft_inv(x, w) if rep else oracle.fft_fwd(x, w)
                assert np.array_equal(got, exp), (rep, log_n)
    finally:
        ctx.close()


@pytest.mark.parametrize("log_n,batches", [(0, (1, 5)), (1, (1, 7)), (2, (1, 255, 257)), (3, (3, 130)), (5, (1, 33)),
                                           (6, (17,)), (8, (5,)), (9, (3,)), (11, (3,)), (12, (1, 2)), (13, (3,)), (15, (5,))])
def test_ntt_ragged_batches(sv, log_n, batches):
    """Batch counts that do not fill the last workgroup (several polynomials share a workgroup when the
    transform is small), including a single polynomial."""
    n = 1 << log_n
    w = root_of_unity(log_n)
    for batch in batches:
        x = synth.field_elements(9000 + 31 * log_n + batch, batch * n).reshape(batch, n, 8)
        for inverse in (False, True):
            got = sv.fft.ntt_batch(x, w, inverse=inverse)
            for b in sorted({0, batch // 2, batch - 1}):
                exp = oracle.fft_inv(x[b], w) if inverse else oracle.fft_fwd(x[b], w)
                assert np.array_equal(got[b], exp), (log_n, batch, inverse, b)
            # every polynomial against the round trip
            if not inverse:
                assert np.array_equal(sv.fft.ntt_batch(got, w, inverse=True), x), (log_n, batch)


def test_ntt_empty_batch_is_a_no_op(sv):
    ctx = sv.default_context()
    w = root_of_unity(4)
    r = sv.fft._limbs(w)
    assert ctx.lib.stark_ntt_batch(ctx.handle, None, 4, 0, sv.fft._ptr(r), 0) == 0
    assert ctx.lib.stark_ntt_batch_dev(ctx.handle, None, None, 4, 0, sv.fft._ptr(r), 0, None) == 0


def test_pseudorandom_indices_match_oracle_and_proof(sv, proof_bytes):
    """utils::get_pseudorandom_indices (src/utils.rs:53-94) on the device: the call sites' shapes
    (40 of deg/4 and 80 of 65536, excluding multiples of 8) on the proof's own roots, plus other shapes."""
    proof, _ = proofio.parse_proof(proof_bytes)
    rng = np.random.default_rng(8)
    seeds = [proof.l_merkle_root, proof.merkle_root] + [l.root2 for l in proof.fri_layers] + \
            [bytes(rng.integers(0, 256, 32, dtype=np.uint8)) for _ in range(200)]
    for count, modulus, excl in ((40, 16384, 8), (80, 65536, 8), (40, 64, 8), (1, 7, None), (8, 2**32 - 1, None), (9, 1000, 2),
                                 (17, 12345, 3), (64, 65536, None), (100, 5, 5)):
        got = sv.merkle_tree.get_pseudorandom_indices(seeds, count, modulus, excl)
        for s, row in zip(seeds, got):
            assert list(row) == [int(x) for x in oracle.get_pseudorandom_indices(s, count, modulus, excl)], (count, modulus, excl)
    one = sv.merkle_tree.get_pseudorandom_indices(proof.l_merkle_root, 80, 65536, 8)
    assert one.shape == (80,) and not (one % 8 == 0).any()
    for bad in ((40, 16384, 1), (40, 1, 8), (40, 0, None)):
        with pytest.raises(sv.InvalidArgument):
            sv.merkle_tree.get_pseudorandom_indices(seeds[:2], *bad)


@pytest.mark.parametrize("log_leaves", [0, 1, 3, 8, 9, 10, 13, 17])
def test_merkle_build_matches_oracle(sv, log_leaves):
    rng = np.random.default_rng(40 + log_leaves)
    leaves = rng.integers(0, 256, size=(1 << log_leaves, 32), dtype=np.uint8)
    got = sv.merkle_tree.merkle_build(leaves)
    exp = oracle.merkle_build(leaves)
    assert np.array_equal(got[1:], exp[1:])
    if log_leaves >= 3:
        # branches cut from the device-built tree verify against its root (permute_4_index leaf order)
        L = 1 << log_leaves
        w = log_leaves - 1
        idx = rng.integers(0, L, size=200, dtype=np.uint32)
        ld4 = L // 4
        pos = (idx // ld4 + 4 * (idx % ld4)).astype(np.int64)
        wit = np.stack([got[((L + pos) >> (k + 1)) ^ 1] for k in range(w)])
        assert sv.merkle_tree.verify_batch(got[L + pos], got[L + (pos ^ 1)], wit, idx, got[1]).all()


@pytest.mark.gpu
def test_host_copies_staged_and_direct_give_the_same_results(monkeypatch):
    """Host-pointer entry points stage copies of 8 MiB and more through pinned buffers; STARKHIP_PINNED=0 (read when a
    context first copies) makes them direct. Both contexts must return the same bytes, and the staged path must handle
    sizes that are not a multiple of its 64 MiB buffers."""
    import stark_verifier_amd as sv
    log_n, batch = 15, 70                       # 70 MiB: two staging buffers and a partial third chunk
    n = 1 << log_n
    w = pow(7, (oracle.P - 1) >> log_n, oracle.P)
    x = synth.field_elements(99, batch * n).reshape(batch, n, 8)
    staged = sv.fft.ntt_batch(x, w, inverse=False)
    monkeypatch.setenv("STARKHIP_PINNED", "0")
    ctx = sv.Context(0)
    direct = sv.fft.ntt_batch(x, w, inverse=False, ctx=ctx)
    ctx.close()
    assert np.array_equal(staged, direct)
    for b in (0, 37, batch - 1):
        assert np.array_equal(staged[b], oracle.fft_fwd(x[b], w))
