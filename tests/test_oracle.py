"""Pins the CPU oracle (oracle/stark_oracle.c) before anything is checked against it.

Anchors: Python big integers (field + direct DFT), hashlib.blake2s and the RFC 7693 vector,
the hex vector in the reference's src/merkle_tree.rs:187-204, the literals asserted in the
reference's src/lib.rs:51,121,123, and every branch plus the accept decision of the
reference's bundled proof.bin (tests/golden/proof.bin).
"""
import hashlib
import random

import numpy as np
import pytest

import oracle
import proofio
import ref_verifier as rv

P = oracle.P
EDGE = [0, 1, 2, P - 1, P - 2, 2**255, 2**256 - 1, P, P + 1, 351 * 2**32 - 1, 2**32 - 1, 2**32,
        (1 << 224) - 1, 0x15EFFFFFFFF, 0x1E140FFFFFD4200000001]


def test_modulus_literal():
    # src/lib.rs:29
    assert P == 115792089237316195423570985008687907853269984665640564039457584006405596119041


def test_field_ops_vs_python_ints():
    rng = random.Random(1)
    xs = EDGE + [rng.getrandbits(256) for _ in range(400)]
    a = [x for x in xs for _ in xs[:25]]
    b = [y for _ in xs for y in (EDGE + xs[-10:])[:25]]
    A, B = oracle.ints_to_limbs(a), oracle.ints_to_limbs(b)
    for op, f in (("add", lambda x, y: (x + y) % P), ("sub", lambda x, y: (x - y) % P),
                  ("mul", lambda x, y: (x * y) % P)):
        got = oracle.limbs_to_ints(oracle.fe_batch(op, A, B))
        assert got == [f(x, y) for x, y in zip(a, b)], op


def test_field_pow():
    rng = random.Random(2)
    for _ in range(20):
        a, e = rng.getrandbits(256), rng.getrandbits(256)
        assert oracle.fe_pow(a, e) == pow(a % P, e, P)
    assert oracle.fe_pow(7, (P - 1) // 65536) == 41913712888260089065520476180880993127517355946012995597287997778376518235852


def test_blake2s_vs_hashlib_and_rfc():
    assert oracle.blake2s(b"abc").hex() == "508c5e8c327c14e2e1a72ba34eeb452f37458b209ed63a294d999b4c86675982"
    assert oracle.blake2s(bytes(64)).hex() == "ae09db7cd54f42b490ef09b6bc541af688e4959bb8c53f359a6f56e38ab454a3"
    rng = random.Random(3)
    for n in [0, 1, 31, 32, 33, 63, 64, 65, 127, 128, 129, 192, 255, 256, 1000]:
        m = bytes(rng.getrandbits(8) for _ in range(n))
        assert oracle.blake2s(m) == hashlib.blake2s(m).digest(), n


KAT_VALUE = "c10b1b74355bf6f5d2085b55bb459f5c14e9e4816c04ec72cca8e8bb5288cb50"
KAT_DIGESTS = """ba5ba7ea7c0fe5ea738a8d207f0797dbb0658bdceeb6be707e5e912f0f7b2728
e845c8789914f8e0b5d11bd7f90471820cfd568e735343bf4661da8b1dda0648
ab163f24c3ba15c5ff71cf5efb50cad7498b82dfa037e436dc05331ef7ebe039
97b87355c6fb49f0993779360f88792d527c572da72bde39305b53de6facda84
37d65601a24040c2faf8c657828174ef9131e0a8c3a30f176fcdeb382a0c2a0d
1325568476a58340b6dd1c0cc2127feff42c8d271aa559802b1b3d93d4899926
3c77b2671efbb23acc796240fc3f4f047bacfc472a3de3d0978540179e4f9e14
92e4884259792a7c891bba4532d25c1dcc74c648553d106344bb10c3840d4c6d
df1a1e30b6fec4aec943937956102149691bc4b06992af48a43694cfa6849165
3cd4cab3eb75edca1054bd5d9673131b834c893ac4997fce779ef295483276a0
04126daaacf18b81a8008309feebb894e58a9bd3051ba8951a90f51e7d46a99f
9e0cf8e9dac36bd52ce3cad4eb4fc0bf930e7543e5c790d61f6bd5c9f85ca0ee
4c9649fbaad59383acc8dc17d6a2b47408e081b7ff955e983580536a2dbc3fd7
89b6f049e518dc76c69f797ce4c7bc9c1b45328ea957b2634f21e7a10af9b0cb""".split()
KAT_ROOT = "f13a4bfaa28c22df47a4e0e89a54736b0a9bedc9727bbc3cc8a0e4237eb59ad9"


def test_merkle_kat_from_reference_unit_test():
    # The data of src/merkle_tree.rs:187-204. In today's struct the first digest is the
    # sibling value and the other 13 are witnesses; 6997 is the permuted tree position of
    # query index 5845 (permute_4_index(5845, 2^14) == 6997).
    value = bytes.fromhex(KAT_VALUE)
    digs = [bytes.fromhex(d) for d in KAT_DIGESTS]
    assert oracle.permute_4_index(5845, 1 << 14) == 6997
    ok, got = oracle.branch_verify(value, digs[0], digs[1:], bytes.fromhex(KAT_ROOT), 5845)
    assert ok and got.hex() == KAT_ROOT
    ok, _ = oracle.branch_verify(value, digs[0], digs[1:], bytes.fromhex(KAT_ROOT), 5846)
    assert not ok
    # pure-hashlib recomputation of the same walk
    pos, acc = 6997, value
    for d in digs:
        acc = hashlib.blake2s(d + acc if pos & 1 else acc + d).digest()
        pos >>= 1
    assert acc.hex() == KAT_ROOT


def test_branch_verify_rejects_reference_panic_shapes():
    with pytest.raises(ValueError):
        oracle.branch_verify(bytes(32), bytes(32), [], bytes(32), 0)  # no witnesses: div by zero
    with pytest.raises(ValueError):
        oracle.branch_verify(bytes(16), bytes(16), [bytes(32)], bytes(32), 0)  # value < 32 bytes


def test_pseudorandom_indices_vs_python():
    rng = random.Random(5)
    for count, modulus, excl in [(40, 16384, 8), (80, 65536, 8), (7, 1000, None), (8, 64, 8), (9, 64, 8), (1, 5, None)]:
        seed = bytes(rng.getrandbits(8) for _ in range(32))
        hashes = seed
        while len(hashes) < 4 * count:
            hashes += hashlib.blake2s(hashes[-32:]).digest()
        real = modulus * (excl - 1) // excl if excl else modulus
        exp = [int.from_bytes(hashes[4 * j:4 * j + 4], "big") % real for j in range(count)]
        if excl:
            exp = [1 + x + x // (excl - 1) for x in exp]
        got = oracle.get_pseudorandom_indices(seed, count, modulus, excl)
        assert list(got) == exp


def _oracle_hooks(counter=None):
    def fft_inv(vals, root):
        return oracle.limbs_to_ints(oracle.fft_inv(oracle.ints_to_limbs(vals), root))

    def multiproof_verify(branches, indices, root):
        out = []
        for br, a in zip(branches, indices):
            ok, _ = oracle.branch_verify(br.value, br.sibling_value, br.witnesses, root, int(a))
            if counter is not None:
                counter[0] += 1
            if not ok:
                return None
            out.append(br.value)
        return out

    def get_indices(seed, count, modulus, exclude):
        return [int(x) for x in oracle.get_pseudorandom_indices(seed, count, modulus, exclude)]

    return {"fft_inv": fft_inv, "multiproof_verify": multiproof_verify, "get_indices": get_indices}


def test_proof_bin_shapes(proof_bytes):
    proof, consumed = proofio.parse_proof(proof_bytes)
    assert consumed == len(proof_bytes) == 566508
    assert proof.merkle_root.hex() == "53a0e380573d3bada3a837e48d93aafa7ef8598cad5164919bbf890f445f04ec"
    assert proof.l_merkle_root.hex() == "f0aeca09558102275bbf9cc82f49d71b37ae96fef1aa3141ae805deb0e80fd14"
    assert [len(l.column_branches.branches) for l in proof.fri_layers] == [40] * 5
    assert [len(l.poly_branches.branches) for l in proof.fri_layers] == [160] * 5
    assert [len(l.column_branches.branches[0].witnesses) for l in proof.fri_layers] == [13, 11, 9, 7, 5]
    assert [len(l.poly_branches.branches[0].witnesses) for l in proof.fri_layers] == [15, 13, 11, 9, 7]
    assert len(proof.merkle_branches.branches) == 160 and len(proof.merkle_branches.branches[0].value) == 96
    assert len(proof.linear_comb_branches.branches) == 80
    # the unit-test vector of src/merkle_tree.rs:187-204 is the first column branch of layer 0
    b0 = proof.fri_layers[0].column_branches.branches[0]
    assert b0.value.hex() == KAT_VALUE and b0.sibling_value.hex() == KAT_DIGESTS[0]
    assert [w.hex() for w in b0.witnesses] == KAT_DIGESTS[1:]
    assert proof.fri_layers[0].root2.hex() == KAT_ROOT


def test_proof_bin_full_verify_with_oracle(proof_bytes):
    """The reference's integration check (README: the binary prints 'proof verified')."""
    proof, _ = proofio.parse_proof(proof_bytes)
    consts = rv.mimc_constants()
    output = rv.mimc(3, 8192, consts)
    assert output == 95224774355499767951968048714566316597785297695903697235130434363122555476056
    counter = [0]
    assert rv.verify_mimc_proof(3, 8192, consts, output, proof, _oracle_hooks(counter))
    assert counter[0] == 1240
    # a wrong public output must be rejected by the constraint checks
    assert not rv.verify_mimc_proof(3, 8192, consts, output + 1, proof, _oracle_hooks())


def test_proof_bin_corruption_rejected(proof_bytes):
    proof, _ = proofio.parse_proof(proof_bytes)
    consts = rv.mimc_constants()
    output = rv.mimc(3, 8192, consts)
    br = proof.fri_layers[2].poly_branches.branches[17]
    w = bytearray(br.witnesses[3])
    w[5] ^= 0x10
    br.witnesses[3] = bytes(w)
    assert not rv.verify_mimc_proof(3, 8192, consts, output, proof, _oracle_hooks())


def test_fft_inv_n64_known_answers():
    # fft_inv(constants, val) as called at src/lib.rs:125; expected values re-derived here by a
    # direct big-integer inverse DFT, and cross-checked against the recorded digests.
    consts = rv.mimc_constants()
    val = 56670364103764250102176604807203318908867195832872336813161821519223575486477
    out = oracle.limbs_to_ints(oracle.fft_inv(oracle.ints_to_limbs(consts), val))
    inv_n, inv_w = pow(64, -1, P), pow(val, -1, P)
    exp = [inv_n * sum(c * pow(inv_w, i * j, P) for j, c in enumerate(consts)) % P for i in range(64)]
    assert out == exp
    assert out[0] == 0x7825542ac5
    assert out[1] == 0x6ba0eba3da5cf73834353119a620cc746ed01e5c49f04aca88c83a388a999588
    assert out[63] == 0xfd0228d8bad580fd57d1b765459647de9d36dd4494d4c94f69da1bed69fc475e
    h = hashlib.blake2s(b"".join(x.to_bytes(32, "big") for x in out)).hexdigest()
    assert h == "978477fb3b799ecb5ed286b82d8e088ea4ce48fe2b46dc5de7bf8c4d24c13c06"


@pytest.mark.parametrize("logn", [0, 1, 2, 3, 4, 5, 8])
def test_fft_vs_direct_dft(logn):
    n = 1 << logn
    rng = random.Random(100 + logn)
    w = pow(7, (P - 1) // n, P)
    v = [rng.getrandbits(256) for _ in range(n)]  # includes non-canonical inputs
    V = oracle.ints_to_limbs(v)
    fwd = oracle.limbs_to_ints(oracle.fft_fwd(V, w))
    assert fwd == [sum(x * pow(w, i * j, P) for j, x in enumerate(v)) % P for i in range(n)]
    inv = oracle.limbs_to_ints(oracle.fft_inv(oracle.ints_to_limbs(fwd), w))
    assert inv == [x % P for x in v]
    for k in (0, n - 1, n // 3):
        assert oracle.dft_at(V, w, k) == fwd[k]


def test_fft_rejects_wrong_order_and_length():
    w8 = pow(7, (P - 1) // 8, P)
    v = oracle.ints_to_limbs(list(range(16)))
    with pytest.raises(ValueError):
        oracle.fft_inv(v, w8)  # order 8, 16 values
    with pytest.raises(ValueError):
        oracle.fft_inv(v[:4], w8)  # order 8, 4 values
    with pytest.raises(ValueError):
        oracle.fft_inv(v[:6], w8)  # not a power of two
    with pytest.raises(ValueError):
        oracle.fft_fwd(v, w8)


def test_roots_of_unity_literals():
    assert pow(7, (P - 1) >> 16, P) == 0x5caa52206ecc9d72853f7c616e06672484967c1e2e5a87d6922259c0982b76cc
    assert oracle.fe_pow(7, (P - 1) >> 20) == 0xd3cd75fbef8d5e5ff367cbc0866ba9b493e153ed47f2c01c2b996aa041d2363b
    assert oracle.fe_pow(7, (P - 1) >> 24) == 0x6b7f42cee2b0355e3c9637dea849609e83b5488d1eb2644e1729a9dc31f6259e


def test_merkle_build_and_batch_roundtrip():
    rng = np.random.default_rng(7)
    w = 5
    L = 1 << (w + 1)
    leaves = rng.integers(0, 256, size=(L, 32), dtype=np.uint8)
    nodes = oracle.merkle_build(leaves)
    root = nodes[1].tobytes()
    n = 50
    idx = rng.integers(0, L, size=n, dtype=np.uint32)
    pos = np.array([oracle.permute_4_index(int(a), L) for a in idx])
    values, sibs = leaves[pos], leaves[pos ^ 1]
    wit = np.stack([nodes[((L + pos) >> (k + 1)) ^ 1] for k in range(w)])
    acc = oracle.merkle_verify_batch(values, sibs, wit, idx, np.frombuffer(root, dtype=np.uint8))
    assert acc.all()
    wit[2, 7, 0] ^= 1
    values[9, 31] ^= 0x80
    acc = oracle.merkle_verify_batch(values, sibs, wit, idx, np.frombuffer(root, dtype=np.uint8))
    assert [i for i in range(n) if not acc[i]] == [7, 9]
