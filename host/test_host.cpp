// Exercises host/stark_host.hpp on a GPU: the reference's unit-test vector
// (reference src/merkle_tree.rs:187-204) and fft_inv as called at reference src/lib.rs:125.
// Build: g++ -std=c++17 -O2 host/test_host.cpp -Lstark_verifier_amd -lstarkhip -Wl,-rpath,$PWD/stark_verifier_amd -o host/test_host
#include <cstdio>
#include <cstdlib>

#include "stark_host.hpp"

using namespace starkhost;

static std::vector<uint8_t> unhex(const char* s) {
    std::vector<uint8_t> o;
    for (size_t i = 0; s[i] && s[i + 1]; i += 2) {
        char b[3] = {s[i], s[i + 1], 0};
        o.push_back((uint8_t)strtoul(b, nullptr, 16));
    }
    return o;
}
static MerkleDigest dig(const char* s) {
    auto v = unhex(s);
    MerkleDigest d;
    std::memcpy(d.data(), v.data(), 32);
    return d;
}
static U256 from_be_hex(const char* s) {  // 64 hex digits, big-endian
    auto v = unhex(s);
    U256 r{};
    for (int i = 0; i < 32; i++) r[(31 - i) / 4] |= (uint32_t)v[i] << (8 * ((31 - i) % 4));
    return r;
}

static std::vector<uint8_t> read_file(const char* path) {
    std::vector<uint8_t> b;
    if (FILE* f = fopen(path, "rb")) {
        uint8_t buf[65536];
        for (size_t n; (n = fread(buf, 1, sizeof buf, f)) > 0;) b.insert(b.end(), buf, buf + n);
        fclose(f);
    }
    return b;
}

int main(int argc, char** argv) {
    Context ctx(0);
    const char* d[14] = {"ba5ba7ea7c0fe5ea738a8d207f0797dbb0658bdceeb6be707e5e912f0f7b2728",
                         "e845c8789914f8e0b5d11bd7f90471820cfd568e735343bf4661da8b1dda0648",
                         "ab163f24c3ba15c5ff71cf5efb50cad7498b82dfa037e436dc05331ef7ebe039",
                         "97b87355c6fb49f0993779360f88792d527c572da72bde39305b53de6facda84",
                         "37d65601a24040c2faf8c657828174ef9131e0a8c3a30f176fcdeb382a0c2a0d",
                         "1325568476a58340b6dd1c0cc2127feff42c8d271aa559802b1b3d93d4899926",
                         "3c77b2671efbb23acc796240fc3f4f047bacfc472a3de3d0978540179e4f9e14",
                         "92e4884259792a7c891bba4532d25c1dcc74c648553d106344bb10c3840d4c6d",
                         "df1a1e30b6fec4aec943937956102149691bc4b06992af48a43694cfa6849165",
                         "3cd4cab3eb75edca1054bd5d9673131b834c893ac4997fce779ef295483276a0",
                         "04126daaacf18b81a8008309feebb894e58a9bd3051ba8951a90f51e7d46a99f",
                         "9e0cf8e9dac36bd52ce3cad4eb4fc0bf930e7543e5c790d61f6bd5c9f85ca0ee",
                         "4c9649fbaad59383acc8dc17d6a2b47408e081b7ff955e983580536a2dbc3fd7",
                         "89b6f049e518dc76c69f797ce4c7bc9c1b45328ea957b2634f21e7a10af9b0cb"};
    merkle_tree::ProofBranch br;
    br.value = unhex("c10b1b74355bf6f5d2085b55bb459f5c14e9e4816c04ec72cca8e8bb5288cb50");
    br.sibling_value = unhex(d[0]);
    for (int i = 1; i < 14; i++) br.witnesses.push_back(dig(d[i]));
    MerkleDigest root = dig("f13a4bfaa28c22df47a4e0e89a54736b0a9bedc9727bbc3cc8a0e4237eb59ad9");
    auto ok = br.verify(ctx, root, 5845);
    if (!ok || *ok != br.value) { printf("FAIL: KAT branch rejected\n"); return 1; }
    if (br.verify(ctx, root, 5846, false)) { printf("FAIL: wrong index accepted\n"); return 1; }
    bool threw = false;
    try { br.verify(ctx, root, 5846); } catch (const MerkleMismatch&) { threw = true; }
    if (!threw) { printf("FAIL: strict mismatch did not throw\n"); return 1; }

    // fft_inv(round_constants, val, modulus), constants i^7 xor 42 (reference src/main.rs:209-212)
    std::vector<U256> consts(64);
    for (uint64_t i = 0; i < 64; i++) {
        unsigned __int128 c = 1;
        for (int k = 0; k < 7; k++) c *= i;
        c ^= 42;
        consts[i] = U256{(uint32_t)c, (uint32_t)(c >> 32), (uint32_t)(c >> 64), (uint32_t)(c >> 96), 0, 0, 0, 0};
    }
    // val = 56670364103764250102176604807203318908867195832872336813161821519223575486477 (src/lib.rs:123)
    U256 val = from_be_hex("7d4a4a16f42aef916f8f1c5979e68b634af29f37230a955a1b1556397163200d");
    std::vector<U256> poly = fft::fft_inv(ctx, consts, val, modulus());
    U256 e0 = from_be_hex("0000000000000000000000000000000000000000000000000000007825542ac5");
    U256 e63 = from_be_hex("fd0228d8bad580fd57d1b765459647de9d36dd4494d4c94f69da1bed69fc475e");
    if (poly[0] != e0 || poly[63] != e63) { printf("FAIL: fft_inv known answers\n"); return 1; }
    if (fft::fft(ctx, poly, val, modulus()) != consts) { printf("FAIL: fft(fft_inv(x)) != x\n"); return 1; }
    threw = false;
    try { fft::fft_inv(ctx, std::vector<U256>(consts.begin(), consts.begin() + 32), val, modulus()); } catch (const std::runtime_error&) { threw = true; }
    if (!threw) { printf("FAIL: wrong-order root accepted\n"); return 1; }
    if (argc > 1) {
        // verify_mimc_proof(3, 8192, constants, mimc(3, 8192, constants), proof, p) as reference src/main.rs:204-222
        // runs it, over a batch: the bundled proof, a copy with one witness bit flipped, the proof again
        std::vector<uint8_t> good = read_file(argv[1]);
        if (good.empty()) { printf("FAIL: cannot read %s\n", argv[1]); return 1; }
        std::vector<uint8_t> bad = good;
        bad[good.size() / 3] ^= 0x20;
        U256 inp{3, 0, 0, 0, 0, 0, 0, 0};
        U256 out = from_be_hex("d2874ff35f56a2f7f5e0cfc8c4e6ae570f3636219c6a3bc8810deec3361ad858");
        if (mimc(ctx, {inp}, 8192, consts)[0] != out) { printf("FAIL: mimc(3, 8192) known answer\n"); return 1; }
        std::vector<uint8_t> status;
        std::vector<bool> acc = verify_mimc_proofs(ctx, inp, 8192, consts, out, {good, bad, good}, &status);
        if (!(acc[0] && !acc[1] && acc[2])) { printf("FAIL: batch accept bits %d %d %d\n", (int)acc[0], (int)acc[1], (int)acc[2]); return 1; }
        if (status[0] != 0 || status[1] == 0) { printf("FAIL: batch status\n"); return 1; }
        // the prover reproduces the bundled proof from the same instance, and its proofs for other inputs verify
        std::vector<U256> outs;
        std::vector<U256> ins = {inp, U256{4, 0, 0, 0, 0, 0, 0, 0}, U256{~0u, ~0u, 0, 0, 0, 0, 0, 1}};
        std::vector<std::vector<uint8_t>> made = mk_mimc_proofs(ctx, ins, 8192, consts, &outs);
        if (made[0] != good || outs[0] != out) { printf("FAIL: prover does not reproduce the bundled proof\n"); return 1; }
        acc = verify_mimc_proofs(ctx, ins, 8192, consts, outs, made, &status);
        if (!(acc[0] && acc[1] && acc[2])) { printf("FAIL: generated proofs rejected\n"); return 1; }
        std::swap(outs[1], outs[2]);
        acc = verify_mimc_proofs(ctx, ins, 8192, consts, outs, made, &status);
        if (!(acc[0] && !acc[1] && !acc[2])) { printf("FAIL: proofs accepted for the wrong output\n"); return 1; }
    }
    printf("host mirror ok\n");
    return 0;
}
