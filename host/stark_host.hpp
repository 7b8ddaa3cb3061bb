// C++ host-side mirror of the reference crate's hot-path interface over libstarkhip's C ABI.
//
// Same names, argument order and meaning as the reference:
//   fft::fft_inv(v, root_of_unity, modulus) -> Vec<BigInt>        reference src/fft.rs:64
//   ProofBranch { witnesses, sibling_value, value }::verify        reference src/merkle_tree.rs:11-15,119
//   MultiProof { branches, root }::verify(indices, rt)             reference src/merkle_tree.rs:18-21,25
//   verify_mimc_proof(inp, num_steps, round_constants, output, proof, modulus), batched   reference src/lib.rs:99
// A BigInt is a U256 here (32 bytes, little-endian limbs, any value; results are canonical).
// Error behaviour: where the reference panics, these throw std::runtime_error (a Merkle
// mismatch throws MerkleMismatch unless `strict` is false, then the result is std::nullopt —
// the reference's `None`).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <map>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../include/starkhip.h"

namespace starkhost {

using U256 = std::array<uint32_t, 8>;  // little-endian limbs
using Value = std::vector<uint8_t>;
using MerkleDigest = std::array<uint8_t, 32>;

inline U256 modulus() { return U256{0x00000001u, 0xFFFFFEA1u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}; }

struct MerkleMismatch : std::runtime_error {
    using std::runtime_error::runtime_error;
};

class Context {
   public:
    explicit Context(int device = 0) {
        if (stark_ctx_create(device, &ctx_) != STARK_OK) throw std::runtime_error("stark_ctx_create failed (no HIP device?)");
    }
    ~Context() { stark_ctx_destroy(ctx_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    stark_ctx* get() const { return ctx_; }
    void check(int rc) const {
        if (rc != STARK_OK) throw std::runtime_error(std::string("libstarkhip: ") + stark_last_error(ctx_));
    }

   private:
    stark_ctx* ctx_ = nullptr;
};

namespace fft {
// reference src/fft.rs:64
inline std::vector<U256> fft_inv(Context& c, const std::vector<U256>& v, const U256& root_of_unity, const U256& mod) {
    std::vector<U256> out(v.size());
    c.check(stark_fft_inv(c.get(), v.empty() ? nullptr : v[0].data(), v.size(), root_of_unity.data(), mod.data(),
                          out.empty() ? nullptr : out[0].data()));
    return out;
}
// the forward transform (the reference's private _fft with the un-reversed root table, src/fft.rs:37)
inline std::vector<U256> fft(Context& c, const std::vector<U256>& v, const U256& root_of_unity, const U256& mod) {
    std::vector<U256> out(v.size());
    c.check(stark_fft(c.get(), v.empty() ? nullptr : v[0].data(), v.size(), root_of_unity.data(), mod.data(),
                      out.empty() ? nullptr : out[0].data()));
    return out;
}
}  // namespace fft

namespace merkle_tree {

struct ProofBranch {
    std::vector<MerkleDigest> witnesses;
    Value sibling_value;
    Value value;
    std::optional<Value> verify(Context& c, const MerkleDigest& root, uint32_t a, bool strict = true) const;
};

struct MultiProof {
    std::vector<ProofBranch> branches;
    MerkleDigest root{};

    // Per-branch accept flags (1 = recomputed root matches), no panic.
    std::vector<uint8_t> accept_bits(Context& c, const std::vector<uint32_t>& indices, const MerkleDigest& rt) const {
        if (branches.size() != indices.size()) throw std::runtime_error("branches len != indices len");  // :28
        std::vector<uint8_t> accept(branches.size(), 0);
        // group by shape, flatten each group into the level-major device layout
        std::map<std::pair<size_t, size_t>, std::vector<size_t>> buckets;
        for (size_t i = 0; i < branches.size(); i++) {
            if (branches[i].sibling_value.size() != branches[i].value.size())
                throw std::runtime_error("value and sibling_value differ in length");
            buckets[{branches[i].value.size(), branches[i].witnesses.size()}].push_back(i);
        }
        for (auto& kv : buckets) {
            const size_t vs = kv.first.first, w = kv.first.second, n = kv.second.size();
            std::vector<uint8_t> values(n * vs), sibs(n * vs), wit(w * n * 32), acc(n);
            std::vector<uint32_t> idx(n);
            for (size_t k = 0; k < n; k++) {
                const ProofBranch& b = branches[kv.second[k]];
                std::memcpy(&values[k * vs], b.value.data(), vs);
                std::memcpy(&sibs[k * vs], b.sibling_value.data(), vs);
                for (size_t l = 0; l < w; l++) std::memcpy(&wit[(l * n + k) * 32], b.witnesses[l].data(), 32);
                idx[k] = indices[kv.second[k]];
            }
            c.check(stark_merkle_verify_batch(c.get(), values.data(), sibs.data(), (uint32_t)vs, wit.data(), (uint32_t)w,
                                              idx.data(), rt.data(), 1, nullptr, n, acc.data(), nullptr));
            for (size_t k = 0; k < n; k++) accept[kv.second[k]] = acc[k];
        }
        return accept;
    }

    // reference src/merkle_tree.rs:25
    std::optional<std::vector<Value>> verify(Context& c, const std::vector<uint32_t>& indices,
                                             const std::optional<MerkleDigest>& rt = std::nullopt, bool strict = true) const {
        const MerkleDigest r = rt ? *rt : root;
        std::vector<uint8_t> acc = accept_bits(c, indices, r);
        for (size_t i = 0; i < acc.size(); i++)
            if (!acc[i]) {
                if (strict) throw MerkleMismatch("values didn't match up (branch " + std::to_string(i) + ")");  // :165
                return std::nullopt;
            }
        std::vector<Value> res;
        for (auto& b : branches) res.push_back(b.value);
        return res;
    }
};

inline std::optional<Value> ProofBranch::verify(Context& c, const MerkleDigest& root, uint32_t a, bool strict) const {
    MultiProof mp;
    mp.branches.push_back(*this);
    mp.root = root;
    auto r = mp.verify(c, {a}, std::nullopt, strict);
    if (!r) return std::nullopt;
    return (*r)[0];
}

}  // namespace merkle_tree

// verify_mimc_proof(inp, num_steps, round_constants, output, proof, modulus) (reference src/lib.rs:99) over a
// batch of serialized proofs (each what deserializer::from_bytes reads, src/deserializer.rs:142). Returns one
// flag per proof: true exactly when the reference returns true; false where it returns false or panics.
// `status`, if given, receives the reason per proof (codes in include/starkhip.h).
inline std::vector<bool> verify_mimc_proofs(Context& c, const U256& inp, size_t num_steps, const std::vector<U256>& round_constants,
                                            const U256& output, const std::vector<std::vector<uint8_t>>& proofs,
                                            std::vector<uint8_t>* status = nullptr) {
    std::vector<uint64_t> offsets(proofs.size() + 1, 0);
    for (size_t i = 0; i < proofs.size(); i++) offsets[i + 1] = offsets[i] + proofs[i].size();
    std::vector<uint8_t> blob(offsets.back());
    for (size_t i = 0; i < proofs.size(); i++)
        if (!proofs[i].empty()) std::memcpy(&blob[offsets[i]], proofs[i].data(), proofs[i].size());
    std::vector<uint8_t> accept(proofs.size()), st(proofs.size());
    c.check(stark_mimc_verify_batch(c.get(), blob.data(), offsets.data(), proofs.size(), inp.data(), num_steps,
                                    round_constants.empty() ? nullptr : round_constants[0].data(), round_constants.size(),
                                    output.data(), accept.data(), st.data()));
    if (status) *status = st;
    return std::vector<bool>(accept.begin(), accept.end());
}

// The same with one (input, output) pair per proof.
inline std::vector<bool> verify_mimc_proofs(Context& c, const std::vector<U256>& inputs, size_t num_steps,
                                            const std::vector<U256>& round_constants, const std::vector<U256>& outputs,
                                            const std::vector<std::vector<uint8_t>>& proofs, std::vector<uint8_t>* status = nullptr) {
    if (inputs.size() != proofs.size() || outputs.size() != proofs.size())
        throw std::runtime_error("verify_mimc_proofs: one input and one output per proof");
    std::vector<uint64_t> offsets(proofs.size() + 1, 0);
    for (size_t i = 0; i < proofs.size(); i++) offsets[i + 1] = offsets[i] + proofs[i].size();
    std::vector<uint8_t> blob(offsets.back());
    for (size_t i = 0; i < proofs.size(); i++)
        if (!proofs[i].empty()) std::memcpy(&blob[offsets[i]], proofs[i].data(), proofs[i].size());
    std::vector<uint8_t> accept(proofs.size()), st(proofs.size());
    c.check(stark_mimc_verify_batch_io(c.get(), blob.data(), offsets.data(), proofs.size(),
                                       inputs.empty() ? nullptr : inputs[0].data(), num_steps,
                                       round_constants.empty() ? nullptr : round_constants[0].data(), round_constants.size(),
                                       outputs.empty() ? nullptr : outputs[0].data(), accept.data(), st.data()));
    if (status) *status = st;
    return std::vector<bool>(accept.begin(), accept.end());
}

// utils::mimc(input, steps, round_constants, modulus) (reference src/utils.rs:8-16) for many inputs at once.
inline std::vector<U256> mimc(Context& c, const std::vector<U256>& inputs, size_t steps, const std::vector<U256>& round_constants) {
    std::vector<U256> out(inputs.size());
    c.check(stark_mimc_batch(c.get(), inputs.empty() ? nullptr : inputs[0].data(), inputs.size(), steps,
                             round_constants.empty() ? nullptr : round_constants[0].data(), round_constants.size(),
                             out.empty() ? nullptr : out[0].data()));
    return out;
}

// Proof generation (the reference's proofs come from an external program, README.md:9): one serialized proof per
// input, in the format deserializer::from_bytes reads; outputs[i] = mimc(inputs[i], num_steps, round_constants)
// (src/utils.rs:8-16), the value verify_mimc_proof is called with.
inline std::vector<std::vector<uint8_t>> mk_mimc_proofs(Context& c, const std::vector<U256>& inputs, size_t num_steps,
                                                        const std::vector<U256>& round_constants, std::vector<U256>* outputs) {
    const size_t size = stark_mimc_proof_size(num_steps, round_constants.size());
    if (size == 0) throw std::runtime_error("mk_mimc_proofs: num_steps / round_constants rejected");
    std::vector<uint8_t> blob(inputs.size() * size);
    std::vector<U256> outs(inputs.size());
    if (!inputs.empty())
        c.check(stark_mimc_prove_batch(c.get(), inputs[0].data(), inputs.size(), num_steps, round_constants[0].data(),
                                       round_constants.size(), blob.data(), outs[0].data()));
    std::vector<std::vector<uint8_t>> proofs(inputs.size());
    for (size_t i = 0; i < inputs.size(); i++) proofs[i].assign(blob.begin() + i * size, blob.begin() + (i + 1) * size);
    if (outputs) *outputs = outs;
    return proofs;
}

}  // namespace starkhost
