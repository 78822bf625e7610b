// keyGen -- drop-in for client/src/keyGen.cpp: `keyGen <cc_path> <pubkey_out> <privkey_out>` (keyGen.cpp:14-22).
// cc->KeyGen() (keyGen.cpp:33) -> mkckks_sample_* + mkckks_keygen, all on the GPU.
#include "hostlib.hpp"
using namespace mkh;

int main(int argc, char *argv[]) {
    if (argc != 4) {
        std::cerr << "Usage: " << argv[0] << " <cc_path> <pubkey_out> <privkey_out>" << std::endl;
        return 1;
    }
    const std::string cc_path = argv[1], pubkey_out = argv[2], privkey_out = argv[3];
    CcFile cc;
    try {
        cc = read_cc(cc_path);
    } catch (const std::exception &) {
        std::cerr << "[keyGen] ERROR: cannot load CryptoContext from " << cc_path << std::endl;
        return 1;
    }
    try {
        Session s(cc);
        std::cout << "[keyGen] CryptoContext loaded from " << cc_path << std::endl;
        const uint32_t N = s.N(), D = s.D();
        // independent 256-bit OS-drawn keys for the secret, the error and the (published) uniform polynomial
        const SamplerKey k_s = fresh_key(), k_e = fresh_key(), k_a = fresh_key();
        int8_t *d_s = s.alloc<int8_t>(N);
        int32_t *d_e = s.alloc<int32_t>(N);
        uint64_t *d_a = s.alloc<uint64_t>((size_t)D * N);
        Session::check(mkckks_sample_ternary(s.ctx(), d_s, N, k_s.bytes, 0));          // secret: uniform ternary
        Session::check(mkckks_sample_gauss(s.ctx(), d_e, N, 3.19, k_e.bytes, 1));       // error: sigma = 3.19
        Session::check(mkckks_sample_uniform(s.ctx(), d_a, 1, s.L(), 1, k_a.bytes, 2));  // a: uniform over QP
        uint64_t *d_pk = s.alloc<uint64_t>((size_t)2 * D * N), *d_sk = s.alloc<uint64_t>((size_t)D * N);
        Session::check(mkckks_keygen(s.ctx(), d_s, d_a, d_e, d_pk, d_sk));
        std::vector<int8_t> sk_t(N);
        s.to_host(sk_t.data(), d_s, N);
        std::vector<uint64_t> pk((size_t)2 * D * N), sk((size_t)D * N);
        s.to_host(pk.data(), d_pk, pk.size());
        s.to_host(sk.data(), d_sk, sk.size());
        std::cout << "[keyGen] Public and Private keys generated" << std::endl;
        try {
            write_key_file(privkey_out, KIND_SK, N, D, 1, sk, &sk_t);
        } catch (const std::exception &) {
            std::cerr << "[keyGen] ERROR: Failed to save private key to " << privkey_out << std::endl;
            return 1;
        }
        try {
            write_key_file(pubkey_out, KIND_PK, N, D, 2, pk);
        } catch (const std::exception &) {
            std::cerr << "[keyGen] ERROR: Failed to save public key to " << pubkey_out << std::endl;
            return 1;
        }
    } catch (const std::exception &e) {
        std::cerr << "[keyGen] ERROR: Key generation failed: " << e.what() << std::endl;
        return 1;
    }
    std::cout << "[keyGen] Keys saved: priv=" << privkey_out << " pub=" << pubkey_out << std::endl;
    return 0;
}
