// REkeyGen -- drop-in for client/src/REkeyGen.cpp:
// `REkeyGen <cc.json> <client_privkey> <peer_pubkey> <rekey_out>` (REkeyGen.cpp:15-25).
// cc->ReKeyGen(privKey, pubKey) (REkeyGen.cpp:52) -> mkckks_rekeygen.
#include "hostlib.hpp"
using namespace mkh;

int main(int argc, char *argv[]) {
    if (argc != 5) {
        std::cerr << "Usage: " << argv[0] << " <cc.json> <client_privkey.json> <peer_pubkey.json> <rekey_out.json>" << std::endl;
        return 1;
    }
    const std::string cc_path = argv[1], sk_path = argv[2], pk_path = argv[3], rk_path = argv[4];
    CcFile cc;
    try {
        cc = read_cc(cc_path);
    } catch (const std::exception &) {
        std::cerr << "Error loading CryptoContext from " << cc_path << std::endl;
        return 1;
    }
    try {
        Session s(cc);
        std::cout << "[ReKeyGen] CryptoContext loaded from " << cc_path << std::endl;
        const uint32_t N = s.N(), D = s.D(), beta = s.beta();
        std::vector<uint64_t> sk, pk;
        std::vector<int8_t> sk_t;
        if (!load_secret_key(s, sk_path, sk, sk_t)) {
            std::cerr << "Error loading Client private key from " << sk_path << std::endl;
            return 1;
        }
        std::cout << "[ReKeyGen] Client Private Key loaded from " << sk_path << std::endl;
        if (!load_public_key(s, pk_path, pk)) {
            std::cerr << "Error loading Peer public key from " << pk_path << std::endl;
            return 1;
        }
        std::cout << "[ReKeyGen] Peer Public Key loaded from " << pk_path << std::endl;
        const SamplerKey k_u = fresh_key(), k_e0 = fresh_key(), k_e1 = fresh_key();
        int8_t *d_u = s.alloc<int8_t>((size_t)beta * N);
        int32_t *d_e0 = s.alloc<int32_t>((size_t)beta * N), *d_e1 = s.alloc<int32_t>((size_t)beta * N);
        Session::check(mkckks_sample_ternary(s.ctx(), d_u, (size_t)beta * N, k_u.bytes, 0));
        Session::check(mkckks_sample_gauss(s.ctx(), d_e0, (size_t)beta * N, 3.19, k_e0.bytes, 1));
        Session::check(mkckks_sample_gauss(s.ctx(), d_e1, (size_t)beta * N, 3.19, k_e1.bytes, 2));
        uint64_t *d_evk = s.alloc<uint64_t>((size_t)beta * 2 * D * N);
        Session::check(mkckks_rekeygen(s.ctx(), s.to_device(sk_t.data(), N), s.to_device(pk.data(), pk.size()), d_u, d_e0,
                                       d_e1, d_evk));
        std::vector<uint64_t> evk((size_t)beta * 2 * D * N);
        s.to_host(evk.data(), d_evk, evk.size());
        std::cout << "[ReKeyGen] Re-encryption key generated successfully" << std::endl;
        try {
            write_key_file(rk_path, KIND_RK, N, D, 2 * beta, evk);
        } catch (const std::exception &) {
            std::cerr << "[ReKeyGen] Failed to save re-encryption key to " << rk_path << std::endl;
            return 1;
        }
    } catch (const std::exception &e) {
        std::cerr << "[ReKeyGen] Re-encryption key generation failed: " << e.what() << std::endl;
        return 1;
    }
    std::cout << "[ReKeyGen] Re-encryption key saved to " << rk_path << std::endl;
    return 0;
}
