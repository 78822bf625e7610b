// decryptModelWeights -- drop-in for client/src/decryptModelWeights.cpp:
// `decryptModelWeights <cc_path> <privkey_path> <input_encfile> <output_file>` (:28-38; caller client_fns.sh:100).
// Decrypt (:81,90,108) -> mkckks_decrypt_batch (c0 + c1*s, INTT) + mkckks_decode_batch (CRT interpolation, embedding);
// mean/std_dev keep slot 0 (SetLength(1), :82-83,91-92); values are concatenated and trimmed to prod(shape) (:100-116).
#include "hostlib.hpp"
using namespace mkh;

int main(int argc, char *argv[]) {
    if (argc != 5) {
        std::cerr << "Usage: " << argv[0] << " <cc_path> <privkey_path> <input_encfile> <output_file>" << std::endl;
        return 1;
    }
    const std::string cc_path = argv[1], privkey_path = argv[2], input_encfile = argv[3], output_file = argv[4];
    CcFile cc;
    try {
        cc = read_cc(cc_path);
    } catch (const std::exception &) {
        std::cerr << "[decrypt] ERROR: Failed to load CryptoContext from " << cc_path << std::endl;
        return 1;
    }
    try {
        Session s(cc);
        std::cout << "[decrypt] CryptoContext loaded\n";
        const uint32_t N = s.N();
        std::vector<uint64_t> sk;
        std::vector<int8_t> sk_t;
        if (!load_secret_key(s, privkey_path, sk, sk_t)) {
            std::cerr << "[decrypt] ERROR: Failed to load private key from " << privkey_path << std::endl;
            return 1;
        }
        std::cout << "[decrypt] Private key loaded\n";
        Json encJson;
        try {
            encJson = read_envelope(input_encfile);
        } catch (const std::exception &) {
            std::cerr << "[decrypt] ERROR: Could not open input file: " << input_encfile << std::endl;
            return 1;
        }
        std::cout << "[decrypt] Encrypted weights loaded\n";
        const std::vector<CtRef> refs = enumerate_cts(encJson);
        std::vector<std::vector<double>> decoded(refs.size());
        if (!refs.empty()) {
            std::vector<Ciphertext> cts;
            for (const CtRef &r : refs) cts.push_back(decode_ct_checked(ct_string(encJson, r), s));
            const uint32_t nl = cts[0].nl;
            for (const Ciphertext &c : cts)
                if (c.nl != nl) throw std::runtime_error("ciphertexts of one file must share a level");
            const size_t words = (size_t)2 * nl * N, B = cts.size();
            std::vector<uint64_t> flat(B * words);
            for (size_t i = 0; i < B; ++i) std::memcpy(&flat[i * words], cts[i].data.data(), words * 8);
            uint64_t *d_m = s.alloc<uint64_t>(B * (size_t)nl * N);
            Session::check(mkckks_decrypt_batch(s.ctx(), s.to_device(flat.data(), flat.size()),
                                                s.to_device(sk.data(), sk.size()), d_m, (uint32_t)B, nl));
            for (const Ciphertext &c : cts)
                if (c.scale != cts[0].scale) throw std::runtime_error("ciphertexts of one file must share a scaling factor");
            const size_t slots = s.slots();
            double *d_vals = s.alloc<double>(B * slots);
            Session::check(mkckks_decode_batch(s.ctx(), d_m, d_vals, (uint32_t)B, nl, cts[0].scale));
            std::vector<double> vals(B * slots);
            s.to_host(vals.data(), d_vals, vals.size());
            for (size_t i = 0; i < B; ++i) decoded[i].assign(vals.begin() + i * slots, vals.begin() + (i + 1) * slots);
        }
        Json plainJson = Json::object();
        plainJson["weights_summary"] = Json::array();
        size_t c = 0;
        for (const Json &encLayer : encJson.at("weights_summary").a) {
            Json plainLayer = Json::object();
            plainLayer["layer"] = encLayer.at("layer");
            plainLayer["shape"] = encLayer.at("shape");
            plainLayer["mean"] = decoded[c++][0];
            plainLayer["std_dev"] = decoded[c++][0];
            size_t expected = 1;
            for (const Json &dim : encLayer.at("shape").a) expected *= (size_t)dim.as_int();
            Json samples = Json::array();
            std::vector<double> all;
            for (size_t k = 0; k < encLayer.at("values").size(); ++k) {
                // encryptModelWeights packs BatchSize values per ciphertext (zero padded to N/2 slots): only those
                // come back (decryptModelWeights.cpp:109-110: GetRealPackedValue of a batchSize-slot plaintext)
                const std::vector<double> &v = decoded[c++];
                all.insert(all.end(), v.begin(), v.begin() + std::min<size_t>(s.batch(), v.size()));
            }
            if (all.size() > expected) all.resize(expected);  // trim the zero padding
            for (double v : all) samples.push_back(Json(v));
            plainLayer["values"] = samples;
            plainJson["weights_summary"].push_back(plainLayer);
        }
        try {
            plainJson.write_file(output_file);
        } catch (const std::exception &) {
            std::cerr << "[decrypt] ERROR: Failed to open output file: " << output_file << std::endl;
            return 1;
        }
    } catch (const std::exception &e) {
        std::cerr << "[decrypt] ERROR: " << e.what() << std::endl;
        return 1;
    }
    std::cout << "[decrypt] Decryption completed successfully. Output: " << output_file << std::endl;
    return 0;
}
