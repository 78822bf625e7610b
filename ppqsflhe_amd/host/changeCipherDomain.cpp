// changeCipherDomain -- drop-in for server/src/changeCipherDomain.cpp:
// `changeCipherDomain <cc_path> <rekey_path> <input_encfile> <output_encfile>` (:19-29; caller server_fns.sh:65,79).
// The reference loops cc->ReEncrypt(ct, reKey) over mean / std_dev / values[] of every layer (:61-117); here all
// ciphertexts of the file go to HBM once and are re-encrypted by ONE mkckks_reencrypt_batch call.
#include "hostlib.hpp"
using namespace mkh;

int main(int argc, char *argv[]) {
    if (argc != 5) {
        std::cerr << "Usage: " << argv[0] << " <cc_path> <rekey_path> <input_encfile> <output_encfile>" << std::endl;
        return 1;
    }
    const std::string cc_path = argv[1], rekey_path = argv[2], input_encfile = argv[3], output_encfile = argv[4];
    CcFile cc;
    try {
        cc = read_cc(cc_path);
    } catch (const std::exception &) {
        std::cerr << "[recrypt] ERROR: Failed to load CryptoContext: " << cc_path << std::endl;
        return 1;
    }
    try {
        Session s(cc);
        std::cout << "[recrypt] CryptoContext loaded\n";
        const uint32_t N = s.N();
        std::vector<uint64_t> evk;
        if (!load_eval_key(s, rekey_path, evk)) {
            std::cerr << "[recrypt] ERROR: Failed to load ReKey from " << rekey_path << std::endl;
            return 1;
        }
        std::cout << "[recrypt] ReKey loaded\n";
        Json inputJson;
        bool binary = false;  // the output keeps the input's envelope form
        try {
            inputJson = read_envelope(input_encfile, &binary);
            raw_blobs() = binary;
        } catch (const std::exception &) {
            std::cerr << "[recrypt] ERROR: Could not open input encrypted weights file\n";
            return 1;
        }
        const std::vector<CtRef> refs = enumerate_cts(inputJson);
        std::vector<Ciphertext> cts;
        cts.reserve(refs.size());
        for (const CtRef &r : refs) cts.push_back(decode_ct_checked(ct_string(inputJson, r), s));
        Json outputJson = inputJson;  // layer / shape carried over; blobs replaced below
        if (!cts.empty()) {
            const uint32_t nl = cts[0].nl;
            for (const Ciphertext &c : cts)
                if (c.nl != nl) throw std::runtime_error("ciphertexts of one file must share a level");
            const size_t words = (size_t)2 * nl * N;
            std::vector<uint64_t> flat(cts.size() * words);
            for (size_t i = 0; i < cts.size(); ++i) std::memcpy(&flat[i * words], cts[i].data.data(), words * 8);
            uint64_t *d_ct = s.to_device(flat.data(), flat.size());
            uint64_t *d_evk = s.to_device(evk.data(), evk.size());
            Session::check(mkckks_reencrypt_batch(s.ctx(), d_ct, d_evk, d_ct, (uint32_t)cts.size(), nl));
            s.to_host(flat.data(), d_ct, flat.size());
            for (size_t i = 0; i < cts.size(); ++i) {
                cts[i].data.assign(flat.begin() + i * words, flat.begin() + (i + 1) * words);
                Json &lay = outputJson["weights_summary"].a[refs[i].layer];
                std::string b64 = encode_ct(cts[i], N);
                if (refs[i].field == 0) lay["mean"] = std::move(b64);
                else if (refs[i].field == 1) lay["std_dev"] = std::move(b64);
                else lay["values"].a[refs[i].idx] = Json(std::move(b64));
            }
        }
        try {
            write_envelope(outputJson, output_encfile, binary);
        } catch (const std::exception &) {
            std::cerr << "[recrypt] ERROR: Failed to open output file\n";
            return 1;
        }
    } catch (const std::exception &e) {
        std::cerr << "[recrypt] ERROR: " << e.what() << std::endl;
        return 1;
    }
    std::cout << "[recrypt] Re-encryption completed successfully. Output: " << output_encfile << std::endl;
    return 0;
}
