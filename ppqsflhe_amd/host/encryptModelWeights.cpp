// encryptModelWeights -- drop-in for client/src/encryptModelWeights.cpp:
// `encryptModelWeights <cc_path> <pubkey_path> <input_weights> <output_encfile>` (:19-29).
// Per layer: {mean}, {std_dev} and the values in chunks of batchSize (zero padded, :100-107) are packed with
// MakeCKKSPackedPlaintext and encrypted (:82-83,90-91,109-110); layers named "optimizer/..." are skipped (:71-74).
// Here all plaintexts of the file are encoded (mkckks_encode_batch) and encrypted (mkckks_encrypt_batch) on the GPU in
// one batch each.
#include "hostlib.hpp"
using namespace mkh;

int main(int argc, char *argv[]) {
    if (argc != 5) {
        std::cerr << "Usage: " << argv[0] << " <cc_path> <pubkey_path> <input_weights> <output_encfile>" << std::endl;
        return 1;
    }
    const std::string cc_path = argv[1], pubkey_path = argv[2], input_weights = argv[3], output_encfile = argv[4];
    CcFile cc;
    try {
        cc = read_cc(cc_path);
    } catch (const std::exception &) {
        std::cerr << "[encrypt] ERROR: Failed to deserialize crypto context from " << cc_path << std::endl;
        return 1;
    }
    try {
        Session s(cc);
        std::cout << "[encrypt] CryptoContext loaded from " << cc_path << std::endl;
        const uint32_t N = s.N(), L = s.L();
        const size_t batchSize = s.batch();
        std::cout << "[encrypt] Batch size from CryptoContext = " << batchSize << std::endl;
        std::vector<uint64_t> pk;
        if (!load_public_key(s, pubkey_path, pk)) {
            std::cerr << "[encrypt] ERROR: Failed to deserialize public key from " << pubkey_path << std::endl;
            return 1;
        }
        std::cout << "[encrypt] Public key loaded from " << pubkey_path << std::endl;
        Json inputJson;
        try {
            inputJson = Json::parse_file(input_weights);
        } catch (const std::exception &) {
            std::cerr << "[encrypt] ERROR: Could not open input weights file: " << input_weights << std::endl;
            return 1;
        }
        std::cout << "[encrypt] Weights loaded from " << input_weights << std::endl;
        raw_blobs() = wants_binary_output(output_encfile);  // MKCKKS_ENVELOPE=binary or a ".mkws" output path

        // gather every plaintext vector of the file
        struct Slot { size_t layer; int field; };
        std::vector<std::vector<double>> plains;
        Json outputJson = Json::object();
        outputJson["weights_summary"] = Json::array();
        std::vector<std::pair<size_t, size_t>> layer_ranges;  // [first, count) of value chunks per kept layer
        for (const Json &weight : inputJson.at("weights_summary").a) {
            const std::string layerName = weight.at("layer").as_string();
            if (layerName.rfind("optimizer/", 0) == 0) {
                std::cout << "[encrypt] Skipping optimizer layer: " << layerName << std::endl;
                continue;
            }
            Json enc = Json::object();
            enc["layer"] = layerName;
            enc["shape"] = weight.at("shape");
            outputJson["weights_summary"].push_back(enc);
            plains.push_back({weight.at("mean").as_double()});
            plains.push_back({weight.at("std_dev").as_double()});
            const auto &vals = weight.at("values").a;
            size_t first = plains.size(), chunks = 0;
            for (size_t i = 0; i < vals.size(); i += batchSize) {
                size_t end = std::min(i + batchSize, vals.size());
                std::vector<double> chunk(batchSize, 0.0);  // zero padded to batchSize
                for (size_t k = i; k < end; ++k) chunk[k - i] = vals[k].as_double();
                plains.push_back(std::move(chunk));
                ++chunks;
            }
            layer_ranges.push_back({first, chunks});
        }
        const size_t n_ct = plains.size();
        // Encode: FLEXIBLEAUTOEXT level-0 plaintexts carry the big scaling factor sf[0]*sf[1], noiseScaleDeg 2
        const double scale = s.sf(0, true);
        const size_t slots = s.slots();
        std::vector<double> slot_vals(n_ct * slots, 0.0);  // zero padded to N/2 slots
        for (size_t c = 0; c < n_ct; ++c) std::copy(plains[c].begin(), plains[c].end(), slot_vals.begin() + c * slots);
        const SamplerKey k_v = fresh_key(), k_e0 = fresh_key(), k_e1 = fresh_key();
        int8_t *d_v = s.alloc<int8_t>(n_ct * N);
        int32_t *d_e0 = s.alloc<int32_t>(n_ct * N), *d_e1 = s.alloc<int32_t>(n_ct * N);
        Session::check(mkckks_sample_ternary(s.ctx(), d_v, n_ct * N, k_v.bytes, 0));
        Session::check(mkckks_sample_gauss(s.ctx(), d_e0, n_ct * N, 3.19, k_e0.bytes, 1));
        Session::check(mkckks_sample_gauss(s.ctx(), d_e1, n_ct * N, 3.19, k_e1.bytes, 2));
        uint64_t *d_pt = s.alloc<uint64_t>(n_ct * L * N), *d_ct = s.alloc<uint64_t>(n_ct * 2 * L * N);
        Session::check(mkckks_encode_batch(s.ctx(), s.to_device(slot_vals.data(), slot_vals.size()), d_pt, (uint32_t)n_ct, L, scale));
        Session::check(mkckks_encrypt_batch(s.ctx(), s.to_device(pk.data(), pk.size()), d_pt, d_v, d_e0, d_e1, d_ct,
                                            (uint32_t)n_ct, L));
        std::vector<uint64_t> all(n_ct * 2 * L * N);
        s.to_host(all.data(), d_ct, all.size());
        auto blob = [&](size_t c) {
            Ciphertext ct;
            ct.nl = L; ct.level = 0; ct.noise_deg = 2; ct.scale = scale; ct.slots = s.slots();
            ct.data.assign(all.begin() + c * 2 * L * N, all.begin() + (c + 1) * 2 * L * N);
            return encode_ct(ct, N);
        };
        size_t c = 0;
        for (size_t l = 0; l < layer_ranges.size(); ++l) {
            Json &enc = outputJson["weights_summary"].a[l];
            enc["mean"] = blob(c++);
            enc["std_dev"] = blob(c++);
            Json arr = Json::array();
            for (size_t k = 0; k < layer_ranges[l].second; ++k) arr.push_back(blob(c++));
            enc["values"] = arr;
        }
        try {
            write_envelope(outputJson, output_encfile, raw_blobs());
        } catch (const std::exception &) {
            std::cerr << "[encrypt] ERROR: Failed to write to output file: " << output_encfile << std::endl;
            return 1;
        }
    } catch (const std::exception &e) {
        std::cerr << "[encrypt] ERROR: " << e.what() << std::endl;
        return 1;
    }
    std::cout << "[encrypt] Encryption completed successfully and saved in " << output_encfile << std::endl;
    return 0;
}
