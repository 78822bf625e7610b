// aggregateEncryptedWeights -- drop-in for server/src/aggregateEncryptedWeights.cpp:
// `aggregateEncryptedWeights <cc_path> <client2_encfile> <client1to2_encfile> <output_aggfile>` (:33-43; caller
// server_fns.sh:72).  For every (w2, w1) pair with equal "layer" and "shape" (:68-72): EvalAdd then EvalMult(.,0.5)
// on mean, std_dev and the first min(|v1|,|v2|) value ciphertexts (:80-109).
// n-client generalisation: any number of further encfiles may follow the output path; every
// matching ciphertext is summed with one mkckks_eval_sum_batch and scaled by 1/n_files.
#include "hostlib.hpp"
using namespace mkh;

int main(int argc, char *argv[]) {
    if (argc < 5) {
        std::cerr << "Usage: " << argv[0] << " <cc_path> <client2_encfile> <client1to2_encfile> <output_aggfile>" << std::endl;
        return 1;
    }
    const std::string cc_path = argv[1], output_file = argv[4];
    std::vector<std::string> in_paths{argv[2], argv[3]};
    for (int i = 5; i < argc; ++i) in_paths.push_back(argv[i]);
    CcFile cc;
    try {
        cc = read_cc(cc_path);
    } catch (const std::exception &) {
        std::cerr << "[agg] ERROR: Failed to load CryptoContext from " << cc_path << std::endl;
        return 1;
    }
    try {
        Session s(cc);
        std::cout << "[agg] CryptoContext loaded\n";
        const uint32_t N = s.N();
        std::vector<Json> files;
        bool binary = false;  // the output keeps the first input's envelope form
        for (const std::string &p : in_paths) {
            bool b = false;
            files.push_back(read_envelope(p, &b));
            if (files.size() == 1) binary = b;
        }
        raw_blobs() = binary;
        const size_t n_files = files.size();

        // match layers: for each layer of file 0 (client 2) the same layer+shape in every other file
        Json outputJson = Json::object();
        outputJson["weights_summary"] = Json::array();
        struct Item { std::vector<const std::string *> blobs; size_t out_layer; int field; size_t idx; };
        std::vector<Item> items;
        for (const Json &w2 : files[0].at("weights_summary").a) {
            std::vector<const Json *> match{&w2};
            for (size_t f = 1; f < n_files; ++f) {
                const Json *hit = nullptr;
                for (const Json &w1 : files[f].at("weights_summary").a)
                    if (w1.at("layer") == w2.at("layer") && w1.at("shape") == w2.at("shape")) { hit = &w1; break; }
                if (hit) match.push_back(hit);
            }
            if (match.size() != n_files) continue;  // reference: no output entry without a matching pair
            Json agg = Json::object();
            agg["layer"] = w2.at("layer");
            agg["shape"] = w2.at("shape");
            size_t nvals = (size_t)-1;
            for (const Json *m : match) nvals = std::min(nvals, m->at("values").size());
            Json vals = Json::array();
            for (size_t k = 0; k < nvals; ++k) vals.push_back(Json(""));
            agg["values"] = vals;
            const size_t out_layer = outputJson["weights_summary"].size();
            outputJson["weights_summary"].push_back(agg);
            Item mean{{}, out_layer, 0, 0}, sd{{}, out_layer, 1, 0};
            for (const Json *m : match) { mean.blobs.push_back(&m->at("mean").as_string()); sd.blobs.push_back(&m->at("std_dev").as_string()); }
            items.push_back(mean);
            items.push_back(sd);
            for (size_t k = 0; k < nvals; ++k) {
                Item v{{}, out_layer, 2, k};
                for (const Json *m : match) v.blobs.push_back(&m->at("values").at(k).as_string());
                items.push_back(v);
            }
        }
        if (!items.empty()) {
            Ciphertext first = decode_ct(*items[0].blobs[0], N);
            const uint32_t nl = first.nl;
            if (nl < 2) throw std::runtime_error("ciphertext has no limb left to rescale");
            const size_t words = (size_t)2 * nl * N, B = items.size();
            std::vector<uint64_t> flat(n_files * B * words);  // [client][ct][2][nl][N]
            for (size_t b = 0; b < B; ++b)
                for (size_t f = 0; f < n_files; ++f) {
                    Ciphertext ct = decode_ct(*items[b].blobs[f], N);
                    if (ct.nl != nl || ct.noise_deg != first.noise_deg || ct.scale != first.scale)
                        throw std::runtime_error("EvalAdd operands differ in level or scale");
                    std::memcpy(&flat[(f * B + b) * words], ct.data.data(), words * 8);
                }
            uint64_t *d_in = s.to_device(flat.data(), flat.size());
            uint64_t *d_sum = s.alloc<uint64_t>(B * words);
            uint64_t *d_out = s.alloc<uint64_t>(B * (size_t)2 * (nl - 1) * N);
            Session::check(mkckks_eval_sum_batch(s.ctx(), d_in, d_sum, (uint32_t)n_files, (uint32_t)B, nl));
            Ciphertext res;
            const double operand = 1.0 / (double)n_files;  // 0.5 for the reference's two clients
            if (first.noise_deg == 2) {
                // EvalMult(ct, double): rescale first (ModReduceInternalInPlace), then the integer constant
                Session::check(mkckks_rescale_mult_const_batch(s.ctx(), d_sum, d_out, (uint32_t)B, nl, operand));
                res.nl = nl - 1;
                res.level = first.level + 1;
                const double sf_new = s.sf(res.level, false);
                res.scale = first.scale / (double)s.moduli()[nl - 1] * sf_new;
                res.noise_deg = 2;
            } else {
                Session::check(mkckks_mult_const_batch(s.ctx(), d_sum, (uint32_t)B, nl, operand));
                d_out = d_sum;
                res.nl = nl;
                res.level = first.level;
                res.scale = first.scale * s.sf(first.level, false);
                res.noise_deg = first.noise_deg + 1;
            }
            res.slots = first.slots;
            const size_t owords = (size_t)2 * res.nl * N;
            std::vector<uint64_t> out(B * owords);
            s.to_host(out.data(), d_out, out.size());
            for (size_t b = 0; b < B; ++b) {
                res.data.assign(out.begin() + b * owords, out.begin() + (b + 1) * owords);
                Json &lay = outputJson["weights_summary"].a[items[b].out_layer];
                std::string b64 = encode_ct(res, N);
                if (items[b].field == 0) lay["mean"] = std::move(b64);
                else if (items[b].field == 1) lay["std_dev"] = std::move(b64);
                else lay["values"].a[items[b].idx] = Json(std::move(b64));
            }
        }
        write_envelope(outputJson, output_file, binary);
    } catch (const std::exception &e) {
        std::cerr << "[agg] ERROR: " << e.what() << std::endl;
        return 1;
    }
    std::cout << "[agg] Aggregation completed successfully. Output: " << output_file << std::endl;
    return 0;
}
