// aggregateEncryptedWeights -- drop-in for server/src/aggregateEncryptedWeights.cpp:
// `aggregateEncryptedWeights <cc_path> <client2_encfile> <client1to2_encfile> <output_aggfile>` (:33-43; caller
// server_fns.sh:72).  For every (w2, w1) pair with equal "layer" and "shape" (:68-72): EvalAdd then EvalMult(.,0.5)
// on mean, std_dev and the first min(|v1|,|v2|) value ciphertexts (:80-109).
// n-client generalisation: any number of further encfiles may follow the output path; every
// matching ciphertext is summed with one mkckks_eval_sum_batch and scaled by 1/n_files.  (serverRound does the
// re-encryptions and this aggregation in one program.)
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

        // one output entry per matching (layer, shape) tuple (aggregateEncryptedWeights.cpp:68-72,115)
        Json outputJson;
        const std::vector<AggItem> items = build_agg_items(files, outputJson);
        if (!items.empty()) {
            std::vector<uint64_t> flat;  // [client][ct][2][nl][N]
            const Ciphertext first = gather_agg_inputs(items, n_files, s, flat);
            const size_t B = items.size(), words = (size_t)2 * first.nl * N;
            uint64_t *d_in = s.to_device(flat.data(), flat.size());
            uint64_t *d_sum = s.alloc<uint64_t>(B * words);
            Session::check(mkckks_eval_sum_batch(s.ctx(), d_in, d_sum, (uint32_t)n_files, (uint32_t)B, first.nl));
            finish_aggregate(s, items, d_sum, first, n_files, outputJson);
        }
        write_envelope(outputJson, output_file, binary);
    } catch (const std::exception &e) {
        std::cerr << "[agg] ERROR: " << e.what() << std::endl;
        return 1;
    }
    std::cout << "[agg] Aggregation completed successfully. Output: " << output_file << std::endl;
    return 0;
}
