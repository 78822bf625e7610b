// serverRound -- the server side of one federated round in ONE program.
//
// The reference's server step is a loop of processes (orchestration/server_fns.sh:62-80, orchestration/run.sh:37-43):
//   changeCipherDomain <cc> <rekey_c> <enc_c> <tmp_c>      for every client c that is not in the target key domain
//   aggregateEncryptedWeights <cc> <enc_target> <tmp_...> <out>
// each of which re-loads the CryptoContext, re-parses base64-in-JSON and moves every ciphertext through the host.  Here:
//   serverRound <cc_path> <output_aggfile> <rekey_1|-> <encfile_1> [<rekey_2|-> <encfile_2> ...]
//               [--back <rekey_back_1> <output_encfile_1> [<rekey_back_2> <output_encfile_2> ...]]
// "-" as the re-encryption key marks a client whose ciphertexts already are in the target domain (the reference's
// client 2).  All ciphertexts go to HBM once; the re-encryption of every re-keyed client and the sum over clients is ONE
// mkckks_reencrypt_sum_batch call (cc->ReEncrypt x n at changeCipherDomain.cpp:74 + the EvalAdd chain of
// aggregateEncryptedWeights.cpp:82), then EvalMult(., 1/n) (:83).  The output file is bit-identical to the one the
// per-client programs produce (PRE is deterministic; tests/test_cli_hosts.py).
// After --back: the n-1 re-encryptions of the aggregate into the other clients' key domains (the reference's
// s_changeCipherDomain_c2_c1, server_fns.sh:76-80: changeCipherDomain <cc> <rekey_back_c> <aggfile> <out_c>) -- the
// aggregate stays in HBM, one mkckks_reencrypt_batch per key; each file equals changeCipherDomain run on <output_aggfile>.
// Binary (MKWS) envelopes take the I/O pipeline of iopipe.hpp: files are indexed, not loaded; reader threads fill pinned
// slots, uploads run beside the reads, residues are range-checked on the device, results are written by pwrite() from
// pinned slots.  MKCKKS_SYNC_IO=1 forces the synchronous path (every ciphertext through read_envelope / decode_ct /
// mkckks_upload); both write the same bytes (tests/test_cli_hosts.py).  A "[round] timing" line reports the phases.
//   serverRound <cc_path> --rounds <file>
// runs one round per line of <file> (a line = the arguments after <cc_path> above) in ONE process -- what the loop of
// orchestration/run.sh:37-43 does with one process per step: the context, the re-encryption keys (by file name), the
// device arrays, the pinned buffers and the resolved kernels stay from round to round.
#include <fstream>
#include <map>
#include <sstream>

#include "iopipe.hpp"
using namespace mkh;

struct RoundArgs {
    std::string output_file;
    std::vector<std::string> rekey_paths, enc_paths, back_keys, back_outs;
};
// tokens: <output_aggfile> <rekey_1|-> <encfile_1> ... [--back <rekey_back_1> <output_encfile_1> ...]
static bool parse_round(const std::vector<std::string> &t, RoundArgs &a) {
    size_t n_args = t.size();
    for (size_t i = 1; i < t.size(); ++i)
        if (t[i] == "--back") {
            n_args = i;
            break;
        }
    const size_t n_back = t.size() - n_args - (n_args < t.size() ? 1 : 0);
    if (n_args < 3 || (n_args - 1) % 2 != 0 || n_back % 2 != 0 || (n_args < t.size() && n_back == 0)) return false;
    a = RoundArgs{};
    a.output_file = t[0];
    for (size_t i = 1; i + 1 < n_args; i += 2) {
        a.rekey_paths.push_back(t[i]);
        a.enc_paths.push_back(t[i + 1]);
    }
    for (size_t i = n_args + 1; i + 1 < t.size(); i += 2) {
        a.back_keys.push_back(t[i]);
        a.back_outs.push_back(t[i + 1]);
    }
    return true;
}

struct ServerState {
    explicit ServerState(Session &s) : cache(s) {}
    std::map<std::string, std::vector<uint64_t>> keys;  // re-encryption keys by file name, loaded once per process
    RoundCache cache;
    double t_ctx = 0;      // ms: context creation
    size_t n_ct = 0;       // ciphertexts re-encrypted / aggregated so far
};

static const std::vector<uint64_t> *cached_key(Session &s, ServerState &st, const std::string &path) {
    auto it = st.keys.find(path);
    if (it != st.keys.end()) return &it->second;
    std::vector<uint64_t> evk;
    if (!load_eval_key(s, path, evk)) return nullptr;
    return &(st.keys[path] = std::move(evk));
}

// one round; 0 on success, 1 after an "[round] ERROR" line
static int run_round(Session &s, ServerState &st, const RoundArgs &a) {
    const double t_start = now_ms();
    const std::vector<std::string> &rekey_paths = a.rekey_paths, &enc_paths = a.enc_paths;
    const uint32_t N = s.N(), D = s.D(), beta = s.beta();
    const size_t n_clients = enc_paths.size(), evk_words = (size_t)beta * 2 * D * N;
    // clients with a re-encryption key first (their ciphertexts feed mkckks_reencrypt_sum_batch), the others after
    std::vector<size_t> order;
    for (size_t c = 0; c < n_clients; ++c)
        if (rekey_paths[c] != "-") order.push_back(c);
    const size_t n_pre = order.size();
    for (size_t c = 0; c < n_clients; ++c)
        if (rekey_paths[c] == "-") order.push_back(c);
    std::vector<const uint64_t *> evk_ptrs;
    std::vector<std::string> evk_names;
    for (size_t k = 0; k < n_pre; ++k) {
        const std::vector<uint64_t> *evk = cached_key(s, st, rekey_paths[order[k]]);
        if (!evk) {
            std::cerr << "[round] ERROR: Failed to load ReKey from " << rekey_paths[order[k]] << std::endl;
            return 1;
        }
        evk_ptrs.push_back(evk->data());
        evk_names.push_back(rekey_paths[order[k]]);
    }
    const double t_keys = now_ms() - t_start;
    std::cout << "[round] " << n_pre << " ReKey(s) loaded\n";
    std::vector<Json> files;
    bool binary = false;  // the output keeps the first input's envelope form
    // binary envelopes: index them (skeleton + blob offsets); anything else goes through read_envelope as before
    const bool want_pipe = !(std::getenv("MKCKKS_SYNC_IO") && std::atoi(std::getenv("MKCKKS_SYNC_IO")) != 0);
    std::vector<EnvelopeIndex> idx(n_clients);
    bool piped = want_pipe;
    const double t_io0 = now_ms();
    for (size_t k = 0; k < n_clients && piped; ++k) {
        try {
            piped = index_envelope(enc_paths[order[k]], idx[k]);
        } catch (const std::exception &) {
            std::cerr << "[round] ERROR: Could not open input encrypted weights file " << enc_paths[order[k]] << std::endl;
            return 1;
        }
    }
    if (piped) {
        binary = true;
        for (size_t k = 0; k < n_clients; ++k) files.push_back(idx[k].doc);
    } else {
        for (size_t k = 0; k < n_clients; ++k) {
            bool b = false;
            try {
                files.push_back(read_envelope(enc_paths[order[k]], &b));
            } catch (const std::exception &) {
                std::cerr << "[round] ERROR: Could not open input encrypted weights file " << enc_paths[order[k]] << std::endl;
                return 1;
            }
            if (k == 0) binary = b;
        }
    }
    raw_blobs() = binary;
    Json outputJson;
    const std::vector<AggItem> items = build_agg_items(files, outputJson);
    AggResult agg;
    const unsigned threads = io_threads();
    bool pinned_out = false;
    if (!items.empty() && piped) {
        RoundPlan plan;
        plan.n_clients = n_clients;
        plan.n_pre = n_pre;
        plan.idx = &idx;
        plan.items = &items;
        plan.evks = evk_ptrs;
        plan.evk_names = &evk_names;
        plan.evk_words = evk_words;
        plan.threads = threads;
        RoundTimes tm;
        agg = run_round_pipeline(s, plan, outputJson, a.output_file, st.cache, tm);
        pinned_out = true;
        const double n_ct = (double)(n_clients * items.size());
        std::cout << "[round] timing: " << n_clients << " clients x " << items.size() << " ciphertexts, chunks of " << tm.chunk
                  << " indices, " << threads << " I/O threads: files to file " << tm.round << " ms -> " << n_ct / tm.round * 1e3
                  << " ciphertexts/s (last upload done at " << tm.last_upload << " ms, last download at " << tm.last_download
                  << " ms; reader threads busy " << tm.read_busy << " ms, writer threads " << tm.write_busy
                  << " ms in total); before it: context (HIP start-up, tables; once per process) " << st.t_ctx
                  << " ms, re-encryption key file(s) not yet loaded " << t_keys << " ms, index + buffers + key upload + warm-up "
                  << now_ms() - t_io0 - tm.round << " ms\n";
    } else if (!items.empty()) {
        const size_t B = items.size();
        const size_t n_plain = n_clients - n_pre;
        std::vector<uint64_t> flat;  // [client in `order`][ct][2][nl][N]
        const Ciphertext first = gather_agg_inputs(items, n_clients, s, flat);
        const size_t words = (size_t)2 * first.nl * N, blk = B * words;
        // device layout: [re-keyed clients][one slot for their re-encrypted sum][clients already in the domain]:
        // the slot and what follows it are the terms of the final n-ary EvalAdd, no copy in between
        uint64_t *d_all = st.cache.grow(st.cache.all, (n_clients + 1) * blk), *d_slot = d_all + n_pre * blk;
        if (n_pre) Session::check(mkckks_upload(s.ctx(), d_all, flat.data(), n_pre * blk * 8));
        if (n_plain) Session::check(mkckks_upload(s.ctx(), d_slot + blk, flat.data() + n_pre * blk, n_plain * blk * 8));
        const uint32_t nl = first.nl;
        uint64_t *d_sum = d_slot;
        if (n_pre) {
            uint64_t *d_evk = st.cache.grow(st.cache.evk, n_pre * evk_words);
            st.cache.evk_names.clear();
            for (size_t k = 0; k < n_pre; ++k) Session::check(mkckks_upload(s.ctx(), d_evk + k * evk_words, evk_ptrs[k], evk_words * 8));
            st.cache.evk_names = evk_names;
            Session::check(mkckks_reencrypt_sum_batch(s.ctx(), d_all, d_evk, d_slot, (uint32_t)n_pre, (uint32_t)B, nl));
        }
        if (n_plain) {
            const uint64_t *d_terms = n_pre ? d_slot : d_slot + blk;
            d_sum = st.cache.grow(st.cache.sum, blk);
            Session::check(mkckks_eval_sum_batch(s.ctx(), d_terms, d_sum, (uint32_t)(n_plain + (n_pre ? 1 : 0)), (uint32_t)B, nl));
        }
        agg = finish_aggregate(s, items, d_sum, first, n_clients, outputJson);
    }
    if (!pinned_out) write_envelope(outputJson, a.output_file, binary);
    for (size_t k = 0; k < a.back_keys.size(); ++k) {
        const std::vector<uint64_t> *evk = cached_key(s, st, a.back_keys[k]);
        if (!evk) {
            std::cerr << "[round] ERROR: Failed to load ReKey from " << a.back_keys[k] << std::endl;
            return 1;
        }
        Json backJson = outputJson;  // layer / shape carried over; blobs replaced below
        if (!items.empty()) {
            const size_t words = items.size() * (size_t)2 * agg.meta.nl * N;
            uint64_t *d_back = st.cache.grow(st.cache.back, words), *d_back_evk = st.cache.grow(st.cache.back_evk, evk_words);
            Session::check(mkckks_upload(s.ctx(), d_back_evk, evk->data(), evk_words * 8));
            Session::check(mkckks_reencrypt_batch(s.ctx(), agg.d_out, d_back_evk, d_back, (uint32_t)items.size(), agg.meta.nl));
            if (pinned_out) {
                write_envelope_from_device(s, *st.cache.ring, items, d_back, agg.meta, backJson, a.back_outs[k], threads);
            } else {
                store_agg_items(s, items, d_back, agg.meta, backJson);
                write_envelope(backJson, a.back_outs[k], binary);
            }
        } else {
            write_envelope(backJson, a.back_outs[k], binary);
        }
        std::cout << "[round] aggregate re-encrypted with " << a.back_keys[k] << " -> " << a.back_outs[k] << "\n";
    }
    st.n_ct += n_clients * items.size();
    std::cout << "[round] Re-encryption and aggregation completed successfully. Output: " << a.output_file << std::endl;
    return 0;
}

int main(int argc, char *argv[]) {
    auto usage = [&] {
        std::cerr << "Usage: " << argv[0] << " <cc_path> <output_aggfile> <rekey_1|-> <encfile_1> [<rekey_2|-> <encfile_2> ...]"
                  << " [--back <rekey_back_1> <output_encfile_1> ...]\n       " << argv[0]
                  << " <cc_path> --rounds <file with one such argument list (after <cc_path>) per line>" << std::endl;
        return 1;
    };
    if (argc < 4) return usage();
    const std::string cc_path = argv[1];
    std::vector<RoundArgs> rounds;
    if (std::string(argv[2]) == "--rounds") {
        if (argc != 4) return usage();
        std::ifstream f(argv[3]);
        if (!f) {
            std::cerr << "[round] ERROR: Could not open rounds file " << argv[3] << std::endl;
            return 1;
        }
        std::string line;
        while (std::getline(f, line)) {
            std::istringstream ls(line);
            std::vector<std::string> t;
            for (std::string w; ls >> w;) t.push_back(w);
            if (t.empty() || t[0][0] == '#') continue;
            RoundArgs a;
            if (!parse_round(t, a)) {
                std::cerr << "[round] ERROR: malformed round (line " << rounds.size() + 1 << " of " << argv[3] << ")" << std::endl;
                return 1;
            }
            rounds.push_back(std::move(a));
        }
        if (rounds.empty()) return usage();
    } else {
        RoundArgs a;
        if (!parse_round(std::vector<std::string>(argv + 2, argv + argc), a)) return usage();
        rounds.push_back(std::move(a));
    }
    const double t_start = now_ms();
    CcFile cc;
    try {
        cc = read_cc(cc_path);
    } catch (const std::exception &) {
        std::cerr << "[round] ERROR: Failed to load CryptoContext from " << cc_path << std::endl;
        return 1;
    }
    try {
        Session s(cc);
        ServerState st(s);
        st.t_ctx = now_ms() - t_start;
        std::cout << "[round] CryptoContext loaded\n";
        const double t_rounds = now_ms();
        for (size_t r = 0; r < rounds.size(); ++r) {
            const double t_r = now_ms();
            if (run_round(s, st, rounds[r])) return 1;
            if (rounds.size() > 1)
                std::cout << "[round] round " << r + 1 << " of " << rounds.size() << ": " << now_ms() - t_r << " ms wall\n";
        }
        if (rounds.size() > 1) {
            const double ms = now_ms() - t_rounds;
            std::cout << "[round] " << rounds.size() << " rounds, " << st.n_ct << " ciphertexts in " << ms << " ms wall after the context ("
                      << st.t_ctx << " ms) = " << (double)st.n_ct / ms * 1e3 << " ciphertexts/s, " << (double)st.n_ct / (ms + st.t_ctx) * 1e3
                      << " with it\n";
        }
    } catch (const std::exception &e) {
        std::cerr << "[round] ERROR: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
