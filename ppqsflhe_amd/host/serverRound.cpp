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
#include "iopipe.hpp"
using namespace mkh;

int main(int argc, char *argv[]) {
    int n_args = argc;
    for (int i = 3; i < argc; ++i)
        if (std::string(argv[i]) == "--back") {
            n_args = i;
            break;
        }
    const int n_back_args = argc - n_args - (n_args < argc ? 1 : 0);
    if (n_args < 5 || (n_args - 3) % 2 != 0 || n_back_args % 2 != 0 || (n_args < argc && n_back_args == 0)) {
        std::cerr << "Usage: " << argv[0] << " <cc_path> <output_aggfile> <rekey_1|-> <encfile_1> [<rekey_2|-> <encfile_2> ...]"
                  << " [--back <rekey_back_1> <output_encfile_1> ...]" << std::endl;
        return 1;
    }
    const std::string cc_path = argv[1], output_file = argv[2];
    std::vector<std::string> rekey_paths, enc_paths, back_keys, back_outs;
    for (int i = 3; i + 1 < n_args; i += 2) {
        rekey_paths.push_back(argv[i]);
        enc_paths.push_back(argv[i + 1]);
    }
    for (int i = n_args + 1; i + 1 < argc; i += 2) {
        back_keys.push_back(argv[i]);
        back_outs.push_back(argv[i + 1]);
    }
    CcFile cc;
    try {
        cc = read_cc(cc_path);
    } catch (const std::exception &) {
        std::cerr << "[round] ERROR: Failed to load CryptoContext from " << cc_path << std::endl;
        return 1;
    }
    try {
        Session s(cc);
        std::cout << "[round] CryptoContext loaded\n";
        const uint32_t N = s.N(), D = s.D(), beta = s.beta();
        const size_t n_clients = enc_paths.size(), evk_words = (size_t)beta * 2 * D * N;
        // clients with a re-encryption key first (their ciphertexts feed mkckks_reencrypt_sum_batch), the others after
        std::vector<size_t> order;
        for (size_t c = 0; c < n_clients; ++c)
            if (rekey_paths[c] != "-") order.push_back(c);
        const size_t n_pre = order.size();
        for (size_t c = 0; c < n_clients; ++c)
            if (rekey_paths[c] == "-") order.push_back(c);
        std::vector<uint64_t> evks(n_pre * evk_words);
        for (size_t k = 0; k < n_pre; ++k) {
            std::vector<uint64_t> evk;
            if (!load_eval_key(s, rekey_paths[order[k]], evk)) {
                std::cerr << "[round] ERROR: Failed to load ReKey from " << rekey_paths[order[k]] << std::endl;
                return 1;
            }
            std::memcpy(&evks[k * evk_words], evk.data(), evk_words * 8);
        }
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
        uint64_t *d_back = nullptr, *d_back_evk = nullptr;
        std::unique_ptr<PinnedRing> ring;
        const unsigned threads = io_threads();
        if (!items.empty() && piped) {
            RoundPlan plan;
            plan.n_clients = n_clients;
            plan.n_pre = n_pre;
            plan.idx = &idx;
            plan.items = &items;
            plan.evks = evks.data();
            plan.evk_words = evk_words;
            plan.threads = threads;
            RoundTimes tm;
            agg = run_round_pipeline(s, plan, outputJson, output_file, ring, tm);
            const double n_ct = (double)(n_clients * items.size());
            std::cout << "[round] timing: " << n_clients << " clients x " << items.size() << " ciphertexts, chunks of " << tm.chunk
                      << " indices, " << threads << " I/O threads: files to file " << tm.round << " ms -> " << n_ct / tm.round * 1e3
                      << " ciphertexts/s (last upload done at " << tm.last_upload << " ms, last download at " << tm.last_download
                      << " ms; reader threads busy " << tm.read_busy << " ms, writer threads " << tm.write_busy
                      << " ms in total); once per process: index + buffers + key upload " << now_ms() - t_io0 - tm.round
                      << " ms\n";
        } else if (!items.empty()) {
            const size_t B = items.size();
            const size_t n_plain = n_clients - n_pre;
            std::vector<uint64_t> flat;  // [client in `order`][ct][2][nl][N]
            const Ciphertext first = gather_agg_inputs(items, n_clients, s, flat);
            const size_t words = (size_t)2 * first.nl * N, blk = B * words;
            // device layout: [re-keyed clients][one slot for their re-encrypted sum][clients already in the domain]:
            // the slot and what follows it are the terms of the final n-ary EvalAdd, no copy in between
            uint64_t *d_all = s.alloc<uint64_t>((n_clients + 1) * blk), *d_slot = d_all + n_pre * blk;
            if (n_pre) Session::check(mkckks_upload(s.ctx(), d_all, flat.data(), n_pre * blk * 8));
            if (n_plain) Session::check(mkckks_upload(s.ctx(), d_slot + blk, flat.data() + n_pre * blk, n_plain * blk * 8));
            const uint32_t nl = first.nl;
            uint64_t *d_sum = d_slot;
            if (n_pre) {
                uint64_t *d_evk = s.to_device(evks.data(), evks.size());
                Session::check(mkckks_reencrypt_sum_batch(s.ctx(), d_all, d_evk, d_slot, (uint32_t)n_pre, (uint32_t)B, nl));
            }
            if (n_plain) {
                const uint64_t *d_terms = n_pre ? d_slot : d_slot + blk;
                d_sum = s.alloc<uint64_t>(blk);
                Session::check(mkckks_eval_sum_batch(s.ctx(), d_terms, d_sum, (uint32_t)(n_plain + (n_pre ? 1 : 0)),
                                                     (uint32_t)B, nl));
            }
            agg = finish_aggregate(s, items, d_sum, first, n_clients, outputJson);
        }
        if (!(piped && !items.empty()))
        write_envelope(outputJson, output_file, binary);
        for (size_t k = 0; k < back_keys.size(); ++k) {
            std::vector<uint64_t> evk;
            if (!load_eval_key(s, back_keys[k], evk)) {
                std::cerr << "[round] ERROR: Failed to load ReKey from " << back_keys[k] << std::endl;
                return 1;
            }
            Json backJson = outputJson;  // layer / shape carried over; blobs replaced below
            if (!items.empty()) {
                const size_t words = items.size() * (size_t)2 * agg.meta.nl * N;
                if (!d_back) d_back = s.alloc<uint64_t>(words);
                if (!d_back_evk) d_back_evk = s.alloc<uint64_t>(evk_words);
                Session::check(mkckks_upload(s.ctx(), d_back_evk, evk.data(), evk_words * 8));
                Session::check(mkckks_reencrypt_batch(s.ctx(), agg.d_out, d_back_evk, d_back, (uint32_t)items.size(), agg.meta.nl));
                if (ring) {
                    write_envelope_from_device(s, *ring, items, d_back, agg.meta, backJson, back_outs[k], threads);
                } else {
                    store_agg_items(s, items, d_back, agg.meta, backJson);
                    write_envelope(backJson, back_outs[k], binary);
                }
            } else {
                write_envelope(backJson, back_outs[k], binary);
            }
            std::cout << "[round] aggregate re-encrypted with " << back_keys[k] << " -> " << back_outs[k] << "\n";
        }
    } catch (const std::exception &e) {
        std::cerr << "[round] ERROR: " << e.what() << std::endl;
        return 1;
    }
    std::cout << "[round] Re-encryption and aggregation completed successfully. Output: " << output_file << std::endl;
    return 0;
}
