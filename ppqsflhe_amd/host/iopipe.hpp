// iopipe.hpp -- I/O pipeline of the server hosts for binary (MKWS) envelopes (second half).
//
// The reference's server programs move every ciphertext through the host one at a time and synchronously: file -> JSON
// string -> Base64Decode -> Serial::Deserialize -> OpenFHE call -> Serialize -> Base64Encode -> JSON -> file
// (server/src/changeCipherDomain.cpp:61-123, server/src/aggregateEncryptedWeights.cpp:18-30,54-119).  With the
// arithmetic on the GPU at ~28 k ciphertexts/s the host side is the wall: a 12 MiB ciphertext is used once, so the
// PCIe link (~50 GB/s, ~4 k ct/s) is the ceiling of a host-resident deployment and every extra copy on the host counts.
// Here, for MKWS envelopes:
//   * the file is INDEXED, not loaded: skeleton JSON + (offset, size) of every ciphertext container;
//   * reader threads pread() ciphertext payloads straight into a ring of PINNED slots (mkckks_host_alloc); the main
//     thread enqueues each filled slot on the context's upload stream (mkckks_upload_async) and recycles it when its
//     ticket is done -- no pageable staging copy in between;
//   * the round runs by chunks of ciphertext indices (run_round_pipeline): re-encryption + sum + scaling of chunk c are
//     enqueued behind its uploads (mkckks_fence_uploads) while chunk c+1 is being read and uploaded;
//   * results come back into a pinned buffer on the download stream (mkckks_fence_compute, mkckks_download_async) and are
//     written with pwrite() by writer threads at offsets known in advance -- the output file is byte-identical to what
//     write_envelope() produces;
//   * residues are range-checked on the DEVICE (mkckks_count_noncanonical) instead of in a host loop, headers on the host.
// One host thread drives the context; the other threads only touch file descriptors and pinned memory.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include "hostlib.hpp"

namespace mkh {

inline double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct BlobRef {
    uint64_t offset = 0, size = 0;  // the container (BlobHeader + residues) inside the file
};
// an MKWS file without its blobs: the skeleton document (ciphertext fields hold "@<index>") and where every blob sits
struct EnvelopeIndex {
    Json doc;
    std::vector<BlobRef> blobs;
    int fd = -1;
    EnvelopeIndex() = default;
    EnvelopeIndex(const EnvelopeIndex &) = delete;
    EnvelopeIndex &operator=(const EnvelopeIndex &) = delete;
    EnvelopeIndex(EnvelopeIndex &&o) noexcept : doc(std::move(o.doc)), blobs(std::move(o.blobs)), fd(o.fd) { o.fd = -1; }
    ~EnvelopeIndex() {
        if (fd >= 0) ::close(fd);
    }
};

inline void pread_all(int fd, void *dst, size_t bytes, uint64_t offset) {
    char *p = static_cast<char *>(dst);
    while (bytes) {
        const ssize_t r = ::pread(fd, p, bytes, (off_t)offset);
        if (r <= 0) throw std::runtime_error("binary envelope: short read");
        p += r;
        offset += (uint64_t)r;
        bytes -= (size_t)r;
    }
}
inline void pwrite_all(int fd, const void *src, size_t bytes, uint64_t offset) {
    const char *p = static_cast<const char *>(src);
    while (bytes) {
        const ssize_t r = ::pwrite(fd, p, bytes, (off_t)offset);
        if (r <= 0) throw std::runtime_error("cannot write output file");
        p += r;
        offset += (uint64_t)r;
        bytes -= (size_t)r;
    }
}

// false: not an MKWS file (the caller takes the JSON path).  Every size field is checked against the file size, as in
// read_envelope(); blob indices in the skeleton are checked to be a permutation-free subset of the blob table.
inline bool index_envelope(const std::string &path, EnvelopeIndex &out) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("cannot open " + path);
    out.fd = fd;
    struct stat st;
    if (::fstat(fd, &st) != 0) throw std::runtime_error("cannot stat " + path);
    const uint64_t file_size = (uint64_t)st.st_size;
    char head[16];
    if (file_size < 24 || (pread_all(fd, head, 16, 0), std::memcmp(head, "MKWS", 4) != 0)) {
        ::close(fd);
        out.fd = -1;
        return false;
    }
    auto fail = [&](const char *why) { throw std::runtime_error(std::string("binary envelope: ") + why); };
    uint32_t version;
    uint64_t skel_len;
    std::memcpy(&version, head + 4, 4);
    std::memcpy(&skel_len, head + 8, 8);
    if (version != 1) fail("unsupported version");
    uint64_t pos = 16;
    if (skel_len > file_size - pos) fail("bad skeleton size");
    std::string skel(skel_len, '\0');
    if (skel_len) pread_all(fd, &skel[0], skel_len, pos);
    pos += skel_len;
    uint64_t n_blobs = 0;
    if (file_size - pos < 8) fail("truncated");
    pread_all(fd, &n_blobs, 8, pos);
    pos += 8;
    if (n_blobs > (file_size - pos) / 8) fail("bad blob count");
    out.blobs.resize(n_blobs);
    for (BlobRef &b : out.blobs) {
        uint64_t sz = 0;
        if (file_size - pos < 8) fail("truncated blob table");
        pread_all(fd, &sz, 8, pos);
        pos += 8;
        if (sz > file_size - pos) fail("bad blob size");
        b.offset = pos;
        b.size = sz;
        pos += sz;
    }
    out.doc = Json::parse(skel);
    std::vector<bool> used(out.blobs.size(), false);
    for_each_ct_field(out.doc, [&](Json &field) {
        const std::string &ref = field.as_string();
        if (ref.size() < 2 || ref[0] != '@') fail("ciphertext field without a blob index");
        const size_t i = std::stoull(ref.substr(1));
        if (i >= used.size() || used[i]) fail("blob index out of range or reused");
        used[i] = true;
    });
    return true;
}
inline size_t blob_index(const std::string &ref) { return std::stoull(ref.substr(1)); }

// run fn(i) for i in [0, n) on up to `threads` threads; the first exception is rethrown on the caller's thread
template <typename F>
inline void parallel_for(size_t n, unsigned threads, F &&fn) {
    threads = (unsigned)std::min<size_t>(std::max(1u, threads), n ? n : 1);
    std::atomic<size_t> next{0};
    std::exception_ptr err;
    std::mutex m;
    auto body = [&] {
        try {
            for (size_t i; (i = next.fetch_add(1)) < n;) fn(i);
        } catch (...) {
            std::lock_guard<std::mutex> g(m);
            if (!err) err = std::current_exception();
            next.store(n);
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads; ++t) pool.emplace_back(body);
    body();
    for (std::thread &t : pool) t.join();
    if (err) std::rethrow_exception(err);
}

inline unsigned io_threads() {
    if (const char *e = std::getenv("MKCKKS_IO_THREADS")) return (unsigned)std::max(1, std::atoi(e));
    const unsigned hw = std::thread::hardware_concurrency();
    return std::max(2u, std::min(8u, hw ? hw : 4u));
}

// A ring of pinned slots shared by the loader and the writer.
class PinnedRing {
public:
    PinnedRing(Session &s, size_t slot_bytes, unsigned n_slots) : s_(s), slot_bytes_(slot_bytes) {
        void *p = nullptr;
        Session::check(mkckks_host_alloc(s.ctx(), slot_bytes * n_slots, &p));
        base_ = static_cast<char *>(p);
        n_ = n_slots;
    }
    ~PinnedRing() {
        if (base_) mkckks_host_free(s_.ctx(), base_);
    }
    PinnedRing(const PinnedRing &) = delete;
    PinnedRing &operator=(const PinnedRing &) = delete;
    char *slot(unsigned i) const { return base_ + (size_t)i * slot_bytes_; }
    unsigned count() const { return n_; }
    size_t slot_bytes() const { return slot_bytes_; }

private:
    Session &s_;
    size_t slot_bytes_;
    char *base_ = nullptr;
    unsigned n_ = 0;
};

// header checks of validate_ct() for a container that went straight to the device; residues are checked there
inline Ciphertext meta_of(const BlobHeader &h, const Session &s) {
    if (std::memcmp(h.magic, "MKCK", 4) || h.version != 1 || h.kind != KIND_CT)
        throw std::runtime_error("not a mkckks ciphertext blob");
    if (h.ring_dim != s.N() || h.parts != 2) throw std::runtime_error("ciphertext does not match the CryptoContext");
    Ciphertext ct;
    ct.nl = h.limbs; ct.level = h.level; ct.noise_deg = h.noise_deg; ct.scale = h.scale; ct.slots = h.slots;
    const uint32_t L = s.L();
    if (ct.nl < 1 || ct.nl > L) throw std::runtime_error("ciphertext: limb count outside [1, L]");
    if (ct.level != L - ct.nl) throw std::runtime_error("ciphertext: level does not match its limb count");
    if (ct.noise_deg != 1 && ct.noise_deg != 2) throw std::runtime_error("ciphertext: noiseScaleDeg must be 1 or 2");
    if (!(ct.scale > 0) || !std::isfinite(ct.scale)) throw std::runtime_error("ciphertext: bad scaling factor");
    if (ct.slots > s.N() / 2) throw std::runtime_error("ciphertext: slot count exceeds N/2");
    return ct;
}

inline BlobHeader header_of(const Ciphertext &meta, uint32_t ring_dim) {
    BlobHeader h{};
    std::memcpy(h.magic, "MKCK", 4);
    h.version = 1; h.kind = KIND_CT; h.ring_dim = ring_dim; h.limbs = meta.nl; h.parts = 2;
    h.level = meta.level; h.noise_deg = meta.noise_deg; h.scale = meta.scale; h.slots = meta.slots;
    return h;
}

// Write an MKWS envelope whose B ciphertexts [B][2][meta.nl][N] sit in HBM: the skeleton is `doc` with the item fields
// replaced by "@<item index>" (items are in for_each_ct_field order: per layer mean, std_dev, values...), the blobs are
// header + residues.  Downloads go through the ring on the download stream; writer threads pwrite() at offsets fixed by
// the skeleton.  Byte-identical to store_agg_items() + write_envelope(.., binary = true).
inline void write_envelope_from_device(Session &s, PinnedRing &ring, const std::vector<AggItem> &items, const uint64_t *d_cts,
                                       const Ciphertext &meta, Json doc, const std::string &path, unsigned threads) {
    const uint32_t N = s.N();
    const size_t B = items.size(), owords = (size_t)2 * meta.nl * N, payload = owords * 8;
    if (payload > ring.slot_bytes()) throw std::logic_error("pinned slot smaller than a ciphertext");
    for (size_t b = 0; b < B; ++b) {
        Json &lay = doc["weights_summary"].a[items[b].out_layer];
        Json ref("@" + std::to_string(b));
        if (items[b].field == 0) lay["mean"] = ref;
        else if (items[b].field == 1) lay["std_dev"] = ref;
        else lay["values"].a[items[b].idx] = ref;
    }
    {   // the writer's blob order is for_each_ct_field order: it must be the item order
        size_t k = 0;
        for_each_ct_field(doc, [&](Json &field) {
            if (field.as_string() != "@" + std::to_string(k)) throw std::logic_error("envelope fields out of item order");
            ++k;
        });
        if (k != B) throw std::logic_error("envelope holds ciphertext fields that are not aggregate items");
    }
    std::string text;
    doc.dump(text, 2);
    const int fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) throw std::runtime_error("cannot write " + path);
    struct Closer {
        int fd;
        ~Closer() { ::close(fd); }
    } closer{fd};
    const uint32_t version = 1;
    const uint64_t skel_len = text.size(), n_blobs = B, blob_size = sizeof(BlobHeader) + payload;
    std::string head("MKWS", 4);
    head.append(reinterpret_cast<const char *>(&version), 4);
    head.append(reinterpret_cast<const char *>(&skel_len), 8);
    head += text;
    head.append(reinterpret_cast<const char *>(&n_blobs), 8);
    const uint64_t blobs0 = head.size();
    if (::ftruncate(fd, (off_t)(blobs0 + B * (8 + blob_size))) != 0) throw std::runtime_error("cannot size " + path);
    pwrite_all(fd, head.data(), head.size(), 0);
    const BlobHeader h = header_of(meta, N);
    Session::check(mkckks_fence_compute(s.ctx()));
    // rounds of ring.count() ciphertexts: download all of a round, then write them in parallel while ... the next round's
    // downloads wait for the slots; with B <= slots (the usual case) it is one round
    for (size_t b0 = 0; b0 < B; b0 += ring.count()) {
        const size_t nb = std::min<size_t>(ring.count(), B - b0);
        std::vector<uint64_t> tickets(nb);
        for (size_t i = 0; i < nb; ++i)
            Session::check(mkckks_download_async(s.ctx(), ring.slot((unsigned)i), d_cts + (b0 + i) * owords, payload, &tickets[i]));
        std::mutex mt;  // the context is driven by one thread at a time: ticket waits are serialised
        parallel_for(nb, threads, [&](size_t i) {
            {
                std::lock_guard<std::mutex> g(mt);
                Session::check(mkckks_copy_wait(s.ctx(), tickets[i]));
            }
            const uint64_t at = blobs0 + (b0 + i) * (8 + blob_size);
            char pre[8 + sizeof(BlobHeader)];
            std::memcpy(pre, &blob_size, 8);
            std::memcpy(pre + 8, &h, sizeof h);
            pwrite_all(fd, pre, sizeof pre, at);
            pwrite_all(fd, ring.slot((unsigned)i), payload, at + sizeof pre);
        });
    }
}

// ---- the whole round as one pipeline ---------------------------------------------------------------------------------
// file -> pinned slot -> HBM -> re-encrypt + sum + scale -> pinned buffer -> file, by CHUNKS of ciphertext indices: while
// chunk c is being re-encrypted the reader threads and the upload stream bring chunk c+1, and the download stream and the
// writer threads take chunk c-1 away.  A 12 MiB ciphertext is used once, so the host link is the wall of this program
// (PCIe: ~3.3 k ciphertexts/s at 12.6 MB each); the point of the chunks is that nothing else adds to it.
// One thread (the caller) drives the context; reader and writer threads only touch file descriptors and pinned memory.
struct RoundTimes {
    double setup = 0, round = 0;                  // ms: buffers + keys (once per process) | first read -> last byte written
    double last_upload = 0, last_download = 0;    // ms since the round started
    double read_busy = 0, write_busy = 0;         // ms summed over the reader / writer threads
    unsigned chunk = 0;
};

struct RoundPlan {
    size_t n_clients = 0, n_pre = 0;          // clients in `order`; the first n_pre carry a re-encryption key
    const std::vector<EnvelopeIndex> *idx = nullptr;
    const std::vector<AggItem> *items = nullptr;
    std::vector<const uint64_t *> evks;       // host, one [beta][2][D][N] per re-keyed client
    const std::vector<std::string> *evk_names = nullptr;  // their file names (what the cache compares)
    size_t evk_words = 0;
    unsigned threads = 4;
};

inline unsigned round_chunk(size_t B) {
    unsigned c = 2;
    if (const char *e = std::getenv("MKCKKS_ROUND_CHUNK")) c = (unsigned)std::max(1, std::atoi(e));
    return (unsigned)std::min<size_t>(c, B);
}

// What a server process keeps from one round to the next (serverRound --rounds): device arrays and pinned buffers grown
// on demand, the re-encryption keys that are resident in HBM (by file name, in order), and whether the kernels of the
// round's shape have been launched once.  A one-shot run simply has a cache that lives for one round.
class RoundCache {
public:
    explicit RoundCache(Session &s) : s_(s) {}
    ~RoundCache() {
        for (Slot *b : {&all, &sum, &out, &evk, &back, &back_evk})
            if (b->p) mkckks_dev_free(s_.ctx(), b->p);
    }
    RoundCache(const RoundCache &) = delete;
    RoundCache &operator=(const RoundCache &) = delete;
    struct Slot {
        uint64_t *p = nullptr;
        size_t words = 0;
    };
    uint64_t *grow(Slot &b, size_t words) {
        if (words > b.words) {
            Session::check(mkckks_sync(s_.ctx()));
            if (b.p) Session::check(mkckks_dev_free(s_.ctx(), b.p));
            b.p = nullptr;
            b.words = 0;
            void *p = nullptr;
            Session::check(mkckks_dev_alloc(s_.ctx(), words * 8, &p));
            b.p = static_cast<uint64_t *>(p);
            b.words = words;
        }
        return b.p;
    }
    PinnedRing &pinned(std::unique_ptr<PinnedRing> &r, size_t slot_bytes, unsigned slots) {
        if (!r || r->slot_bytes() < slot_bytes || r->count() < slots) {
            r.reset();
            r.reset(new PinnedRing(s_, slot_bytes, slots));
        }
        return *r;
    }
    Slot all, sum, out, evk, back, back_evk;
    std::vector<std::string> evk_names;  // keys resident in evk.p, in order
    std::unique_ptr<PinnedRing> ring, out_pin;
    std::string warm_shape;              // "<nl>/<chunk>/<n_pre>/<n_plain>/<rescale>" the kernels were launched with

private:
    Session &s_;
};

// aggregate of all clients' ciphertexts, scaled by 1/n, left in HBM ([B][2][meta.nl][N], for the --back leg) and written
// to `path` as an MKWS envelope -- the same bytes as the synchronous path (tests/test_cli_hosts.py)
inline AggResult run_round_pipeline(Session &s, const RoundPlan &plan, Json doc, const std::string &path, RoundCache &cache,
                                    RoundTimes &tm) {
    const double t_setup0 = now_ms();
    const std::vector<AggItem> &items = *plan.items;
    const std::vector<EnvelopeIndex> &idx = *plan.idx;
    const uint32_t N = s.N();
    const size_t B = items.size(), n_clients = plan.n_clients, n_pre = plan.n_pre, n_plain = n_clients - n_pre;
    // shape of the round from the first container's header; every other one must carry the same payload size
    BlobHeader h0;
    const BlobRef &blob0 = idx[0].blobs.at(blob_index(*items[0].blobs[0]));
    if (blob0.size < sizeof(BlobHeader)) throw std::runtime_error("ciphertext blob too short");
    pread_all(idx[0].fd, &h0, sizeof h0, blob0.offset);
    const Ciphertext first = meta_of(h0, s);
    const uint32_t nl = first.nl;
    const size_t words = (size_t)2 * nl * N, in_bytes = words * 8;
    const unsigned Bc = round_chunk(B);
    tm.chunk = Bc;
    // result shape (scale_aggregate's rules)
    AggResult agg;
    Ciphertext &res = agg.meta;
    const bool rescale = first.noise_deg == 2;
    if (rescale && nl < 2) throw std::runtime_error("ciphertext has no limb left to rescale");
    res.nl = rescale ? nl - 1 : nl;
    res.level = rescale ? first.level + 1 : first.level;
    res.scale = rescale ? first.scale / (double)s.moduli()[nl - 1] * s.sf(res.level, false) : first.scale * s.sf(first.level, false);
    res.noise_deg = rescale ? 2 : first.noise_deg + 1;
    res.slots = first.slots;
    const size_t owords = (size_t)2 * res.nl * N, out_bytes = owords * 8;
    // device: per chunk of cnt indices [re-keyed clients][slot for their sum][clients already in the domain] x [cnt], so that
    // every chunk is what mkckks_reencrypt_sum_batch / mkckks_eval_sum_batch take; sums and outputs [B]
    uint64_t *d_all = cache.grow(cache.all, (n_clients + 1) * B * words);
    uint64_t *d_sum = cache.grow(cache.sum, B * words);  // sums over all clients, [B]
    uint64_t *d_out = rescale ? cache.grow(cache.out, B * owords) : nullptr;
    uint64_t *d_evk = nullptr;
    if (n_pre) {  // keys that are already resident (same files, same order) are not uploaded again
        d_evk = cache.grow(cache.evk, n_pre * plan.evk_words);
        if (cache.evk_names != *plan.evk_names) {
            cache.evk_names.clear();
            for (size_t k = 0; k < n_pre; ++k)
                Session::check(mkckks_upload(s.ctx(), d_evk + k * plan.evk_words, plan.evks[k], plan.evk_words * 8));
            cache.evk_names = *plan.evk_names;
        }
    }
    std::unique_ptr<PinnedRing> &ring = cache.ring;
    cache.pinned(cache.ring, in_bytes, std::max(4u, std::min<unsigned>(16u, 2 * plan.threads)));
    PinnedRing &out_pin = cache.pinned(cache.out_pin, out_bytes, (unsigned)B);
    // output file: skeleton with "@<item>" fields, sized in advance, blobs written at fixed offsets
    for (size_t b = 0; b < B; ++b) {
        Json &lay = doc["weights_summary"].a[items[b].out_layer];
        Json ref("@" + std::to_string(b));
        if (items[b].field == 0) lay["mean"] = ref;
        else if (items[b].field == 1) lay["std_dev"] = ref;
        else lay["values"].a[items[b].idx] = ref;
    }
    {
        size_t k = 0;
        for_each_ct_field(doc, [&](Json &field) {
            if (field.as_string() != "@" + std::to_string(k)) throw std::logic_error("envelope fields out of item order");
            ++k;
        });
        if (k != B) throw std::logic_error("envelope holds ciphertext fields that are not aggregate items");
    }
    std::string head("MKWS", 4);
    {
        std::string text;
        doc.dump(text, 2);
        const uint32_t version = 1;
        const uint64_t skel_len = text.size(), n_blobs = B;
        head.append(reinterpret_cast<const char *>(&version), 4);
        head.append(reinterpret_cast<const char *>(&skel_len), 8);
        head += text;
        head.append(reinterpret_cast<const char *>(&n_blobs), 8);
    }
    const uint64_t blobs0 = head.size(), blob_size = sizeof(BlobHeader) + out_bytes;
    const BlobHeader out_hdr = header_of(res, N);
    const std::string shape = std::to_string(nl) + "/" + std::to_string(Bc) + "/" + std::to_string(n_pre) + "/" +
                              std::to_string(n_plain) + "/" + std::to_string((int)rescale);
    if (cache.warm_shape != shape) {   // process warm-up, the last part of the setup: copy streams, the workspace of one chunk and the first launch of every
        // kernel of the round (the HIP runtime resolves a kernel when it is first launched: ~30 ms for this path), on
        // whatever the fresh buffers hold -- none of the kernels addresses memory by data, and chunk 0 is overwritten below
        uint64_t ticket = 0;
        Session::check(mkckks_upload_async(s.ctx(), d_all, ring->slot(0), in_bytes, &ticket));  // full-size copies: the first large one of a direction costs ~8 ms
        Session::check(mkckks_fence_uploads(s.ctx()));
        const size_t cnt = std::min<size_t>(Bc, B);
        uint64_t *slot = d_all + n_pre * cnt * words;
        if (n_pre) Session::check(mkckks_reencrypt_sum_batch(s.ctx(), d_all, d_evk, n_plain ? slot : d_sum, (uint32_t)n_pre, (uint32_t)cnt, nl));
        if (n_plain)
            Session::check(mkckks_eval_sum_batch(s.ctx(), n_pre ? slot : slot + cnt * words, d_sum, (uint32_t)(n_plain + (n_pre ? 1 : 0)), (uint32_t)cnt, nl));
        if (rescale) Session::check(mkckks_rescale_mult_const_batch(s.ctx(), d_sum, d_out, (uint32_t)cnt, nl, 1.0 / (double)n_clients));
        else Session::check(mkckks_mult_const_batch(s.ctx(), d_sum, (uint32_t)cnt, nl, 1.0 / (double)n_clients));
        Session::check(mkckks_fence_compute(s.ctx()));
        Session::check(mkckks_download_async(s.ctx(), out_pin.slot(0), rescale ? d_out : d_sum, out_bytes, &ticket));
        Session::check(mkckks_copy_wait(s.ctx(), ticket));
        Session::check(mkckks_sync(s.ctx()));
        cache.warm_shape = shape;
    }
    tm.setup = now_ms() - t_setup0;

    const double t0 = now_ms();
    const int fd = ::open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) throw std::runtime_error("cannot write " + path);
    struct Closer {
        int fd;
        ~Closer() { ::close(fd); }
    } closer{fd};
    // The output goes through a shared mapping of the file: write() holds the inode lock, so parallel pwrite()s into ONE
    // tmpfs file run one after the other (~3 GB/s in total, measured); copies into the mapping do not.  The pages are
    // reserved first (fallocate, by the first writer thread, while the uploads run): no SIGBUS for a full file system later.
    const size_t file_bytes = blobs0 + B * (8 + blob_size);
    if (::ftruncate(fd, (off_t)file_bytes) != 0) throw std::runtime_error("cannot size " + path);
    char *const map = static_cast<char *>(::mmap(nullptr, file_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
    if (map == MAP_FAILED) throw std::runtime_error("cannot map " + path);
    struct Unmapper {
        char *p;
        size_t n;
        ~Unmapper() { ::munmap(p, n); }
    } unmapper{map, file_bytes};

    // jobs in chunk order: chunk, client, index within the chunk
    struct Job {
        int fd;
        BlobRef blob;
        uint64_t *d_dst;
        BlobHeader hdr;
        unsigned chunk;
        double t_read0 = 0, t_read1 = 0, t_enq = 0, t_done = 0;  // MKCKKS_IO_TRACE
    };
    std::vector<Job> jobs;
    jobs.reserve(n_clients * B);
    std::vector<size_t> chunk_b0;
    for (size_t b0 = 0; b0 < B; b0 += Bc) chunk_b0.push_back(b0);
    const size_t n_chunks = chunk_b0.size();
    auto chunk_cnt = [&](size_t c) { return std::min<size_t>(Bc, B - chunk_b0[c]); };
    auto chunk_base = [&](size_t c) { return d_all + (n_clients + 1) * chunk_b0[c] * words; };
    for (size_t c = 0; c < n_chunks; ++c) {
        const size_t cnt = chunk_cnt(c);
        for (size_t k = 0; k < n_clients; ++k)
            for (size_t i = 0; i < cnt; ++i) {
                const size_t b = chunk_b0[c] + i, bi = blob_index(*items[b].blobs[k]);
                const size_t slot_k = k < n_pre ? k : k + 1;  // position n_pre is the sum slot
                jobs.push_back(Job{idx[k].fd, idx[k].blobs.at(bi), chunk_base(c) + (slot_k * cnt + i) * words, BlobHeader{}, (unsigned)c, 0, 0, 0, 0});
            }
    }

    std::mutex m;
    std::condition_variable cv_free, cv_event, cv_write;
    std::deque<unsigned> free_slots;
    std::deque<std::pair<unsigned, size_t>> filled;  // (slot, job)
    std::deque<size_t> to_write;                      // (output index b, part) of a downloaded ciphertext
    for (unsigned i = 0; i < ring->count(); ++i) free_slots.push_back(i);
    std::atomic<size_t> next{0};
    std::exception_ptr err;
    bool stop = false, writes_closed = false;
    double read_busy = 0, write_busy = 0;
    auto fail_from_thread = [&] {
        std::lock_guard<std::mutex> g(m);
        if (!err) err = std::current_exception();
        stop = true;
        cv_event.notify_all();
        cv_free.notify_all();
        cv_write.notify_all();
    };
    auto reader = [&] {
        double busy = 0;
        try {
            for (size_t j; (j = next.fetch_add(1)) < jobs.size();) {
                unsigned slot;
                {
                    std::unique_lock<std::mutex> g(m);
                    cv_free.wait(g, [&] { return stop || !free_slots.empty(); });
                    if (stop) return;
                    slot = free_slots.front();
                    free_slots.pop_front();
                }
                const double tb = now_ms();
                Job &job = jobs[j];
                if (job.blob.size != sizeof(BlobHeader) + in_bytes) throw std::runtime_error("ciphertext blob has the wrong size");
                pread_all(job.fd, &job.hdr, sizeof(BlobHeader), job.blob.offset);
                pread_all(job.fd, ring->slot(slot), in_bytes, job.blob.offset + sizeof(BlobHeader));
                busy += now_ms() - tb;
                job.t_read0 = tb - t0;
                job.t_read1 = now_ms() - t0;
                {
                    std::lock_guard<std::mutex> g(m);
                    filled.emplace_back(slot, j);
                }
                cv_event.notify_one();
            }
        } catch (...) {
            fail_from_thread();
        }
        std::lock_guard<std::mutex> g(m);
        read_busy += busy;
    };
    // a downloaded ciphertext is written in WRITE_PARTS pieces by different threads: the last chunk's write is the tail of
    // the round.  The first writer reserves the file's pages (fallocate: tmpfs allocates and clears them) while the uploads
    // run, so that the writes proper are plain copies.
    constexpr unsigned WRITE_PARTS = 4;
    std::atomic<bool> reserve_taken{false};
    bool reserved = false;
    auto writer = [&] {
        double busy = 0;
        try {
            if (!reserve_taken.exchange(true)) {
                const int rc = ::posix_fallocate(fd, 0, (off_t)file_bytes);
                if (rc != 0) throw std::runtime_error("cannot reserve " + path + ": " + std::strerror(rc));
                std::memcpy(map, head.data(), head.size());
                {
                    std::lock_guard<std::mutex> g(m);
                    reserved = true;
                }
                cv_write.notify_all();
            }
            for (;;) {
                size_t w;
                {
                    std::unique_lock<std::mutex> g(m);
                    cv_write.wait(g, [&] { return stop || (reserved && (writes_closed || !to_write.empty())); });
                    if (stop) return;
                    if (to_write.empty()) break;  // closed and drained
                    w = to_write.front();
                    to_write.pop_front();
                }
                const double tb = now_ms();
                const size_t b = w / WRITE_PARTS, part = w % WRITE_PARTS;
                char *at = map + blobs0 + b * (8 + blob_size);
                if (part == 0) {
                    std::memcpy(at, &blob_size, 8);
                    std::memcpy(at + 8, &out_hdr, sizeof out_hdr);
                }
                const size_t lo = out_bytes * part / WRITE_PARTS, hi = out_bytes * (part + 1) / WRITE_PARTS;
                std::memcpy(at + 8 + sizeof out_hdr + lo, out_pin.slot((unsigned)b) + lo, hi - lo);
                busy += now_ms() - tb;
            }
        } catch (...) {
            fail_from_thread();
        }
        std::lock_guard<std::mutex> g(m);
        write_busy += busy;
    };
    std::vector<std::thread> pool;
    const unsigned n_readers = (unsigned)std::min<size_t>(std::max(1u, plan.threads), jobs.size());
    const unsigned n_writers = std::max(1u, std::min(4u, plan.threads));
    for (unsigned t = 0; t < n_readers; ++t) pool.emplace_back(reader);
    for (unsigned t = 0; t < n_writers; ++t) pool.emplace_back(writer);

    std::exception_ptr main_err;
    try {
        std::deque<std::pair<uint64_t, unsigned>> up_flight;   // (ticket, ring slot)
        std::deque<size_t> up_jobs;                            // jobs of up_flight, for the trace
        std::vector<std::pair<double, double>> chunk_enq(n_chunks);
        double call_ms[6] = {0, 0, 0, 0, 0, 0};  // chunk 0's calls, for the trace
        std::deque<std::pair<uint64_t, size_t>> down_flight;   // (ticket, output index)
        std::vector<size_t> enq(n_chunks, 0);
        size_t uploaded = 0, computed = 0, downloaded = 0;
        const double operand = 1.0 / (double)n_clients;
        while (downloaded < B) {
            bool progress = false;
            std::pair<unsigned, size_t> got{0, 0};
            bool have = false;
            {
                std::unique_lock<std::mutex> g(m);
                if (stop) break;
                if (!filled.empty()) {
                    got = filled.front();
                    filled.pop_front();
                    have = true;
                }
            }
            if (have) {
                uint64_t ticket = 0;
                Session::check(mkckks_upload_async(s.ctx(), jobs[got.second].d_dst, ring->slot(got.first), in_bytes, &ticket));
                up_flight.emplace_back(ticket, got.first);
                up_jobs.push_back(got.second);
                jobs[got.second].t_enq = now_ms() - t0;
                ++enq[jobs[got.second].chunk];
                ++uploaded;
                progress = true;
            }
            while (!up_flight.empty()) {  // uploads complete in ticket order
                int done = 0;
                Session::check(mkckks_copy_done(s.ctx(), up_flight.front().first, &done));
                if (!done) break;
                {
                    std::lock_guard<std::mutex> g(m);
                    free_slots.push_back(up_flight.front().second);
                }
                cv_free.notify_one();
                up_flight.pop_front();
                jobs[up_jobs.front()].t_done = now_ms() - t0;
                up_jobs.pop_front();
                progress = true;
                if (up_flight.empty() && uploaded == jobs.size()) tm.last_upload = now_ms() - t0;
            }
            // the next chunk in order whose uploads are all enqueued: everything below is asynchronous
            if (computed < n_chunks && enq[computed] == n_clients * chunk_cnt(computed)) {
                const size_t c = computed, cnt = chunk_cnt(c), b0 = chunk_b0[c];
                chunk_enq[c].first = now_ms() - t0;
                uint64_t *base = chunk_base(c), *slot = base + n_pre * cnt * words, *sum = d_sum + b0 * words;
                double tc = now_ms();
                auto lap = [&](int i) {
                    if (c == 0) call_ms[i] = now_ms() - tc;
                    tc = now_ms();
                };
                Session::check(mkckks_fence_uploads(s.ctx()));
                lap(0);
                if (n_pre)  // with clients already in the domain the re-encrypted sum is one more term of their EvalAdd
                    Session::check(mkckks_reencrypt_sum_batch(s.ctx(), base, d_evk, n_plain ? slot : sum, (uint32_t)n_pre, (uint32_t)cnt, nl));
                if (n_plain) {
                    const uint64_t *terms = n_pre ? slot : slot + cnt * words;
                    Session::check(mkckks_eval_sum_batch(s.ctx(), terms, sum, (uint32_t)(n_plain + (n_pre ? 1 : 0)), (uint32_t)cnt, nl));
                }
                lap(1);
                uint64_t *outp;
                if (rescale) {
                    outp = d_out + b0 * owords;
                    Session::check(mkckks_rescale_mult_const_batch(s.ctx(), sum, outp, (uint32_t)cnt, nl, operand));
                } else {
                    Session::check(mkckks_mult_const_batch(s.ctx(), sum, (uint32_t)cnt, nl, operand));
                    outp = sum;
                }
                lap(2);
                Session::check(mkckks_fence_compute(s.ctx()));
                lap(3);
                for (size_t i = 0; i < cnt; ++i) {
                    uint64_t ticket = 0;
                    Session::check(mkckks_download_async(s.ctx(), out_pin.slot((unsigned)(b0 + i)), outp + i * owords, out_bytes, &ticket));
                    down_flight.emplace_back(ticket, b0 + i);
                }
                lap(4);
                chunk_enq[c].second = now_ms() - t0;
                ++computed;
                progress = true;
            }
            while (!down_flight.empty()) {
                int done = 0;
                Session::check(mkckks_copy_done(s.ctx(), down_flight.front().first, &done));
                if (!done) break;
                {
                    std::lock_guard<std::mutex> g(m);
                    for (unsigned part = 0; part < WRITE_PARTS; ++part) to_write.push_back(down_flight.front().second * WRITE_PARTS + part);
                }
                cv_write.notify_all();
                down_flight.pop_front();
                ++downloaded;
                progress = true;
                if (downloaded == B) tm.last_download = now_ms() - t0;
            }
            if (!progress) {
                std::unique_lock<std::mutex> g(m);
                if (filled.empty() && !stop) cv_event.wait_for(g, std::chrono::microseconds(40));
            }
        }
        bool stopped;
        {
            std::lock_guard<std::mutex> g(m);
            stopped = stop;
        }
        if (const char *tr = std::getenv("MKCKKS_IO_TRACE")) {
            if (FILE *f = std::fopen(tr, "w")) {
                std::fprintf(f, "# job chunk read_start read_end upload_enqueued upload_seen_done (ms since the round started)\n");
                for (size_t j = 0; j < jobs.size(); ++j)
                    std::fprintf(f, "%zu %u %.3f %.3f %.3f %.3f\n", j, jobs[j].chunk, jobs[j].t_read0, jobs[j].t_read1, jobs[j].t_enq, jobs[j].t_done);
                std::fprintf(f, "# chunk compute_enqueue_start compute_enqueue_end\n");
                for (size_t c = 0; c < n_chunks; ++c) std::fprintf(f, "c%zu %.3f %.3f\n", c, chunk_enq[c].first, chunk_enq[c].second);
                std::fprintf(f, "# chunk 0 calls: fence_uploads %.3f, reencrypt_sum(+eval_sum) %.3f, scale %.3f, fence_compute %.3f, download_async %.3f ms\n",
                             call_ms[0], call_ms[1], call_ms[2], call_ms[3], call_ms[4]);
                std::fclose(f);
            }
        }
        // headers, then the residues of every input (the kernels above do no data-dependent addressing; a bad residue only
        // makes bad numbers, and the output is withdrawn below before anyone can read it)
        for (size_t j = 0; j < jobs.size() && !stopped; ++j) {
            const Ciphertext mj = meta_of(jobs[j].hdr, s);
            if (mj.nl != first.nl || mj.noise_deg != first.noise_deg || mj.scale != first.scale)
                throw std::runtime_error("EvalAdd operands differ in level or scale");
        }
        uint64_t bad_total = 0;
        for (size_t c = 0; c < n_chunks && !stopped; ++c) {
            const size_t cnt = chunk_cnt(c);
            uint64_t bad = 0;
            if (n_pre) {
                Session::check(mkckks_count_noncanonical(s.ctx(), chunk_base(c), (uint32_t)(n_pre * cnt), nl, &bad));
                bad_total += bad;
            }
            if (n_plain) {
                Session::check(mkckks_count_noncanonical(s.ctx(), chunk_base(c) + (n_pre + 1) * cnt * words, (uint32_t)(n_plain * cnt), nl, &bad));
                bad_total += bad;
            }
        }
        if (bad_total) throw std::runtime_error("ciphertext: residue not below its modulus");
    } catch (...) {
        main_err = std::current_exception();
    }
    {
        std::lock_guard<std::mutex> g(m);
        if (main_err) stop = true;
        writes_closed = true;
    }
    cv_free.notify_all();
    cv_write.notify_all();
    for (std::thread &t : pool) t.join();
    if (main_err || err) {
        mkckks_sync(s.ctx());       // nothing of this round may still be running when the buffers go away
        ::unlink(path.c_str());
        std::rethrow_exception(main_err ? main_err : err);
    }
    tm.round = now_ms() - t0;
    tm.read_busy = read_busy;
    tm.write_busy = write_busy;
    agg.d_out = rescale ? d_out : d_sum;  // the aggregate as one array for the --back leg
    return agg;
}

}  // namespace mkh
