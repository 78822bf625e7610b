// hostlib.hpp -- shared plumbing of the CLI hosts: CryptoContext file, key files, ciphertext blobs, the
// weights_summary envelope, and a Session that owns the device context (C-ABI: include/mkckks.h).
//
// What it replaces in the reference (paths relative to /root/reference):
//   Serial::DeserializeFromFile(cc_path, cc, SerType::JSON)          changeCipherDomain.cpp:32-36 (and every main)
//   Serial::{De,}Serialize(ct, ss, SerType::BINARY) + Base64          changeCipherDomain.cpp:69-78,
//                                                                     aggregateEncryptedWeights.cpp:18-30
//   Serial::{SerializeTo,DeserializeFrom}File(key, SerType::JSON)     keyGen.cpp:41-48, REkeyGen.cpp:37-60
// Blob formats are this project's own (header + raw limb-major residues): OpenFHE's cereal encodings cannot be
// tested offline (all ciphertext/key blobs are in .MISSING_LARGE_BLOBS) -- SURVEY.md 8b "Serialization note".
// A CC.json written by OpenFHE IS accepted: its parameters are read and the moduli re-derived and checked.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mkckks.h"
#include "base64.hpp"
#include "json.hpp"
#include "sampler.hpp"

namespace mkh {

// ------------------------------------------------------------------------------------------------
// CryptoContext description (what genCC writes and every other program reads)
// ------------------------------------------------------------------------------------------------
struct CcFile {
    mkckks_params p{};
    uint32_t batch = 0;
    std::string pre_mode = "INDCPA";
    std::vector<uint64_t> moduli;  // as listed in the file (Q limbs), for cross-checking
};

// [upstream] lattice/stdlatticeparms: HEStd_128_classic maximum log2(QP) per ring dimension
inline uint32_t ring_dim_for_security(uint32_t log_qp) {
    static const struct { uint32_t n, max_bits; } tbl[] = {{1024, 27}, {2048, 54}, {4096, 109}, {8192, 218},
                                                           {16384, 438}, {32768, 881}, {65536, 1747}, {131072, 3523}};
    for (auto &e : tbl)
        if (log_qp <= e.max_bits) return e.n;
    throw std::runtime_error("parameters exceed the largest supported ring dimension (2^17)");
}

inline uint32_t ilog2(uint64_t v) { return 63 - (uint32_t)__builtin_clzll(v); }
inline uint32_t bit_len(uint64_t v) { return 64 - (uint32_t)__builtin_clzll(v); }

// default number of large digits: [upstream] ComputeNumLargeDigits
inline uint32_t default_dnum(uint32_t depth) { return depth > 3 ? 3 : (depth > 0 ? 2 : 1); }

inline CcFile read_cc(const std::string &path) {
    Json j = Json::parse_file(path);
    CcFile cc;
    if (j.contains("mkckks_cc")) {
        const Json &c = j.at("mkckks_cc");
        cc.p.log_n = (uint32_t)c.at("log_n").as_int();
        cc.p.mult_depth = (uint32_t)c.at("MultiplicativeDepth").as_int();
        cc.p.scaling_bits = (uint32_t)c.at("ScalingModSize").as_int();
        cc.p.first_bits = (uint32_t)c.at("FirstModSize").as_int();
        cc.p.dnum = (uint32_t)c.at("NumLargeDigits").as_int();
        cc.p.aux_bits = (uint32_t)c.at("AuxBits").as_int();
        cc.p.extra_bits = (uint32_t)c.at("ExtraBits").as_int();
        cc.batch = (uint32_t)c.at("BatchSize").as_int();
        cc.pre_mode = c.at("PREMode").as_string();
        for (const Json &m : c.at("moduli").a) cc.moduli.push_back(m.as_u64());
        return cc;
    }
    // OpenFHE cereal JSON (server/storage/CC.json): CryptoParametersCKKSRNS nested three levels deep
    const Json &d = j.at("value0").at("ptr_wrapper").at("data").at("cc").at("ptr_wrapper").at("data");
    const Json &rns = d.at("value0");
    const Json &rlwe = rns.at("value0");
    const Json &base = rlwe.at("value0");
    const Json &elp = base.at("elp").at("ptr_wrapper").at("data");
    const Json &enp = base.at("enp").at("ptr_wrapper").at("data");
    if (rns.at("ks").as_int() != 2) throw std::runtime_error("only HYBRID key switching (ks=2) is supported");
    if (rns.at("rs").as_int() != 3) throw std::runtime_error("only FLEXIBLEAUTOEXT scaling (rs=3) is supported");
    for (const Json &lim : elp.at("p").a)
        cc.moduli.push_back(lim.at("ptr_wrapper").at("data").at("value0").at("cm").at("v").as_u64());
    if (cc.moduli.size() < 3) throw std::runtime_error("CC.json: too few RNS limbs");
    cc.p.log_n = ilog2((uint64_t)elp.at("value0").at("rd").as_int());
    cc.p.mult_depth = (uint32_t)cc.moduli.size() - 2;
    cc.p.scaling_bits = (uint32_t)enp.at("m").as_int();
    cc.p.first_bits = bit_len(cc.moduli[0]);
    cc.p.dnum = (uint32_t)rns.at("dnum").as_int();
    cc.p.aux_bits = (uint32_t)rns.at("ab").as_int();
    cc.p.extra_bits = (uint32_t)rns.at("eb").as_int();
    cc.batch = (uint32_t)enp.at("bs").as_int();
    return cc;
}

// ------------------------------------------------------------------------------------------------
// binary containers
// ------------------------------------------------------------------------------------------------
struct BlobHeader {
    char magic[4];       // "MKCK"
    uint32_t version;    // 1
    uint32_t kind;       // 1 ciphertext, 2 public key, 3 secret key, 4 re-encryption key
    uint32_t ring_dim;
    uint32_t limbs;      // ciphertext: nl; keys: D
    uint32_t parts;      // ciphertext: 2; pk: 2; sk: 1; rekey: 2*beta
    uint32_t level;      // ciphertext: #dropped limbs
    uint32_t noise_deg;  // ciphertext: noiseScaleDeg
    double scale;        // ciphertext: scaling factor
    uint32_t slots;
    uint32_t reserved;
};
static_assert(sizeof(BlobHeader) == 48, "blob header layout");

enum : uint32_t { KIND_CT = 1, KIND_PK = 2, KIND_SK = 3, KIND_RK = 4 };

struct Ciphertext {
    uint32_t nl = 0, level = 0, noise_deg = 0, slots = 0;
    double scale = 0;
    std::vector<uint64_t> data;  // [2][nl][N]
};

// Binary envelope: when set, ciphertext fields hold the raw container bytes in memory (no
// base64) and write_envelope() stores them behind the JSON skeleton; see read_envelope / write_envelope below.
inline bool &raw_blobs() {
    static bool on = false;
    return on;
}

inline std::string encode_ct(const Ciphertext &ct, uint32_t ring_dim) {
    BlobHeader h{};
    std::memcpy(h.magic, "MKCK", 4);
    h.version = 1; h.kind = KIND_CT; h.ring_dim = ring_dim; h.limbs = ct.nl; h.parts = 2;
    h.level = ct.level; h.noise_deg = ct.noise_deg; h.scale = ct.scale; h.slots = ct.slots;
    std::string bin(sizeof h + ct.data.size() * 8, '\0');
    std::memcpy(&bin[0], &h, sizeof h);
    std::memcpy(&bin[sizeof h], ct.data.data(), ct.data.size() * 8);
    return raw_blobs() ? bin : Base64Encode(bin);
}

inline Ciphertext decode_ct(const std::string &b64, uint32_t ring_dim) {
    // a raw container starts with the magic itself; its base64 form starts with "TUtD"
    const bool raw = b64.size() >= 4 && !std::memcmp(b64.data(), "MKCK", 4);
    const std::string decoded = raw ? std::string() : Base64Decode(b64);
    const std::string &bin = raw ? b64 : decoded;
    if (bin.size() < sizeof(BlobHeader)) throw std::runtime_error("ciphertext blob too short");
    BlobHeader h;
    std::memcpy(&h, bin.data(), sizeof h);
    if (std::memcmp(h.magic, "MKCK", 4) || h.version != 1 || h.kind != KIND_CT)
        throw std::runtime_error("not a mkckks ciphertext blob");
    if (h.ring_dim != ring_dim || h.parts != 2) throw std::runtime_error("ciphertext does not match the CryptoContext");
    Ciphertext ct;
    ct.nl = h.limbs; ct.level = h.level; ct.noise_deg = h.noise_deg; ct.scale = h.scale; ct.slots = h.slots;
    const size_t words = (size_t)2 * h.limbs * ring_dim;
    if (bin.size() != sizeof h + words * 8) throw std::runtime_error("ciphertext blob has the wrong size");
    ct.data.resize(words);
    std::memcpy(ct.data.data(), bin.data() + sizeof h, words * 8);
    return ct;
}

inline void write_key_file(const std::string &path, uint32_t kind, uint32_t ring_dim, uint32_t limbs, uint32_t parts,
                           const std::vector<uint64_t> &data, const std::vector<int8_t> *ternary = nullptr) {
    BlobHeader h{};
    std::memcpy(h.magic, "MKCK", 4);
    h.version = 1; h.kind = kind; h.ring_dim = ring_dim; h.limbs = limbs; h.parts = parts;
    h.reserved = ternary ? 1 : 0;
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    std::fwrite(&h, sizeof h, 1, f);
    std::fwrite(data.data(), 8, data.size(), f);
    if (ternary) std::fwrite(ternary->data(), 1, ternary->size(), f);
    std::fclose(f);
}

inline bool read_key_file(const std::string &path, uint32_t kind, uint32_t ring_dim, uint32_t limbs, uint32_t parts,
                          std::vector<uint64_t> &data, std::vector<int8_t> *ternary = nullptr) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    BlobHeader h;
    bool ok = std::fread(&h, sizeof h, 1, f) == 1 && !std::memcmp(h.magic, "MKCK", 4) && h.version == 1 &&
              h.kind == kind && h.ring_dim == ring_dim && h.limbs == limbs && h.parts == parts;
    if (ok) {
        data.resize((size_t)parts * limbs * ring_dim);
        ok = std::fread(data.data(), 8, data.size(), f) == data.size();
    }
    if (ok && ternary) {
        ternary->resize(ring_dim);
        ok = h.reserved == 1 && std::fread(ternary->data(), 1, ring_dim, f) == ring_dim;
    }
    std::fclose(f);
    return ok;
}

// ------------------------------------------------------------------------------------------------
// Session: CryptoContext on the device
// ------------------------------------------------------------------------------------------------
class Session {
public:
    explicit Session(const CcFile &cc) : cc_(cc) {
        mkckks_params p = cc.p;
        p.device = 0;
        if (const char *e = std::getenv("MKCKKS_DEVICE")) p.device = std::atoi(e);
        check(mkckks_ctx_create(&p, &ctx_));
        check(mkckks_ctx_info(ctx_, &info_));
        moduli_.resize(info_.num_q + info_.num_p);
        check(mkckks_ctx_moduli(ctx_, moduli_.data()));
        for (size_t i = 0; i < cc.moduli.size() && i < info_.num_q; ++i)
            if (cc.moduli[i] != moduli_[i])
                throw std::runtime_error("CryptoContext file lists moduli that differ from the derived ones");
    }
    ~Session() {
        for (void *p : bufs_) mkckks_dev_free(ctx_, p);
        if (ctx_) mkckks_ctx_destroy(ctx_);
    }
    Session(const Session &) = delete;
    Session &operator=(const Session &) = delete;

    static void check(int rc) {
        if (rc != MKCKKS_OK) throw std::runtime_error(std::string("mkckks: ") + mkckks_last_error());
    }
    mkckks_ctx *ctx() { return ctx_; }
    uint32_t N() const { return info_.ring_dim; }
    uint32_t L() const { return info_.num_q; }
    uint32_t K() const { return info_.num_p; }
    uint32_t D() const { return info_.num_q + info_.num_p; }
    uint32_t beta() const { return info_.beta; }
    uint32_t slots() const { return info_.slots; }
    uint32_t batch() const { return cc_.batch ? cc_.batch : info_.slots; }
    const std::vector<uint64_t> &moduli() const { return moduli_; }
    double sf(uint32_t level, bool big) const {
        double v = 0;
        check(mkckks_scaling_factor(ctx_, level, big ? 1 : 0, &v));
        return v;
    }

    template <typename T>
    T *alloc(size_t count) {
        void *p = nullptr;
        check(mkckks_dev_alloc(ctx_, count * sizeof(T), &p));
        bufs_.push_back(p);
        return static_cast<T *>(p);
    }
    template <typename T>
    T *to_device(const T *h, size_t count) {
        T *d = alloc<T>(count);
        check(mkckks_upload(ctx_, d, h, count * sizeof(T)));
        return d;
    }
    template <typename T>
    void to_host(T *h, const T *d, size_t count) { check(mkckks_download(ctx_, h, d, count * sizeof(T))); }

private:
    CcFile cc_;
    mkckks_ctx *ctx_ = nullptr;
    mkckks_info info_{};
    std::vector<uint64_t> moduli_;
    std::vector<void *> bufs_;
};

// ------------------------------------------------------------------------------------------------
// Secret key: this project's container, or a private key written by OpenFHE's
// Serial::SerializeToFile(path, kp.secretKey, SerType::JSON) (keyGen.cpp:45; fixture client_1-private.key:
// value0.ptr_wrapper.data.s = DCRTPoly{v:[{v:{ptr_wrapper:{data:{v:[residues], m:{v:modulus}}}}, f:format}], f}).
// Import: the Q limbs are checked against the context, limb 0 is brought to COEFFICIENT form on the device, the
// coefficients must be ternary and every other limb must agree with them; the key over all D = L + K limbs is then
// rebuilt from the ternary polynomial (OpenFHE stores s over Q only, hybrid re-key generation needs it over QP).
// ------------------------------------------------------------------------------------------------
inline bool import_openfhe_secret_key(Session &s, const std::string &path, std::vector<uint64_t> &sk,
                                      std::vector<int8_t> &sk_t) {
    Json j;
    try {
        j = Json::parse_file(path);
    } catch (const std::exception &) {
        return false;
    }
    if (!j.contains("value0")) return false;
    const Json &poly = j.at("value0").at("ptr_wrapper").at("data").at("s");
    const uint32_t N = s.N(), L = s.L(), D = s.D();
    const std::vector<Json> &limbs = poly.at("v").a;
    if (limbs.size() != L) throw std::runtime_error("private key: limb count differs from the CryptoContext");
    const bool eval = limbs[0].at("f").as_int() == 0;
    std::vector<uint64_t> res((size_t)L * N);
    for (uint32_t i = 0; i < L; ++i) {
        const Json &dat = limbs[i].at("v").at("ptr_wrapper").at("data");
        const Json &mod = dat.at("m").at("v");
        const uint64_t m = mod.kind == Json::Str ? std::stoull(mod.as_string()) : mod.as_u64();
        if (m != s.moduli()[i]) throw std::runtime_error("private key: modulus differs from the CryptoContext");
        if ((limbs[i].at("f").as_int() == 0) != eval) throw std::runtime_error("private key: mixed limb formats");
        const std::vector<Json> &v = dat.at("v").a;
        if (v.size() != N) throw std::runtime_error("private key: ring dimension differs from the CryptoContext");
        for (uint32_t k = 0; k < N; ++k) {
            const uint64_t r = v[k].kind == Json::Str ? std::stoull(v[k].as_string()) : v[k].as_u64();
            if (r >= m) throw std::runtime_error("private key: residue out of range");
            res[(size_t)i * N + k] = r;
        }
    }
    uint64_t *d_res = s.to_device(res.data(), res.size());
    if (eval) Session::check(mkckks_ntt_inverse_batch(s.ctx(), d_res, 1, L, 0));
    s.to_host(res.data(), d_res, res.size());
    sk_t.resize(N);
    for (uint32_t k = 0; k < N; ++k) {
        const uint64_t c = res[k], q0 = s.moduli()[0];
        if (c > 1 && c != q0 - 1) throw std::runtime_error("private key: not a ternary secret");
        sk_t[k] = c == 0 ? 0 : (c == 1 ? 1 : -1);
        for (uint32_t i = 1; i < L; ++i) {
            const uint64_t want = sk_t[k] == 0 ? 0 : (sk_t[k] == 1 ? 1 : s.moduli()[i] - 1);
            if (res[(size_t)i * N + k] != want) throw std::runtime_error("private key: limbs disagree");
        }
    }
    // NTT of the ternary polynomial over all D limbs: KeyGen's s-path with a = 0, e = 0 (the pk half is discarded)
    std::vector<uint64_t> zeros_a((size_t)D * N, 0);
    std::vector<int32_t> zeros_e(N, 0);
    uint64_t *d_pk = s.alloc<uint64_t>((size_t)2 * D * N), *d_sk = s.alloc<uint64_t>((size_t)D * N);
    Session::check(mkckks_keygen(s.ctx(), s.to_device(sk_t.data(), N), s.to_device(zeros_a.data(), zeros_a.size()),
                                 s.to_device(zeros_e.data(), N), d_pk, d_sk));
    sk.resize((size_t)D * N);
    s.to_host(sk.data(), d_sk, sk.size());
    if (eval)  // the rebuilt Q limbs must reproduce the file bit for bit
        for (uint32_t i = 0; i < L; ++i) {
            const Json &dat = limbs[i].at("v").at("ptr_wrapper").at("data");
            const std::vector<Json> &v = dat.at("v").a;
            for (uint32_t k = 0; k < N; k += 97) {
                const uint64_t r = v[k].kind == Json::Str ? std::stoull(v[k].as_string()) : v[k].as_u64();
                if (sk[(size_t)i * N + k] != r) throw std::runtime_error("private key: transform convention mismatch");
            }
        }
    return true;
}

// ------------------------------------------------------------------------------------------------
// Public keys and re-encryption keys written by OpenFHE's JSON serialiser (keyGen.cpp:41,
// REkeyGen.cpp:60: Serial::SerializeToFile(path, key, SerType::JSON)).  The reference ships no such file (its key blobs
// are in .MISSING_LARGE_BLOBS), so the layout below is the private-key fixture's nesting applied to the other key
// classes -- PARITY UNPINNED, stated in INTEGRATION.md:
//   public key      value0.ptr_wrapper.data.h = [ DCRTPoly b, DCRTPoly a ]                    (PublicKeyImpl::m_h)
//   re-encryption   value0.ptr_wrapper.data.k = [ [ a_0 .. a_{beta-1} ], [ b_0 .. b_{beta-1} ] ]  (EvalKeyRelinImpl::m_rKey:
//                   A vector then B vector; EvalFastKeySwitchCoreExt multiplies the digits with B for c0 and A for c1)
// Every DCRTPoly is { v: [ { v: { ptr_wrapper: { data: { v: [residues], m: { v: modulus } } } }, f: format } ... ], f }
// over the D = L + K limbs of Q P in the context's order; format 0 = EVALUATION, otherwise the limbs are transformed on
// the device.  Moduli and residue ranges are checked against the context.
// ------------------------------------------------------------------------------------------------
inline void import_openfhe_dcrtpoly(Session &s, const Json &poly, uint64_t *dst /* [D][N] */) {
    const uint32_t N = s.N(), D = s.D();
    const std::vector<Json> &limbs = poly.at("v").a;
    if (limbs.size() != D) throw std::runtime_error("key: element is not over the D = L + K limbs of Q P");
    const bool eval = limbs[0].at("f").as_int() == 0;
    for (uint32_t i = 0; i < D; ++i) {
        const Json &dat = limbs[i].at("v").at("ptr_wrapper").at("data");
        const Json &mod = dat.at("m").at("v");
        const uint64_t m = mod.kind == Json::Str ? std::stoull(mod.as_string()) : mod.as_u64();
        if (m != s.moduli()[i]) throw std::runtime_error("key: modulus differs from the CryptoContext");
        if ((limbs[i].at("f").as_int() == 0) != eval) throw std::runtime_error("key: mixed limb formats");
        const std::vector<Json> &v = dat.at("v").a;
        if (v.size() != N) throw std::runtime_error("key: ring dimension differs from the CryptoContext");
        for (uint32_t k = 0; k < N; ++k) {
            const uint64_t r = v[k].kind == Json::Str ? std::stoull(v[k].as_string()) : v[k].as_u64();
            if (r >= m) throw std::runtime_error("key: residue out of range");
            dst[(size_t)i * N + k] = r;
        }
    }
    if (!eval) {  // COEFFICIENT format on file: bring to EVALUATION on the device
        uint64_t *d = s.to_device(dst, (size_t)D * N);
        Session::check(mkckks_ntt_forward_batch(s.ctx(), d, 1, s.L(), 1));
        s.to_host(dst, d, (size_t)D * N);
    }
}

inline bool import_openfhe_public_key(Session &s, const std::string &path, std::vector<uint64_t> &pk) {
    Json j;
    try {
        j = Json::parse_file(path);
    } catch (const std::exception &) {
        return false;
    }
    if (!j.contains("value0")) return false;
    const Json &data = j.at("value0").at("ptr_wrapper").at("data");
    if (!data.contains("h")) return false;
    const std::vector<Json> &h = data.at("h").a;
    if (h.size() != 2) throw std::runtime_error("public key: expected the two elements (b, a)");
    const size_t poly = (size_t)s.D() * s.N();
    pk.resize(2 * poly);
    import_openfhe_dcrtpoly(s, h[0], &pk[0]);
    import_openfhe_dcrtpoly(s, h[1], &pk[poly]);
    return true;
}

inline bool import_openfhe_eval_key(Session &s, const std::string &path, std::vector<uint64_t> &evk) {
    Json j;
    try {
        j = Json::parse_file(path);
    } catch (const std::exception &) {
        return false;
    }
    if (!j.contains("value0")) return false;
    const Json &data = j.at("value0").at("ptr_wrapper").at("data");
    if (!data.contains("k")) return false;
    const std::vector<Json> &k = data.at("k").a;
    const uint32_t beta = s.beta();
    if (k.size() != 2 || k[0].a.size() != beta || k[1].a.size() != beta)
        throw std::runtime_error("re-encryption key: expected the A and B vectors with one element per digit");
    const size_t poly = (size_t)s.D() * s.N();
    evk.resize((size_t)beta * 2 * poly);  // [digit][b then a][D][N]
    for (uint32_t d = 0; d < beta; ++d) {
        import_openfhe_dcrtpoly(s, k[1].a[d], &evk[((size_t)d * 2 + 0) * poly]);  // B vector: multiplies into c0
        import_openfhe_dcrtpoly(s, k[0].a[d], &evk[((size_t)d * 2 + 1) * poly]);  // A vector: multiplies into c1
    }
    return true;
}

// this project's container first, OpenFHE's JSON second
inline bool load_public_key(Session &s, const std::string &path, std::vector<uint64_t> &pk) {
    if (read_key_file(path, KIND_PK, s.N(), s.D(), 2, pk)) return true;
    return import_openfhe_public_key(s, path, pk);
}
inline bool load_eval_key(Session &s, const std::string &path, std::vector<uint64_t> &evk) {
    if (read_key_file(path, KIND_RK, s.N(), s.D(), 2 * s.beta(), evk)) return true;
    return import_openfhe_eval_key(s, path, evk);
}

inline bool load_secret_key(Session &s, const std::string &path, std::vector<uint64_t> &sk, std::vector<int8_t> &sk_t) {
    if (read_key_file(path, KIND_SK, s.N(), s.D(), 1, sk, &sk_t)) return true;
    return import_openfhe_secret_key(s, path, sk, sk_t);
}

// ------------------------------------------------------------------------------------------------
// weights_summary envelope helpers: the ciphertext fields of one file in a fixed order
// ------------------------------------------------------------------------------------------------
struct CtRef {
    size_t layer;     // index into weights_summary
    int field;        // 0 mean, 1 std_dev, 2 values[idx]
    size_t idx;
};

inline std::vector<CtRef> enumerate_cts(const Json &file) {
    std::vector<CtRef> refs;
    const Json &ws = file.at("weights_summary");
    for (size_t l = 0; l < ws.size(); ++l) {
        refs.push_back({l, 0, 0});
        refs.push_back({l, 1, 0});
        for (size_t k = 0; k < ws.at(l).at("values").size(); ++k) refs.push_back({l, 2, k});
    }
    return refs;
}
inline const std::string &ct_string(const Json &file, const CtRef &r) {
    const Json &lay = file.at("weights_summary").at(r.layer);
    if (r.field == 0) return lay.at("mean").as_string();
    if (r.field == 1) return lay.at("std_dev").as_string();
    return lay.at("values").at(r.idx).as_string();
}

// ------------------------------------------------------------------------------------------------
// Envelope files.  JSON form: the reference's weights_summary document with base64 ciphertext strings
// (aggregateEncryptedWeights.cpp:64-65,119).  Binary form ("MKWS"): the same document with every ciphertext string
// replaced by "@<index>", followed by the raw ciphertext containers -- no base64 (x4/3) and no JSON string
// escaping/parsing of hundreds of MiB.  Layout: magic "MKWS", u32 version = 1, u64 skeleton bytes, skeleton JSON,
// u64 blob count, then per blob u64 size + bytes.  Readers detect the form from the first four bytes; writers
// follow the input's form (encryptModelWeights: MKCKKS_ENVELOPE=binary, or an output path ending in ".mkws").
// ------------------------------------------------------------------------------------------------
template <typename F>
inline void for_each_ct_field(Json &file, F &&fn) {
    for (Json &lay : file["weights_summary"].a) {
        if (lay.contains("mean") && lay.at("mean").is_string()) fn(lay["mean"]);
        if (lay.contains("std_dev") && lay.at("std_dev").is_string()) fn(lay["std_dev"]);
        if (lay.contains("values"))
            for (Json &v : lay["values"].a)
                if (v.is_string()) fn(v);
    }
}

inline bool wants_binary_output(const std::string &out_path) {
    const char *e = std::getenv("MKCKKS_ENVELOPE");
    if (e && std::string(e) == "binary") return true;
    if (e && std::string(e) == "json") return false;
    return out_path.size() > 5 && out_path.compare(out_path.size() - 5, 5, ".mkws") == 0;
}

struct FileCloser {
    void operator()(FILE *f) const {
        if (f) std::fclose(f);
    }
};
using FilePtr = std::unique_ptr<FILE, FileCloser>;

inline Json read_envelope(const std::string &path, bool *was_binary = nullptr) {
    FilePtr fp(std::fopen(path.c_str(), "rb"));
    if (!fp) throw std::runtime_error("cannot open " + path);
    FILE *f = fp.get();
    char magic[4] = {0, 0, 0, 0};
    const bool bin = std::fread(magic, 1, 4, f) == 4 && !std::memcmp(magic, "MKWS", 4);
    if (was_binary) *was_binary = bin;
    if (!bin) {
        fp.reset();
        return Json::parse_file(path);
    }
    auto fail = [&](const char *why) { throw std::runtime_error(std::string("binary envelope: ") + why); };
    // every size field is checked against what the file can still hold: a hostile header cannot make us allocate
    if (std::fseek(f, 0, SEEK_END) != 0) fail("not seekable");
    const long long file_size = std::ftell(f);
    if (file_size < 0 || std::fseek(f, 4, SEEK_SET) != 0) fail("not seekable");
    auto remaining = [&]() -> uint64_t { return (uint64_t)(file_size - std::ftell(f)); };
    uint32_t version = 0;
    uint64_t skel_len = 0, n_blobs = 0;
    if (std::fread(&version, 4, 1, f) != 1 || version != 1) fail("unsupported version");
    if (std::fread(&skel_len, 8, 1, f) != 1 || skel_len > remaining()) fail("bad skeleton size");
    std::string skel(skel_len, '\0');
    if (skel_len && std::fread(&skel[0], 1, skel_len, f) != skel_len) fail("truncated skeleton");
    if (std::fread(&n_blobs, 8, 1, f) != 1 || n_blobs > remaining() / 8) fail("bad blob count");
    std::vector<std::string> blobs(n_blobs);
    for (std::string &b : blobs) {
        uint64_t sz = 0;
        if (std::fread(&sz, 8, 1, f) != 1 || sz > remaining()) fail("bad blob size");
        b.resize(sz);
        if (sz && std::fread(&b[0], 1, sz, f) != sz) fail("truncated blob");
    }
    fp.reset();
    Json doc = Json::parse(skel);
    std::vector<bool> used(blobs.size(), false);
    for_each_ct_field(doc, [&](Json &field) {
        const std::string &ref = field.as_string();
        if (ref.size() < 2 || ref[0] != '@') throw std::runtime_error("binary envelope: ciphertext field without a blob index");
        const size_t i = std::stoull(ref.substr(1));
        if (i >= blobs.size() || used[i]) throw std::runtime_error("binary envelope: blob index out of range or reused");
        used[i] = true;
        field = Json(std::move(blobs[i]));
    });
    return doc;
}

inline void write_envelope(const Json &doc, const std::string &path, bool binary) {
    if (!binary) {
        doc.write_file(path);
        return;
    }
    Json skel = doc;
    std::vector<std::string> blobs;
    for_each_ct_field(skel, [&](Json &field) {
        std::string ref = "@" + std::to_string(blobs.size());
        blobs.push_back(field.as_string());
        field = Json(std::move(ref));
    });
    for (const std::string &b : blobs)
        if (b.size() < 4 || std::memcmp(b.data(), "MKCK", 4))
            throw std::runtime_error("binary envelope: ciphertext field is not a raw container");
    std::string text;
    skel.dump(text, 2);
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    const uint32_t version = 1;
    const uint64_t skel_len = text.size(), n_blobs = blobs.size();
    std::fwrite("MKWS", 1, 4, f);
    std::fwrite(&version, 4, 1, f);
    std::fwrite(&skel_len, 8, 1, f);
    std::fwrite(text.data(), 1, text.size(), f);
    std::fwrite(&n_blobs, 8, 1, f);
    for (const std::string &b : blobs) {
        const uint64_t sz = b.size();
        std::fwrite(&sz, 8, 1, f);
        std::fwrite(b.data(), 1, b.size(), f);
    }
    if (std::fclose(f) != 0) throw std::runtime_error("cannot write " + path);
}

// ------------------------------------------------------------------------------------------------
// Ciphertexts arrive from clients: nothing in a blob is trusted.  The kernels assume canonical residues (lazy sums of
// <= 4 terms in k_sum, 30-bit halves below 2^30 in the conversions, values below 2^51 on the fp64 limbs) and the level
// the header claims decides which constants a rescale uses, so a file that breaks either is refused here instead of
// silently producing a wrong aggregate for every client.
// ------------------------------------------------------------------------------------------------
inline void validate_ct(const Ciphertext &ct, const Session &s) {
    const uint32_t N = s.N(), L = s.L();
    if (ct.nl < 1 || ct.nl > L) throw std::runtime_error("ciphertext: limb count outside [1, L]");
    if (ct.level != L - ct.nl) throw std::runtime_error("ciphertext: level does not match its limb count");
    if (ct.noise_deg != 1 && ct.noise_deg != 2) throw std::runtime_error("ciphertext: noiseScaleDeg must be 1 or 2");
    if (!(ct.scale > 0) || !std::isfinite(ct.scale)) throw std::runtime_error("ciphertext: bad scaling factor");
    if (ct.slots > N / 2) throw std::runtime_error("ciphertext: slot count exceeds N/2");
    if (ct.data.size() != (size_t)2 * ct.nl * N) throw std::runtime_error("ciphertext: wrong payload size");
    for (uint32_t comp = 0; comp < 2; ++comp)
        for (uint32_t i = 0; i < ct.nl; ++i) {
            const uint64_t q = s.moduli()[i];
            const uint64_t *p = &ct.data[((size_t)comp * ct.nl + i) * N];
            uint64_t bad = 0;
            for (uint32_t k = 0; k < N; ++k) bad |= (uint64_t)(p[k] >= q);
            if (bad) throw std::runtime_error("ciphertext: residue not below its modulus");
        }
}
inline Ciphertext decode_ct_checked(const std::string &blob, const Session &s) {
    Ciphertext ct = decode_ct(blob, s.N());
    validate_ct(ct, s);
    return ct;
}

// ------------------------------------------------------------------------------------------------
// Aggregation over n encrypted-weights files (aggregateEncryptedWeights.cpp:68-115): one output entry per tuple of
// entries with equal "layer" and "shape" -- for two files exactly the reference's nested loops (every matching (w2, w1)
// pair, duplicates included), for more files the same nesting continued file by file.
// ------------------------------------------------------------------------------------------------
struct AggItem {
    std::vector<const std::string *> blobs;  // one per file
    size_t out_layer;
    int field;  // 0 mean, 1 std_dev, 2 values[idx]
    size_t idx;
};
inline void match_layers(const std::vector<Json> &files, size_t f, std::vector<const Json *> &cur,
                         std::vector<std::vector<const Json *>> &out) {
    if (f == files.size()) {
        out.push_back(cur);
        return;
    }
    for (const Json &w : files[f].at("weights_summary").a) {
        if (f > 0 && !(w.at("layer") == cur[0]->at("layer") && w.at("shape") == cur[0]->at("shape"))) continue;
        cur.push_back(&w);
        match_layers(files, f + 1, cur, out);
        cur.pop_back();
    }
}
inline std::vector<AggItem> build_agg_items(const std::vector<Json> &files, Json &outputJson) {
    outputJson = Json::object();
    outputJson["weights_summary"] = Json::array();
    std::vector<std::vector<const Json *>> tuples;
    std::vector<const Json *> cur;
    match_layers(files, 0, cur, tuples);
    std::vector<AggItem> items;
    for (const auto &match : tuples) {
        Json agg = Json::object();
        agg["layer"] = match[0]->at("layer");
        agg["shape"] = match[0]->at("shape");
        size_t nvals = (size_t)-1;
        for (const Json *m : match) nvals = std::min(nvals, m->at("values").size());
        Json vals = Json::array();
        for (size_t k = 0; k < nvals; ++k) vals.push_back(Json(""));
        agg["values"] = vals;
        const size_t out_layer = outputJson["weights_summary"].size();
        outputJson["weights_summary"].push_back(agg);
        AggItem mean{{}, out_layer, 0, 0}, sd{{}, out_layer, 1, 0};
        for (const Json *m : match) {
            mean.blobs.push_back(&m->at("mean").as_string());
            sd.blobs.push_back(&m->at("std_dev").as_string());
        }
        items.push_back(mean);
        items.push_back(sd);
        for (size_t k = 0; k < nvals; ++k) {
            AggItem v{{}, out_layer, 2, k};
            for (const Json *m : match) v.blobs.push_back(&m->at("values").at(k).as_string());
            items.push_back(v);
        }
    }
    return items;
}
// the ciphertexts of every item, file-major: flat [file][item][2][nl][N]; all must share level and scale
inline Ciphertext gather_agg_inputs(const std::vector<AggItem> &items, size_t n_files, const Session &s,
                                    std::vector<uint64_t> &flat) {
    const uint32_t N = s.N();
    Ciphertext first = decode_ct_checked(*items[0].blobs[0], s);
    const uint32_t nl = first.nl;
    const size_t words = (size_t)2 * nl * N, B = items.size();
    flat.resize(n_files * B * words);
    for (size_t b = 0; b < B; ++b)
        for (size_t f = 0; f < n_files; ++f) {
            Ciphertext ct = decode_ct_checked(*items[b].blobs[f], s);
            if (ct.nl != nl || ct.noise_deg != first.noise_deg || ct.scale != first.scale)
                throw std::runtime_error("EvalAdd operands differ in level or scale");
            std::memcpy(&flat[(f * B + b) * words], ct.data.data(), words * 8);
        }
    first.data.clear();
    return first;
}
// EvalMult(sum, 1/n) (aggregateEncryptedWeights.cpp:83,92,107) on d_sum [B][2][nl][N] and the output document
// the aggregate as it sits in HBM after EvalMult(., 1/n): B ciphertexts [B][2][meta.nl][N] (+ the header fields every
// one of them carries)
struct AggResult {
    uint64_t *d_out = nullptr;
    Ciphertext meta;
};

inline AggResult scale_aggregate(Session &s, size_t B, uint64_t *d_sum, const Ciphertext &first, size_t n_clients) {
    const uint32_t N = s.N(), nl = first.nl;
    AggResult r;
    Ciphertext &res = r.meta;
    r.d_out = d_sum;
    const double operand = 1.0 / (double)n_clients;  // 0.5 for the reference's two clients
    if (first.noise_deg == 2) {
        // EvalMult(ct, double): rescale first (ModReduceInternalInPlace), then the integer constant
        if (nl < 2) throw std::runtime_error("ciphertext has no limb left to rescale");
        r.d_out = s.alloc<uint64_t>(B * (size_t)2 * (nl - 1) * N);
        Session::check(mkckks_rescale_mult_const_batch(s.ctx(), d_sum, r.d_out, (uint32_t)B, nl, operand));
        res.nl = nl - 1;
        res.level = first.level + 1;
        res.scale = first.scale / (double)s.moduli()[nl - 1] * s.sf(res.level, false);
        res.noise_deg = 2;
    } else {
        Session::check(mkckks_mult_const_batch(s.ctx(), d_sum, (uint32_t)B, nl, operand));
        res.nl = nl;
        res.level = first.level;
        res.scale = first.scale * s.sf(first.level, false);
        res.noise_deg = first.noise_deg + 1;
    }
    res.slots = first.slots;
    return r;
}

// download B ciphertexts [B][2][meta.nl][N] and put them where `items` say in the envelope
inline void store_agg_items(Session &s, const std::vector<AggItem> &items, const uint64_t *d_cts, Ciphertext meta,
                            Json &outputJson) {
    const uint32_t N = s.N();
    const size_t B = items.size(), owords = (size_t)2 * meta.nl * N;
    std::vector<uint64_t> out(B * owords);
    s.to_host(out.data(), d_cts, out.size());
    for (size_t b = 0; b < B; ++b) {
        meta.data.assign(out.begin() + b * owords, out.begin() + (b + 1) * owords);
        Json &lay = outputJson["weights_summary"].a[items[b].out_layer];
        std::string b64 = encode_ct(meta, N);
        if (items[b].field == 0) lay["mean"] = std::move(b64);
        else if (items[b].field == 1) lay["std_dev"] = std::move(b64);
        else lay["values"].a[items[b].idx] = Json(std::move(b64));
    }
}

inline AggResult finish_aggregate(Session &s, const std::vector<AggItem> &items, uint64_t *d_sum, const Ciphertext &first,
                                  size_t n_clients, Json &outputJson) {
    AggResult r = scale_aggregate(s, items.size(), d_sum, first, n_clients);
    store_agg_items(s, items, r.d_out, r.meta, outputJson);
    return r;
}

}  // namespace mkh
