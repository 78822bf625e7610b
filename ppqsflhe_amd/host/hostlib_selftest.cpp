// hostlib_selftest -- the hosts' parsing paths without a device: JSON parser, base64, weights_summary envelopes (JSON and
// binary "MKWS"), ciphertext / key containers.  Built plain and with -fsanitize=address,undefined (`make asan`); the CPU
// test suite drives both builds with well-formed and hostile inputs (tests/test_sanitizers.py).
//   hostlib_selftest json <file>        parse, re-serialise, parse again, compare
//   hostlib_selftest envelope <file>    read_envelope, decode every ciphertext container, re-write both envelope forms
//   hostlib_selftest cc <file>          read_cc (this project's or OpenFHE's CryptoContext JSON)
//   hostlib_selftest roundtrip          synthetic ciphertext -> blob -> base64 -> blob, binary envelope write + read
#include "hostlib.hpp"
using namespace mkh;

static int run(int argc, char **argv) {
    const std::string cmd = argc > 1 ? argv[1] : "";
    if (cmd == "json" && argc == 3) {
        Json a = Json::parse_file(argv[2]);
        std::string text;
        a.dump(text, 2);
        Json b = Json::parse(text);
        std::string again;
        b.dump(again, 2);
        if (text != again) throw std::runtime_error("json: dump/parse is not a fixed point");
        std::cout << "ok json " << text.size() << "\n";
        return 0;
    }
    if (cmd == "envelope" && argc == 3) {
        bool binary = false;
        Json doc = read_envelope(argv[2], &binary);
        size_t n = 0, bytes = 0;
        uint32_t ring = 0;
        for_each_ct_field(doc, [&](Json &f) {
            const std::string &blob = f.as_string();
            const bool raw = blob.size() >= 4 && !std::memcmp(blob.data(), "MKCK", 4);
            const std::string bin = raw ? blob : Base64Decode(blob);
            if (bin.size() < sizeof(BlobHeader)) throw std::runtime_error("blob too short");
            BlobHeader h;
            std::memcpy(&h, bin.data(), sizeof h);
            if (!ring) ring = h.ring_dim;
            Ciphertext ct = decode_ct(blob, ring);
            bytes += ct.data.size() * 8;
            ++n;
        });
        std::cout << "ok envelope " << (binary ? "binary " : "json ") << n << " ciphertexts " << bytes << " bytes\n";
        return 0;
    }
    if (cmd == "cc" && argc == 3) {
        CcFile cc = read_cc(argv[2]);
        std::cout << "ok cc log_n=" << cc.p.log_n << " depth=" << cc.p.mult_depth << " moduli=" << cc.moduli.size() << "\n";
        return 0;
    }
    if (cmd == "roundtrip" && argc == 3) {
        const uint32_t N = 64;
        Ciphertext ct;
        ct.nl = 3; ct.level = 1; ct.noise_deg = 2; ct.slots = 32; ct.scale = 1099511627776.0;
        ct.data.resize((size_t)2 * ct.nl * N);
        for (size_t i = 0; i < ct.data.size(); ++i) ct.data[i] = 0x9E3779B97F4A7C15ull * (i + 1);
        raw_blobs() = false;
        const std::string b64 = encode_ct(ct, N);
        Ciphertext back = decode_ct(b64, N);
        if (back.data != ct.data || back.nl != ct.nl || back.scale != ct.scale) throw std::runtime_error("blob round trip");
        raw_blobs() = true;
        Json doc = Json::object();
        doc["weights_summary"] = Json::array();
        Json lay = Json::object();
        lay["layer"] = "l";
        lay["shape"] = Json::array();
        lay["mean"] = encode_ct(ct, N);
        lay["std_dev"] = encode_ct(ct, N);
        Json vals = Json::array();
        vals.push_back(Json(encode_ct(ct, N)));
        lay["values"] = vals;
        doc["weights_summary"].push_back(lay);
        const std::string path = argv[2];
        write_envelope(doc, path, true);
        bool binary = false;
        Json rd = read_envelope(path, &binary);
        if (!binary) throw std::runtime_error("binary envelope not recognised");
        Ciphertext again = decode_ct(rd.at("weights_summary").at(0).at("values").at(0).as_string(), N);
        if (again.data != ct.data) throw std::runtime_error("envelope round trip");
        std::cout << "ok roundtrip\n";
        return 0;
    }
    std::cerr << "usage: hostlib_selftest json|envelope|cc <file> | roundtrip <tmpfile>\n";
    return 2;
}

int main(int argc, char **argv) {
    try {
        return run(argc, argv);
    } catch (const std::exception &e) {
        std::cerr << "ERROR: " << e.what() << std::endl;
        return 1;
    }
}
