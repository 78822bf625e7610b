// genCC -- drop-in for the reference's server/src/genCC.cpp (argv: none; compile-time CONFIG_PATH / OUTPUT_PATH,
// genCC.cpp:10-16; optional argv[1], argv[2] override them here).
// Reads the four knobs of server/config/config_cc.json:2-5 (MultiplicativeDepth, ScalingModSize, BatchSize,
// PREMode; plus optional RingDim, NumLargeDigits, FirstModSize), derives the CKKS parameter set the way
// GenCryptoContext does (FLEXIBLEAUTOEXT, HYBRID, HEStd_128_classic ring dimension) and writes CC.json.
#include "hostlib.hpp"

#ifndef CONFIG_PATH
#define CONFIG_PATH "server/config/config_cc.json"
#endif
#ifndef OUTPUT_PATH
#define OUTPUT_PATH "server/storage/CC.json"
#endif

using namespace mkh;

int main(int argc, char *argv[]) {
    const std::string configPath = argc > 1 ? argv[1] : CONFIG_PATH;
    const std::string outputPath = argc > 2 ? argv[2] : OUTPUT_PATH;
    Json config;
    try {
        config = Json::parse_file(configPath);
    } catch (const std::exception &) {
        std::cerr << "Failed to open " << configPath << std::endl;
        return 1;
    }
    try {
        mkckks_params p{};
        p.mult_depth = config.contains("MultiplicativeDepth") ? (uint32_t)config.at("MultiplicativeDepth").as_int() : 1;
        p.scaling_bits = config.contains("ScalingModSize") ? (uint32_t)config.at("ScalingModSize").as_int() : 50;
        p.first_bits = config.contains("FirstModSize") ? (uint32_t)config.at("FirstModSize").as_int() : 60;
        p.dnum = config.contains("NumLargeDigits") ? (uint32_t)config.at("NumLargeDigits").as_int() : default_dnum(p.mult_depth);
        p.aux_bits = 60;
        p.extra_bits = 20;
        p.device = -1;  // parameter generation needs no GPU
        std::string mode = "INDCPA";
        if (config.contains("PREMode")) {
            mode = config.at("PREMode").as_string();
            if (mode != "INDCPA") {  // genCC.cpp:59-65: FIXED_NOISE silently switches the scaling technique; reject here
                std::cerr << "Unknown PREMode in config: " << mode << std::endl;
                return 1;
            }
        }
        // ring dimension: smallest N whose HEStd_128_classic bound admits log2(QP); QP depends on N only through
        // the prime search, so size it from the bit budget: Q = first + depth*scaling + extra, P = ceil(digit/aux)*aux
        const uint32_t L = p.mult_depth + 2;
        const uint32_t alpha = (L + p.dnum - 1) / p.dnum;
        const uint32_t q_bits = p.first_bits + p.mult_depth * p.scaling_bits + p.extra_bits;
        const uint32_t digit_bits = p.first_bits + (alpha - 1) * p.scaling_bits;
        const uint32_t k = (digit_bits + p.aux_bits - 1) / p.aux_bits;
        uint32_t ring = config.contains("RingDim") ? (uint32_t)config.at("RingDim").as_int()
                                                    : ring_dim_for_security(q_bits + k * p.aux_bits);
        p.log_n = ilog2(ring);
        const uint32_t batch = config.contains("BatchSize") ? (uint32_t)config.at("BatchSize").as_int() : ring / 2;
        if (batch > ring / 2) throw std::runtime_error("BatchSize exceeds N/2");

        mkckks_ctx *ctx = nullptr;
        Session::check(mkckks_ctx_create(&p, &ctx));
        mkckks_info info;
        Session::check(mkckks_ctx_info(ctx, &info));
        std::vector<uint64_t> mod(info.num_q + info.num_p), roots(mod.size());
        Session::check(mkckks_ctx_moduli(ctx, mod.data()));
        Session::check(mkckks_ctx_roots(ctx, roots.data()));
        mkckks_ctx_destroy(ctx);

        Json c = Json::object();
        c["log_n"] = (int)p.log_n;
        c["ring_dim"] = (int)ring;
        c["MultiplicativeDepth"] = (int)p.mult_depth;
        c["ScalingModSize"] = (int)p.scaling_bits;
        c["FirstModSize"] = (int)p.first_bits;
        c["NumLargeDigits"] = (int)p.dnum;
        c["AuxBits"] = (int)p.aux_bits;
        c["ExtraBits"] = (int)p.extra_bits;
        c["BatchSize"] = (int)batch;
        c["PREMode"] = mode;
        c["ScalingTechnique"] = "FLEXIBLEAUTOEXT";
        c["KeySwitchTechnique"] = "HYBRID";
        c["SecretKeyDist"] = "UNIFORM_TERNARY";
        Json mq = Json::array(), mp = Json::array(), rq = Json::array();
        for (uint32_t i = 0; i < info.num_q; ++i) { mq.push_back(Json((unsigned long long)mod[i])); rq.push_back(Json((unsigned long long)roots[i])); }
        for (uint32_t i = info.num_q; i < mod.size(); ++i) mp.push_back(Json((unsigned long long)mod[i]));
        c["moduli"] = mq;
        c["roots"] = rq;
        c["special_moduli"] = mp;
        Json out = Json::object();
        out["mkckks_cc"] = c;
        out.write_file(outputPath);
    } catch (const std::exception &e) {
        std::cerr << "Failed to serialize CryptoContext to CC.json: " << e.what() << std::endl;
        return 1;
    }
    std::cout << "CryptoContext Generated and saved to : " << outputPath << std::endl;
    return 0;
}
