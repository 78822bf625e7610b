// base64.hpp -- standard base64 without line breaks.  Same contract as the reference's
// lib/base64_utils.h (Base64Encode :10-27 / Base64Decode :30-49, OpenSSL BIO with BIO_FLAGS_BASE64_NO_NL),
// written from the RFC 4648 alphabet; no OpenSSL dependency.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>

namespace mkh {

inline std::string Base64Encode(const unsigned char *data, size_t len) {
    static const char tbl[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    std::string out;
    out.resize(((len + 2) / 3) * 4);
    size_t o = 0, i = 0;
    for (; i + 2 < len; i += 3) {
        uint32_t v = (data[i] << 16) | (data[i + 1] << 8) | data[i + 2];
        out[o++] = tbl[v >> 18]; out[o++] = tbl[(v >> 12) & 63]; out[o++] = tbl[(v >> 6) & 63]; out[o++] = tbl[v & 63];
    }
    if (i + 1 == len) {
        uint32_t v = data[i] << 16;
        out[o++] = tbl[v >> 18]; out[o++] = tbl[(v >> 12) & 63]; out[o++] = '='; out[o++] = '=';
    } else if (i + 2 == len) {
        uint32_t v = (data[i] << 16) | (data[i + 1] << 8);
        out[o++] = tbl[v >> 18]; out[o++] = tbl[(v >> 12) & 63]; out[o++] = tbl[(v >> 6) & 63]; out[o++] = '=';
    }
    return out;
}
inline std::string Base64Encode(const std::string &in) {
    return Base64Encode(reinterpret_cast<const unsigned char *>(in.data()), in.size());
}

inline std::string Base64Decode(const std::string &in) {
    static int8_t rev[256];
    static bool init = false;
    if (!init) {
        for (int i = 0; i < 256; ++i) rev[i] = -1;
        const char *tbl = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
        for (int i = 0; i < 64; ++i) rev[(unsigned char)tbl[i]] = (int8_t)i;
        init = true;
    }
    std::string out;
    out.reserve(in.size() / 4 * 3);
    uint32_t acc = 0;
    int bits = 0;
    for (unsigned char c : in) {
        if (c == '=' || c == '\n' || c == '\r') continue;
        int v = rev[c];
        if (v < 0) throw std::runtime_error("base64: invalid character");
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            out.push_back((char)((acc >> bits) & 0xFF));
        }
    }
    return out;
}

}  // namespace mkh
