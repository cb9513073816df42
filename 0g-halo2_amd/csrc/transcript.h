// Host side of the Fiat-Shamir transcript: Keccak-256 and snark-verifier's EvmTranscript
// (system/halo2/transcript/evm.rs at v2023_04_20; used by the reference at
// /root/reference/src/wnn.rs:21,241,249,260).  Points are absorbed / written as x || y, 32-byte
// big-endian canonical coordinates; scalars as 32-byte big-endian; a challenge is
// keccak256(buffer ++ [1 if the buffer is exactly the previous 32-byte state]) reduced mod r.
#pragma once

#include <cstdint>
#include <cstring>
#include <vector>

#include "curve.h"

namespace zg {

void keccak256(const uint8_t* data, size_t len, uint8_t out[32]);

inline void fe_to_be_bytes_fr(const Fe& a, uint8_t out[32]) {
    Fe r = Fr::to_raw(a);
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 4; j++) out[31 - (4 * i + j)] = (uint8_t)(r.l[i] >> (8 * j));
}
inline void fe_to_be_bytes_fq(const Fe& a, uint8_t out[32]) {
    Fe r = Fq::to_raw(a);
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 4; j++) out[31 - (4 * i + j)] = (uint8_t)(r.l[i] >> (8 * j));
}

class EvmTranscript {
   public:
    std::vector<uint8_t> buf;     // pending hash input
    std::vector<uint8_t> stream;  // proof bytes
    bool failed = false;          // an identity point was offered (EvmTranscript refuses those)

    void common_scalar(const Fe& s) {
        uint8_t b[32];
        fe_to_be_bytes_fr(s, b);
        buf.insert(buf.end(), b, b + 32);
    }
    void write_scalar(const Fe& s) {
        uint8_t b[32];
        fe_to_be_bytes_fr(s, b);
        buf.insert(buf.end(), b, b + 32);
        stream.insert(stream.end(), b, b + 32);
    }
    // p: normalised Jacobian as the MSM returns it (z = 1, or z = 0 for the identity)
    void write_point(const Jac& p) {
        if (fe_is_zero(p.z)) {
            failed = true;
            return;
        }
        uint8_t b[64];
        fe_to_be_bytes_fq(p.x, b);
        fe_to_be_bytes_fq(p.y, b + 32);
        buf.insert(buf.end(), b, b + 64);
        stream.insert(stream.end(), b, b + 64);
    }
    Fe squeeze() {
        if (buf.size() == 32) buf.push_back(1);
        uint8_t h[32];
        keccak256(buf.data(), buf.size(), h);
        buf.assign(h, h + 32);
        // 256-bit big-endian integer mod r: at most five subtractions (2^256 < 6r)
        Fe v;
        for (int i = 0; i < 8; i++)
            v.l[i] = (uint32_t)h[31 - 4 * i] | ((uint32_t)h[30 - 4 * i] << 8) | ((uint32_t)h[29 - 4 * i] << 16) |
                     ((uint32_t)h[28 - 4 * i] << 24);
        uint32_t pm[8], t[8];
        for (int i = 0; i < 8; i++) pm[i] = FrParams::p(i);
        for (;;) {
            uint32_t borrow = sub8(t, v.l, pm);
            if (borrow) break;
            for (int i = 0; i < 8; i++) v.l[i] = t[i];
        }
        return Fr::from_raw(v);
    }
};

}  // namespace zg
