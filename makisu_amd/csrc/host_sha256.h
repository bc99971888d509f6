// host_sha256.h -- streaming SHA-256 on a host core (SHA-NI when the CPU has it) for the layer
// digests.  The reference keeps two serial hash.Hash streams per layer (tarDigester,
// gzipDigester -- lib/builder/step/common.go:44-45); a single Merkle-Damgard stream cannot be
// spread over a GPU, so those two stay on CPU threads, overlapped with the GPU content scan.
// Product code: independent of oracle/.
#pragma once

#include <stddef.h>
#include <stdint.h>
#include <string.h>

#if defined(__x86_64__)
#include <cpuid.h>
#include <immintrin.h>
#endif

namespace mi_host {

class Sha256 {
public:
    Sha256() { reset(); }
    void reset() {
        static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,
                                       0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
        memcpy(h_, iv, sizeof iv);
        total_ = 0;
        fill_ = 0;
    }
    void update(const void* data, size_t len) {
        const uint8_t* p = (const uint8_t*)data;
        total_ += len;
        if (fill_) {
            size_t take = 64 - fill_;
            if (take > len) take = len;
            memcpy(buf_ + fill_, p, take);
            fill_ += take;
            p += take;
            len -= take;
            if (fill_ < 64) return;
            blocks(buf_, 1);
            fill_ = 0;
        }
        const size_t whole = len / 64;
        if (whole) {
            blocks(p, whole);
            p += whole * 64;
            len -= whole * 64;
        }
        if (len) {
            memcpy(buf_, p, len);
            fill_ = len;
        }
    }
    void final(uint8_t out[32]) {
        const uint64_t bits = total_ * 8;
        uint8_t tail[128];
        size_t n = fill_;
        memcpy(tail, buf_, n);
        tail[n++] = 0x80;
        const size_t padded = n <= 56 ? 64 : 128;
        memset(tail + n, 0, padded - n);
        for (int i = 0; i < 8; ++i) tail[padded - 1 - i] = (uint8_t)(bits >> (8 * i));
        blocks(tail, padded / 64);
        for (int i = 0; i < 8; ++i) {
            out[4 * i + 0] = (uint8_t)(h_[i] >> 24);
            out[4 * i + 1] = (uint8_t)(h_[i] >> 16);
            out[4 * i + 2] = (uint8_t)(h_[i] >> 8);
            out[4 * i + 3] = (uint8_t)(h_[i]);
        }
    }
    static bool have_shani() {
#if defined(__x86_64__)
        static const int cached = [] {
            unsigned a, b, c, d;
            if (!__get_cpuid_count(7, 0, &a, &b, &c, &d) || !((b >> 29) & 1)) return 0;   // CPUID.7.0:EBX.SHA
            if (!__get_cpuid(1, &a, &b, &c, &d)) return 0;
            return (int)(((c >> 19) & 1) && ((c >> 9) & 1));                               // SSE4.1, SSSE3
        }();
        return cached != 0;
#else
        return false;
#endif
    }

private:
    uint32_t h_[8];
    uint64_t total_;
    uint8_t buf_[64];
    size_t fill_;

    static const uint32_t* k() {
        static const uint32_t K[64] = {
            0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
            0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
            0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
            0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
            0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
            0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
            0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
            0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
        return K;
    }

    void blocks(const uint8_t* p, size_t n) {
#if defined(__x86_64__)
        if (have_shani()) { blocks_ni(p, n); return; }
#endif
        blocks_portable(p, n);
    }

    static uint32_t ror(uint32_t x, int s) { return (x >> s) | (x << (32 - s)); }

    void blocks_portable(const uint8_t* p, size_t n) {
        const uint32_t* K = k();
        for (; n; --n, p += 64) {
            uint32_t w[16], v[8];
            memcpy(v, h_, sizeof v);
            for (int t = 0; t < 64; ++t) {
                uint32_t wt;
                if (t < 16) {
                    wt = ((uint32_t)p[4 * t] << 24) | ((uint32_t)p[4 * t + 1] << 16) | ((uint32_t)p[4 * t + 2] << 8) | p[4 * t + 3];
                } else {
                    const uint32_t a = w[(t + 1) & 15], b = w[(t + 14) & 15];
                    wt = w[t & 15] + (ror(a, 7) ^ ror(a, 18) ^ (a >> 3)) + w[(t + 9) & 15] + (ror(b, 17) ^ ror(b, 19) ^ (b >> 10));
                }
                w[t & 15] = wt;
                const uint32_t e = v[4], a = v[0];
                const uint32_t t1 = v[7] + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & v[5]) ^ (~e & v[6])) + K[t] + wt;
                const uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & v[1]) ^ (a & v[2]) ^ (v[1] & v[2]));
                v[7] = v[6]; v[6] = v[5]; v[5] = v[4]; v[4] = v[3] + t1;
                v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = t1 + t2;
            }
            for (int i = 0; i < 8; ++i) h_[i] += v[i];
        }
    }

#if defined(__x86_64__)
    // SHA extensions: state kept as {ABEF, CDGH}; four rounds per sha256rnds2 pair, the message
    // schedule advanced with sha256msg1 / sha256msg2 (Intel SHA extensions programming model).
    __attribute__((target("sha,sse4.1,ssse3")))
    void blocks_ni(const uint8_t* p, size_t n) {
        const __m128i* K = (const __m128i*)k();
        const __m128i flip = _mm_set_epi8(12, 13, 14, 15, 8, 9, 10, 11, 4, 5, 6, 7, 0, 1, 2, 3);
        __m128i lo = _mm_loadu_si128((const __m128i*)&h_[0]);           // a b c d (d high)
        __m128i hi = _mm_loadu_si128((const __m128i*)&h_[4]);           // e f g h
        __m128i t = _mm_shuffle_epi32(lo, 0xB1);                        // b a d c
        hi = _mm_shuffle_epi32(hi, 0x1B);                               // h g f e
        __m128i abef = _mm_alignr_epi8(t, hi, 8);
        __m128i cdgh = _mm_blend_epi16(hi, t, 0xF0);
        for (; n; --n, p += 64) {
            const __m128i abef0 = abef, cdgh0 = cdgh;
            __m128i w0 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 0)), flip);
            __m128i w1 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 16)), flip);
            __m128i w2 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 32)), flip);
            __m128i w3 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 48)), flip);
#define MI_SHA_QROUND(W, idx)                                                    \
    do {                                                                         \
        __m128i m_ = _mm_add_epi32((W), _mm_loadu_si128(K + (idx)));             \
        cdgh = _mm_sha256rnds2_epu32(cdgh, abef, m_);                            \
        abef = _mm_sha256rnds2_epu32(abef, cdgh, _mm_shuffle_epi32(m_, 0x0E));   \
    } while (0)
#define MI_SHA_SCHED(Wa, Wb, Wc, Wd)                                             \
    Wa = _mm_sha256msg2_epu32(_mm_add_epi32(_mm_sha256msg1_epu32(Wa, Wb), _mm_alignr_epi8(Wd, Wc, 4)), Wd)
            MI_SHA_QROUND(w0, 0);
            MI_SHA_QROUND(w1, 1);
            MI_SHA_QROUND(w2, 2);
            MI_SHA_QROUND(w3, 3);
            for (int q = 4; q < 16; q += 4) {
                MI_SHA_SCHED(w0, w1, w2, w3); MI_SHA_QROUND(w0, q);
                MI_SHA_SCHED(w1, w2, w3, w0); MI_SHA_QROUND(w1, q + 1);
                MI_SHA_SCHED(w2, w3, w0, w1); MI_SHA_QROUND(w2, q + 2);
                MI_SHA_SCHED(w3, w0, w1, w2); MI_SHA_QROUND(w3, q + 3);
            }
#undef MI_SHA_QROUND
#undef MI_SHA_SCHED
            abef = _mm_add_epi32(abef, abef0);
            cdgh = _mm_add_epi32(cdgh, cdgh0);
        }
        t = _mm_shuffle_epi32(abef, 0x1B);                              // f e b a
        cdgh = _mm_shuffle_epi32(cdgh, 0xB1);                           // d c h g
        lo = _mm_blend_epi16(t, cdgh, 0xF0);
        hi = _mm_alignr_epi8(cdgh, t, 8);
        _mm_storeu_si128((__m128i*)&h_[0], lo);
        _mm_storeu_si128((__m128i*)&h_[4], hi);
    }
#endif
};

}  // namespace mi_host
