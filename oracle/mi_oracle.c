/*
 * mi_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see mi_oracle.h).
 *
 * Plain C restatement of the arithmetic on the makisu snapshot/digest path and
 * of this repo's own Gear-CDC spec.  Reference citations are file:line under
 * the uber/makisu tree.
 */
#define _GNU_SOURCE
#include "mi_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#if defined(__x86_64__)
#include <cpuid.h>
#include <immintrin.h>
#define MI_X86 1
#endif

/* ======================================================================= */
/* SHA-256 (FIPS 180-4).  Reference call sites: lib/builder/step/common.go:44-45
 * (tarDigester/gzipDigester), lib/docker/image/digester.go:33-37
 * (crypto.SHA256.New()), lib/docker/image/digest.go:42-50 (Digest.Equals).    */
/* ======================================================================= */

static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1,
    0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
    0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786,
    0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b,
    0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a,
    0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

static inline uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static void sha256_blocks_c(uint32_t st[8], const uint8_t* p, size_t nblk) {
    uint32_t w[64];
    while (nblk--) {
        for (int i = 0; i < 16; i++)
            w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) |
                   ((uint32_t)p[4 * i + 2] << 8) | (uint32_t)p[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            uint32_t s0 = rotr32(w[i - 15], 7) ^ rotr32(w[i - 15], 18) ^ (w[i - 15] >> 3);
            uint32_t s1 = rotr32(w[i - 2], 17) ^ rotr32(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = st[0], b = st[1], c = st[2], d = st[3];
        uint32_t e = st[4], f = st[5], g = st[6], h = st[7];
        for (int i = 0; i < 64; i++) {
            uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
            uint32_t ch = (e & f) ^ (~e & g);
            uint32_t t1 = h + S1 + ch + K256[i] + w[i];
            uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
            uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
            uint32_t t2 = S0 + mj;
            h = g; g = f; f = e; e = d + t1;
            d = c; c = b; b = a; a = t1 + t2;
        }
        st[0] += a; st[1] += b; st[2] += c; st[3] += d;
        st[4] += e; st[5] += f; st[6] += g; st[7] += h;
        p += 64;
    }
}

#ifdef MI_X86
__attribute__((target("sha,sse4.1,ssse3")))
static void sha256_blocks_shani(uint32_t st[8], const uint8_t* p, size_t nblk) {
    const __m128i bswap = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
    __m128i tmp = _mm_loadu_si128((const __m128i*)&st[0]);   /* DCBA */
    __m128i s1 = _mm_loadu_si128((const __m128i*)&st[4]);    /* HGFE */
    tmp = _mm_shuffle_epi32(tmp, 0xB1);                       /* CDAB */
    s1 = _mm_shuffle_epi32(s1, 0x1B);                         /* EFGH */
    __m128i s0 = _mm_alignr_epi8(tmp, s1, 8);                 /* ABEF */
    s1 = _mm_blend_epi16(s1, tmp, 0xF0);                      /* CDGH */
    while (nblk--) {
        const __m128i save0 = s0, save1 = s1;
        __m128i m[4];
        for (int i = 0; i < 16; i++) {
            if (i < 4) {
                m[i] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 16 * i)), bswap);
            } else {
                __m128i t = _mm_sha256msg1_epu32(m[i & 3], m[(i + 1) & 3]);
                t = _mm_add_epi32(t, _mm_alignr_epi8(m[(i + 3) & 3], m[(i + 2) & 3], 4));
                m[i & 3] = _mm_sha256msg2_epu32(t, m[(i + 3) & 3]);
            }
            __m128i msg = _mm_add_epi32(m[i & 3], _mm_loadu_si128((const __m128i*)&K256[4 * i]));
            s1 = _mm_sha256rnds2_epu32(s1, s0, msg);
            msg = _mm_shuffle_epi32(msg, 0x0E);
            s0 = _mm_sha256rnds2_epu32(s0, s1, msg);
        }
        s0 = _mm_add_epi32(s0, save0);
        s1 = _mm_add_epi32(s1, save1);
        p += 64;
    }
    tmp = _mm_shuffle_epi32(s0, 0x1B);                        /* FEBA */
    s1 = _mm_shuffle_epi32(s1, 0xB1);                         /* DCHG */
    s0 = _mm_blend_epi16(tmp, s1, 0xF0);                      /* DCBA */
    s1 = _mm_alignr_epi8(s1, tmp, 8);                         /* HGFE */
    _mm_storeu_si128((__m128i*)&st[0], s0);
    _mm_storeu_si128((__m128i*)&st[4], s1);
}
#endif

int mi_ref_have_shani(void) {
#ifdef MI_X86
    static int cached = -1;
    if (cached < 0) {
        unsigned a, b, c, d;
        int ok = 0;
        if (__get_cpuid_count(7, 0, &a, &b, &c, &d)) ok = (b >> 29) & 1;   /* CPUID.7.0:EBX.SHA */
        unsigned a1, b1, c1, d1;
        if (ok && __get_cpuid(1, &a1, &b1, &c1, &d1)) ok = ((c1 >> 19) & 1) && ((c1 >> 9) & 1); /* SSE4.1, SSSE3 */
        cached = ok;
    }
    return cached;
#else
    return 0;
#endif
}

static void sha256_blocks(mi_ref_sha256_ctx* c, const uint8_t* p, size_t nblk) {
#ifdef MI_X86
    if (c->use_shani) { sha256_blocks_shani(c->h, p, nblk); return; }
#endif
    sha256_blocks_c(c->h, p, nblk);
}

void mi_ref_sha256_init(mi_ref_sha256_ctx* c, int allow_shani) {
    static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,
                                   0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(c->h, iv, sizeof iv);
    c->nbytes = 0;
    c->buflen = 0;
    c->use_shani = allow_shani && mi_ref_have_shani();
}

void mi_ref_sha256_update(mi_ref_sha256_ctx* c, const void* data, size_t len) {
    const uint8_t* p = (const uint8_t*)data;
    c->nbytes += len;
    if (c->buflen) {
        size_t take = 64 - c->buflen;
        if (take > len) take = len;
        memcpy(c->buf + c->buflen, p, take);
        c->buflen += (uint32_t)take;
        p += take; len -= take;
        if (c->buflen == 64) { sha256_blocks(c, c->buf, 1); c->buflen = 0; }
    }
    if (len >= 64) {
        size_t nblk = len / 64;
        sha256_blocks(c, p, nblk);
        p += nblk * 64; len -= nblk * 64;
    }
    if (len) { memcpy(c->buf, p, len); c->buflen = (uint32_t)len; }
}

void mi_ref_sha256_final(mi_ref_sha256_ctx* c, uint8_t out[32]) {
    uint64_t bits = c->nbytes * 8;
    uint8_t pad[72];
    size_t padlen = (c->buflen < 56) ? (56 - c->buflen) : (120 - c->buflen);
    memset(pad, 0, sizeof pad);
    pad[0] = 0x80;
    for (int i = 0; i < 8; i++) pad[padlen + i] = (uint8_t)(bits >> (56 - 8 * i));
    uint64_t keep = c->nbytes;
    mi_ref_sha256_update(c, pad, padlen + 8);
    c->nbytes = keep;
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)(c->h[i] >> 24); out[4 * i + 1] = (uint8_t)(c->h[i] >> 16);
        out[4 * i + 2] = (uint8_t)(c->h[i] >> 8); out[4 * i + 3] = (uint8_t)c->h[i];
    }
}

void mi_ref_sha256(const void* data, size_t len, uint8_t out[32], int allow_shani) {
    mi_ref_sha256_ctx c;
    mi_ref_sha256_init(&c, allow_shani);
    mi_ref_sha256_update(&c, data, len);
    mi_ref_sha256_final(&c, out);
}

/* ======================================================================= */
/* CRC32-IEEE.  Reference: crc32.NewIEEE() in addCopyStep.SetCacheID,
 * lib/builder/step/add_copy_step.go:102-122; bytes fed by
 * checksumPathContents :194-238.                                            */
/* ======================================================================= */

#define CRC_POLY 0xEDB88320u
static uint32_t crc_tab[8][256];
static pthread_once_t crc_once = PTHREAD_ONCE_INIT;

static void crc_init(void) {
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ CRC_POLY : c >> 1;
        crc_tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++)
        for (int t = 1; t < 8; t++)
            crc_tab[t][i] = (crc_tab[t - 1][i] >> 8) ^ crc_tab[0][crc_tab[t - 1][i] & 0xFF];
}

uint32_t mi_ref_crc32(uint32_t crc, const void* data, size_t len) {
    pthread_once(&crc_once, crc_init);
    const uint8_t* p = (const uint8_t*)data;
    uint32_t c = ~crc;
    while (len && ((uintptr_t)p & 7)) { c = (c >> 8) ^ crc_tab[0][(c ^ *p++) & 0xFF]; len--; }
    while (len >= 8) {
        uint64_t v;
        memcpy(&v, p, 8);
        v ^= c;
        c = crc_tab[7][v & 0xFF] ^ crc_tab[6][(v >> 8) & 0xFF] ^ crc_tab[5][(v >> 16) & 0xFF] ^
            crc_tab[4][(v >> 24) & 0xFF] ^ crc_tab[3][(v >> 32) & 0xFF] ^ crc_tab[2][(v >> 40) & 0xFF] ^
            crc_tab[1][(v >> 48) & 0xFF] ^ crc_tab[0][(v >> 56) & 0xFF];
        p += 8; len -= 8;
    }
    while (len--) c = (c >> 8) ^ crc_tab[0][(c ^ *p++) & 0xFF];
    return ~c;
}

/* a(x)*b(x) mod P(x), reflected bit order (bit 31 = x^0) */
static uint32_t gf2_mulmod(uint32_t a, uint32_t b) {
    uint32_t prod = 0;
    for (uint32_t m = 1u << 31; m; m >>= 1) {
        if (a & m) prod ^= b;
        b = (b & 1) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return prod;
}

uint32_t mi_ref_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2) {
    /* x^(8*len2) mod P by square-and-multiply, then crc1 * that ^ crc2 */
    uint32_t result = 1u << 31;           /* x^0 */
    uint32_t sq = 1u << 23;               /* x^8 */
    for (uint64_t n = len2; n; n >>= 1) {
        if (n & 1) result = gf2_mulmod(sq, result);
        sq = gf2_mulmod(sq, sq);
    }
    return gf2_mulmod(result, crc1) ^ crc2;
}

/* ======================================================================= */
/* Gear CDC -- this repo's own spec (no reference counterpart).   */
/* ======================================================================= */

static inline uint64_t splitmix64_mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
#define SM_GAMMA 0x9E3779B97F4A7C15ULL

void mi_ref_gear_table(uint64_t seed, uint64_t table[256]) {
    uint64_t s = seed;
    for (int i = 0; i < 256; i++) { s += SM_GAMMA; table[i] = splitmix64_mix(s); }
}

static inline int gear_pass(uint64_t h, uint32_t mask_bits) {
    return mask_bits == 0 ? 1 : (h >> (64 - mask_bits)) == 0;
}

size_t mi_ref_gear_candidates(const uint8_t* data, size_t len, const uint8_t* halo,
                              size_t halo_len, const uint64_t table[256],
                              uint32_t mask_bits, uint64_t* out_pos, size_t cap) {
    uint64_t h = 0;
    size_t n = 0;
    /* h_i = sum_{k<64} G[b_{i-k}] << k : the running form drops older bytes by itself */
    for (size_t i = 0; i < halo_len; i++) h = (h << 1) + table[halo[i]];
    for (size_t i = 0; i < len; i++) {
        h = (h << 1) + table[data[i]];
        if (gear_pass(h, mask_bits)) {
            if (n < cap) out_pos[n] = (uint64_t)i + 1;
            n++;
        }
    }
    return n;
}

size_t mi_ref_cdc_select(const uint64_t* cand, size_t n_cand, uint64_t len, uint32_t min_size,
                         uint32_t max_size, uint64_t* out_ends, size_t cap) {
    uint64_t last = 0;
    size_t n = 0;
#define EMIT(e) do { if (n < cap) out_ends[n] = (e); n++; } while (0)
    for (size_t i = 0; i < n_cand; i++) {
        uint64_t e = cand[i];
        if (e > len) break;
        while (e - last > max_size) { last += max_size; EMIT(last); }
        if (e - last >= min_size) { last = e; EMIT(e); }
    }
    while (len - last > max_size) { last += max_size; EMIT(last); }
    if (len > last) EMIT(len);
#undef EMIT
    return n;
}

size_t mi_ref_cdc_two_phase(const uint8_t* data, size_t len, const mi_ref_cdc_params* p,
                            uint64_t* out_ends, size_t cap) {
    uint64_t table[256];
    mi_ref_gear_table(p->gear_seed, table);
    size_t ccap = len ? len : 1;
    uint64_t* cand = (uint64_t*)malloc(ccap * sizeof(uint64_t));
    if (!cand) return 0;
    size_t nc = mi_ref_gear_candidates(data, len, NULL, 0, table, p->mask_bits, cand, ccap);
    size_t n = mi_ref_cdc_select(cand, nc, len, p->min_size, p->max_size, out_ends, cap);
    free(cand);
    return n;
}

size_t mi_ref_cdc_classic(const uint8_t* data, size_t len, const mi_ref_cdc_params* p,
                          uint64_t* out_ends, size_t cap) {
    uint64_t table[256];
    mi_ref_gear_table(p->gear_seed, table);
    size_t n = 0, s = 0;
    while (s < len) {
        size_t limit = len - s > p->max_size ? s + p->max_size : len;
        size_t cut = limit;
        uint64_t h = 0;   /* hash restarts with the chunk */
        for (size_t i = s; i < limit; i++) {
            h = (h << 1) + table[data[i]];
            if (i + 1 - s >= p->min_size && gear_pass(h, p->mask_bits)) { cut = i + 1; break; }
        }
        if (n < cap) out_ends[n] = cut;
        n++;
        s = cut;
    }
    return n;
}

/* ======================================================================= */
/* Synthetic content (counter-mode PRNG keyed by
 * (seed, content id, offset/8)).                                            */
/* ======================================================================= */

void mi_ref_synth_fill(uint64_t seed, uint64_t content_id, uint64_t offset, uint64_t len,
                       uint8_t* out) {
    const uint64_t base = splitmix64_mix(seed + (content_id + 1) * SM_GAMMA);
    uint64_t o = offset, end = offset + len;
    while (o < end && (o & 7)) {                       /* unaligned head */
        uint64_t v = splitmix64_mix(base + ((o >> 3) + 1) * SM_GAMMA);
        *out++ = (uint8_t)(v >> (8 * (o & 7)));
        o++;
    }
    while (o + 8 <= end) {                             /* whole little-endian words */
        uint64_t v = splitmix64_mix(base + ((o >> 3) + 1) * SM_GAMMA);
        memcpy(out, &v, 8);                            /* x86-64: little-endian */
        out += 8;
        o += 8;
    }
    if (o < end) {
        uint64_t v = splitmix64_mix(base + ((o >> 3) + 1) * SM_GAMMA);
        while (o < end) { *out++ = (uint8_t)(v >> (8 * (o & 7))); o++; }
    }
}

/* ======================================================================= */
/* Batch scan (the twin of the C-ABI's mi_batch_run)                          */
/* ======================================================================= */

typedef struct {
    const uint8_t* data;        /* NULL: synthetic content, generated per file by the worker */
    const uint64_t* offsets;
    const uint64_t* sizes;
    const uint64_t* content_ids;/* synthetic mode: content id per file (NULL = file index)   */
    uint64_t synth_seed;
    uint64_t n_files;
    const mi_ref_cdc_params* p;
    int allow_shani;
    int flags;
    mi_ref_file* files;
    mi_ref_chunk* slots;        /* per-file slot regions */
    const uint64_t* slot_base;
    uint64_t table[256];
    uint64_t next;              /* atomic file cursor */
    mi_ref_chunk* chunks;       /* gather phase */
    uint64_t chunk_cap;
} scan_job;

/* chunk_root: SHA-256 over the concatenated chunk digests for files of up to 64 chunks;
 * beyond that a tree with fan-out 64 (each node = SHA-256 over up to 64 consecutive child
 * digests), repeated until at most 64 nodes remain, whose concatenation is hashed.  The tree
 * keeps the root of a multi-GiB file from being one serial million-block SHA-256 stream. */
void mi_ref_chunk_root(const uint8_t* digests, uint64_t n, uint8_t out[32], int allow_shani) {
    const uint64_t F = MI_REF_ROOT_FANOUT;
    uint8_t* owned = NULL;
    const uint8_t* cur = digests;
    while (n > F) {
        uint64_t m = (n + F - 1) / F;
        uint8_t* next = (uint8_t*)malloc((size_t)m * 32);
        for (uint64_t g = 0; g < m; g++) {
            uint64_t cnt = n - g * F < F ? n - g * F : F;
            mi_ref_sha256(cur + g * F * 32, (size_t)cnt * 32, next + g * 32, allow_shani);
        }
        free(owned);
        owned = next;
        cur = next;
        n = m;
    }
    mi_ref_sha256(cur, (size_t)n * 32, out, allow_shani);
    free(owned);
}

static void scan_one(scan_job* j, uint64_t f, const uint8_t* d) {
    const uint64_t len = j->sizes[f];
    const mi_ref_cdc_params* p = j->p;
    mi_ref_file* fo = &j->files[f];
    mi_ref_chunk* out = j->slots + j->slot_base[f];
    /* streaming two-phase: mark + select in one pass, O(1) memory */
    uint64_t h = 0, last = 0, n = 0;
#define CUT(e) do { \
        out[n].file_index = f; out[n].offset = last; out[n].length = (uint32_t)((e) - last); \
        out[n].dup_of = -1; \
        mi_ref_sha256(d + last, (size_t)((e) - last), out[n].sha256, j->allow_shani); \
        n++; last = (e); } while (0)
    for (uint64_t i = 0; i < len; i++) {
        h = (h << 1) + j->table[d[i]];
        uint64_t e = i + 1;
        if (e - last > p->max_size) { uint64_t c = last + p->max_size; CUT(c); }
        if (gear_pass(h, p->mask_bits) && e - last >= p->min_size) CUT(e);
    }
    while (len - last > p->max_size) { uint64_t c = last + p->max_size; CUT(c); }
    if (len > last) CUT(len);
#undef CUT
    fo->n_chunks = n;
    {
        uint8_t* dg = (uint8_t*)malloc((size_t)(n ? n : 1) * 32);
        for (uint64_t k = 0; k < n; k++) memcpy(dg + 32 * k, out[k].sha256, 32);
        mi_ref_chunk_root(dg, n, fo->chunk_root, j->allow_shani);
        free(dg);
    }
    if (j->flags & MI_REF_FILE_SHA256) mi_ref_sha256(d, (size_t)len, fo->file_sha256, j->allow_shani);
    if (j->flags & MI_REF_FILE_CRC32) fo->crc32 = mi_ref_crc32(0, d, (size_t)len);
}

static void* scan_worker(void* arg) {
    scan_job* j = (scan_job*)arg;
    uint8_t* buf = NULL;
    uint64_t buf_cap = 0;
    for (;;) {
        uint64_t f = __atomic_fetch_add(&j->next, 1, __ATOMIC_RELAXED);
        if (f >= j->n_files) break;
        if (j->data) {
            scan_one(j, f, j->data + j->offsets[f]);
        } else {                                   /* synthetic: generate, scan, forget */
            if (j->sizes[f] > buf_cap) {
                free(buf);
                buf_cap = j->sizes[f] + (j->sizes[f] >> 2) + 64;
                buf = (uint8_t*)malloc((size_t)buf_cap);
            }
            mi_ref_synth_fill(j->synth_seed, j->content_ids ? j->content_ids[f] : f, 0, j->sizes[f], buf);
            scan_one(j, f, buf);
        }
    }
    free(buf);
    return NULL;
}

/* gather: files' slot regions -> the compact chunk table, one file per worker at a time */
static void* gather_worker(void* arg) {
    scan_job* j = (scan_job*)arg;
    for (;;) {
        uint64_t f0 = __atomic_fetch_add(&j->next, 256, __ATOMIC_RELAXED);
        if (f0 >= j->n_files) break;
        uint64_t f1 = f0 + 256 < j->n_files ? f0 + 256 : j->n_files;
        for (uint64_t f = f0; f < f1; f++) {
            uint64_t at = j->files[f].first_chunk, n = j->files[f].n_chunks;
            if (at >= j->chunk_cap) continue;
            if (at + n > j->chunk_cap) n = j->chunk_cap - at;
            memcpy(j->chunks + at, j->slots + j->slot_base[f], (size_t)n * sizeof(mi_ref_chunk));
        }
    }
    return NULL;
}

static void run_threads(void* (*fn)(void*), void* arg, int n_threads) {
    if (n_threads <= 1) { fn(arg); return; }
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
    int started = 0;
    for (int t = 0; t < n_threads; t++)
        if (pthread_create(&th[started], NULL, fn, arg) == 0) started++;
    if (started == 0) fn(arg);
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    free(th);
}

/* ---- duplicate marking ---------------------------------------------------------------- *
 * dup_of[i] = smallest j < i with an equal digest, else -1.  Digests are uniform, so the rows
 * are partitioned by their first 12 bits into 4096 buckets (a counting pass and a scatter pass,
 * both over contiguous row ranges per thread); every bucket is then sorted by (digest, row) and
 * marked independently -- the buckets are the parallel unit.                                  */
#define DD_BUCKET_BITS 12
#define DD_BUCKETS (1u << DD_BUCKET_BITS)
typedef struct {
    const uint8_t* base;        /* digest of row i at base + i * stride */
    size_t stride;
    uint64_t n;
    int64_t* dup_of;            /* dup_of of row i at (uint8_t*)dup_of + i * dup_stride */
    size_t dup_stride;
    int n_threads;
    uint64_t* counts;           /* n_threads x DD_BUCKETS */
    uint64_t* bucket_start;     /* DD_BUCKETS + 1 */
    uint64_t* idx;
    uint64_t next;              /* atomic cursor (thread ids, then buckets) */
    uint64_t uniq;
} dedup_job;

static inline uint32_t dd_bucket(const uint8_t* d) { return ((uint32_t)d[0] << 4) | (d[1] >> 4); }
static inline const uint8_t* dd_row(const dedup_job* j, uint64_t i) { return j->base + i * j->stride; }
static inline int64_t* dd_dup(const dedup_job* j, uint64_t i) {
    return (int64_t*)((uint8_t*)j->dup_of + i * j->dup_stride);
}

static void* dd_count_worker(void* arg) {
    dedup_job* j = (dedup_job*)arg;
    uint64_t t = __atomic_fetch_add(&j->next, 1, __ATOMIC_RELAXED);
    uint64_t per = (j->n + (uint64_t)j->n_threads - 1) / (uint64_t)j->n_threads;
    uint64_t lo = t * per, hi = lo + per < j->n ? lo + per : j->n;
    uint64_t* c = j->counts + t * DD_BUCKETS;
    for (uint64_t i = lo; i < hi; i++) c[dd_bucket(dd_row(j, i))]++;
    return NULL;
}

static void* dd_scatter_worker(void* arg) {
    dedup_job* j = (dedup_job*)arg;
    uint64_t t = __atomic_fetch_add(&j->next, 1, __ATOMIC_RELAXED);
    uint64_t per = (j->n + (uint64_t)j->n_threads - 1) / (uint64_t)j->n_threads;
    uint64_t lo = t * per, hi = lo + per < j->n ? lo + per : j->n;
    uint64_t* c = j->counts + t * DD_BUCKETS;      /* now: this thread's write cursor per bucket */
    for (uint64_t i = lo; i < hi; i++) j->idx[c[dd_bucket(dd_row(j, i))]++] = i;
    return NULL;
}

static int dd_cmp(const void* a, const void* b, void* ctx) {
    const dedup_job* j = (const dedup_job*)ctx;
    uint64_t ia = *(const uint64_t*)a, ib = *(const uint64_t*)b;
    int c = memcmp(dd_row(j, ia), dd_row(j, ib), 32);
    if (c) return c;
    return ia < ib ? -1 : (ia > ib);
}

static void* dd_mark_worker(void* arg) {
    dedup_job* j = (dedup_job*)arg;
    uint64_t uniq = 0;
    for (;;) {
        uint64_t b = __atomic_fetch_add(&j->next, 1, __ATOMIC_RELAXED);
        if (b >= DD_BUCKETS) break;
        uint64_t lo = j->bucket_start[b], hi = j->bucket_start[b + 1];
        if (hi - lo > 1) qsort_r(j->idx + lo, (size_t)(hi - lo), sizeof(uint64_t), dd_cmp, j);
        uint64_t head = lo;
        for (uint64_t k = lo; k < hi; k++) {
            if (k == lo || memcmp(dd_row(j, j->idx[k]), dd_row(j, j->idx[head]), 32) != 0) {
                head = k; uniq++;
                *dd_dup(j, j->idx[k]) = -1;
            } else {
                *dd_dup(j, j->idx[k]) = (int64_t)j->idx[head];   /* rows ascend within a group */
            }
        }
    }
    __atomic_fetch_add(&j->uniq, uniq, __ATOMIC_RELAXED);
    return NULL;
}

static uint64_t dedup_strided(const uint8_t* base, size_t stride, uint64_t n, int64_t* dup_of,
                              size_t dup_stride, int n_threads) {
    if (n == 0) return 0;
    if (n_threads < 1) n_threads = 1;
    if ((uint64_t)n_threads > n) n_threads = (int)n;
    dedup_job j;
    memset(&j, 0, sizeof j);
    j.base = base; j.stride = stride; j.n = n; j.dup_of = dup_of; j.dup_stride = dup_stride;
    j.n_threads = n_threads;
    j.counts = (uint64_t*)calloc((size_t)n_threads * DD_BUCKETS, sizeof(uint64_t));
    j.bucket_start = (uint64_t*)malloc((DD_BUCKETS + 1) * sizeof(uint64_t));
    j.idx = (uint64_t*)malloc((size_t)n * sizeof(uint64_t));
    run_threads(dd_count_worker, &j, n_threads);
    uint64_t at = 0;                              /* bucket-major, thread-minor write cursors */
    for (uint32_t b = 0; b < DD_BUCKETS; b++) {
        j.bucket_start[b] = at;
        for (int t = 0; t < n_threads; t++) {
            uint64_t c = j.counts[(size_t)t * DD_BUCKETS + b];
            j.counts[(size_t)t * DD_BUCKETS + b] = at;
            at += c;
        }
    }
    j.bucket_start[DD_BUCKETS] = at;
    j.next = 0;
    run_threads(dd_scatter_worker, &j, n_threads);  /* thread t scatters its own row range in order */
    j.next = 0;
    run_threads(dd_mark_worker, &j, n_threads);
    free(j.counts); free(j.bucket_start); free(j.idx);
    return j.uniq;
}

uint64_t mi_ref_dedup(const uint8_t* digests, uint64_t n, int64_t* dup_of) {
    return dedup_strided(digests, 32, n, dup_of, sizeof(int64_t), 1);
}

uint64_t mi_ref_dedup_mt(const uint8_t* digests, uint64_t n, int64_t* dup_of, int n_threads) {
    return dedup_strided(digests, 32, n, dup_of, sizeof(int64_t), n_threads);
}

static double g_phase_s[3];
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
void mi_ref_last_phase_seconds(double out[3]) { memcpy(out, g_phase_s, sizeof g_phase_s); }

static uint64_t scan_batch_impl(scan_job* j, int n_threads, mi_ref_chunk* chunks, uint64_t chunk_cap,
                                uint64_t* n_unique) {
    const mi_ref_cdc_params* p = j->p;
    const uint64_t n_files = j->n_files;
    if (p->min_size < 64 || p->max_size < p->min_size) return (uint64_t)-1;
    uint64_t* slot_base = (uint64_t*)malloc((n_files + 1) * sizeof(uint64_t));
    uint64_t total_slots = 0;
    for (uint64_t f = 0; f < n_files; f++) {
        slot_base[f] = total_slots;
        total_slots += j->sizes[f] / p->min_size + 2;
    }
    slot_base[n_files] = total_slots;
    mi_ref_chunk* slots = (mi_ref_chunk*)malloc((total_slots ? total_slots : 1) * sizeof(mi_ref_chunk));
    j->slots = slots;
    j->slot_base = slot_base;
    j->next = 0;
    memset(j->files, 0, sizeof(mi_ref_file) * n_files);
    mi_ref_gear_table(p->gear_seed, j->table);
    double t0 = now_s();
    run_threads(scan_worker, j, n_threads);
    double t1 = now_s();
    uint64_t total = 0;
    for (uint64_t f = 0; f < n_files; f++) {
        j->files[f].first_chunk = total;
        total += j->files[f].n_chunks;
    }
    j->chunks = chunks;
    j->chunk_cap = chunk_cap;
    j->next = 0;
    run_threads(gather_worker, j, n_threads);
    free(slots);
    free(slot_base);
    double t2 = now_s();
    uint64_t uniq = total;
    if (total <= chunk_cap && total && !(j->flags & MI_REF_NO_DEDUP))
        uniq = dedup_strided(chunks[0].sha256, sizeof(mi_ref_chunk), total, &chunks[0].dup_of,
                             sizeof(mi_ref_chunk), n_threads);
    double t3 = now_s();
    g_phase_s[0] = t1 - t0; g_phase_s[1] = t2 - t1; g_phase_s[2] = t3 - t2;
    if (n_unique) *n_unique = uniq;
    return total;
}

uint64_t mi_ref_scan_batch(const uint8_t* data, const uint64_t* offsets, const uint64_t* sizes,
                           uint64_t n_files, const mi_ref_cdc_params* p, int allow_shani,
                           int n_threads, int flags, mi_ref_file* files,
                           mi_ref_chunk* chunks, uint64_t chunk_cap) {
    scan_job j;
    memset(&j, 0, sizeof j);
    j.data = data; j.offsets = offsets; j.sizes = sizes; j.n_files = n_files; j.p = p;
    j.allow_shani = allow_shani; j.flags = flags; j.files = files;
    return scan_batch_impl(&j, n_threads, chunks, chunk_cap, NULL);
}

uint64_t mi_ref_scan_synthetic(uint64_t seed, const uint64_t* content_ids, const uint64_t* sizes,
                               uint64_t n_files, const mi_ref_cdc_params* p, int allow_shani,
                               int n_threads, int flags, mi_ref_file* files,
                               mi_ref_chunk* chunks, uint64_t chunk_cap, uint64_t* n_unique) {
    scan_job j;
    memset(&j, 0, sizeof j);
    j.data = NULL; j.content_ids = content_ids; j.synth_seed = seed; j.sizes = sizes;
    j.n_files = n_files; j.p = p; j.allow_shani = allow_shani; j.flags = flags; j.files = files;
    return scan_batch_impl(&j, n_threads, chunks, chunk_cap, n_unique);
}

/* Synthetic files into one host buffer, files spread over threads (bench.py's CPU sample). */
typedef struct {
    uint64_t seed; const uint64_t* cids; const uint64_t* sizes; const uint64_t* offsets;
    uint64_t n; uint8_t* out; uint64_t next;
} fill_job;
static void* fill_worker(void* arg) {
    fill_job* j = (fill_job*)arg;
    for (;;) {
        uint64_t f = __atomic_fetch_add(&j->next, 1, __ATOMIC_RELAXED);
        if (f >= j->n) break;
        mi_ref_synth_fill(j->seed, j->cids ? j->cids[f] : f, 0, j->sizes[f], j->out + j->offsets[f]);
    }
    return NULL;
}
void mi_ref_synth_fill_many(uint64_t seed, const uint64_t* content_ids, const uint64_t* sizes,
                            const uint64_t* offsets, uint64_t n_files, uint8_t* out, int n_threads) {
    fill_job j = {seed, content_ids, sizes, offsets, n_files, out, 0};
    run_threads(fill_worker, &j, n_threads);
}

/* ======================================================================= */
/* Reference-shaped layer scanner: what makisu does today on a commit with
 * --compression no.  Dataflow restated from
 *   lib/builder/step/common.go:35-63   (tar.Writer -> tee -> sha256)
 *   lib/snapshot/mem_layer.go:232-244  (sorted order -- caller passes it sorted)
 *   lib/tario/write.go:28-68           (header, then io.CopyN in <=32 KiB writes)
 * Tar byte parity with Go's archive/tar is UNPINNED.                         */
/* ======================================================================= */

static void tar_octal(uint8_t* dst, int width, uint64_t v) {
    /* width-1 octal digits, zero padded, NUL terminated */
    for (int i = width - 2; i >= 0; i--) { dst[i] = (uint8_t)('0' + (v & 7)); v >>= 3; }
    dst[width - 1] = 0;
}

int mi_ref_tar_header(const char* name, uint64_t size, uint64_t mtime, uint32_t mode,
                      char typeflag, const char* linkname, uint8_t out[512]) {
    memset(out, 0, 512);
    while (*name == '/') name++;                 /* lib/tario/write.go:57 */
    size_t nl = strlen(name);
    if (nl > 100) {
        /* ustar prefix split at a '/' */
        size_t split = nl;
        for (size_t i = (nl > 155 ? 155 : nl - 1); i > 0; i--)
            if (name[i] == '/' && nl - i - 1 <= 100) { split = i; break; }
        if (split == nl || split > 155) return -1;
        memcpy(out + 345, name, split);
        memcpy(out, name + split + 1, nl - split - 1);
    } else {
        memcpy(out, name, nl);
    }
    tar_octal(out + 100, 8, mode & 07777);
    tar_octal(out + 108, 8, 0);                  /* uid */
    tar_octal(out + 116, 8, 0);                  /* gid */
    tar_octal(out + 124, 12, size);
    tar_octal(out + 136, 12, mtime);             /* seconds: lib/tario/write.go:62 truncates */
    out[156] = (uint8_t)typeflag;
    if (linkname) strncpy((char*)out + 157, linkname, 100);
    memcpy(out + 257, "ustar\0" "00", 8);
    /* uname/gname cleared: lib/snapshot/mem_layer.go:161-162 */
    tar_octal(out + 329, 8, 0);
    tar_octal(out + 337, 8, 0);
    memset(out + 148, ' ', 8);
    uint32_t sum = 0;
    for (int i = 0; i < 512; i++) sum += out[i];
    tar_octal(out + 148, 7, sum);
    out[155] = ' ';
    return 0;
}

uint64_t mi_ref_layer_scan(const uint8_t* data, const uint64_t* offsets, const uint64_t* sizes,
                           const char* const* names, uint64_t n_files, int allow_shani,
                           uint8_t tar_sha256[32]) {
    static const uint8_t zeros[1024] = {0};
    mi_ref_sha256_ctx c;
    mi_ref_sha256_init(&c, allow_shani);
    uint64_t total = 0;
    uint8_t hdr[512];
    char nbuf[32];
    for (uint64_t f = 0; f < n_files; f++) {
        const char* nm = names ? names[f] : NULL;
        if (!nm) { snprintf(nbuf, sizeof nbuf, "f%08llu", (unsigned long long)f); nm = nbuf; }
        mi_ref_tar_header(nm, sizes[f], 0, 0644, '0', NULL, hdr);
        mi_ref_sha256_update(&c, hdr, 512);
        const uint8_t* p = data + offsets[f];
        uint64_t left = sizes[f];
        while (left) {                              /* io.CopyN: 32 KiB writes */
            size_t n = left > 32768 ? 32768 : (size_t)left;
            mi_ref_sha256_update(&c, p, n);
            p += n; left -= n;
        }
        uint64_t pad = (512 - (sizes[f] & 511)) & 511;
        if (pad) mi_ref_sha256_update(&c, zeros, (size_t)pad);
        total += 512 + sizes[f] + pad;
    }
    mi_ref_sha256_update(&c, zeros, 1024);          /* tar.Writer.Close trailer */
    total += 1024;
    mi_ref_sha256_final(&c, tar_sha256);
    return total;
}
