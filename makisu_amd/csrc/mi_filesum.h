// mi_filesum.h -- the end-to-end byte sums of a content-aware commit.
//
// The commit frames the layer tar from bytes that crossed PCIe twice: file -> pinned slab -> arena (HBM) -> pinned window ->
// tar block.  TarDigest is computed over what came back, the chunk root over what landed: a fault on either hop would give a
// self-consistent corrupted layer.  The reference has no such exposure -- io.CopyN (lib/tario/write.go:43-45) hands the
// writer the bytes read() returned, or an error.  So every file row of a batch that keeps sums carries, per 1 MiB CHUNK of the
// file, a 128-bit sum taken where the bytes were READ (the reader thread's pinned slab right after its pread; the directory
// reader's block for small files; the caller's buffer for mi_batch_add_bytes), and the layer writer takes the same sum over
// the bytes it is about to frame: equal, or the chunk is fetched from HBM once more, or the commit fails with MI_ERR_IO
// naming the file, the arena range and the hop.
//
// The sum of a chunk: its bytes as little-endian 64-bit words w_0 .. w_{n-1}, the last word zero-padded;
//     a = sum w_i,   b = sum i * w_i      (mod 2^64).
// Both are plain sums over the words, so pieces of a chunk that are read by different threads, in any order, add up
// (FileSum's members are atomics); within a piece b costs no multiplication per word: the loop is `s1 += w; s2 += s1`
// (what stage_sum_host of MI_FLAG_VERIFY_STAGING runs too), which leaves s2 = sum (m - j) w_j over the piece's m words, and
// sum (k + j) w_j = k * s1 + m * s1 - s2 for a piece that begins at word k (four such chains in a vector where the CPU has
// AVX2).  One flipped bit changes a; two words exchanged, or a run that moved, change b; a range that reads as zeros (round 2's
// unexplained slab) changes a unless it WAS zeros.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <atomic>
#include <memory>
#include <vector>

namespace mi_sum {

constexpr uint64_t kChunk = 1ull << 20;

struct FileSum {
    std::atomic<uint64_t> a{0}, b{0};
};

// m words at q: *a = sum w_i, *c = sum i * w_i (i from 0).  The plain loop is `s1 += w; s2 += s1` -- one word per cycle; with AVX2
// (asked of the CPU at run time) four such chains run in the four lanes of a vector over words 4g + j, two vector adds per 32 bytes,
// and  sum (4g + j) w  =  4 (G s1_j - s2_j) + j s1_j  per lane puts them together -- the reader threads sum what they read at memory
// speed, not at a third of their pread's (profiles/r06_host_feed_with_file_sums.txt: 41 -> 38 GB/s with the plain loop).
inline void words_sum_plain(const uint8_t* q, size_t m, uint64_t* a, uint64_t* c) {
    uint64_t s1 = 0, s2 = 0;
    for (size_t j = 0; j < m; ++j) {
        uint64_t w;
        memcpy(&w, q + 8 * j, 8);
        s1 += w;
        s2 += s1;
    }
    *a = s1;
    *c = (uint64_t)m * s1 - s2;
}
#if defined(__x86_64__)
__attribute__((target("avx2"))) inline void words_sum_avx2(const uint8_t* q, size_t m, uint64_t* a, uint64_t* c) {
    const size_t G = m / 4;
    __m256i s1 = _mm256_setzero_si256(), s2 = _mm256_setzero_si256();
    for (size_t g = 0; g < G; ++g) {
        const __m256i w = _mm256_loadu_si256((const __m256i*)(q + 32 * g));
        s1 = _mm256_add_epi64(s1, w);
        s2 = _mm256_add_epi64(s2, s1);
    }
    uint64_t l1[4], l2[4];
    _mm256_storeu_si256((__m256i*)l1, s1);
    _mm256_storeu_si256((__m256i*)l2, s2);
    uint64_t sa = 0, sc = 0;
    for (int j = 0; j < 4; ++j) {
        sa += l1[j];
        sc += 4 * ((uint64_t)G * l1[j] - l2[j]) + (uint64_t)j * l1[j];
    }
    for (size_t i = 4 * G; i < m; ++i) {                       // the last one to three words
        uint64_t w;
        memcpy(&w, q + 8 * i, 8);
        sa += w;
        sc += (uint64_t)i * w;
    }
    *a = sa;
    *c = sc;
}
inline bool have_avx2() {
    static const bool yes = __builtin_cpu_supports("avx2");
    return yes;
}
#endif
inline void words_sum(const uint8_t* q, size_t m, uint64_t* a, uint64_t* c) {
#if defined(__x86_64__)
    if (m >= 16 && have_avx2()) { words_sum_avx2(q, m, a, c); return; }
#endif
    words_sum_plain(q, m, a, c);
}

// adds bytes [off, off + len) of ONE chunk (off a multiple of 8; len a multiple of 8 unless the piece ends the chunk) to (a, b)
inline void chunk_add(const void* p, size_t len, size_t off, uint64_t* a, uint64_t* b) {
    const uint8_t* q = (const uint8_t*)p;
    const size_t m = len / 8;
    uint64_t s1 = 0, c = 0;
    words_sum(q, m, &s1, &c);
    if (len & 7) {
        uint64_t w = 0;
        memcpy(&w, q + 8 * m, len & 7);
        s1 += w;
        c += (uint64_t)m * w;
    }
    *a += s1;
    *b += (uint64_t)(off / 8) * s1 + c;
}

inline uint64_t chunks_of(uint64_t size) { return size ? (size + kChunk - 1) / kChunk : 1; }

// adds bytes [row_off, row_off + len) of a file row to the row's chunk sums (row_off a multiple of 8; len a multiple of 8
// unless the piece ends the file)
inline void row_add(const void* p, uint64_t len, uint64_t row_off, FileSum* sums) {
    const uint8_t* q = (const uint8_t*)p;
    while (len) {
        const uint64_t k = row_off / kChunk, in = row_off - k * kChunk;
        const uint64_t take = len < kChunk - in ? len : kChunk - in;
        uint64_t a = 0, b = 0;
        chunk_add(q, (size_t)take, (size_t)in, &a, &b);
        sums[k].a.fetch_add(a, std::memory_order_relaxed);
        sums[k].b.fetch_add(b, std::memory_order_relaxed);
        q += take;
        row_off += take;
        len -= take;
    }
}

// Rows' sums live in slabs that never move (a reader thread holds a pointer into one while the adder's thread appends rows).
struct Pool {
    static constexpr size_t kSlab = 1u << 16;
    std::vector<std::unique_ptr<FileSum[]>> slabs;
    size_t left = 0;
    FileSum* next = nullptr;
    FileSum* take(size_t n) {
        if (n > left) {
            const size_t cap = n > kSlab ? n : kSlab;
            slabs.emplace_back(new FileSum[cap]);
            next = slabs.back().get();
            left = cap;
        }
        FileSum* r = next;
        next += n;
        left -= n;
        return r;
    }
    void clear() { slabs.clear(); left = 0; next = nullptr; }
};

}  // namespace mi_sum
