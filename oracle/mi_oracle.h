/*
 * mi_oracle.h -- CPU oracle for the makisu_amd snapshot/dedup hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (makisu_amd/, the
 * C-ABI library libmakisu_mi.so) may include, link or call this.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and there
 * only as the checker / the timed CPU baseline.
 *
 * What it restates (reference = uber/makisu, paths relative to its root):
 *   - SHA-256 (FIPS 180-4): the arithmetic behind crypto/sha256 used at
 *     lib/builder/step/common.go:44-45, lib/docker/image/digester.go:28-60,
 *     lib/docker/image/digest.go:42-50, bin/makisu/cmd/push.go:207,230.
 *     Go's stdlib is not under /root/reference; SHA-256 is standardised, and
 *     the oracle is PINNED against the reference's own fixtures
 *     (lib/utils/testutil/constants.go:25-28, lib/docker/image/const_linux.go:18,
 *     const_darwin.go:18, digest.go:25) and NIST vectors -- tests/test_oracle.py.
 *   - CRC32-IEEE: hash/crc32 as used by checksumPathContents,
 *     lib/builder/step/add_copy_step.go:102-122,194-238.  Pinned against zlib.
 *   - Gear CDC: the reference has NO content-defined chunking,
 *     so there is no reference CPU path for cut points: PARITY UNPINNED w.r.t.
 *     the reference.  The spec is this repo's own (DESIGN.md "Gear-CDC spec");
 *     the oracle implements it twice -- the classic sequential streaming
 *     chunker and the two-phase (mark, then select) form the GPU uses -- and
 *     the tests require both to agree.
 *   - The reference-shaped scanner (one running SHA-256 over a tar-framed
 *     stream, lib/builder/step/common.go:35-63 + lib/tario/write.go:28-68 +
 *     lib/snapshot/mem_layer.go:232-244) used as the timed CPU baseline.
 *     Tar byte parity with Go archive/tar is UNPINNED (no Go toolchain here).
 */
#ifndef MI_ORACLE_H
#define MI_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- SHA-256 ---------------------------------------------------------- */
typedef struct {
    uint32_t h[8];
    uint64_t nbytes;
    uint8_t  buf[64];
    uint32_t buflen;
    int      use_shani;   /* 0 = portable C rounds, 1 = x86 SHA-NI rounds */
} mi_ref_sha256_ctx;

void mi_ref_sha256_init(mi_ref_sha256_ctx* c, int allow_shani);
void mi_ref_sha256_update(mi_ref_sha256_ctx* c, const void* data, size_t len);
void mi_ref_sha256_final(mi_ref_sha256_ctx* c, uint8_t out[32]);
/* one-shot; allow_shani=0 forces the portable implementation */
void mi_ref_sha256(const void* data, size_t len, uint8_t out[32], int allow_shani);
int  mi_ref_have_shani(void);

/* ---- CRC32 (IEEE 802.3, reflected, poly 0xEDB88320) -------------------- */
uint32_t mi_ref_crc32(uint32_t crc, const void* data, size_t len);
/* crc of A||B from crc(A), crc(B), len(B) */
uint32_t mi_ref_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2);

/* ---- Gear CDC ---------------------------------------------------------- */
typedef struct {
    uint64_t gear_seed;   /* table = first 256 outputs of splitmix64(seed)   */
    uint32_t mask_bits;   /* candidate iff top mask_bits bits of h are zero  */
    uint32_t min_size;    /* >= 64                                            */
    uint32_t max_size;    /* >= min_size                                      */
} mi_ref_cdc_params;

void mi_ref_gear_table(uint64_t seed, uint64_t table[256]);

/* Phase 1: every position e in (0,len] (chunk END offsets, e = i+1) whose
 * windowed Gear hash h_i passes the mask.  `halo` = up to 63 bytes that
 * precede data[0] in the file (NULL/0 at file start).  Returns the number of
 * candidates; writes at most cap of them (ascending). */
size_t mi_ref_gear_candidates(const uint8_t* data, size_t len,
                              const uint8_t* halo, size_t halo_len,
                              const uint64_t table[256], uint32_t mask_bits,
                              uint64_t* out_pos, size_t cap);

/* Phase 2: sequential selection over an ascending candidate list.
 * Writes chunk END offsets (ascending, last == len); returns their count
 * (0 for len == 0). */
size_t mi_ref_cdc_select(const uint64_t* cand, size_t n_cand, uint64_t len,
                         uint32_t min_size, uint32_t max_size,
                         uint64_t* out_ends, size_t cap);

/* Two-phase chunker (phase 1 + phase 2). */
size_t mi_ref_cdc_two_phase(const uint8_t* data, size_t len,
                            const mi_ref_cdc_params* p,
                            uint64_t* out_ends, size_t cap);

/* Classic streaming Gear chunker: hash reset to 0 at every chunk start, the
 * first min_size bytes of a chunk are skipped, cut at the first masked hit or
 * at max_size.  Must equal mi_ref_cdc_two_phase whenever min_size >= 64. */
size_t mi_ref_cdc_classic(const uint8_t* data, size_t len,
                          const mi_ref_cdc_params* p,
                          uint64_t* out_ends, size_t cap);

/* ---- synthetic file content ------------------------------------------- */
/* byte o of content `content_id` under `seed`: little-endian bytes of
 * splitmix64 stream keyed by (seed, content_id).  Fills out[0..len) with the
 * bytes at offsets [offset, offset+len). */
void mi_ref_synth_fill(uint64_t seed, uint64_t content_id, uint64_t offset,
                       uint64_t len, uint8_t* out);

/* n files at once, file f at out + offsets[f], spread over n_threads */
void mi_ref_synth_fill_many(uint64_t seed, const uint64_t* content_ids, const uint64_t* sizes,
                            const uint64_t* offsets, uint64_t n_files, uint8_t* out, int n_threads);

/* ---- whole-batch scan: the twin of the C-ABI mi_batch_run -------------- */
typedef struct {
    uint64_t file_index;
    uint64_t offset;
    uint32_t length;
    int64_t  dup_of;       /* smallest earlier global chunk index with the same digest, or -1 */
    uint8_t  sha256[32];
} mi_ref_chunk;

typedef struct {
    uint64_t n_chunks;
    uint64_t first_chunk;
    uint8_t  chunk_root[32];   /* mi_ref_chunk_root over the file's chunk digests */
    uint8_t  file_sha256[32];  /* SHA-256 of the file bytes */
    uint32_t crc32;            /* CRC32-IEEE of the file bytes */
} mi_ref_file;

/* Scans n_files files (file f = data + offsets[f], sizes[f] bytes).
 * chunks must have room for sum(size/min+2).  Returns total chunks.
 * n_threads > 1 spreads files over pthreads (one file per thread at a time).
 * flags: which optional columns to compute (the GPU engine's MI_FLAG_* twins). */
#define MI_REF_FILE_SHA256 0x1
#define MI_REF_FILE_CRC32  0x2
#define MI_REF_NO_DEDUP    0x4
uint64_t mi_ref_scan_batch(const uint8_t* data, const uint64_t* offsets,
                           const uint64_t* sizes, uint64_t n_files,
                           const mi_ref_cdc_params* p, int allow_shani,
                           int n_threads, int flags, mi_ref_file* files,
                           mi_ref_chunk* chunks, uint64_t chunk_cap);

/* The same scan over SYNTHETIC files (mi_ref_synth_fill(seed, content_ids[f] or f, 0, sizes[f])):
 * every worker generates one file at a time into its own buffer, so a full-size config never
 * exists in host memory.  n_unique (optional) receives the number of dup_of == -1 rows. */
uint64_t mi_ref_scan_synthetic(uint64_t seed, const uint64_t* content_ids, const uint64_t* sizes,
                               uint64_t n_files, const mi_ref_cdc_params* p, int allow_shani,
                               int n_threads, int flags, mi_ref_file* files,
                               mi_ref_chunk* chunks, uint64_t chunk_cap, uint64_t* n_unique);
/* seconds the last mi_ref_scan_* call spent in: [0] scan (Gear + SHA-256 per chunk + roots, all
 * threads), [1] gather into the chunk table, [2] duplicate marking */
void mi_ref_last_phase_seconds(double out[3]);

/* chunk_root of n chunk digests (n x 32 bytes): SHA-256 over their concatenation when
 * n <= 64, else a fan-out-64 tree of SHA-256 nodes (DESIGN.md "chunk_root"). */
#define MI_REF_ROOT_FANOUT 64
void mi_ref_chunk_root(const uint8_t* digests, uint64_t n, uint8_t out[32], int allow_shani);

/* Marks dup_of over an arbitrary digest list (n x 32 bytes): dup_of[i] =
 * smallest j < i with equal digest, else -1.  Returns the unique count. */
uint64_t mi_ref_dedup(const uint8_t* digests, uint64_t n, int64_t* dup_of);
/* the same, rows partitioned into 4096 digest-prefix buckets sorted and marked by n_threads */
uint64_t mi_ref_dedup_mt(const uint8_t* digests, uint64_t n, int64_t* dup_of, int n_threads);

/* ---- reference-shaped layer scanner (CPU baseline) ---------------------- */
/* One running SHA-256 over a tar-framed stream of the files in the given
 * order: 512-byte ustar header, file bytes fed in <=32 KiB writes, zero pad
 * to 512, 1024-byte zero trailer.  names[f] may be NULL ("f%08llu").
 * Returns stream length; tar_sha256 receives the digest. */
uint64_t mi_ref_layer_scan(const uint8_t* data, const uint64_t* offsets,
                           const uint64_t* sizes, const char* const* names,
                           uint64_t n_files, int allow_shani,
                           uint8_t tar_sha256[32]);

/* Builds one 512-byte ustar header (regular file, mode 0644, uid/gid 0,
 * mtime given, empty uname/gname) -- restates the fields
 * lib/snapshot/mem_layer.go:152-190 and lib/tario/write.go:55-68 set. */
int mi_ref_tar_header(const char* name, uint64_t size, uint64_t mtime,
                      uint32_t mode, char typeflag, const char* linkname,
                      uint8_t out[512]);

#ifdef __cplusplus
}
#endif
#endif
