/*
 * makisu_mi.h -- C ABI of libmakisu_mi.so, the MI355X (gfx950) layer-snapshot
 * content-scan and dedup engine for uber/makisu.
 *
 * The reference has no cgo/FFI today; this header is the
 * boundary a ~100-line cgo shim in lib/snapshot would bind (INTEGRATION.md shows
 * it).  Each entry point cites the reference seam it serves (paths relative to
 * the makisu tree).  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions (mirror Go's "wrap the error with context" style):
 *   - every function returns int: 0 = MI_OK, <0 = error code;
 *   - mi_last_error(ctx) returns a NUL-terminated message -- the calling thread's own copy, valid until
 *     that thread asks again -- so the shim can do fmt.Errorf("gpu scan: %s", C.GoString(..));
 *   - a ctx is not re-entrant but may be called from any OS thread (the engine
 *     calls hipSetDevice itself and keeps no thread-local state -- goroutines
 *     migrate between threads); several ctxs may run concurrently;
 *   - the engine never retains caller memory after a call returns (cgo pointer
 *     rule): mi_batch_add_bytes copies before returning;
 *   - no callbacks, no exceptions/longjmp across the boundary.
 *   - there is NO CPU fallback: if no gfx950 device is usable, mi_ctx_create
 *     fails with MI_ERR_NO_DEVICE.
 */
#ifndef MAKISU_MI_H
#define MAKISU_MI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_ABI_VERSION 6

/* ---- THREE KINDS OF EXPORT (round 6) ------------------------------------------------------------------------------------- *
 * Every prototype below carries one of three (empty) tags; tests/test_abi.py checks that each export has exactly one, and
 * tests/test_integration_shim.py that the cgo shim of INTEGRATION.md binds CORE calls only.
 *
 * MI_CORE   what a first cgo shim in lib/snapshot binds -- the seam as the reference has it (about thirty calls):
 *             ctx        mi_abi_version, mi_config_default, mi_ctx_create / _destroy, mi_last_error
 *             MemFS      mi_memfs_create / _free / _error / _set_clock / _reset / _update_from_entries / _entries / _root_of
 *             commit     mi_memfs_commit_layer (step.commitLayer in one call: walk, GPU scan, content-aware diff, tar from HBM,
 *                        DigestPair), mi_memfs_commit_layer_n (the same over several GPUs), mi_memfs_commit_stats, mi_memfs_set_options / _set_index / _reserve_device /
 *                        _release_device, mi_layer_config_default, mi_copy_layer_entries / _roots / _free
 *             cache      mi_cache_key / _create_entry / _parse_entry / _parse_entry_str   (cache.Manager's strings)
 *             index      mi_index_create / _free / _count / _export / _import               (keyvalue.Store seam)
 *             push       mi_sha256_many                                                     (image.Digester, batched)
 * MI_BLOCK  the building blocks the commit is made of, for a host that wants to arrange them itself (batches and their
 *           results, parts of large files, the RCCL exchange, tree walks, the tar reader and the layer writer, the stateless
 *           forms of the diff, the context checksum): everything INTEGRATION.md's later sections use.
 * MI_DIAG   measurement and diagnosis (stats, roofs, staging counters, wave records): no product flow depends on them.      */
#define MI_CORE
#define MI_BLOCK
#define MI_DIAG

enum {
    MI_OK = 0,
    MI_ERR_INVALID = -1,     /* bad argument / bad state                         */
    MI_ERR_NO_DEVICE = -2,   /* no usable gfx950 device (there is no CPU path)   */
    MI_ERR_HIP = -3,         /* a HIP runtime call failed (message has detail)   */
    MI_ERR_NOMEM = -4,       /* host or device allocation failed                 */
    MI_ERR_IO = -5,          /* open/pread of a path failed (mi_batch_add_path)  */
    MI_ERR_STATE = -6,       /* call out of order (e.g. results before run)      */
    MI_ERR_CAPACITY = -7     /* caller buffer too small                          */
};

/* mi_config.flags */
#define MI_FLAG_FILE_SHA256 0x1u  /* also compute SHA-256 of each whole file: what
                                     image.Digester.FromReader returns per file in
                                     `makisu push` (bin/makisu/cmd/push.go:207,230;
                                     lib/docker/image/digester.go:45-52).  One GPU lane per
                                     file -- 13.5 MB/s a lane, 1.77 TB/s when every lane has
                                     one -- EXCEPT files that would hold the pass up: a single
                                     SHA-256 stream is serial and belongs
                                     on a host core with SHA-NI (2.4 GB/s); such files are hashed
                                     by the library's reader threads out of HBM while the GPU
                                     takes the rest (csrc/mi_stage.hip route_long_strings: the
                                     length classes that make max(host time, GPU time) smallest;
                                     a 128 MiB file 0.06 s instead of 10 s).  Not a fallback: the
                                     same digests, each where it is computed fastest          */
#define MI_FLAG_FILE_CRC32  0x2u  /* also compute CRC32-IEEE of each whole file: the
                                     per-file term of checksumPathContents
                                     (lib/builder/step/add_copy_step.go:194-238)     */
#define MI_FLAG_NO_DEDUP    0x4u  /* skip in-batch duplicate marking                 */
#define MI_FLAG_PREFETCH_ROWS 0x8u /* mi_batch_wait also brings the result rows to the host
                                      (one packed copy, ~1 ms per 700 k chunks, while the
                                      other batch in flight keeps the GPU busy): the
                                      following mi_batch_chunks_view is a pointer.  The steps
                                      with the rows delivered run 0.5-2.2 % below the steps
                                      without (once 5.8 %: profiles/r05_with_rows_repeats.txt,
                                      `with_rows_on_host` in every bench line)             */
#define MI_FLAG_VERIFY_STAGING 0x10u /* host-fed batches check their own staged bytes: every
                                      span a reader thread (or the inline window) copies is
                                      summed on the host while it sits in the pinned slab and
                                      summed again on the GPU where it landed -- right after the
                                      copy and once more, every span of the batch, when staging
                                      ends (after any arena growth).  A span that differs is
                                      copied again from the slab; one that still differs, or
                                      differs at the end, fails the run with MI_ERR_IO naming
                                      the arena range, the reader thread and what the GPU holds
                                      there.  io.CopyN must deliver exactly the file's bytes
                                      (lib/tario/write.go:43-45).  New arena memory is filled
                                      with 0xA5 first, so a byte that never arrived does not
                                      read as a plausible zero.  Counters: mi_batch_stage_stats */
#define MI_FLAG_FILE_SUMS 0x20u     /* every host-fed file of a batch carries 128-bit sums of its
                                      bytes, per 1 MiB of the file, taken WHERE THE BYTES WERE READ
                                      (the reader thread's pinned slab right after its pread, the
                                      caller's buffer for mi_batch_add_bytes); mi_layer_add_batch_file
                                      takes the same sums over the bytes it is about to frame and
                                      accepts them only if equal: a chunk that differs is fetched from
                                      HBM once more, one that still differs fails the layer with
                                      MI_ERR_IO naming the file, the arena range and the hop (the
                                      bytes made two PCIe crossings; io.CopyN, lib/tario/write.go:
                                      43-45, hands over the bytes read() returned or an error).
                                      mi_memfs_commit_layer keeps these sums whatever the ctx's
                                      flags say (MI_COMMIT_VERIFY=0 in the environment: off, for
                                      measurements).  About 0.06 ns per byte on each side -- beside a
                                      2.45 GB/s SHA-256 stream.  The heavier MI_FLAG_VERIFY_STAGING
                                      checks every staged span on the GPU itself: a diagnosis tool */

typedef struct mi_ctx mi_ctx;
typedef struct mi_batch mi_batch;

/* Gear-CDC parameters + engine resources.  The reference has no CDC; the spec is
 * DESIGN.md "Gear-CDC spec".  Defaults (mi_config_default): seed 0x4D414B49,
 * mask_bits 13 (8 KiB mean gap), min 2 KiB, max 64 KiB. */
typedef struct {
    uint32_t struct_size;     /* sizeof(mi_config), for ABI evolution              */
    int32_t  device;          /* HIP device ordinal                                 */
    uint64_t gear_seed;       /* Gear table = first 256 outputs of splitmix64(seed) */
    uint32_t mask_bits;       /* 0..32: candidate iff top mask_bits bits of h == 0  */
    uint32_t min_size;        /* >= 64                                              */
    uint32_t max_size;        /* >= min_size, <= 2^30                               */
    uint32_t flags;           /* MI_FLAG_*                                          */
    uint64_t staging_bytes;   /* bytes per pinned staging slab (0 = 8 MiB)          */
    uint32_t n_streams;       /* reader threads for host-fed batches, one pinned slab
                                 and one copy stream each (0 = 8, at most 64)       */
    /* SHA-256 pass tuning, per ctx (0 = the engine's default).  The defaults
     * can also be moved for a whole process by MI_SHA_BLOCKS_PER_CU / MI_SHA_COOP_MIN_GIB /
     * MI_SHA_COOP_MIN_GIB_PIECES / MI_SHA_COOP_BLOCKS_PER_CU, read at mi_ctx_create; a non-zero field here wins.      */
    uint32_t sha_blocks_per_cu;      /* workgroups per CU of the hashing kernels (default 2, 1..8) */
    uint32_t sha_load_scheme;        /* MI_SHA_LOADS_*: how a lane fetches its next block      */
    uint32_t sha_coop_min_gib;       /* MI_SHA_LOADS_AUTO: arena footprint in GiB from which the
                                        quad-cooperative loads are used (default 9; on an arena
                                        mapped in pieces under 256 MiB -- every walk-fed batch's --
                                        from 1 GiB on: a quarter of the address translations)  */
    uint32_t sha_coop_blocks_per_cu; /* workgroups per CU with cooperative loads (default: 3 from
                                        24 GiB up, else sha_blocks_per_cu; 1..3)               */
    uint32_t sha_sched;              /* MI_SHA_SCHED_*: how the strings of a hashing launch are shared
                                        out among the waves of a SIMD (0 = default; this field was
                                        `reserved`, always 0, before ABI 3's round-3 library)     */
} mi_config;
/* sha_sched.  Default: the wave that arrives FIRST on a SIMD takes the longest quarter of the launch's
 * strings at issue priority, the other wave(s) the rest; whoever runs dry continues in the other range
 * (one string is a serial chain, and the older wave of a SIMD runs three times as fast).   */
#define MI_SHA_SCHED_FLAT      1u            /* one range, every wave equal (the scheme of rounds 1-2)      */
#define MI_SHA_SCHED_LONG_SHIFT(k) ((((k) & 15u) + 1u) << 8)   /* the long range = the first n >> k strings
                                                (k = 0..15; default 2)                                       */
#define MI_SHA_LOADS_AUTO 0u   /* by footprint (sha_coop_min_gib)                            */
#define MI_SHA_LOADS_LANE 1u   /* every lane loads its own block, byte-aligned               */
#define MI_SHA_LOADS_COOP 2u   /* the four lanes of a quad fetch one owner's block together  */

/* One result row per file, in the order files were added.  This is what a
 * content-aware MemFS.isUpdated (lib/snapshot/mem_fs.go:487-503) would compare
 * next to tario.IsSimilarHeader's metadata (lib/tario/compare.go:104-120).       */
typedef struct {
    uint64_t user_tag;         /* caller's tag from mi_batch_add_*                  */
    uint64_t size;             /* bytes                                             */
    uint64_t first_chunk;      /* index of the file's first row in the chunk table  */
    uint32_t n_chunks;
    uint32_t crc32;            /* CRC32-IEEE of the bytes (MI_FLAG_FILE_CRC32)      */
    uint8_t  chunk_root[32];   /* SHA-256 over the concatenated chunk digests       */
    uint8_t  file_sha256[32];  /* SHA-256 of the bytes (MI_FLAG_FILE_SHA256)        */
} mi_file_result;

/* One row per chunk, files in add order, chunks in file order.                    */
typedef struct {
    uint64_t file_index;       /* index into the file table                         */
    uint64_t offset;           /* byte offset inside the file                       */
    uint32_t length;           /* bytes (min_size..max_size except a file's last)   */
    uint32_t reserved;
    int64_t  dup_of;           /* smallest chunk index (global index after
                                  mi_dedup_mark_global) with the same digest, -1 if
                                  this row is the first occurrence                  */
    uint8_t  sha256[32];       /* what sha256.Sum256(chunk) gives in Go             */
} mi_chunk_result;

/* Per-ctx counters for roofline reporting.                          */
typedef struct {
    uint64_t bytes_in;         /* file bytes scanned by the last mi_batch_run       */
    uint64_t n_files;
    uint64_t n_chunks;
    uint64_t n_unique;         /* chunks with dup_of == -1 after the last dedup     */
    double   ms_h2d;           /* host->device staging (0 for device-resident)      */
    double   ms_cdc;           /* Gear marking + cut selection kernels              */
    double   ms_sort;          /* chunk-table compaction + length binning           */
    double   ms_sha_chunks;    /* SHA-256-per-chunk kernel (the dominant kernel)    */
    double   ms_sha_files;     /* chunk_root (+ whole-file SHA-256 / CRC32) kernels */
    double   ms_dedup;         /* duplicate marking                                 */
    double   ms_total;         /* first kernel start to last kernel end             */
} mi_stats;

/* Staging counters of one batch (MI_FLAG_VERIFY_STAGING; all zero without the flag except
 * spans / bytes).                                                                   */
typedef struct {
    uint64_t spans;            /* host->device copies issued (reader-thread runs + inline flushes) */
    uint64_t bytes;            /* bytes they carried                                       */
    uint64_t verified_spans;   /* spans summed on both sides right after their copy        */
    uint64_t mismatches;       /* ... whose sums differed                                  */
    uint64_t repaired;         /* ... and matched after one more copy from the slab        */
    uint64_t final_spans;      /* spans summed again when staging ended                    */
    uint64_t final_mismatches; /* ... that no longer matched (the run fails)               */
    double   ms_verify;        /* host time spent summing + waiting for the device sums    */
} mi_stage_stats;

/* ---- context ----------------------------------------------------------------- */
MI_CORE int  mi_abi_version(void);
/* Diagnostics: from now on every chunk-pass launch of this ctx runs the recording instantiation of the hashing kernel
 * and appends one record per wave to `path` (where it ran, when, how much it hashed: tools/sha_wave_stats.py reads it);
 * each such launch is followed by a stream synchronize -- never on a ctx whose time is measured.  NULL or "" turns it
 * off.  MI_SHA_WAVE_STATS=<file> in the environment of mi_ctx_create does the same for the new ctx.              */
MI_DIAG int  mi_debug_sha_wave_stats(mi_ctx* ctx, const char* path);
MI_CORE int  mi_config_default(mi_config* cfg);
MI_CORE int  mi_ctx_create(const mi_config* cfg, mi_ctx** out);
/* Batches and indexes hold a pointer to their ctx: free them first (mi_batch_free /
 * mi_index_free).  With live children the call changes nothing and returns MI_ERR_STATE (the
 * count is in the message) -- a finalizer that runs in the wrong order gets an error, not a
 * use-after-free.  NULL is MI_OK.                                                       */
MI_CORE int  mi_ctx_destroy(mi_ctx* ctx);
/* Optional: pays NOW what the ctx's first content-aware commit would pay otherwise -- the reader threads with their pinned
 * slabs and streams, the kernels' code objects (a four-file synthetic batch): 0.05-0.06 s of a first commit's 0.07 where a
 * later one takes 0.011 (tools/first_commit_probe.py, profiles/r06_first_commit_probe.txt).  Blocking; a host runs it beside its own start-up work (a goroutine)
 * and joins it before the ctx's next call.  Changes no result and no statistic.                                          */
MI_BLOCK int mi_ctx_warm(mi_ctx* ctx);
/* ctx may be NULL: returns the message of the last failed mi_ctx_create.           */
MI_CORE const char* mi_last_error(mi_ctx* ctx);
MI_DIAG int  mi_get_stats(mi_ctx* ctx, mi_stats* out);
/* device properties as measured by hipGetDeviceProperties: CUs, clock, HBM bytes   */
MI_DIAG int  mi_device_info(mi_ctx* ctx, int32_t* n_cu, int32_t* clock_mhz, uint64_t* hbm_bytes,
                    char* name, size_t name_cap);

/* The integer-VALU roof of SHA-256 on this device, measured now: every lane runs `blocks` 64-round
 * compressions over register data (the very function the hashing kernels inline; no memory traffic),
 * waves_per_simd waves on every SIMD; *bytes_per_second = 64 bytes per compression over the best of
 * three launches.  0 = defaults (8 waves, 512 blocks: ~10 ms).  bench.py quotes the SHA pass against
 * this number from the same run (SURVEY.md 8d "second roof that actually binds SHA-256").          */
MI_DIAG int  mi_sha_valu_roof(mi_ctx* ctx, uint32_t waves_per_simd, uint32_t blocks, double* bytes_per_second);

/* ---- batch: a set of files scanned in one pass --------------------------------- *
 * Serves the per-entry loop of MemFS.commitLayer -> contentMemFile.commit ->
 * tario.WriteEntry (lib/snapshot/mem_fs.go:424-433, mem_layer.go:83-88,
 * lib/tario/write.go:28-52): the shim registers each regular file it is about to
 * write to the layer tar; directories/links/whiteouts have no bytes and are not
 * added.  Order of results == order of adds (the caller adds in sorted-path order,
 * lib/snapshot/mem_layer.go:232-244).                                              */
MI_BLOCK int mi_batch_begin(mi_ctx* ctx, uint64_t n_files_hint, uint64_t bytes_hint, mi_batch** out);
/* The bytes have been consumed when the call returns (len may be 0): small buffers are copied
 * into the batch's own pinned window by the calling thread, buffers of 1 MiB and more by the
 * ctx's reader threads in parallel.  Two batches of one ctx may be filled at the same time.  */
MI_BLOCK int mi_batch_add_bytes(mi_batch* b, const void* data, uint64_t len, uint64_t user_tag);
/* The engine opens `path` and reads exactly `size` bytes (the size at stat time,
 * like io.CopyN(w, f, h.Size) at lib/tario/write.go:43-45).  Short files are an
 * error, extra appended bytes are ignored.  The open and the size check happen in this call;
 * the bytes are read by the ctx's reader threads (several files at once, consecutive small
 * files share one PCIe transfer), so a file that shrinks afterwards fails mi_batch_run /
 * mi_batch_submit with MI_ERR_IO.                                                    */
MI_BLOCK int mi_batch_add_path(mi_batch* b, const char* path, uint64_t size, uint64_t user_tag);
/* Bulk form for MANY files (a layer is mostly small ones): nothing is opened in this call, the reader
 * threads open, read and close the files themselves, several at a time.  The price of the deferred
 * open: a missing or short file does not fail this call but mi_batch_run / mi_batch_submit
 * (MI_ERR_IO naming the path).  user_tags may be NULL (tags 0).                              */
MI_BLOCK int mi_batch_add_paths(mi_batch* b, uint64_t n, const char* const* paths, const uint64_t* sizes,
                       const uint64_t* user_tags);
/* Room for what is known to come (more_files files of more_bytes bytes in total): the arena grows once, now.  Growing
 * under way is correct but costs -- the reader threads are drained first and what the arena holds is moved -- so a
 * caller that knows a layer's size (after its walk; mi_batch_begin's hints serve the same purpose for a fresh batch)
 * says so.  mi_batch_add_tree does it by itself: its enumeration runs ahead of the files it hands over.            */
MI_BLOCK int mi_batch_reserve(mi_batch* batch, uint64_t more_files, uint64_t more_bytes);
/* The same for a byte range of a file -- a member of an uncompressed layer tar, whose ranges
 * mi_tar_entries lists: the file's bytes are [offset, offset + size) of `path`.             */
MI_BLOCK int mi_batch_add_path_range(mi_batch* b, const char* path, uint64_t offset, uint64_t size,
                            uint64_t user_tag);
/* Device-generated synthetic files (bench / roofline runs):
 * file i has sizes[i] bytes of the counter-mode stream keyed by (seed,
 * content_ids[i]); equal content ids give byte-identical files.  content_ids may be
 * NULL (ids = running file index).  user_tag = content id.                         */
MI_BLOCK int mi_batch_add_synthetic(mi_batch* b, uint64_t n_files, const uint64_t* sizes,
                           const uint64_t* content_ids, uint64_t seed);
/* Blocking: stage -> Gear CDC -> SHA-256 per chunk -> per-file roots -> dedup.     */
MI_BLOCK int mi_batch_run(mi_batch* b);
/* Re-runs the device pipeline on data already resident from a previous run (bench
 * steps; no re-staging / re-generation).                                            */
MI_BLOCK int mi_batch_rerun(mi_batch* b);
/* Asynchronous form: mi_batch_submit stages the batch on first use and ENQUEUES the
 * whole pipeline on the batch's own HIP stream without any host synchronisation, then
 * returns; mi_batch_wait blocks until it has finished and publishes counts and stats.
 * Two batches may be in flight on one ctx: the Gear pass of one overlaps the SHA-256
 * pass of the other (they bind different units of the CU).  mi_batch_run ==
 * submit + wait.  A batch may be submitted again after its wait (same data).         */
MI_BLOCK int mi_batch_submit(mi_batch* b);
MI_BLOCK int mi_batch_wait(mi_batch* b);
MI_BLOCK int mi_batch_counts(mi_batch* b, uint64_t* n_files, uint64_t* n_chunks, uint64_t* n_bytes);
MI_BLOCK int mi_batch_files(mi_batch* b, mi_file_result* out, uint64_t cap);
MI_BLOCK int mi_batch_chunks(mi_batch* b, mi_chunk_result* out, uint64_t cap);
/* The same rows without the copy into caller memory: *rows points at the batch's own pinned host
 * buffer (packed on the device, one device-to-host copy), valid until the batch is submitted
 * again, marked globally, reset or freed.  What a cgo shim reads through unsafe.Slice.          */
MI_BLOCK int mi_batch_chunks_view(mi_batch* b, const mi_chunk_result** rows, uint64_t* n_chunks);
/* ... and the file rows the same way (they are packed on the device too and arrive with the same wait).                  */
MI_BLOCK int mi_batch_files_view(mi_batch* b, const mi_file_result** rows, uint64_t* n_files);
/* The per-file chunk roots alone, 32 bytes per file in add order (cap = rows `out` has room for): what a content-aware
 * MemFS.isUpdated compares (lib/snapshot/mem_fs.go:487-503) -- one copy of n_files x 32 bytes, none of the chunk rows.   */
MI_BLOCK int mi_batch_roots(mi_batch* b, uint8_t* out, uint64_t cap);
/* Bytes [offset, offset + len) of file `file_index` as they lie in HBM -- the bytes the scan saw -- through a pinned
 * window of the batch (a fetch brings neighbours along: files staged together are asked for together).  From the moment
 * the batch is staged (mi_batch_run / _submit + _wait / _scan_cuts returned) until it is reset or freed; not while in
 * flight; not for parts.  What the layer writer reads when a commit takes its files from the batch
 * (mi_layer_add_batch_file) instead of reading them from disk a second time (lib/tario/write.go:43-45 reads once).     */
MI_BLOCK int mi_batch_read_file(mi_batch* b, uint64_t file_index, uint64_t offset, void* dst, uint64_t len);
/* Device pointer to the batch's n_chunks x 32-byte digest array (valid until
 * mi_batch_free); what a rank contributes to the all-gather.        */
MI_BLOCK int mi_batch_device_digests(mi_batch* b, const void** d_digests, uint64_t* n_chunks);
/* Copies the batch's file bytes back to the host (tests; cap >= total bytes, files
 * concatenated in add order, no padding).                                           */
MI_DIAG int mi_batch_read_back(mi_batch* b, void* out, uint64_t cap);
/* Empties the batch (files, results, recorded walk) but keeps its device memory and pinned window
 * for the next set of files: the way to scan layer after layer without paying allocation -- and
 * the driver's clearing of fresh device memory, which slows the first host-to-device copies into
 * it -- every time.  Not while in flight.                                                  */
MI_BLOCK int mi_batch_reset(mi_batch* b);
MI_BLOCK int mi_batch_free(mi_batch* b);
/* Staging counters since mi_batch_begin / mi_batch_reset (MI_FLAG_VERIFY_STAGING).  A staging
 * failure (a missing, short or unreadable file, a failed copy, a span that does not verify) is
 * STICKY: every later mi_batch_run / _submit / _scan_cuts of the batch returns MI_ERR_IO with the
 * first failure's message until mi_batch_reset -- a failed batch never scans half-staged bytes.   */
MI_DIAG int mi_batch_stage_stats(mi_batch* b, mi_stage_stats* out);
/* What the first verification mismatch of the batch looked like -- arena range, reader thread, both
 * pairs of sums, how many bytes differed and what the GPU held there (zeros / the 0xA5 fill / other
 * data), whether a second copy repaired it; "" if every span verified.  Call it after staging has
 * ended (mi_batch_run / _submit / _scan_cuts returned): the reader threads write the note while they
 * stage.  The pointer is valid until the batch is reset or freed.                                */
MI_DIAG const char* mi_batch_stage_note(mi_batch* b);

/* ---- parts: ONE file split across batches / GPUs (files >= 256 MiB) ------------- *
 * No reference counterpart: the reference streams a file through one goroutine
 * (lib/tario/write.go:28-52).  A part is the byte range [begin, end) of a file; begin and end are
 * multiples of MI_PART_ALIGN (end may also be the file size).  The engine stages the part behind a
 * halo of max_size bytes (rounded up to MI_PART_ALIGN) so that the chunk straddling `begin` lies
 * inside the item; a part OWNS the chunks that END in (begin, end], wherever they begin.
 * The only thing the parts' owners exchange is 8 bytes per boundary -- the previous part's last cut:
 *     every owner:  mi_batch_scan_cuts(batch)          cuts under an assumed entry (the halo's own
 *                   mi_batch_parts(batch, ...)         selection: right unless it never re-synchronised)
 *     exchange exits; for every part but a file's first:
 *                   mi_batch_set_part_entry(batch, file_index, exit of the previous part)
 *                   mi_batch_fix_cuts(batch)           re-selects only where the entry differed
 *     repeat while some part's exit changed (at most parts-per-file rounds; one on ordinary data)
 *     then mi_batch_submit / mi_batch_run as usual.
 * Submitting a batch that holds a part with an unconfirmed entry is MI_ERR_STATE.
 * Results: mi_chunk_result.offset is the offset inside the WHOLE file; mi_file_result.size is
 * end - begin, chunk_root covers the part's own chunks only (the file's root: mi_chunk_root over
 * the parts' digests), file_sha256 / crc32 are zero.                                           */
#define MI_PART_ALIGN 262144u
typedef struct {
    uint64_t file_index;       /* the part's row in this batch's file table                       */
    uint64_t file_size, begin, end;
    uint64_t entry;            /* file offset of the cut the part's first chunk starts at          */
    uint64_t exit;             /* file offset of the part's last cut (<= end): the next part's entry */
    uint32_t entry_confirmed;  /* mi_batch_set_part_entry was called (always 1 for begin == 0)     */
    uint32_t cuts_current;     /* 0: the confirmed entry differs, mi_batch_fix_cuts is due         */
} mi_part_state;
MI_BLOCK int mi_batch_add_path_part(mi_batch* b, const char* path, uint64_t file_size, uint64_t begin,
                           uint64_t end, uint64_t user_tag);
/* the part of the synthetic file (seed, content_id) of file_size bytes: same bytes as the range
 * [begin, end) of what mi_batch_add_synthetic generates for that content id                     */
MI_BLOCK int mi_batch_add_synthetic_part(mi_batch* b, uint64_t file_size, uint64_t content_id, uint64_t seed,
                                uint64_t begin, uint64_t end);
/* The chunk root of a file from its chunk digests (n x 32 bytes, file order) on the host: SHA-256
 * over the concatenation when n <= 64, else the fan-out-64 tree the engine computes per file.  A
 * split file's root = mi_chunk_root over its parts' digests put end to end.                     */
MI_BLOCK int mi_chunk_root(const uint8_t* digests, uint64_t n, uint8_t* root_out);
/* Blocking: stages the batch and runs Gear marking + cut selection only.                         */
MI_BLOCK int mi_batch_scan_cuts(mi_batch* b);
MI_BLOCK int mi_batch_parts(mi_batch* b, mi_part_state* out, uint64_t cap, uint64_t* n_parts);
MI_BLOCK int mi_batch_set_part_entry(mi_batch* b, uint64_t file_index, uint64_t entry);
/* Blocking: re-selects the parts whose confirmed entry differs from the one their cuts were made
 * with and refreshes their exits.                                                              */
MI_BLOCK int mi_batch_fix_cuts(mi_batch* b);

/* ---- cross-batch / cross-GPU dedup --------------------------------------------- *
 * Plugs in where the reference dedups layer blobs by digest
 * (lib/builder/step/common.go:88-91 LinkStoreFileFrom && !os.IsExist;
 * lib/cache/cache_manager.go:239-252 entry codec): here the unit is a chunk.
 * d_digests: device pointer to n x 32 bytes (e.g. the all-gathered digest set of
 * every rank).  d_dup_of: device pointer to n x int64, receives for every row the
 * smallest row index with an equal digest, or -1 for first occurrences.
 * n_unique (host, optional) receives the number of -1 rows.                        */
MI_BLOCK int mi_dedup_mark(mi_ctx* ctx, const void* d_digests, uint64_t n, void* d_dup_of,
                  uint64_t* n_unique);
/* The same marking for ONE rank of a multi-GPU job: d_digests holds the job-wide, rank-major
 * digest set (n_total rows, after the all-gather); only the rank's own rows
 * [own_first, own_first + own_n) are answered: d_dup_of_own[i] = smallest GLOBAL index with
 * the digest of row own_first + i, or -1.  Identical values to mi_dedup_mark over the whole
 * set, at a fraction of the work: own rows build the table, rows of earlier ranks probe it,
 * rows of later ranks cannot be a minimum and are not touched.  n_own_first = own rows that
 * are the job-wide first occurrence (their sum over ranks = the unique count).          */
MI_BLOCK int mi_dedup_mark_range(mi_ctx* ctx, const void* d_digests, uint64_t n_total, uint64_t own_first,
                        uint64_t own_n, void* d_dup_of_own, uint64_t* n_own_first);
/* Rewrites the batch's dup_of column from a global marking: row i of the batch is
 * global row first_global + i of d_dup_of_global (device, int64).                  */
MI_BLOCK int mi_batch_set_global_dedup(mi_batch* b, const void* d_dup_of_global, uint64_t first_global);
/* mi_dedup_mark_range for a batch's own chunks, written straight into its dup_of column (no
 * intermediate array, no copy): the batch's rows are rows [own_first, own_first + n_chunks) of
 * the job-wide set d_digests_all.                                                        */
MI_BLOCK int mi_batch_mark_global(mi_batch* b, const void* d_digests_all, uint64_t n_total,
                         uint64_t own_first, uint64_t* n_own_first);
/* Device pointer to the batch's dup_of column (n_chunks x int64; valid until mi_batch_free). */
MI_BLOCK int mi_batch_device_dup_of(mi_batch* b, const void** d_dup_of, uint64_t* n_chunks);

/* ---- the digest exchange inside the library: RCCL all-gather over xGMI ---------------- *
 * For hosts without torch (the Go shim).  RCCL is loaded at run time (dlopen), so these
 * fail with MI_ERR_NO_DEVICE where no librccl exists (MI_RCCL_LIB=<path> names another library with
 * the same nccl* entry points: tests/rccl_stub is one, for n ranks on a single GPU).  Multi-process: rank 0 obtains the
 * id, ships it to the peers, every rank calls mi_comm_init_rank.  Single process driving n
 * devices (one ctx each): mi_comm_init_all + mi_dedup_allgather_all.
 * mi_dedup_allgather: all-gathers the batches' digest arrays (counts first, then slabs
 * padded to the largest count), marks this rank's chunks against the gathered set
 * (mi_batch_mark_global) so the batch's dup_of holds GLOBAL row indices (rank-major), and
 * sums the ranks' first-occurrence counts into n_unique; collective: every rank must call
 * it.  Outputs are optional.                                                            */
#define MI_COMM_ID_BYTES 128
MI_BLOCK int mi_comm_unique_id(void* id_out /* MI_COMM_ID_BYTES */);
MI_BLOCK int mi_comm_init_rank(mi_ctx* ctx, int nranks, int rank, const void* id);
MI_BLOCK int mi_comm_init_all(mi_ctx** ctxs, int n);
MI_BLOCK int mi_comm_destroy(mi_ctx* ctx);
/* How many ranks the ctx's communicator spans (ncclCommCount); 0 without a communicator.  What a
 * scaling run records next to its number: a job that believes it ran on 8 GPUs can prove it.     */
MI_BLOCK int mi_comm_ranks(mi_ctx* ctx, int* n_ranks);
MI_BLOCK int mi_dedup_allgather(mi_batch* b, uint64_t* n_total, uint64_t* n_unique,
                       uint64_t* first_global);
MI_BLOCK int mi_dedup_allgather_all(mi_batch** batches, int n, uint64_t* n_total, uint64_t* n_unique);
/* The HASH-PARTITIONED form of the same exchange (SURVEY 8e's "optimisation"): same arguments, same results bit for bit,
 * two all-to-alls (grouped ncclSend / ncclRecv) instead of the all-gather.  Every digest has one OWNER rank, chosen by its
 * second 8 bytes; a rank sends each owner its share (32-byte digest + 4-byte row, split stably on the device), the owner
 * marks what it received -- n_total / n rows, each distinct digest of the job counted once -- and sends 8 bytes per row
 * back.  Per own row at 8 ranks: 38.5 bytes over xGMI instead of 224 received, and a table of 1/8 of the job's rows instead
 * of own rows probed by every earlier rank's.  At most 64 ranks; MI_ERR_NO_DEVICE if the collective library has no
 * ncclSend / ncclRecv.  The all-gather form stays the default of bench.py: the exchange is 3-4 % of a C4 step
 * (DESIGN.md 5), and this form has met only the double's ranks (tests/test_gpu_native_exchange.py), never 8 real ones. */
MI_BLOCK int mi_dedup_alltoall(mi_batch* b, uint64_t* n_total, uint64_t* n_unique, uint64_t* first_global);
MI_BLOCK int mi_dedup_alltoall_all(mi_batch** batches, int n, uint64_t* n_total, uint64_t* n_unique);
/* Device time of the ctx's LAST exchange, from HIP events on the ctx stream (what a scaling line
 * reports beside its rate): ms_gather = the slab all-gather (xGMI time), ms_marking = squeezing the padding
 * out + the job-wide marking of this rank's rows.  Both 0 before the first exchange with rows.  After the all-to-all
 * form: ms_gather = both all-to-alls, ms_marking = the split + the owner's marking and answers + the scatter.  */
MI_DIAG int mi_comm_exchange_ms(mi_ctx* ctx, double* ms_gather, double* ms_marking);

/* ---- COPY/ADD context checksum (addCopyStep.SetCacheID seam) ------------------------ *
 * Reproduces the ONE running CRC32-IEEE the reference feeds at plan time
 * (lib/builder/step/add_copy_step.go:102-122,153-238): `prefix` = seed + directive +
 * args (:104-105), then for every walked path, in filepath.Walk order: its relpath
 * (:211), and for a symlink its target (:221-227), for a regular file all of its bytes
 * (:230-236); directories contribute only their relpath (:216-218), special files are
 * skipped by the caller (:198-203).  File bytes are reduced on the GPU (the batch must
 * come from a ctx with MI_FLAG_FILE_CRC32 and have run); the strings are spliced in on
 * the host with crc(A||B) = crc(A)*x^(8|B|) + crc(B).  The shim prints the result with
 * fmt.Sprintf("%x", crc) (:119) -- unpadded hex -- to get the reference's cacheID.    */
typedef struct {
    const char* relpath;       /* filepath.Rel(ctx.ContextDir, path)                     */
    const char* link_target;   /* non-NULL for a symlink: os.Readlink(path)              */
    int64_t     file_index;    /* regular file: its index in the batch; otherwise -1     */
} mi_ctx_entry;
MI_BLOCK int mi_context_checksum(mi_batch* b, const void* prefix, uint64_t prefix_len,
                        const mi_ctx_entry* entries, uint64_t n, uint32_t* crc_out);

/* ---- host-side walks: which files reach the batch, in what order -------------------- *
 * C++ restatement of the two walks that feed the hot path (filepath.Walk order: lexical
 * per directory, a directory before its children, symlinks never followed):
 *   MI_TREE_CONTEXT  checksumPathContents' walk (lib/builder/step/add_copy_step.go:
 *                    153-169,194-238): only special files are skipped;
 *   MI_TREE_SCAN     the snapshot walk (lib/snapshot/utils.go:37-75): also skips
 *                    ".wh..wh."-prefixed names, blacklist descendants
 *                    (lib/pathutils/path.go:24-35) and mountpoints
 *                    (lib/mountutils/mountutils.go:54-93; the table is read once per
 *                    process from /proc/mounts, or from the file MI_MOUNTS_FILE names -- the
 *                    reference's tests swap mountInfo.mountsFile the same way; a malformed
 *                    table fails the walk with MI_ERR_IO); skipped directories are pruned.
 *                    As in memLayer.createHeader (lib/snapshot/mem_layer.go:171-185) an
 *                    ABSOLUTE symlink target loses the rel_base prefix (pathutils.TrimRoot,
 *                    path.go:63-68: plain string prefix, then AbsPath); a target outside
 *                    rel_base fails the walk with MI_ERR_INVALID, as it fails the scan there.
 * Every regular file is added to the batch (stat-time size, user_tag = entry index);
 * relpath = filepath.Rel(rel_base, path) (rel_base NULL = root).  May be called several
 * times (one per COPY source, add_copy_step.go:158-167); entries accumulate.          */
#define MI_TREE_CONTEXT 0u
#define MI_TREE_SCAN    1u
typedef struct {
    const char* relpath;       /* valid until mi_batch_free                              */
    const char* link_target;   /* symlinks only: os.Readlink (context walk) / root-trimmed (scan) */
    int64_t     file_index;    /* regular files: index in the batch, else -1             */
    uint64_t    size;
    int64_t     mtime_sec;     /* truncated to seconds like tario.WriteHeader (write.go:62) */
    uint32_t    mode;          /* st_mode                                                */
    uint8_t     kind;          /* 0 directory, 1 regular file, 2 symlink, 3 hard link
                                  (3 is never produced by the walks; callers that track
                                  inodes like the reference's snapshot may set it)       */
    uint32_t    uid, gid;
} mi_tree_entry;
/* Regular files reach the batch while the walk goes on.  Files up to 16 KiB (MI_WALK_INLINE_MAX_KIB) are read by the
 * walk's own directory readers where they are listed -- one block of host memory per directory, one piece of the arena
 * each; those threads work on a file-descriptor table of their own (unshare(CLONE_FILES); where that is refused, on
 * the shared one) -- and a file that cannot be opened, or has shrunk since its lstat, fails THIS call with MI_ERR_IO
 * naming it, at its place in walk order.  Larger files are handed over in bulk as paths (mi_batch_add_paths: the
 * reader threads open them); one of those that vanishes or shrinks fails mi_batch_run / mi_batch_submit.
 * MI_WALK_INLINE=0: every file as a path.  Where a file lies in the arena is independent of its index.            */
MI_BLOCK int mi_batch_add_tree(mi_batch* b, const char* root, const char* rel_base,
                      const char* const* blacklist, uint64_t n_blacklist, uint32_t mode,
                      uint64_t* n_entries);
MI_BLOCK int mi_batch_tree_entries(mi_batch* b, mi_tree_entry* out, uint64_t cap);
/* The same walk on its own -- no ctx, no GPU: lists what mi_batch_add_tree WOULD add and in
 * which order (file_index = running ordinal of the regular files).  Host logic only.      */
typedef struct mi_tree mi_tree;
MI_BLOCK int  mi_tree_walk(const char* root, const char* rel_base, const char* const* blacklist,
                  uint64_t n_blacklist, uint32_t mode, mi_tree** out, uint64_t* n_entries);
MI_BLOCK int  mi_tree_entries(const mi_tree* tree, mi_tree_entry* out, uint64_t cap);
MI_BLOCK void mi_tree_free(mi_tree* tree);
/* mi_context_checksum over the recorded walk (the batch must have run).                */
MI_BLOCK int mi_context_checksum_tree(mi_batch* b, const void* prefix, uint64_t prefix_len,
                             uint32_t* crc_out);

/* The order MemFS.commitLayer writes entries in (memLayer.rangeFiles, lib/snapshot/
 * mem_layer.go:232-244: sort.Strings over the absolute dst paths; a whiteout marker
 * ".wh.<name>" sorts under the path it deletes, addHeader :190-211) -- not the walk order
 * ("a-b" < "a/x").  order_out[k] = index of the k-th entry to commit; equal keys keep their input
 * order.  The input's sorted runs are merged (a walk's order is nearly this one: 0.13 us per entry);
 * from 131 072 entries on, blocks are sorted on up to 16 threads of the library's own
 * (MI_WALK_THREADS) that end with the call.  Host logic.                                    */
MI_BLOCK int mi_entries_commit_order(const mi_tree_entry* entries, uint64_t n, uint64_t* order_out);

/* "Did this path change?" -- tario.IsSimilarHeader (lib/tario/compare.go:24-117), the test
 * behind MemFS.isUpdated (lib/snapshot/mem_fs.go:487-503), on walk entries: two entries with
 * empty relpaths are similar; otherwise the kinds must match and symlinks compare their
 * targets; hard links mtime+target+uid+gid+mode; directories mtime+uid+gid+mode; regular
 * files mtime+uid+gid+size+mode (mtime at second resolution, skipped when ignore_time;
 * mode = permission, setuid/setgid/sticky bits).  Content-aware extension: when BOTH roots
 * are given (32-byte chunk_root of each file) regular files are similar only if the roots
 * are equal too -- the reference "ignores path and content" (compare.go:101-103) because it
 * has no cheap content identity; with NULL roots the answer is exactly the reference's.
 * Unknown kind -> MI_ERR_INVALID (the reference's "unsupported type" error).  Host logic.  */
MI_BLOCK int mi_entry_similar(const mi_tree_entry* a, const mi_tree_entry* b, int ignore_time,
                     const uint8_t* root_a, const uint8_t* root_b, int* similar);

/* ---- a layer tar as a source: entries and the byte range of every file, no extraction ---- *
 * MemFS.UpdateFromTarReader (lib/snapshot/mem_fs.go:165-255) fills the in-memory tree from the
 * headers of base / cached layers.  mi_tar_open reads the headers of an UNCOMPRESSED tar (ustar,
 * pax 'x' records, GNU long names / base-256 numbers -- what Go's archive/tar and docker
 * write) and lists them as mi_tree_entry rows: relpath = the cleaned header name ("." for the
 * root, no leading or trailing "/"), kind 0 directory / 1 regular / 2 symlink / 3 hard link /
 * 4 other (device, fifo, a pax GLOBAL header -- which archive/tar hands to its caller as a member
 * and applies to nobody), file_index = ordinal among the regular files.  What ends an archive and
 * what fails it follow archive/tar's reader, not POSIX (csrc/mi_tar.hip lists the rules): one zero
 * block + end of data is an end, a zero block + a header or a last block of 1..511 bytes is
 * MI_ERR_INVALID ("read header: ...", mem_fs.go:185), as are malformed numbers, pax records and
 * pax times.  data_offsets[i]
 * (optional) = where entry i's bytes start in the archive (regular files only): a file is one
 * contiguous range of the tar.  Whiteout markers (".wh.<name>") are listed as the entries they
 * are.  Strings live until mi_tar_free.  Host logic.                                        */
typedef struct mi_tar mi_tar;
MI_BLOCK int  mi_tar_open(const char* path, mi_tar** out, uint64_t* n_entries);
/* The same with the failure reason as text (err, err_cap; nothing is printed anywhere) and
 * *is_gzip = 1 when `path` is a gzip blob -- the form layers are stored and pulled in
 * (tario.NewGzipReader, lib/tario/gzip.go:50-53; lib/builder/build_node.go:133-148).  A gzip blob is
 * listed through a streaming inflate; its data offsets are offsets in the UNCOMPRESSED tar.   */
MI_BLOCK int  mi_tar_open_ex(const char* path, mi_tar** out, uint64_t* n_entries, int* is_gzip, char* err,
                    uint64_t err_cap);
/* gzip blob -> the uncompressed tar at tar_path_out (NULL = digests only), its size and SHA-256
 * (the layer's TarDigest / diffID) and, optionally, the blob's own SHA-256 (GzipDescriptor.Digest):
 * the reference's fixture pair 393ccd5c... / 4ac76077... (lib/utils/testutil/constants.go:28) is
 * the test.  Then mi_tar_open(tar_path_out) + mi_batch_add_path_range scan the members.  A plain
 * tar is copied through unchanged.  Host logic (zlib, SHA-NI).                                */
MI_BLOCK int  mi_tar_inflate(const char* blob_path, const char* tar_path_out, uint64_t* tar_bytes,
                    uint8_t* tar_sha256, uint8_t* blob_sha256, char* err, uint64_t err_cap);
MI_BLOCK int  mi_tar_entries(const mi_tar* tar, mi_tree_entry* out, uint64_t* data_offsets, uint64_t cap);
MI_BLOCK void mi_tar_free(mi_tar* tar);

/* The layer diff of a scan, on two walks -- what MemFS.createLayerByScan + maybeAddToLayer
 * (lib/snapshot/mem_fs.go:315-341, 440-480) decide against the in-memory tree:
 *   after_flags[i]      MI_DIFF_CHANGED  the path is new or mi_entry_similar says it changed
 *                                        (content-aware when both sides carry chunk roots);
 *                       MI_DIFF_ANCESTOR an unchanged directory carried along because something
 *                                        below it changed or was deleted (addAncestors);
 *                       MI_DIFF_SAME     not part of the layer.  The root ("." ) never is.
 *   before_whiteout[j]  1 = write a whiteout for this path: it is gone, and it is the top of
 *                       the deleted subtree under a parent that still is a directory
 *                       ("only one whiteout file is needed for a deleted subtree", :461).
 * A side's roots: 32-byte chunk roots indexed by entry.file_index with a byte stride (pass
 * &files[0].chunk_root and sizeof(mi_file_result)), or NULL for the reference's
 * metadata-only rule.  Host logic.                                                       */
#define MI_DIFF_SAME     0u
#define MI_DIFF_CHANGED  1u
#define MI_DIFF_ANCESTOR 2u
typedef struct {
    const mi_tree_entry* entries;
    uint64_t             n;
    const void*          roots;        /* may be NULL */
    uint64_t             root_stride;
    const char*          disk_root;    /* `after` side only, may be NULL: the directory that was walked.
                                          When given, a path missing from `after` is whited out only
                                          if it is really gone from disk (memFSNode.isOnDisk,
                                          mem_fs.go:49-57,466) -- a path the walk now SKIPS (a new
                                          mountpoint, a blacklisted directory) still exists and gets
                                          no whiteout.                                              */
} mi_snapshot_side;
MI_BLOCK int mi_snapshot_diff(const mi_snapshot_side* before, const mi_snapshot_side* after, int ignore_time,
                     uint8_t* after_flags, uint8_t* before_whiteout);

/* ---- the layer of a COPY / ADD step: MemFS.AddLayerByCopyOps on entry lists ---------------- *
 * addToLayer + maybeAddToLayer + addAncestors (lib/snapshot/mem_fs.go:276-289, 343-421, 440-566)
 * and CopyOperation's source handling (lib/snapshot/copy_op.go:29-100, utils.go:249-327):
 *   tree / tree_root  the merged view so far (entries with relpaths below tree_root, e.g. the
 *                     result of mi_memfs_entries) and the directory it describes;
 *   ops               one per COPY/ADD: srcs relative to src_root (symlinks inside src_root are
 *                     resolved, one leaving it is an error), dst absolute, "dir/" = copy INTO it;
 *                     a single non-directory source with a dst not ending in "/" copies onto dst;
 *   now_sec           mtime of the directories the step has to create.
 * The destination directory chain is ensured first (existing ancestors are carried into the layer,
 * symlinks on the way followed, missing directories created with the op's uid/gid), then every
 * walked source path (snapshot walk rules, no blacklist) gets its header with the op's uid/gid and
 * is added iff tario.IsSimilarHeader says it differs from what the tree holds.  A single FILE
 * source skips the first step, so its dst stays as spelled: right below a symlink it becomes a
 * child of the link's node (`COPY f /lib/f` with /lib -> usr/lib: the layer holds lib, lib/f, usr,
 * usr/lib), deeper it fails as in the reference ("missing intermediate directory").  No scan
 * whiteouts: a copy never deletes -- except by NAME: a source called ".wh.<x>" is a whiteout of its
 * sibling <x>, filed under that path (memLayer.addHeader, mem_layer.go:197-212).  Result: the layer's entries in commit order (mi_copy_layer_entries), each
 * with the path its bytes are read from ("/" for created directories, as in the reference) -- feed them to mi_layer_add
 * and the regular files to a batch.  The caller's tree is not modified.  Host logic.           */
typedef struct {
    const char*        src_root;
    const char* const* srcs;
    uint64_t           n_srcs;
    const char*        dst;
    uint32_t           uid, gid;
} mi_copy_op;
/* NewCopyOperation's parameter check and destination (lib/snapshot/copy_op.go:44-81, 149-180): no sources, several
 * sources with a dst that is not in directory format (trailing "/", "." or ".."), or a relative dst without an
 * absolute work_dir are MI_ERR_INVALID ("check copy param: ..."); otherwise dst_out = dst if absolute, else
 * filepath.Join(work_dir, dst) with a trailing "/" kept.  mi_memfs_add_layer_by_copy_ops applies the same check to the
 * (already resolved, hence absolute) dst of every op.  Host logic.                                                        */
MI_BLOCK int  mi_copy_op_resolve(uint64_t n_srcs, const char* work_dir, const char* dst, char* dst_out, uint64_t cap,
                        char* err, uint64_t err_cap);
typedef struct mi_copy_layer mi_copy_layer;
MI_CORE int  mi_copy_layer_entries(const mi_copy_layer* layer, mi_tree_entry* out, const char** src_paths,
                           uint64_t cap);
MI_CORE void mi_copy_layer_free(mi_copy_layer* layer);

/* ---- MemFS as a handle: the reference's type (lib/snapshot/mem_fs.go:59-125) ------------------------------------- *
 * One tree for the life of a build, nodes of the reference's shape (header + children + the path the content came
 * from).  This is the ONE implementation of the layer merge and of the copy-op layer behind this header (until ABI 3
 * there were stateless twins on entry lists, mi_entries_apply_layer and mi_snapshot_copy_ops; they could not name the
 * directories addAncestors creates and are gone).  The handle keeps what lies between a build's steps -- those
 * directories (nodes like any other, part of mi_memfs_entries) and memFSNode.src, which is what isOnDisk asks the
 * disk about (mem_fs.go:49-57: a file a COPY step added is "on disk" while its SOURCE exists, whether or not the step
 * modified the file system).  mi_snapshot_diff above stays as the stateless form of ONE question (the scan diff of two
 * walks), held against mi_memfs_add_layer_by_scan on generated trees.
 *   mi_memfs_create           NewMemFS(clk, root, blacklist): the root's header from lstat(root); now_sec = the clock
 *                             (mtime of created directories; mi_memfs_set_clock moves it).
 *   mi_memfs_update_from_entries  UpdateFromTarReader with untar = false on a layer's entries (mi_tar_entries): the
 *                             per-header filter (shouldSkip with the blacklist, IsMounted), hard links in a second
 *                             pass, *n_merged = "Merged %d headers from tar to memfs".  A merged node's source is
 *                             filepath.Join(root, name) -- the reference passes AbsPath(name), the same path under
 *                             the root "/" of every real build; under another root isOnDisk must look below it.
 *   mi_memfs_add_layer_by_scan    createLayerByScan on a walk of the root (mi_tree_walk / mi_batch_add_tree with
 *                             MI_TREE_SCAN, rel_base = root, the same blacklist): every walked path through
 *                             maybeAddToLayer with createWhiteout -- changed paths with their ancestors, one whiteout
 *                             per deleted subtree (if the child's src is really gone).  roots / root_stride: chunk roots
 *                             by file_index, kept in the tree -- for changed paths with their new node, for unchanged
 *                             files that had none as the root they have now -- so that the NEXT scan's isUpdated is
 *                             content-aware (NULL = the reference's metadata-only rule).  mi_memfs_commit_layer with
 *                             a ctx does walk, GPU scan, this call and the layer tar in one.
 *   mi_memfs_add_layer_by_copy_ops   AddLayerByCopyOps (mem_fs.go:276-289): addToLayer per mi_copy_op against this tree.
 * Both return the layer in commit order as an mi_copy_layer (mi_copy_layer_entries: headers + the path each entry's
 * bytes are read from; a whiteout is an entry named ".wh.<x>" without content) and fold it into the tree.  A failing
 * call returns the error code, mi_memfs_error() the reference's message, and leaves the handle usable.
 *   mi_memfs_entries          the tree in sorted-path order (src_paths may be NULL; cap 0 sizes).
 *   mi_memfs_reset            MemFS.Reset: the tree is emptied, the root stays.
 * A handle is not re-entrant (the reference serialises MemFS with one mutex); handles are independent.  Host logic. */
typedef struct mi_memfs mi_memfs;
typedef struct mi_index mi_index;         /* the chunk index, below */
MI_CORE int  mi_memfs_create(const char* root, const char* const* blacklist, uint64_t n_blacklist, int64_t now_sec,
                     mi_memfs** out);
MI_CORE void mi_memfs_free(mi_memfs* fs);
MI_CORE const char* mi_memfs_error(const mi_memfs* fs);
MI_CORE int  mi_memfs_set_clock(mi_memfs* fs, int64_t now_sec);
MI_CORE int  mi_memfs_reset(mi_memfs* fs);
MI_CORE int  mi_memfs_update_from_entries(mi_memfs* fs, const mi_tree_entry* layer, uint64_t n_layer, uint64_t* n_merged);
MI_BLOCK int  mi_memfs_add_layer_by_scan(mi_memfs* fs, const mi_tree_entry* walked, uint64_t n, const void* roots,
                                uint64_t root_stride, mi_copy_layer** out, uint64_t* n_entries);
MI_BLOCK int  mi_memfs_add_layer_by_copy_ops(mi_memfs* fs, const mi_copy_op* ops, uint64_t n_ops, mi_copy_layer** out,
                                    uint64_t* n_entries);
MI_CORE int  mi_memfs_entries(const mi_memfs* fs, mi_tree_entry* out, const char** src_paths, uint64_t cap, uint64_t* n_out);

/* ---- the layer writer: tar framing + the two serial layer digests (host threads) ---------- *
 * step.tarAndGzipDiffs + MemFS.commitLayer (lib/builder/step/common.go:35-111,
 * lib/snapshot/mem_fs.go:424-433) behind the ABI: the shim hands over the layer's entries in
 * commit order (mi_entries_commit_order) and gets the DigestPair's numbers back, while the GPU
 * scans the same files' content (an mi_batch fed with the same paths).  Pipeline, one thread
 * per stage, 1 MiB blocks, every sink sees every block (stream.ConcurrentMultiWriter,
 * lib/stream/multi_writer.go:35-66):
 *     framer --tee--> SHA-256 of the tar stream                      (tarDigester, common.go:45)
 *               '---> gzip (zlib) --> out_fd + SHA-256 of the blob   (gzipper/gzipDigester, :44-52)
 * Entries: memLayer.createHeader's rules (lib/snapshot/mem_layer.go:152-190: Name = dst without
 * the leading "/", directories with a trailing "/", Uname/Gname empty, symlink target as given --
 * mi_tree_walk already root-trims it) written the way tario.WriteEntry / WriteHeader do
 * (lib/tario/write.go:28-68: mtime in whole seconds; regular files followed by EXACTLY size bytes
 * read from src_path -- io.CopyN; a shorter file is MI_ERR_IO; directories, symlinks and hard
 * links header-only; any other kind MI_ERR_INVALID "unsupported type").  The header bytes follow
 * Go's archive/tar Writer (USTAR, PAX when a field does not fit); byte parity with Go is
 * UNPINNED here (no Go toolchain) -- see csrc/mi_layer.hip.  mi_layer_finish writes
 * tar.Writer.Close's 1024-byte trailer: an empty layer is 1024 zero bytes, digest 5f70bf18...
 * (lib/docker/image/const_darwin.go:18).  Not thread-safe per layer; layers are independent. */
#define MI_GZIP_OFF     (-2)   /* no gzip leg: out_fd receives the tar itself                    */
#define MI_GZIP_DEFAULT (-1)   /* "default"; 0 = "no", 1 = "speed", 9 = "size" (lib/tario/gzip.go:26-43) */
typedef struct mi_layer mi_layer;
typedef struct {
    uint32_t struct_size;
    int32_t  gzip_level;       /* MI_GZIP_OFF, MI_GZIP_DEFAULT or 0..9.  At every level but 0 a 1 MiB
                                  block that will not compress (three 8 KiB samples of it through
                                  level 1) is stored, not searched: one valid member either way    */
    int32_t  out_fd;           /* where the layer blob is written; -1 = digests only           */
    uint32_t flags;            /* MI_LAYER_*                                                    */
} mi_layer_config;
/* The header's Mode field keeps the file-type bits of the entry's st_mode (0100755, 040755,
 * 0120777): what tar.FileInfoHeader produced up to Go 1.8 (Mode |= c_ISREG / c_ISDIR / c_ISLNK) and
 * what the Go-written layer tars the reference holds as fixtures carry (testdata/files/busybox/
 * 393ccd5c.../layer.tar).  Without the flag: the Go >= 1.9 rule -- permission bits + setuid / setgid
 * / sticky only -- which is what createHeader (lib/snapshot/mem_layer.go:152-154) hands to the
 * writer under the reference's Go 1.14 toolchain.  Everything else in the header is the same, and
 * with the flag this writer reproduces that fixture byte for byte (tests/test_host_layer.py).     */
#define MI_LAYER_MODE_WITH_TYPE 0x1u
typedef struct {
    uint8_t  tar_sha256[32];   /* DigestPair.TarDigest      (common.go:97)                      */
    uint8_t  gzip_sha256[32];  /* GzipDescriptor.Digest     (:98-102); zero with MI_GZIP_OFF   */
    uint64_t tar_bytes;
    uint64_t gzip_bytes;       /* GzipDescriptor.Size                                           */
    uint64_t n_entries;
} mi_layer_result;
MI_CORE int  mi_layer_config_default(mi_layer_config* cfg);
MI_BLOCK int  mi_layer_begin(const mi_layer_config* cfg, mi_layer** out);
/* e->relpath = the entry's dst path; src_path = where a regular file's bytes are read from.  A dst whose
 * base name carries the whiteout prefix ".wh." is written as a whiteout -- a zero header with only that name,
 * no content -- whatever the entry is (memLayer.addHeader, lib/snapshot/mem_layer.go:197-212).                */
MI_BLOCK int  mi_layer_add(mi_layer* layer, const mi_tree_entry* e, const char* src_path);
/* The same entry with its content taken from a staged batch: the header as mi_layer_add writes it, then file
 * `file_index` of `batch` -- the bytes the GPU scanned, read back from HBM (mi_batch_read_file) -- instead of a second
 * read of the path.  The tar then holds exactly the bytes the file's chunk root describes, whatever has happened to the
 * file since it was staged; e->size must be the staged size (MI_ERR_INVALID otherwise).  Entries without content
 * (directories, links, whiteouts by name) are written as by mi_layer_add.                                              */
MI_BLOCK int  mi_layer_add_batch_file(mi_layer* layer, const mi_tree_entry* e, mi_batch* batch, uint64_t file_index);
/* What this writer read from disk itself so far: files it opened, bytes it read (mi_layer_add with a src_path).        */
MI_DIAG int  mi_layer_io_counts(mi_layer* layer, uint64_t* files_opened, uint64_t* file_bytes_read);
/* whiteoutMemFile.commit (mem_layer.go:101-132): a zero header named <dir>/.wh.<base>.          */
MI_BLOCK int  mi_layer_add_whiteout(mi_layer* layer, const char* deleted_path);
MI_BLOCK int  mi_layer_finish(mi_layer* layer, mi_layer_result* out);
MI_BLOCK const char* mi_layer_error(mi_layer* layer);
MI_BLOCK void mi_layer_free(mi_layer* layer);
/* step.commitLayer (lib/builder/step/common.go:67-111) in one call on a MemFS handle: the step's layer by scan
 * (must_scan: the root is walked here, with the handle's blacklist) or by its copy operations, written through the layer
 * writer configured by cfg (tarAndGzipDiffs: tar framing, TarDigest, the gzip leg with its digest and size), folded into
 * the tree.  Neither a scan nor ops: "Nothing to do" -- *committed = 0, res untouched.  layer_out (may be NULL): the
 * layer's entries and source paths (mi_copy_layer_entries) and chunk roots (mi_copy_layer_roots); free with
 * mi_copy_layer_free.  Errors carry the reference's chain in mi_memfs_error ("failed to generate diff layer: write
 * diffs: ...").  MemFS.sync's one-second wait stays with the caller.
 *
 * ctx == NULL: the reference's commit, byte for byte -- tario.IsSimilarHeader decides what changed
 * (lib/tario/compare.go:24-120), the writer reads the changed files from disk (lib/tario/write.go:28-52).  Host logic.
 *
 * ctx != NULL: the CONTENT-AWARE commit -- the seam this library exists for (lib/snapshot/mem_fs.go:315-341,440-503 +
 * lib/builder/step/common.go:67-111 as one flow):
 *   1. the root (must_scan) or the ops' sources are walked and every regular file listed is staged into one batch of
 *      `ctx` while the walk goes on: each file is opened once and read once;
 *   2. the GPU cuts and hashes them (Gear CDC, SHA-256 per chunk, one chunk root per file);
 *   3. createLayerByScan / addToLayer run with the roots: a path is in the layer if its header changed (the reference's
 *      rule) OR its content did -- a same-size edit within the same second, invisible to the reference, is caught.  A
 *      file the tree holds without a root (merged from a base layer, committed with ctx == NULL) and whose header is
 *      unchanged takes the root it has now; from the next commit on its content is watched;
 *   4. the layer writer frames the tar; file content comes from HBM (mi_layer_add_batch_file): the tar holds the bytes
 *      the stored root describes, even if the file was written to after it was staged;
 *   5. with an index set (mi_memfs_set_index) the batch's chunk digests are added to it (mi_index_add_batch).
 * Steps 2-4 overlap: the scan runs on a thread of the library's own while the committing thread computes the layer and
 * frames the tar from the bytes that have landed; the diff waits for the scan only where a root DECIDES (a file the tree
 * holds with a root and an unchanged header).  A file that vanishes or shrinks between the walk and its staging therefore
 * fails the commit AFTER the tree took the layer -- as a failing tar write does in the reference (MemFS.AddLayerByScan
 * updates the tree, then writes; lib/snapshot/mem_fs.go:260-274).  MI_COMMIT_PIPELINE=0: one step after the other.
 * The handle keeps the batch (device memory sized by the largest commit so far) for its next commit; it belongs to
 * `ctx`: free the handle, or call mi_memfs_release_device, before mi_ctx_destroy.  A scanned tree larger than the device's
 * free memory is not refused: its roots are computed in windows (runs of files the device has room for; MI_COMMIT_WINDOW_MB)
 * and the writer reads the layer's files from disk -- a second read for those, mi_commit_stats.n_windows says so; copy ops
 * whose sources do not fit go the same way (planned again without a batch).  mi_memfs_commit_stats: what the last
 * commit did.                                                                                                            */
typedef struct {
    uint64_t n_walked;           /* paths the walk(s) listed                                                    */
    uint64_t n_scanned_files;    /* regular files staged and scanned on the GPU (0 with ctx == NULL)            */
    uint64_t scanned_bytes;
    uint64_t n_chunks;
    uint64_t n_layer_entries;    /* headers in the layer                                                        */
    uint64_t n_layer_files;      /* ... of them regular files with content                                      */
    uint64_t layer_file_bytes;
    uint64_t n_content_changed;  /* files IsSimilarHeader calls similar whose chunk roots differ: in the layer  */
    uint64_t n_roots_learned;    /* unchanged files that had no root in the tree and have one now               */
    uint64_t n_content_trusted;  /* MI_MEMFS_TRUST_CTIME: files that were not read again (their inode is as it was)  */
    uint64_t n_index_new, n_index_known;   /* mi_index_add_batch's counts (index set)                           */
    uint64_t index_new_bytes;    /* ... and the bytes of the chunks no earlier commit held: what a chunk-addressed
                                    store would have to take in for this commit                                 */
    uint64_t files_opened;       /* file descriptors whose content was read, by every thread of the library ... */
    uint64_t file_bytes_read;    /* ... and the bytes read from them, during this commit (process-wide counters:
                                    a commit running beside another one counts both)                            */
    uint64_t pipelined;          /* 1: the scan ran beside the diff and the tar writer (its time is not in the sum) */
    uint64_t n_windows;          /* 0 normally; k: the tree did not fit the device and was scanned in k windows (the
                                    layer's files were then read from disk by the writer: a second read for those)   */
    double   s_walk_stage;       /* walk (+ staging, which goes on behind it)                                   */
    double   s_scan;             /* end of staging + the GPU passes + the roots' way back (pipelined: on a thread
                                    of its own, overlapping s_diff and s_write)                                 */
    double   s_diff;             /* createLayerByScan / addToLayer + commit order                               */
    double   s_write;            /* tar framing, digests, gzip leg                                              */
    double   s_total;
    uint64_t n_verified_files;   /* layer files whose framed bytes were held against the sums taken where they were read
                                    (MI_FLAG_FILE_SUMS: every file the writer took out of HBM) ...                   */
    uint64_t verified_bytes;     /* ... and their bytes                                                          */
    uint64_t n_refetched;        /* 1 MiB chunks that differed and were right at the second fetch from HBM (a commit
                                    with a chunk that differs twice fails)                                         */
    uint64_t arena_bytes;        /* device memory behind the handle's arena after this commit ...                */
    uint64_t arena_pieces;       /* ... in how many pieces (the arena of a commit is an address range mapped piece by
                                    piece behind the walk: csrc/mi_arena.hip) ...                                  */
    uint64_t arena_moves;        /* ... and how often its base address changed during this commit -- each time the reader
                                    threads were drained and, until round 5, the arena copied.  0, unless the tree
                                    outgrew the address range (four times the first estimate, 32 GiB at least)      */
    uint64_t n_ctxs;             /* GPUs the commit ran on (mi_memfs_commit_layer_n; 1 otherwise, 0 with ctx == NULL) ... */
    uint64_t ctx_bytes_max, ctx_bytes_min;   /* ... and the bytes the fullest and the emptiest of them were handed        */
    uint64_t n_split_files;      /* files of 256 MiB and more (MI_COMMIT_SPLIT_MIB) that were split over the GPUs as parts     */
} mi_commit_stats;
MI_CORE int  mi_memfs_commit_layer(mi_memfs* fs, mi_ctx* ctx, int must_scan, const mi_copy_op* ops, uint64_t n_ops,
                           const mi_layer_config* cfg, mi_layer_result* res, mi_copy_layer** layer_out, int* committed);
/* The same commit over n_ctx GPUs of one node, one ctx each (north_star: "file batches shard across the 8 GPUs"): what the walk
 * hands over goes to the GPU with the fewest bytes so far, every GPU stages through its own reader threads and PCIe link and scans
 * its share; roots come back in the walk's order, the tar writer reads each file from the GPU that holds it, the chunk index (on
 * any one of the ctxs) takes the other GPUs' digests through the host.  Layer, roots and DigestPair are those of the one-GPU
 * commit, byte for byte.  n_ctx = 1: mi_memfs_commit_layer; n_ctx = 0: the reference's commit.  A handle keeps its batches
 * between commits as long as it is called with the same ctxs.  A file of 256 MiB and more (MI_COMMIT_SPLIT_MIB) is split over the
 * GPUs as parts (the parts protocol below: a halo of 256 KiB per boundary is read twice, the file is opened once per part); its root
 * is mi_chunk_root over the parts' digests, its bytes are checked chunk by chunk like everybody's.  UNMEASURED on more than one
 * physical GPU (tested with n ctxs on one device, and on the HIP double): mi_commit_stats.n_ctxs / ctx_bytes_max / _min.        */
MI_CORE int  mi_memfs_commit_layer_n(mi_memfs* fs, mi_ctx* const* ctxs, uint32_t n_ctx, int must_scan, const mi_copy_op* ops,
                             uint64_t n_ops, const mi_layer_config* cfg, mi_layer_result* res, mi_copy_layer** layer_out,
                             int* committed);
MI_CORE int  mi_memfs_commit_stats(const mi_memfs* fs, mi_commit_stats* out);
/* Options of a handle's content-aware commits (default: none).
 * MI_MEMFS_TRUST_CTIME: a scan commit does not read a regular file again whose inode is what it was when the tree's root for
 * it was computed -- same device, inode number, size, mtime and ctime, to the nanosecond.  ctime is the kernel's own record
 * of the last change of an inode and cannot be set from user space (utimes sets it to "now"), so the same-size same-SECOND
 * rewrite the reference misses still changes it.  What git calls "racily clean" is handled as git does: a file whose ctime is
 * not safely older than the moment its content was read (MI_TRUST_CTIME_SLACK_MS, default 20: the kernel's timestamps tick
 * every 1-4 ms; two seconds for a ctime without a sub-second part -- a file system that keeps whole seconds) is read again.
 * With the option a commit that changed nothing costs a walk and a diff (the reference's price) instead of a read of the
 * whole tree; the layer, the roots and the DigestPair are the same unless the kernel's timestamps lie (a clock set back
 * between a write and the next one to the same file).  Scan commits only.                                               */
#define MI_MEMFS_TRUST_CTIME 0x1u
MI_CORE int  mi_memfs_set_options(mi_memfs* fs, uint32_t options);
/* From now on every content-aware commit of this handle adds its batch to `index` (NULL: stop).  The index must belong
 * to the ctx the commits run on and outlive them; the handle does not own it.                                          */
MI_CORE int  mi_memfs_set_index(mi_memfs* fs, mi_index* index);
/* The handle's batch ahead of its first content-aware commit, with room for `files` files of `bytes` bytes in total: a
 * ctx's first use costs (the reader threads: 40-55 ms; device memory: by the byte on some boxes, 47-68 ms per
 * GiB at every allocation) -- a host that knows what is coming, e.g. the size of the base image it is pulling, pays them beside its own work.
 * Optional.                                                                                                              */
MI_CORE int  mi_memfs_reserve_device(mi_memfs* fs, mi_ctx* ctx, uint64_t files, uint64_t bytes);
/* Gives back the batch a content-aware commit left with the handle (its arena holds the scanned tree's bytes).          */
MI_CORE int  mi_memfs_release_device(mi_memfs* fs);
/* The chunk root the tree holds for `path` ("/"-rooted, relative to the handle's root): *has_root = 0 when the path was
 * never scanned; MI_ERR_INVALID when the tree does not hold the path.                                                  */
MI_CORE int  mi_memfs_root_of(const mi_memfs* fs, const char* path, uint8_t* root_out, int* has_root);
/* The chunk roots of a layer's entries, in mi_copy_layer_entries' order: roots = n x 32 bytes, has_root = n flags
 * (regular files of a content-aware commit carry one; everything else zeros).                                          */
MI_CORE int  mi_copy_layer_roots(const mi_copy_layer* layer, uint8_t* roots, uint8_t* has_root, uint64_t cap);
/* The header block(s) mi_layer_add would write for `e` (512 bytes, or 1536+ with a PAX record) in
 * a layer begun with `layer_flags` (MI_LAYER_*).                                                 */
MI_BLOCK int  mi_layer_header_bytes(const mi_tree_entry* e, uint32_t layer_flags, uint8_t* out, uint64_t cap, uint64_t* n);

/* ---- cache entry codec (cache.Manager seam, lib/cache/cache_manager.go:34-35,239-252) -------- *
 * key   = "makisu_builder_cache_" + cacheID;
 * entry = "<tarHex>,<gzipHex>" (createEntry), or "MAKISU_CACHE_EMPTY" for a step that produced no
 *         layer (pass NULL, NULL); parse is parseEntry + PullCache's empty-entry case: *is_empty
 *         = 1 means "cached, and there is no layer"; an entry without "," is MI_ERR_INVALID
 *         (the reference wraps it as ErrorLayerNotFound).                                      */
MI_CORE int mi_cache_key(const char* cache_id, char* out, uint64_t cap);
MI_CORE int mi_cache_create_entry(const uint8_t* tar_sha256, const uint8_t* gzip_sha256, char* out, uint64_t cap);
MI_CORE int mi_cache_parse_entry(const char* entry, int* is_empty, uint8_t* tar_sha256, uint8_t* gzip_sha256);
/* parseEntry to the letter (cache_manager.go:239-245): the two halves around the FIRST comma, each
 * behind "sha256:", whatever they contain -- the reference does not validate them (the strict form
 * above needs two 64-digit hex halves because it hands out raw digests).  MI_ERR_INVALID without a
 * comma, MI_ERR_CAPACITY if a half does not fit.                                                 */
MI_CORE int mi_cache_parse_entry_str(const char* entry, char* tar_digest, uint64_t tar_cap, char* gzip_digest,
                             uint64_t gzip_cap);

/* ---- standalone digests (image.Digester seam) ---------------------------------- *
 * n independent byte strings -> n SHA-256 digests: the batched form of
 * image.NewDigester().FromBytes / FromReader (lib/docker/image/digester.go:45-60;
 * `makisu push`: bin/makisu/cmd/push.go:207,230).  data: host pointer, string i =
 * data[offsets[i] .. offsets[i]+lens[i]).  out: n x 32 bytes.
 * Many short strings (manifests, configs, small files): one GPU lane each.  Long ones -- a
 * `docker save` layer tar -- on host threads with SHA-NI, one stream per thread, straight from
 * `data`, beside the GPU's launch: one lane does 13.5 MB/s, one core 2.4 GB/s, and a
 * Merkle-Damgard stream cannot be split ("must stay on a CPU core").
 * route_long_strings (csrc/mi_stage.hip) sends the length classes to the host that make
 * max(host time, GPU time) smallest: eight 128 MiB blobs take 0.06-0.12 s on eight cores
 * instead of 10 s on eight lanes; 100 000 x 64 KiB stay on the GPU.  MI_SHA_HOST_THREADS
 * (default: the cores the process may use, at most 16); MI_SHA_LONG_ON_GPU=1 keeps every
 * string on a lane (tests).                                                                */
MI_CORE int mi_sha256_many(mi_ctx* ctx, const void* data, const uint64_t* offsets,
                   const uint64_t* lens, uint64_t n, uint8_t* out);

/* ---- chunk index: dedup across batches (keyvalue.Store seam) -------------------- *
 * A device-resident set of chunk digests that outlives a batch.  The reference
 * remembers earlier work as content-addressed layers (IsExist + CAS link,
 * lib/builder/step/common.go:88-91) and as cacheID -> "tarHex,gzipHex" strings
 * behind keyvalue.Store (lib/cache/keyvalue/store.go:22-26, written at
 * lib/cache/cache_manager.go:239-252); this is the chunk-granular analogue:
 * "which chunks of this layer did an earlier layer already hold?".
 *
 * mi_index_add_batch: for every chunk of a batch that has run, known[i] = 1 if its
 * digest was in the index BEFORE the call (an in-batch repeat of a new digest stays
 * 0 - dup_of already says so), then the new digests are added.  known may be NULL.
 * mi_index_export / mi_index_import move the set as a flat blob of 32-byte digests
 * (order unspecified) so the shim can keep it behind keyvalue.Store.Put/Get.      */
MI_CORE int  mi_index_create(mi_ctx* ctx, uint64_t capacity_hint, mi_index** out);
MI_CORE void mi_index_free(mi_index* index);
MI_CORE int  mi_index_count(mi_index* index, uint64_t* n_digests);
MI_BLOCK int  mi_index_add_batch(mi_index* index, mi_batch* b, uint8_t* known, uint64_t cap,
                        uint64_t* n_new, uint64_t* n_known);
MI_CORE int  mi_index_export(mi_index* index, void* out_digests, uint64_t cap_digests);
MI_CORE int  mi_index_import(mi_index* index, const void* digests, uint64_t n, uint64_t* n_new);

#ifdef __cplusplus
}
#endif
#endif /* MAKISU_MI_H */
