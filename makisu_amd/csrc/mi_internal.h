// mi_internal.h -- engine-internal state shared by the host translation units
// (mi_api.hip: pipeline and C ABI; mi_comm.hip: RCCL digest exchange).  Not installed.
#pragma once

#include "../../include/makisu_mi.h"
#include "mi_common.h"
#include "mi_local.h"
#include "mi_filesum.h"

#include <deque>
#include <memory>

#include <mutex>
#include <set>
#include <string>
#include <vector>

namespace mi {

// mi_alloc.hip: hipMalloc / hipFree, or -- MI_GUARD_ALLOC=1, the over-read audit -- allocations that end on an
// unmapped page, sized exactly as asked
bool       guard_alloc();
hipError_t dev_alloc(void** p, size_t bytes);
hipError_t dev_free(void* p);

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t want) {
        if (want <= bytes) return hipSuccess;
        if (p) { (void)dev_free(p); p = nullptr; bytes = 0; }
        // 256 bytes of slack behind every buffer: the hashing kernels read up to 67 bytes past a string's end
        // (sha256.hip); one eighth on top so that a buffer that grows does not grow every time (not under the guard)
        size_t alloc = guard_alloc() ? ((want + 255) & ~(size_t)255) + 256 : want + want / 8 + 256;
        hipError_t e = dev_alloc(&p, alloc);
        if (e == hipSuccess) bytes = alloc;
        return e;
    }
    void release() { if (p) (void)dev_free(p); p = nullptr; bytes = 0; }
    template <typename T> T* as() const { return (T*)p; }
};

// The arena of a batch (mi_arena.hip): a reserved address range whose front is MAPPED PIECE BY PIECE by a thread of its own, so that
// an arena that grows never moves (no drain of the reader threads, no device-to-device copy of what it holds) and nobody waits
// for device memory that is not needed yet: [0, bytes) is PROMISED -- mapped, or about to be; whoever writes or reads device
// memory at arena offset x first waits for arena_wait_mapped(x).  Under MI_GUARD_ALLOC (the over-read audit) and with
// MI_ARENA=malloc (A/B measurements) the arena is one dev_alloc that moves when it grows, as it was until round 5 (vm == nullptr).
struct ArenaVm;
struct Arena {
    void* p = nullptr;
    size_t bytes = 0;
    ArenaVm* vm = nullptr;
    u64 moves = 0;            // times the arena's base address changed while it held bytes (plain: every growth; piecewise: a range outgrown)
    template <typename T> T* as() const { return (T*)p; }
};
bool arena_is_plain();                                      // MI_GUARD_ALLOC or MI_ARENA=malloc: the moving arena
bool arena_always_pieces();                                 // MI_ARENA=pieces (A/B runs): batches that are told their size are piecewise too
u64  arena_piece_bytes(const Arena* a);                     // 0: one allocation
// the promise grows to `want` bytes.  outgrown (see arena_outgrown): the caller has drained everything that targets the arena
int  arena_promise(mi_ctx* c, Arena* a, u64 want, bool* no_addresses = nullptr);   // *no_addresses: the FIRST reservation found no address range
bool arena_outgrown(const Arena* a, u64 want);              // `want` does not fit the reserved range: the pieces move to a larger one
int  arena_wait_mapped(mi_ctx* c, Arena* a, u64 upto, std::string* msg = nullptr);   // MI_ERR_NOMEM when the device ran out under the mapper
                                                            // (msg: where the message goes instead of the ctx)
void arena_release(Arena* a);                               // unmaps and frees the pieces; the address range is retired, not reused
void arena_counts(const Arena* a, u64* mapped, u64* pieces, u64* reserved);

struct SynthSpec { u64 f0, n, seed; std::vector<u64> cids; u64 unit0 = 0; };   // unit0: first 16-byte unit (parts)

// A part: bytes [begin, end) of a file that is split across batches / GPUs, staged behind a halo of
// whole 256 KiB groups (gear_cdc.hip "parts").  Offsets with _rel are relative to the item's first
// byte, i.e. to file offset begin - halo_bytes.
struct PartRec {
    u64 file_index, file_size, begin, end, halo_bytes;
    u32 halo_groups = 0, group0 = 0, n_groups = 0;
    u64 entry_rel = 0, exit_rel = 0;   // as of the last read-back (scan / fix)
    u64 set_entry_rel = ~0ull;         // what the caller confirmed (~0: nothing yet)
    bool confirmed = false;
};

// sets the ctx's (or, for c == nullptr, the create-time) error message; returns code
int fail(mi_ctx* c, int code, const char* fmt, ...);

// One host->device copy of a host-fed batch, with the sums of the bytes it carried
// (MI_FLAG_VERIFY_STAGING): words w_0..w_{n-1} = the span as little-endian u64, the tail
// zero-padded; s1 = sum w_i, s2 = sum (n - i) w_i (mod 2^64) -- what `s1 += w; s2 += s1` leaves.
struct StageSpan { u64 off, len, s1, s2; u32 thread; };
constexpr u32 kStageInlineThread = 0xFFFFu;        // the calling thread's inline window
void stage_sum_host(const void* p, u64 len, u64* s1, u64* s2);
// device sums of n spans of `base` (offsets 8-byte aligned) into d_out[2 * i .. +2), zeroed here
void launch_stage_sums(const u8* base, const u64* d_off, const u64* d_len, u64 n, u64 max_len, u64* d_out,
                       hipStream_t s);
// every recorded span of the batch summed again where it lies now (after the readers drained, after
// any arena growth); MI_ERR_IO naming the first span that differs
int stage_verify_final(mi_batch* b);

// mi_stage.hip: reader threads + pinned slabs behind mi_batch_add_path / large mi_batch_add_bytes
struct Stager;
Stager* stager_create(mi_ctx* c, u32 n_threads, u64 slab_bytes);   // returns at once: the readers set up behind it
bool    stager_ready(Stager* st);                                   // waits for them; false: none got slab + stream
bool stager_ready_all(Stager* st);
void    stager_destroy(Stager* st);
int     stager_put_bytes(Stager* st, mi_batch* b, u64 arena_off, const void* src, u64 len, mi_sum::FileSum* sums);
int     stager_put_file(Stager* st, mi_batch* b, u64 arena_off, int fd, u64 file_off, u64 len, const char* path, mi_sum::FileSum* sums,
                        u64 row_off0 = 0);      // row_off0: where on the sums' grid the first byte lies (a part: origin mod 1 MiB)
int     stager_put_paths(Stager* st, mi_batch* b, u64 n, const char* const* paths, const u64* arena_off,
                         const u64* len, mi_sum::FileSum* const* sums);   // sums: NULL, or one pointer per file (the row's chunk sums)
int     stager_put_block(Stager* st, mi_batch* b, u64 arena_off, const void* src, u64 len, std::shared_ptr<void> keep);
int     stager_drain(Stager* st, mi_batch* b);
void    stager_pause(Stager* st, int ms);            // up to `ms` milliseconds, or until a run has landed
// blocks until every byte of the batch's arena below `upto` that was queued so far has landed in HBM (pieces of one batch are
// queued in arena order), or the batch's staging failed (MI_ERR_IO with the first failure's message)
// *landed_out (optional): how far the landed prefix reaches by now (~0: everything queued so far)
int     stager_wait_landed(Stager* st, mi_batch* b, u64 upto, u64* landed_out);
u64     stager_landed(Stager* st, mi_batch* b);      // the same number, now, without waiting
// Whole-string SHA-256 of n ranges of the batch's arena on the reader threads (each a SHA-NI stream over its pinned slab, the
// next piece on its way out of HBM while one is hashed): what the hashing pass hands over when a string is too long for a GPU
// lane (route_long_strings).  Returns at once; wait with stager_hash_wait (MI_ERR_HIP / the first failure's message on the ctx).
struct HashLatch;
HashLatch* stager_hash_ranges(Stager* st, mi_batch* b, u64 n, const u64* arena_off, const u64* len, u8* out32);
int        stager_hash_wait(mi_ctx* c, HashLatch* latch);      // frees the latch

// Which of n strings go to host SHA-NI streams instead of GPU lanes?  One lane hashes 13.5 MB/s (the pass's 1.77 TB/s over
// 131 072 lanes), one host core 2.2 GB/s: a pass is as long as its longest string on ONE lane unless there are enough strings to
// keep every lane busy anyway.  The strings are looked at as classes of equal length, longest first; the classes that leave the
// GPU are those for which max(host time, GPU time of the rest) is smallest.  to_host: indexes, longest first (empty: nothing).
// host_threads: streams the host side can run at once; h2d: the rest has to cross PCIe first (mi_sha256_many)
void route_long_strings(const u64* lens, u64 n, u32 host_threads, bool h2d, std::vector<u32>* to_host);

}  // namespace mi

struct mi_ctx {
    mi_config cfg;
    int device = 0;
    hipDeviceProp_t prop;
    hipStream_t stream = nullptr;        // ctx-level work (mi_dedup_mark, mi_sha256_many, uploads)
    mi::Stager* stager = nullptr;        // reader threads + pinned slabs, created on the first host-fed add
    bool stager_checked = false;         // ... and found able to take work (stager_ready)
    mi::u32 stage_threads = 0;
    size_t staging_bytes = 0;            // bytes per pinned slab (reader threads, per-batch inline ring)
    int live_children = 0;               // batches + indexes that still point at this ctx
    int batches_in_flight = 0;           // submitted, not yet waited for
    mi::DevBuf gear_table, heads, crc_consts;
    mi::DevBuf dd_table, dd_slot, dd_nuniq;             // dedup scratch of mi_dedup_mark
    mi::DevBuf dd_tag;                                  // ... and of mi_dedup_mark_range
    mi::u64* h_word = nullptr;           // pinned: small read-backs on the ctx stream
    hipEvent_t ev[2];
    hipEvent_t sha_done = nullptr;       // end of the last chunk pass submitted on this ctx, whatever the batch
    bool sha_done_set = false;           // (the next one waits for it: mi_api.hip submit_pipeline)
    bool serialize_sha = false;          // MI_SHA_SERIALIZE=0: let the chunk passes of two batches overlap
    mi::ShaTune sha;                     // per ctx (mi_config.sha_*), not per process
    std::string sha_wave_stats;          // mi_debug_sha_wave_stats: the file sha.wave_stats_path points at
    bool verify_staging = false;         // MI_FLAG_VERIFY_STAGING
    // fault injection for the tests of that flag (MI_STAGE_FAULT=copy:N | final:N): the N-th span a
    // reader copies loses 4 KiB right after its copy / just before the end-of-staging pass
    long long fault_copy = -1, fault_final = -1;
    // ... and MI_STAGE_FAULT=readback:N[:K]: the N-th (.. N+K-1-th) copy into a read-back window arrives with one byte flipped
    long long fault_readback = -1, fault_readback_n = 1;
    bool file_sums = false;              // MI_FLAG_FILE_SUMS: batches of this ctx keep their files' source sums
    mi::CdcParams cdc;
    void* comm = nullptr;                // ncclComm_t when mi_comm_init_* was called (mi_comm.hip)
    int comm_rank = 0, comm_nranks = 1;
    void* comm_scratch = nullptr;        // exchange buffers (mi_comm.hip)
    std::string err;
    mi_stats stats;
};

struct mi_batch {
    mi_ctx* ctx;
    struct FileRec { mi::u64 off, size, tag; int part = -1; mi_sum::FileSum* sums = nullptr; mi::u64 origin = 0; };   // part: index into `parts`; sums: per 1 MiB
                                                                   // chunk, taken where the bytes were read (keep_sums); origin: the FILE
                                                                   // offset of the row's first staged byte (a part: begin - halo; else 0) --
                                                                   // sums[] lie on the file's 1 MiB grid: sums[k - origin / 1 MiB] is chunk k
    std::vector<FileRec> files;
    std::vector<mi::SynthSpec> synth;
    std::vector<mi::PartRec> parts;          // split files (mi_batch_add_*_part)
    mi::DevBuf file_flags, part_file, part_group0, part_halo, part_entry;
    bool cuts_ready = false;                 // mi_batch_scan_cuts ran: the next submit reuses its cuts
    bool parts_dirty = false;                // a confirmed entry differs from the one the cuts were made with
    mi::u64 total_bytes = 0;     // sum of sizes
    mi::u64 arena_used = 0;      // next free arena offset
    // A GROUP HEAD (mi_batch_group_begin; mi_memfs_commit_layer_n): no arena, no stream -- a handle behind which the file rows of a
    // walk are spread over one batch per ctx (one per GPU), each block / file going to the member with the fewest bytes so far.  The
    // walk and the commit see ONE batch: group row g is row row_row[g] of members[row_member[g]].
    std::vector<mi_batch*> members;
    std::vector<mi::u32> row_member;
    std::vector<mi::u64> row_row;
    std::vector<mi::u64> member_bytes;
    // a file of MI_COMMIT_SPLIT_MIB (256) MiB and more is SPLIT over the members as parts (mi_batch_add_path_part: the parts protocol):
    // group row g with row_member[g] == kGroupSplit is splits[row_row[g]]
    struct SplitPart { mi::u32 member; mi::u64 row, begin, end; };
    struct Split { mi::u64 size; std::vector<SplitPart> parts; };
    std::vector<Split> splits;
    mi::Arena arena;
    bool keep_sums = false;      // every host-fed file row carries the sums of its bytes as they were READ (mi_filesum.h): what the layer
    mi_sum::Pool sum_pool;       // writer checks the bytes it frames against (MI_FLAG_FILE_SUMS; always for a MemFS handle's batch)
    bool arena_plain = false;    // decided when the arena is first made (mi_api.hip arena_reserve): one allocation, or piecewise
    // inline staging window for small mi_batch_add_bytes calls: the batch's OWN two pinned slabs
    // (two batches may be filled at the same time), copied on the batch's own copy stream
    void* ring[2] = {nullptr, nullptr};
    hipEvent_t ring_ev[2] = {nullptr, nullptr};
    hipStream_t ring_stream = nullptr;
    int cur = 0;             // slab being filled
    mi::u64 win_start = 0;       // arena offset the current slab maps to
    mi::u64 win_fill = 0;        // bytes valid in it
    bool staged_any = false;
    // reader-thread staging (mi_stage.hip); guarded by the stager's mutex
    mi::u64 stage_pending = 0;   // queued pieces not yet in HBM
    std::multiset<mi::u64> stage_inflight;   // arena offsets at which the runs a reader thread holds right now begin
    std::deque<mi::u64> stage_queued;        // arena offsets of this batch's queued pieces, in queue order -- ascending (every adder
                                             // enqueues in arena order; stage_unordered is set if one ever does not): the front is
    bool stage_unordered = false;            // the lowest offset still queued, which is what stager_wait_landed asks for
    int stage_waiters = 0;       // threads in stager_wait_landed (every landed run wakes them, not just the last)
    std::string stage_err;       // first read / copy / verification error: STICKY until mi_batch_reset
    std::string stage_note;      // what the first verification mismatch looked like (even if repaired)
    std::mutex span_mu;          // guards the three below (reader threads + the inline window)
    std::vector<mi::StageSpan> stage_spans;   // MI_FLAG_VERIFY_STAGING: every copy, for the final pass
    mi_stage_stats stage_stats;
    mi::DevBuf span_off, span_len, span_sums;
    double ms_h2d = 0;
    // pipeline state: every batch owns a stream, so two batches can be in flight and the
    // Gear pass of one overlaps the SHA pass of the other (they bind different units)
    hipStream_t stream = nullptr;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    mi::u64* h_counts = nullptr;             // pinned: [0] = chunk count, [1] = unique count
    bool staged = false, in_flight = false, ran = false, results_valid = false;
    mi::u64 n_chunks = 0, total_slots = 0;
    mi_stats stats;
    // CDC segments (gear_cdc.hip): a small file or one 256 KiB group of a large file
    mi::DevBuf small_list;                   // segments that are files <= one tile (wave per file)
    mi::DevBuf dense_list;                   // ... of them, those the bitmap-free kernel could not list (+ count)
    mi::DevBuf seg_file, seg_slot, seg_n, seg_first, seg_group, file_seg0, ends32;
    mi::DevBuf group_file, group_index, group_recs, tile_lists, tile_fast, large_list, large_group0;
    mi::u32 n_small = 0, n_groups = 0, n_large = 0;
    mi::u64 n_segs = 0, ends_total = 0;
    mi::DevBuf file_off, file_size, cids, n_chunks_d, first, scratch;
    mi::DevBuf ctl;                          // control block: counts, queue heads, bin cursor, histogram
    mi::DevBuf chunk_off, chunk_len, chunk_file, chunk_start, digests;
    mi::DevBuf q_off, q_len, q_id;           // SHA queue descriptors, longest chunk first
    mi::DevBuf item_off, item_len, roots, file_sha, dup_of;
    // MI_FLAG_FILE_SHA256: files too long for a GPU lane are hashed by the reader threads (route_long_strings)
    std::vector<mi::u32> fsha_host;              // their rows, longest first (decided when the batch is staged)
    std::vector<mi::u8> fsha_host_out;           // 32 bytes each, as of the last run
    mi::DevBuf fsha_len;                         // the GPU pass's lengths: 0 for those rows
    mi::HashLatch* fsha_latch = nullptr;         // the host side of a run in flight
    mi::DevBuf root_addr, root_cnt, rseg_cnt, rseg_first, rseg_total, root_items_off, root_items_len;
    mi::DevBuf root_level[mi::kMaxRootPasses];   // node digests of the reduction passes
    mi::DevBuf root_addr2, root_cnt2;    // ping-pong partner of root_addr / root_cnt
    int root_passes = 0;                 // reduction passes this batch can need (from max file size)
    mi::u64 max_file_size = 0;
    mi::DevBuf tile_file, first_tile, tile_raw, crc_d;   // MI_FLAG_FILE_CRC32
    mi::u64 n_tiles = 0;
    void* tree = nullptr;                // host-side walk record (mi_tree.hip)
    mi::DevBuf dd_table, dd_slot;
    mi_file_result* h_files = nullptr;   // the file rows, packed on the device, in pinned host memory (n_h_files valid rows)
    size_t n_h_files = 0, h_files_cap = 0;
    mi::DevBuf file_rows_d;
    mi::DevBuf rows_d, file_base;        // chunk rows packed on the device; per-file offset base (parts)
    void* rows_h = nullptr;              // ... and in pinned host memory (what mi_batch_chunks_view hands out)
    size_t rows_h_bytes = 0;
    // mi_batch_read_file: the pinned window staged bytes come back through (the layer writer's source when a commit
    // reads its files from HBM instead of a second time from disk)
    hipStream_t rb_stream = nullptr;     // ... on a stream of its own: a read-back does not queue behind the batch's kernels
    // two windows: the one reads are served from, and the one the NEXT range is on its way into while a streaming reader (the
    // tar writer) consumes the first -- the writer then never waits for a copy, also while eight reader threads keep PCIe busy
    struct ReadWin { void* p = nullptr; mi::u64 start = 0, len = 0; bool pending = false; hipEvent_t ev = nullptr; };
    ReadWin rb[2];
    int rb_cur = 0;
    mi::u64 rb_next = 0;                 // next copy's length: doubles while reads continue where the last window ended, else back to the minimum
    mi::u64 rb_fetches = 0, rb_bytes = 0;
    mi::u64 rb_copies = 0;               // copies into the windows since the ctx was made... per batch (fault injection counts them)
    double rb_wait_s = 0, rb_fetch_s = 0;   // waited for bytes to land / for the window's copies (MI_LAYER_TIMING)
    std::vector<mi::u8> h_roots;         // mi_batch_roots: the 32 bytes per file isUpdated needs, nothing else
    bool h_roots_valid = false;
};



#define HIPCHK(c, call)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return mi::fail((c), e_ == hipErrorOutOfMemory ? MI_ERR_NOMEM : MI_ERR_HIP,      \
                            "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                            __LINE__);                                                      \
    } while (0)
