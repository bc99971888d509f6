// mi_api.hip -- host pipeline behind the C ABI of include/makisu_mi.h.
//
// Data layout in HBM (DESIGN.md "HBM layout"):
//   arena      : all file bytes of a batch, every file on a 256-byte boundary
//   file table : file_off[], file_size[], file_seg0[]            (u64 SoA)
//   segments   : a small file or one 256 KiB group of a large file; ends32[seg_slot[s] ..] = its
//                chunk END offsets (u32, relative to the segment start), seg_n[s] their count
//   chunk table: chunk_off[] (arena offset), chunk_len[], chunk_start[] (u64),
//                chunk_file[] (u32), digests[] (32 B)
//   SHA queue  : q_off[], q_len[] (u64), q_id[] (u32): chunk descriptors, longest first
//   file out   : roots[] (32 B), file_sha[] (32 B, optional), crc[] (u32, optional)
// One HIP stream per BATCH for its pipeline (two batches may be in flight), one per ctx for
// ctx-level work, n_streams copy streams for staging.
// There is no CPU fallback anywhere in this file.
#include "mi_internal.h"
#include "mi_hostpath.h"      // mi_io
#include "host_sha256.h"      // mi_sha256_many: strings too long for a GPU lane

#include <errno.h>
#include <fcntl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/stat.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

using namespace mi;

namespace {

std::mutex g_err_mu;
std::string g_create_err;

}  // namespace

namespace mi {

int fail(mi_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    // (one writer at a time: a pipelined commit has two threads of the library on one ctx -- its scan and its tar writer --
    //  and a staging failure reaches both)
    std::lock_guard<std::mutex> g(g_err_mu);
    if (c) c->err = buf; else g_create_err = buf;
    return code;
}

}  // namespace mi

namespace {

u64 align_up(u64 v, u64 a) { return (v + a - 1) / a * a; }

// per-batch control block (device): see submit_pipeline
constexpr size_t kCtlHeadsOff = 16;
constexpr size_t kCtlCursorOff = kCtlHeadsOff + 8 * kShaHeadWords * sizeof(u32);
constexpr size_t kCtlHistOff = kCtlCursorOff + 1024 * sizeof(u32);
constexpr size_t kCtlRolesOff = kCtlHistOff + 1024 * sizeof(u32);       // SIMD arrival counters: chunk pass | file pass
constexpr size_t kCtlBytes = kCtlRolesOff + 2 * kShaRoleWords * sizeof(u32);

// ---- arena + staging ------------------------------------------------------------
// Two ways into the arena: small mi_batch_add_bytes calls are copied inline into the batch's own
// pinned window (two slabs, the copy of one in flight while the other fills); files and large
// buffers go through the ctx's reader threads (mi_stage.hip).
int staging_sync(mi_batch* b) {
    mi_ctx* c = b->ctx;
    if (b->ring_stream) HIPCHK(c, hipStreamSynchronize(b->ring_stream));
    if (c->stager) return stager_drain(c->stager, b);
    return MI_OK;
}

// Room for `want` bytes.  TWO KINDS OF ARENA, decided when a batch's arena is first made:
//  * PIECEWISE (mi_arena.hip): a reserved address range mapped piece by piece by a thread of its own.  Growing it moves a
//    number and nobody waits here; whoever touches device memory waits for the mapper where it touches (arena_wait_mapped).
//    For batches that LEARN their size as they go -- a tree walk, paths or buffers added without a hint: every content-aware
//    commit -- where an arena that moves would drain the reader threads and copy itself at every step, on the thread the
//    commit's tar writer is waiting for.  Only an arena that outgrows its whole address range is drained (its pieces are
//    mapped again elsewhere).
//  * PLAIN: one allocation (made again, larger, and copied if it has to grow after all) -- for batches that are TOLD their size
//    before their first byte (mi_batch_begin's hints, mi_batch_reserve, synthetic files): the headline configurations.  The
//    hashing kernel's lane-owned 64-byte loads touch 64 pages per wave instruction and run at the speed of the page-table
//    fragments behind them: one hipMalloc 4.13-4.15 ms per C2 chunk pass, pieces of 1 GiB 4.27, of 256 MiB 4.32, of 32 MiB
//    5.0-5.3, of 2 MiB 5.83 (profiles/r06_arena_ab.txt) -- a commit, bound by its 2.45 GB/s TarDigest, does not notice; the
//    headline would.  Also under MI_GUARD_ALLOC (the over-read audit needs the arena to END on an unmapped page) and with
//    MI_ARENA=malloc (A/B).  told: an eighth on top instead of a half.
// ahead: the caller is a walk whose enumeration runs ahead of what it hands over (mi_batch_reserve_ahead): told, and piecewise.
int arena_reserve(mi_batch* b, u64 want, bool told = false, bool ahead = false) {
    mi_ctx* c = b->ctx;
    want += 4096;                                   // slack: tile loads may touch 15 B past a file
    if (want <= b->arena.bytes) return MI_OK;
    if (!b->arena.p) b->arena_plain = arena_is_plain() || (told && !ahead && !arena_always_pieces());
    if (!b->arena_plain) {
        if (arena_outgrown(&b->arena, want)) {
            int rc = staging_sync(b);
            if (rc) return rc;
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        // Address ranges are never given back (mi_arena.hip): a process that has gone through thousands of walk-fed batches may
        // find none left.  Such a batch takes ONE allocation that moves when it grows, as every batch did until round 5.
        bool no_addresses = false;
        const int rc = arena_promise(c, &b->arena, want, &no_addresses);
        if (!no_addresses) return rc;
        static const bool say = [] { const char* v = getenv("MI_ARENA_TRACE"); return v && *v == '1'; }();
        if (say) fprintf(stderr, "mi_arena: no address range left (%s): this batch's arena is one allocation\n", mi_last_error(c));
        b->arena_plain = true;
    }
    int rc = staging_sync(b);                       // copies in flight target the old arena
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    u64 alloc = guard_alloc() ? ((want + 255) & ~255ull) : want + (told ? want / 8 : want / 2);   // under the guard: the 4 KiB and no more
    void* np = nullptr;
    static const bool trace = [] { const char* v = getenv("MI_ARENA_TRACE"); return v && *v == '1'; }();
    if (trace) fprintf(stderr, "mi_arena: %s %.1f MB -> %.1f MB (used %.1f)\n", told ? "told" : "grow", b->arena.bytes / 1e6, alloc / 1e6, b->arena_used / 1e6);
    hipError_t e = dev_alloc(&np, alloc);
    if (e != hipSuccess) { alloc = want; HIPCHK(c, dev_alloc(&np, alloc)); }
    if (b->arena.p && b->arena_used) {
        const u64 keep = b->arena_used < b->arena.bytes ? b->arena_used : b->arena.bytes;
        HIPCHK(c, hipMemcpy(np, b->arena.p, keep, hipMemcpyDeviceToDevice));
        ++b->arena.moves;
    }
    if (b->arena.p) (void)dev_free(b->arena.p);
    b->arena.p = np;
    b->arena.bytes = alloc;
    if (c->verify_staging) {
        // a byte that never arrives must not read as a plausible zero: fill what no copy has written
        // yet, and be done with it before the first host-to-device copy may target it
        const u64 keep = b->arena_used < alloc ? b->arena_used : alloc;
        HIPCHK(c, hipMemsetAsync((u8*)np + keep, 0xA5, alloc - keep, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return MI_OK;
}

// the inline window: buffers below kInlineBytes only, so 2 MiB per slab is plenty (pinned
// allocation is what makes a fresh batch expensive: ~3 ms per 8 MiB)
constexpr u64 kRingBytes = 2ull << 20;

int ensure_ring(mi_batch* b) {
    mi_ctx* c = b->ctx;
    if (b->ring[0]) return MI_OK;
    HIPCHK(c, hipStreamCreateWithFlags(&b->ring_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        HIPCHK(c, hipHostMalloc(&b->ring[i], kRingBytes, hipHostMallocDefault));
        HIPCHK(c, hipEventCreateWithFlags(&b->ring_ev[i], hipEventDisableTiming));
    }
    return MI_OK;
}

int ensure_stager(mi_ctx* c) {
    if (!c->stager) c->stager = stager_create(c, c->stage_threads, c->staging_bytes);
    if (!c->stager_checked) {
        if (!stager_ready(c->stager)) {
            stager_destroy(c->stager);
            c->stager = nullptr;
            return fail(c, MI_ERR_NOMEM, "host-fed staging: none of the %u reader threads could allocate its "
                        "%llu-byte pinned slab and copy stream", c->stage_threads, (unsigned long long)c->staging_bytes);
        }
        c->stager_checked = true;
    }
    return MI_OK;
}
// host-fed bytes are on their way (a tree walk has begun): the reader threads set up while the walk lists its first
// directories; the first block or path that reaches them waits for whoever is not ready yet
// ---- a group of batches behind one handle (mi_internal.h: members) -----------------------------------------------------------
static inline bool is_group(const mi_batch* b) { return !b->members.empty(); }
constexpr u32 kGroupSplit = 0xFFFFFFFFu;             // row_member of a file that is split over the members as parts
constexpr u64 kGroupAtShift = 48, kGroupAtMask = (1ull << kGroupAtShift) - 1;       // a group's "arena offset": member << 48 | offset
static int group_fail(mi_batch* h, size_t k, int rc) {
    std::string m;
    { std::lock_guard<std::mutex> g(g_err_mu); m = h->members[k]->ctx->err; }
    return mi::fail(h->ctx, rc, "gpu %zu of %zu: %s", k, h->members.size(), m.c_str());
}
static size_t group_least_loaded(const mi_batch* h) {
    size_t k = 0;
    for (size_t i = 1; i < h->members.size(); ++i) if (h->member_bytes[i] < h->member_bytes[k]) k = i;
    return k;
}

extern "C" void mi_batch_expect_host_bytes(mi_batch* b) {
    if (is_group(b)) { for (mi_batch* m : b->members) mi_batch_expect_host_bytes(m); return; }
    mi_ctx* c = b->ctx;
    if (!c->stager) c->stager = stager_create(c, c->stage_threads, c->staging_bytes);
}

int staging_flush(mi_batch* b) {
    mi_ctx* c = b->ctx;
    if (b->win_fill == 0) return MI_OK;
    {
        StageSpan sp{b->win_start, b->win_fill, 0, 0, kStageInlineThread};
        if (c->verify_staging) stage_sum_host(b->ring[b->cur], b->win_fill, &sp.s1, &sp.s2);
        std::lock_guard<std::mutex> g(b->span_mu);
        ++b->stage_stats.spans;
        b->stage_stats.bytes += b->win_fill;
        if (c->verify_staging) b->stage_spans.push_back(sp);   // summed on the GPU when staging ends
    }
    {
        const int rc = arena_wait_mapped(c, &b->arena, b->win_start + b->win_fill);
        if (rc) return rc;
    }
    HIPCHK(c, hipMemcpyAsync((u8*)b->arena.p + b->win_start, b->ring[b->cur], b->win_fill,
                             hipMemcpyHostToDevice, b->ring_stream));
    HIPCHK(c, hipEventRecord(b->ring_ev[b->cur], b->ring_stream));
    b->staged_any = true;
    b->cur ^= 1;
    HIPCHK(c, hipEventSynchronize(b->ring_ev[b->cur]));        // the slab we are about to reuse
    b->win_start += b->win_fill;
    b->win_fill = 0;
    return MI_OK;
}

// Appends `len` bytes of caller memory at arena offset `at` through the batch's pinned window.
int staging_append(mi_batch* b, u64 at, const u8* src, u64 len) {
    int rc = ensure_ring(b);
    if (rc) return rc;
    if (b->win_fill == 0) b->win_start = at;
    if (at != b->win_start + b->win_fill) {
        // only alignment padding may be bridged: a larger gap is a file the reader threads are
        // staging, and zero-filling across it would overwrite their bytes
        const u64 gap = at - (b->win_start + b->win_fill);
        if (at < b->win_start + b->win_fill || gap >= kFileAlign || b->win_fill + gap > kRingBytes) {
            rc = staging_flush(b);
            if (rc) return rc;
            b->win_start = at;
        } else {
            memset((u8*)b->ring[b->cur] + b->win_fill, 0, gap);
            b->win_fill += gap;
        }
    }
    while (len) {
        if (b->win_fill == kRingBytes) {
            rc = staging_flush(b);
            if (rc) return rc;
        }
        u64 take = kRingBytes - b->win_fill;
        if (take > len) take = len;
        memcpy((u8*)b->ring[b->cur] + b->win_fill, src, take);
        src += take;
        b->win_fill += take;
        len -= take;
    }
    return MI_OK;
}

int batch_add_common(mi_batch* b, u64 len, u64 tag, u64* at, u64 origin = 0) {
    if (!b) return MI_ERR_INVALID;
    if (b->staged) return fail(b->ctx, MI_ERR_STATE, "batch already ran; begin a new batch");
    *at = align_up(b->arena_used, kFileAlign);
    int rc = arena_reserve(b, *at + align_up(len, kFileAlign));
    if (rc) return rc;
    b->files.push_back({*at, len, tag});
    b->files.back().origin = origin;
    if (b->keep_sums) b->files.back().sums = b->sum_pool.take(mi_sum::chunks_of(origin % mi_sum::kChunk + len));   // (the FILE's 1 MiB grid)
    b->arena_used = *at + len;
    b->total_bytes += len;
    return MI_OK;
}

template <typename T>
int upload(mi_ctx* c, DevBuf& d, const std::vector<T>& h) {
    HIPCHK(c, d.ensure(h.size() * sizeof(T) + 16));
    if (!h.empty())
        HIPCHK(c, hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice,
                                 c->stream));
    return MI_OK;
}

float ev_ms(hipEvent_t a, hipEvent_t b) {
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}

// ---- cut selection stage ------------------------------------------------------------------
GearLaunch gear_args(mi_batch* b) {
    mi_ctx* c = b->ctx;
    GearLaunch g;
    g.data = b->arena.as<u8>();
    g.file_off = b->file_off.as<u64>();
    g.file_size = b->file_size.as<u64>();
    g.file_seg0 = b->file_seg0.as<u64>();
    g.seg_file = b->seg_file.as<u32>();
    g.seg_slot = b->seg_slot.as<u64>();
    g.ends32 = b->ends32.as<u32>();
    g.seg_n = b->seg_n.as<u32>();
    g.small_list = b->small_list.as<u32>();
    g.n_small = b->n_small;
    g.dense_list = b->dense_list.as<u32>();
    g.dense_count = b->dense_list.p ? b->dense_list.as<u32>() + b->n_small : nullptr;
    g.group_file = b->group_file.as<u32>();
    g.group_index = b->group_index.as<u32>();
    g.n_groups = b->n_groups;
    g.large_list = b->large_list.as<u32>();
    g.large_group0 = b->large_group0.as<u32>();
    g.n_large = b->n_large;
    g.group_recs = b->group_recs.p;
    g.tile_lists = b->tile_lists.as<u32>();
    g.tile_fast = b->tile_fast.as<u32>();
    g.file_flags = b->parts.empty() ? nullptr : b->file_flags.as<u32>();
    g.gear_table = c->gear_table.as<u64>();
    return g;
}

int ensure_cut_buffers(mi_batch* b) {
    mi_ctx* c = b->ctx;
    HIPCHK(c, b->ends32.ensure(b->ends_total * 4 + 16));
    HIPCHK(c, b->seg_n.ensure(b->n_segs * 4 + 16));
    if (b->n_small) HIPCHK(c, b->dense_list.ensure(((size_t)b->n_small + 1) * 4));
    if (b->n_groups) {
        HIPCHK(c, b->group_recs.ensure(gear_group_rec_bytes() * (size_t)b->n_groups));
        HIPCHK(c, b->tile_lists.ensure(1024ull * b->n_groups));
        HIPCHK(c, b->tile_fast.ensure(16ull * b->n_groups));
    }
    return MI_OK;
}

// the parts' confirmed entries -> device (the vector lives in the batch: pageable async copy)
int upload_part_entries(mi_batch* b) {
    mi_ctx* c = b->ctx;
    std::vector<u64> e(b->parts.size());
    for (size_t i = 0; i < e.size(); ++i) e[i] = b->parts[i].set_entry_rel;
    HIPCHK(c, b->part_entry.ensure(e.size() * 8 + 16));
    HIPCHK(c, hipMemcpy(b->part_entry.p, e.data(), e.size() * 8, hipMemcpyHostToDevice));
    return MI_OK;
}

// Gear marking + cut selection of the whole batch on the batch's stream -- unless mi_batch_scan_cuts
// already made the cuts, then only what the parts still need (entries confirmed since).
int enqueue_cuts(mi_batch* b) {
    mi_ctx* c = b->ctx;
    hipStream_t s = b->stream;
    int rc = ensure_cut_buffers(b);
    if (rc) return rc;
    const GearLaunch g = gear_args(b);
    const bool fresh = !b->cuts_ready;
    if (fresh) launch_gear_cdc(g, c->cdc, c->prop.multiProcessorCount, s);
    if (!b->parts.empty()) {
        const bool refix = fresh || b->parts_dirty;
        if (refix && (rc = upload_part_entries(b))) return rc;
        launch_gear_parts(g, b->part_file.as<u32>(), b->part_group0.as<u32>(), b->part_halo.as<u32>(),
                          b->part_entry.as<u64>(), (u32)b->parts.size(), refix, c->cdc, s);
    }
    b->cuts_ready = false;
    b->parts_dirty = false;
    return MI_OK;
}

// entry of the first own group and exit of the last group of every part, after the stream drained
int read_part_states(mi_batch* b) {
    mi_ctx* c = b->ctx;
    HIPCHK(c, hipStreamSynchronize(b->stream));
    const size_t rb = gear_group_rec_bytes();
    for (PartRec& p : b->parts) {
        GroupRec r;
        const u8* recs = b->group_recs.as<u8>();
        HIPCHK(c, hipMemcpy(&r, recs + rb * (p.group0 + p.n_groups - 1), rb, hipMemcpyDeviceToHost));
        p.exit_rel = r.final_exit;
        if (p.halo_groups) {
            HIPCHK(c, hipMemcpy(&r, recs + rb * (p.group0 + p.halo_groups), rb, hipMemcpyDeviceToHost));
            p.entry_rel = r.entry;
        } else {
            p.entry_rel = 0;
        }
    }
    return MI_OK;
}

// ---- the device pipeline ---------------------------------------------------------
// submit_pipeline only ENQUEUES (kernels, memsets, two 8-byte async copies into pinned
// memory) on the batch's stream: there is no host synchronisation between the stages.  Table
// sizes come from host-side upper bounds (slots = sum(size/min_size + 2)); the real chunk
// count lives in device memory (the control block) and every kernel that needs it reads it there.
int submit_pipeline_enqueue(mi_batch* b);

// in_flight is set only when everything was enqueued: a submit that failed half-way waits for what it
// did launch and leaves the batch as it was (a following mi_batch_wait is MI_ERR_STATE, not an empty
// "successful" run)
int submit_pipeline(mi_batch* b) {
    b->in_flight = false;
    const int rc = submit_pipeline_enqueue(b);
    if (rc == MI_OK) { b->in_flight = true; ++b->ctx->batches_in_flight; }
    else (void)hipStreamSynchronize(b->stream);
    return rc;
}

int submit_pipeline_enqueue(mi_batch* b) {
    mi_ctx* c = b->ctx;
    hipStream_t s = b->stream;
    const u64 nf = b->files.size();
    b->results_valid = false;
    b->h_roots_valid = false;
    memset(&b->stats, 0, sizeof b->stats);
    b->stats.bytes_in = b->total_bytes;
    b->stats.n_files = nf;
    b->stats.ms_h2d = b->ms_h2d;
    b->n_chunks = 0;
    b->h_counts[0] = b->h_counts[1] = 0;
    if (nf == 0) return MI_OK;
    if (nf >= 0x7FFFFFFFull) return fail(c, MI_ERR_INVALID, "too many files in one batch");
    for (const PartRec& p : b->parts)
        if (p.halo_groups && !p.confirmed) {
            return fail(c, MI_ERR_STATE, "file %llu is a part whose entry cut is not confirmed: "
                        "mi_batch_scan_cuts, exchange the exits, mi_batch_set_part_entry",
                        (unsigned long long)p.file_index);
        }
    const u64 cap = b->total_slots;                     // upper bound of the chunk count
    if (cap >= 0xFFFFFFFFull) return fail(c, MI_ERR_INVALID, "batch too large: %llu chunk slots",
                                          (unsigned long long)cap);
    // length bins for the longest-first order: SHA block counts >> bin_shift, <= 1024 bins
    u32 bin_shift = 2;
    while (((c->cfg.max_size / 64 + 3) >> bin_shift) + 1 > 1024) ++bin_shift;
    const u32 n_bins = ((c->cfg.max_size / 64 + 3) >> bin_shift) + 1;
    u64 dd_cap = 1024;
    while (dd_cap < 2 * cap) dd_cap <<= 1;
    const bool dedup = !(c->cfg.flags & MI_FLAG_NO_DEDUP);

    HIPCHK(c, b->seg_first.ensure(b->n_segs * 8 + 16));
    HIPCHK(c, b->n_chunks_d.ensure(nf * 4));
    HIPCHK(c, b->first.ensure(nf * 8));
    HIPCHK(c, b->ctl.ensure(kCtlBytes));
    HIPCHK(c, b->scratch.ensure(scan_scratch_elems(b->n_segs > nf ? b->n_segs : nf) * 8));
    HIPCHK(c, b->chunk_off.ensure(cap * 8));
    HIPCHK(c, b->chunk_len.ensure(cap * 8));
    HIPCHK(c, b->chunk_start.ensure(cap * 8));
    HIPCHK(c, b->chunk_file.ensure(cap * 4));
    HIPCHK(c, b->q_off.ensure(cap * 8));
    HIPCHK(c, b->q_len.ensure(cap * 8));
    HIPCHK(c, b->q_id.ensure(cap * 4));
    HIPCHK(c, b->digests.ensure(cap * 32));
    HIPCHK(c, b->item_off.ensure(nf * 8));
    HIPCHK(c, b->item_len.ensure(nf * 8));
    HIPCHK(c, b->roots.ensure(nf * 32));
    HIPCHK(c, b->dup_of.ensure(cap * 8));
    if (c->cfg.flags & MI_FLAG_FILE_SHA256) HIPCHK(c, b->file_sha.ensure(nf * 32));
    if (dedup) {
        HIPCHK(c, b->dd_table.ensure(dd_cap * 8));
        HIPCHK(c, b->dd_slot.ensure(cap * 4));
    }

    const u64* d_off = b->file_off.as<u64>();
    const u64* d_size = b->file_size.as<u64>();
    // control block: {u64 total, u64 n_unique | SHA queue heads, one set per launch | bin cursor |
    // length histogram} -- everything the pipeline needs zeroed, cleared by ONE memset
    u8* ctl = b->ctl.as<u8>();
    u64* d_total = (u64*)ctl;
    u64* d_nuniq = (u64*)(ctl + 8);
    auto heads = [&](int set) { return (u32*)(ctl + kCtlHeadsOff) + set * kShaHeadWords; };
    auto roles = [&](int set) { return (u32*)(ctl + kCtlRolesOff) + set * kShaRoleWords; };
    u32* d_cursor = (u32*)(ctl + kCtlCursorOff);
    u32* d_hist = (u32*)(ctl + kCtlHistOff);
    const u64* d_n = d_total;
    const int ncu = c->prop.multiProcessorCount;
    // Pin the hashing workgroups to their CUs (sha256.hip launch_sha256_items) when this batch has the GPU to
    // itself: a crowded CU then costs the launch up to 25 %.  With another batch in flight the passes of the
    // two fill each other's gaps, the step is the same either way (5.8 ms on C2), and the unused LDS the pin
    // reserves would only keep the other batch's Gear workgroups off the CU (measured: -1.3 %).
    ShaTune sha = c->sha;
    sha.pin_blocks_per_cu = c->sha.pin_blocks_per_cu && c->batches_in_flight == 0;
    // an arena of small pieces: the cooperative loads from a much smaller footprint on (ShaTune::coop_min_bytes_pieces)
    if (const u64 piece = arena_piece_bytes(&b->arena); piece && piece < (256ull << 20) && sha.coop_min_bytes_pieces < sha.coop_min_bytes)
        sha.coop_min_bytes = sha.coop_min_bytes_pieces;

    HIPCHK(c, hipEventRecord(b->ev[0], s));
    HIPCHK(c, hipMemsetAsync(ctl, 0, kCtlBytes, s));
    const u32 region = (u32)gear_group_region(c->cfg.min_size);
    {
        int rc = enqueue_cuts(b);
        if (rc) return rc;
    }
    launch_scan_counts(b->seg_n.as<u32>(), b->seg_first.as<u64>(), d_total, b->n_segs,
                       b->scratch.as<u64>(), s);
    HIPCHK(c, hipEventRecord(b->ev[1], s));
    const bool flat_roots = b->root_passes == 0;             // every file's root is one string
    launch_compact_chunks(d_off, b->file_seg0.as<u64>(), b->seg_file.as<u32>(), b->seg_slot.as<u64>(),
                          b->ends32.as<u32>(), b->seg_first.as<u64>(),
                          b->n_groups ? b->seg_group.as<u32>() : nullptr, b->group_recs.p, region, nf,
                          b->n_segs, cap, d_n, b->chunk_off.as<u64>(), b->chunk_len.as<u64>(),
                          b->chunk_file.as<u32>(), b->chunk_start.as<u64>(), b->first.as<u64>(),
                          b->n_chunks_d.as<u32>(), d_hist, n_bins, bin_shift, b->digests.as<u8>(),
                          flat_roots ? b->item_off.as<u64>() : nullptr,
                          flat_roots ? b->item_len.as<u64>() : nullptr, s);
    launch_bin_order(b->chunk_off.as<u64>(), b->chunk_len.as<u64>(), (u32)cap, d_n,
                     d_hist, d_cursor, n_bins, bin_shift,
                     b->q_off.as<u64>(), b->q_len.as<u64>(), b->q_id.as<u32>(), s);
    // MI_SHA_SERIALIZE=1 (experiments; off by default): one chunk pass at a time on the device.  A chunk pass
    // is a persistent grid that fills every SIMD; the chunk pass of ANOTHER batch in flight fits beside it
    // (138 VGPRs: three waves per SIMD) and the two then stretch each other to 6-8 ms apiece.  Making a
    // batch's chunk pass wait for the previous one of this ctx gives clean 4.45 ms launches again -- and
    // costs 2-4 % of the two-batch throughput (5.94 vs 5.77 ms per C2 step: the tails and the table kernels
    // of one batch are no longer filled by the other's hashing), which is what one batch at a time does too.
    if (c->serialize_sha && c->sha_done_set) HIPCHK(c, hipStreamWaitEvent(s, c->sha_done, 0));
    HIPCHK(c, hipEventRecord(b->ev[2], s));
    launch_sha256_items(kShaChunks, b->arena.as<u8>(), b->q_off.as<u64>(), b->q_len.as<u64>(),
                        b->q_id.as<u32>(), (u32)cap, d_n, heads(0), roles(0), false,
                        b->digests.as<u8>(), sha, ncu, b->arena_used, s);
    HIPCHK(c, hipEventRecord(b->ev[3], s));
    if (c->serialize_sha) {
        HIPCHK(c, hipEventRecord(c->sha_done, s));
        c->sha_done_set = true;
    }
    // per-file chunk roots: fan-out-64 tree; reduction passes only exist for files with more than
    // 64 chunks (> ~0.5 MiB), the final pass hashes every file's <= 64 nodes
    if (!flat_roots) {
        HIPCHK(c, b->root_addr.ensure(nf * 8));
        HIPCHK(c, b->root_cnt.ensure(nf * 4));
        launch_root_init(b->digests.as<u8>(), b->first.as<u64>(), b->n_chunks_d.as<u32>(), nf,
                         b->root_addr.as<u64>(), b->root_cnt.as<u32>(), s);
        HIPCHK(c, b->root_addr2.ensure(nf * 8));
        HIPCHK(c, b->root_cnt2.ensure(nf * 4));
        u64* cur_addr = b->root_addr.as<u64>();
        u32* cur_cnt = b->root_cnt.as<u32>();
        u64* next_addr = b->root_addr2.as<u64>();
        u32* next_cnt = b->root_cnt2.as<u32>();
        u64 nodes_ub = cap;                              // upper bound of nodes entering a pass
        for (int r = 0; r < b->root_passes; ++r) {
            const u64 out_ub = nodes_ub / kChunkRootFanout + nf;     // nodes it can produce
            if (out_ub >= 0xFFFFFFFFull) return fail(c, MI_ERR_INVALID, "batch too large for the root tree");
            HIPCHK(c, b->rseg_cnt.ensure(nf * 4));
            HIPCHK(c, b->rseg_first.ensure(nf * 8));
            HIPCHK(c, b->rseg_total.ensure(8));
            HIPCHK(c, b->root_items_off.ensure(out_ub * 8));
            HIPCHK(c, b->root_items_len.ensure(out_ub * 8));
            HIPCHK(c, b->root_level[r].ensure(out_ub * 32));
            launch_root_level(nf, out_ub, cur_addr, cur_cnt, next_addr, next_cnt, b->rseg_cnt.as<u32>(),
                              b->rseg_first.as<u64>(), b->rseg_total.as<u64>(), b->scratch.as<u64>(),
                              b->root_level[r].as<u8>(), b->root_items_off.as<u64>(),
                              b->root_items_len.as<u64>(), s);
            launch_sha256_items(kShaRoots, nullptr, b->root_items_off.as<u64>(),
                                b->root_items_len.as<u64>(), nullptr, (u32)out_ub,
                                b->rseg_total.as<u64>(), heads(3 + r), nullptr, false,
                                b->root_level[r].as<u8>(), sha, ncu, 0, s);
            nodes_ub = out_ub;
            std::swap(cur_addr, next_addr);
            std::swap(cur_cnt, next_cnt);
        }
        launch_root_final_items(cur_addr, cur_cnt, nf, b->item_off.as<u64>(), b->item_len.as<u64>(), s);
    }
    launch_sha256_items(kShaRoots, nullptr, b->item_off.as<u64>(), b->item_len.as<u64>(), nullptr,
                        (u32)nf, nullptr, heads(1), nullptr, false, b->roots.as<u8>(),
                        sha, ncu, 0, s);
    if (c->cfg.flags & MI_FLAG_FILE_SHA256)
        launch_sha256_items(kShaFiles, b->arena.as<u8>(), d_off, b->fsha_len.as<u64>(), nullptr, (u32)nf, nullptr,
                            heads(2), nullptr, false, b->file_sha.as<u8>(), sha, ncu, b->arena_used, s);   // files come in
                            // arrival order, not longest-first: no "long" range to hand to a SIMD's first wave (flat sharing)
    if (c->cfg.flags & MI_FLAG_FILE_CRC32) {
        HIPCHK(c, b->tile_raw.ensure(b->n_tiles * 4 + 16));
        HIPCHK(c, b->crc_d.ensure(nf * 4));
        launch_crc32_files(b->arena.as<u8>(), d_off, d_size, b->tile_file.as<u32>(),
                           b->first_tile.as<u64>(), b->n_tiles, nf, c->crc_consts.as<u32>(),
                           b->tile_raw.as<u32>(), b->crc_d.as<u32>(), s);
    }
    HIPCHK(c, hipEventRecord(b->ev[4], s));
    if (dedup) {
        launch_dedup_mark(b->digests.as<u8>(), cap, d_n, b->dd_table.as<u32>(), b->dd_slot.as<u32>(), dd_cap,
                          b->dup_of.as<i64>(), d_nuniq, false, s);
    } else {
        HIPCHK(c, hipMemsetAsync(b->dup_of.p, 0xFF, cap * 8, s));
    }
    HIPCHK(c, hipMemcpyAsync(&b->h_counts[0], ctl, 16, hipMemcpyDeviceToHost, s));   // {total, n_unique}
    HIPCHK(c, hipEventRecord(b->ev[5], s));
    HIPCHK(c, hipGetLastError());
    if ((c->cfg.flags & MI_FLAG_FILE_SHA256) && !b->fsha_host.empty()) {
        // ... and beside the passes, the long files on the reader threads (they are idle: staging has ended)
        const size_t k = b->fsha_host.size();
        std::vector<u64> ho(k), hl(k);
        for (size_t i = 0; i < k; ++i) { ho[i] = b->files[b->fsha_host[i]].off; hl[i] = b->files[b->fsha_host[i]].size; }
        b->fsha_host_out.assign(k * 32, 0);
        if (b->fsha_latch) (void)stager_hash_wait(c, b->fsha_latch);
        b->fsha_latch = stager_hash_ranges(c->stager, b, k, ho.data(), hl.data(), b->fsha_host_out.data());
    }
    return MI_OK;
}

int fetch_results(mi_batch* b);

int wait_pipeline(mi_batch* b) {
    mi_ctx* c = b->ctx;
    if (!b->in_flight) return fail(c, MI_ERR_STATE, "mi_batch_wait without a submitted run");
    b->in_flight = false;
    if (c->batches_in_flight > 0) --c->batches_in_flight;
    HIPCHK(c, hipStreamSynchronize(b->stream));
    HIPCHK(c, hipGetLastError());
    if (b->fsha_latch) {
        mi::HashLatch* l = b->fsha_latch;
        b->fsha_latch = nullptr;
        const int rc = stager_hash_wait(c, l);
        if (rc) return rc;
    }
    const u64 total = b->h_counts[0];
    if (total > b->total_slots)
        return fail(c, MI_ERR_HIP, "chunk count %llu exceeds its bound %llu",
                    (unsigned long long)total, (unsigned long long)b->total_slots);
    b->n_chunks = total;
    b->stats.n_chunks = total;
    b->stats.n_unique = (c->cfg.flags & MI_FLAG_NO_DEDUP) ? total : b->h_counts[1];
    if (!b->files.empty()) {
        b->stats.ms_cdc = ev_ms(b->ev[0], b->ev[1]);
        b->stats.ms_sort = ev_ms(b->ev[1], b->ev[2]);
        b->stats.ms_sha_chunks = ev_ms(b->ev[2], b->ev[3]);
        b->stats.ms_sha_files = ev_ms(b->ev[3], b->ev[4]);
        b->stats.ms_dedup = ev_ms(b->ev[4], b->ev[5]);
        b->stats.ms_total = ev_ms(b->ev[0], b->ev[5]);
    }
    c->stats = b->stats;
    b->ran = true;
    if (c->cfg.flags & MI_FLAG_PREFETCH_ROWS) return fetch_results(b);
    return MI_OK;
}

int fetch_results(mi_batch* b) {
    mi_ctx* c = b->ctx;
    if (!b->ran) return fail(c, MI_ERR_STATE, "results requested before mi_batch_run");
    if (b->results_valid) return MI_OK;
    const u64 nf = b->files.size(), nc = b->n_chunks;
    b->n_h_files = 0;
    if (nf) {
        // the file rows are packed by a kernel too: one copy behind the chunk rows' (same stream, one wait for both) instead
        // of three to five synchronous column copies and a repacking loop on the host
        static_assert(sizeof(mi_file_result) == 96, "pack_file_rows_kernel writes 96-byte rows");
        if (nf > b->h_files_cap) {
            if (b->h_files) (void)hipHostFree(b->h_files);
            b->h_files = nullptr;
            b->h_files_cap = 0;
            const size_t want = nf + nf / 8 + 16;
            HIPCHK(c, hipHostMalloc((void**)&b->h_files, want * sizeof(mi_file_result), hipHostMallocDefault));
            b->h_files_cap = want;
        }
        HIPCHK(c, b->file_rows_d.ensure(nf * sizeof(mi_file_result)));
        launch_pack_file_rows(nf, b->file_size.as<u64>(), b->first.as<u64>(), b->n_chunks_d.as<u32>(),
                              (c->cfg.flags & MI_FLAG_FILE_CRC32) ? b->crc_d.as<u32>() : nullptr, b->roots.as<u8>(),
                              (c->cfg.flags & MI_FLAG_FILE_SHA256) ? b->file_sha.as<u8>() : nullptr, b->file_rows_d.p, b->stream);
        HIPCHK(c, hipMemcpyAsync(b->h_files, b->file_rows_d.p, nf * sizeof(mi_file_result), hipMemcpyDeviceToHost, b->stream));
    }
    if (nc) {
        // rows are packed by a kernel; ONE device-to-host copy into the batch's pinned buffer
        static_assert(sizeof(mi_chunk_result) == 64, "pack_chunk_rows_kernel writes 64-byte rows");
        const size_t bytes = nc * sizeof(mi_chunk_result);
        HIPCHK(c, b->rows_d.ensure(bytes));
        if (bytes > b->rows_h_bytes) {
            if (b->rows_h) (void)hipHostFree(b->rows_h);
            b->rows_h = nullptr;
            b->rows_h_bytes = 0;
            const size_t want = bytes + bytes / 8;
            HIPCHK(c, hipHostMalloc(&b->rows_h, want, hipHostMallocDefault));
            b->rows_h_bytes = want;
        }
        const u64* d_base = nullptr;
        if (!b->parts.empty()) {                         // parts report offsets inside the whole file
            std::vector<u64> base(nf, 0);
            for (const PartRec& p : b->parts) base[p.file_index] = p.begin - p.halo_bytes;
            HIPCHK(c, b->file_base.ensure(nf * 8));
            HIPCHK(c, hipMemcpy(b->file_base.p, base.data(), nf * 8, hipMemcpyHostToDevice));
            d_base = b->file_base.as<u64>();
        }
        launch_pack_chunk_rows(nc, b->chunk_file.as<u32>(), b->chunk_start.as<u64>(), b->chunk_len.as<u64>(),
                               b->dup_of.as<i64>(), b->digests.as<u8>(), d_base, b->rows_d.p, b->stream);
        HIPCHK(c, hipMemcpyAsync(b->rows_h, b->rows_d.p, bytes, hipMemcpyDeviceToHost, b->stream));
    }
    if (nf || nc) {
        HIPCHK(c, hipStreamSynchronize(b->stream));
        HIPCHK(c, hipGetLastError());
    }
    if ((c->cfg.flags & MI_FLAG_FILE_SHA256) && b->fsha_host_out.size() == b->fsha_host.size() * 32)
        for (size_t i = 0; i < b->fsha_host.size(); ++i)   // the long files' digests come from the host side of the pass
            memcpy(b->h_files[b->fsha_host[i]].file_sha256, b->fsha_host_out.data() + 32 * i, 32);
    for (u64 f = 0; f < nf; ++f) {                      // what only the host knows: the caller's tag; a part's own range
        mi_file_result& r = b->h_files[f];
        r.user_tag = b->files[f].tag;
        if (b->files[f].part >= 0) {                    // a part: its own range; no whole-file values
            const PartRec& p = b->parts[b->files[f].part];
            r.size = p.end - p.begin;
            r.crc32 = 0;
            memset(r.file_sha256, 0, 32);
        }
    }
    b->n_h_files = nf;
    b->results_valid = true;
    return MI_OK;
}

}  // namespace

// ================================ C ABI ============================================
extern "C" {


int mi_abi_version(void) { return MI_ABI_VERSION; }

int mi_debug_sha_wave_stats(mi_ctx* c, const char* path) {
    if (!c) return MI_ERR_INVALID;
    c->sha_wave_stats = path ? path : "";
    c->sha.wave_stats_path = c->sha_wave_stats.empty() ? nullptr : c->sha_wave_stats.c_str();
    return MI_OK;
}

int mi_config_default(mi_config* cfg) {
    if (!cfg) return MI_ERR_INVALID;
    memset(cfg, 0, sizeof *cfg);
    cfg->struct_size = sizeof *cfg;
    cfg->device = 0;
    cfg->gear_seed = 0x4D414B49ull;
    cfg->mask_bits = 13;
    cfg->min_size = 2048;
    cfg->max_size = 65536;
    cfg->flags = 0;
    cfg->staging_bytes = 0;                          // 0 = engine defaults: 8 MiB slabs,
    cfg->n_streams = 0;                              //     8 reader threads
    return MI_OK;
}

// (a copy of the caller's own, taken under the lock every writer holds: a pipelined commit has two threads of the library on one
//  ctx -- its scan and its tar writer -- and a failure can reach both while the host asks; valid until the calling thread asks again)
const char* mi_last_error(mi_ctx* ctx) {
    static thread_local std::string mine;
    std::lock_guard<std::mutex> g(g_err_mu);
    mine = ctx ? ctx->err : g_create_err;
    return mine.c_str();
}

int mi_ctx_create(const mi_config* cfg, mi_ctx** out) {
    if (!cfg || !out) return fail(nullptr, MI_ERR_INVALID, "mi_ctx_create: null argument");
    if (cfg->struct_size != sizeof(mi_config))
        return fail(nullptr, MI_ERR_INVALID, "mi_ctx_create: struct_size %u != %zu",
                    cfg->struct_size, sizeof(mi_config));
    if (cfg->mask_bits > 32 || cfg->min_size < 64 || cfg->max_size < cfg->min_size ||
        cfg->max_size > (1u << 30))
        return fail(nullptr, MI_ERR_INVALID,
                    "mi_ctx_create: need mask_bits<=32, 64<=min_size<=max_size<=2^30");
    if (cfg->sha_load_scheme > MI_SHA_LOADS_COOP)
        return fail(nullptr, MI_ERR_INVALID, "mi_ctx_create: sha_load_scheme %u is not an MI_SHA_LOADS_* value",
                    cfg->sha_load_scheme);
    if ((cfg->sha_sched & ~(MI_SHA_SCHED_FLAT | 0x1F00u)) || ((cfg->sha_sched >> 8) & 0x1Fu) > 16)
        return fail(nullptr, MI_ERR_INVALID, "mi_ctx_create: sha_sched 0x%x is not a combination of MI_SHA_SCHED_* values",
                    cfg->sha_sched);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, MI_ERR_NO_DEVICE,
                    "no HIP device visible; this engine has no CPU path");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, MI_ERR_NO_DEVICE, "device %d not present (%d visible)", cfg->device,
                    ndev);
    mi_ctx* c = new mi_ctx();
    c->cfg = *cfg;
    c->device = cfg->device;
#define CREATE_CHK(call)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            int rc_ = fail(nullptr, MI_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
            mi_ctx_destroy(c);                                                                \
            return rc_;                                                                       \
        }                                                                                     \
    } while (0)
    CREATE_CHK(hipSetDevice(c->device));
    CREATE_CHK(hipGetDeviceProperties(&c->prop, c->device));
    if (strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
        int rc = fail(nullptr, MI_ERR_NO_DEVICE, "device %d is %s; kernels are built for gfx950 only",
                      c->device, c->prop.gcnArchName);
        mi_ctx_destroy(c);
        return rc;
    }
    // Plain creation on purpose: a stream made with hipStreamCreateWithPriority -- even at the
    // default priority -- changes how the runtime spreads the later (batch) streams over the
    // hardware queues, and the batches in flight stop overlapping (measured: 6.2 vs 5.87 ms
    // per C2 step under the ROCm 7.0 runtime PyTorch bundles).
    CREATE_CHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CREATE_CHK(hipHostMalloc((void**)&c->h_word, 64, hipHostMallocDefault));
    for (auto& e : c->ev) e = nullptr;
    for (auto& e : c->ev) CREATE_CHK(hipEventCreate(&e));
    CREATE_CHK(hipEventCreateWithFlags(&c->sha_done, hipEventDisableTiming));
    if (const char* e = getenv("MI_SHA_SERIALIZE")) c->serialize_sha = atoi(e) != 0;
    // host-fed staging (mi_stage.hip): slab bytes and reader threads; both lazily allocated
    c->staging_bytes = cfg->staging_bytes ? cfg->staging_bytes : (8ull << 20);
    if (c->staging_bytes < (1ull << 16)) c->staging_bytes = 1ull << 16;
    if (c->staging_bytes > (1ull << 30)) c->staging_bytes = 1ull << 30;
    c->staging_bytes = (c->staging_bytes + 4095) / 4096 * 4096;   // spans start 8-byte aligned (stage_sum_kernel)
    c->stage_threads = cfg->n_streams ? cfg->n_streams : 8;    // 8 readers already saturate PCIe Gen5 x16
    if (const char* e = getenv("MI_STAGE_THREADS")) {
        int v = atoi(e);
        if (v >= 1 && v <= 64) c->stage_threads = (u32)v;
    }
    // every reader owns a pinned slab and a stream: a byte-like number from an old caller must not
    // spawn thousands of them
    if (c->stage_threads > 64) c->stage_threads = 64;
    c->verify_staging = (cfg->flags & MI_FLAG_VERIFY_STAGING) != 0;
    if (const char* e = getenv("MI_VERIFY_STAGING")) c->verify_staging = atoi(e) != 0;
    if (const char* e = getenv("MI_STAGE_FAULT")) {
        if (!strncmp(e, "copy:", 5)) c->fault_copy = atoll(e + 5);
        if (!strncmp(e, "final:", 6)) c->fault_final = atoll(e + 6);
        if (!strncmp(e, "readback:", 9)) {
            c->fault_readback = atoll(e + 9);
            if (const char* k = strchr(e + 9, ':')) c->fault_readback_n = atoll(k + 1);
        }
    }
    c->file_sums = (cfg->flags & MI_FLAG_FILE_SUMS) != 0;
    // Gear table: first 256 outputs of splitmix64(seed)
    u64 table[256];
    u64 st = cfg->gear_seed;
    for (int i = 0; i < 256; ++i) { st += kSmGamma; table[i] = splitmix64_mix(st); }
    CREATE_CHK(c->gear_table.ensure(sizeof table));
    CREATE_CHK(hipMemcpy(c->gear_table.p, table, sizeof table, hipMemcpyHostToDevice));
    CREATE_CHK(c->heads.ensure(sizeof(u32) * (kShaHeadWords + kShaRoleWords)));
    {
        std::vector<u32> consts(kCrcConstWords);
        crc32_build_tables(consts.data());
        CREATE_CHK(c->crc_consts.ensure(consts.size() * 4));
        CREATE_CHK(hipMemcpy(c->crc_consts.p, consts.data(), consts.size() * 4, hipMemcpyHostToDevice));
    }
    c->cdc.thresh_m1 = cfg->mask_bits == 0 ? 0xFFFFFFFFu : (u32)((1ull << (32 - cfg->mask_bits)) - 1);
    c->cdc.min_size = cfg->min_size;
    c->cdc.max_size = cfg->max_size;
    c->cdc.pad = 0;
    // hashing launches: mi_config.sha_* per ctx; the environment only moves the defaults
    if (const char* e = getenv("MI_SHA_BLOCKS_PER_CU")) {
        int v = atoi(e);
        if (v >= 1 && v <= 8) c->sha.blocks_per_cu = v;
    }
    if (const char* e = getenv("MI_SHA_COOP_MIN_GIB")) c->sha.coop_min_bytes = (u64)(atof(e) * 1073741824.0);
    if (const char* e = getenv("MI_SHA_COOP_MIN_GIB_PIECES")) c->sha.coop_min_bytes_pieces = (u64)(atof(e) * 1073741824.0);
    if (const char* e = getenv("MI_SHA_COOP_BLOCKS_PER_CU")) {
        int v = atoi(e);
        if (v >= 1 && v <= 3) c->sha.coop_blocks_per_cu = v;
    }
    if (const char* e = getenv("MI_SHA_PIN_BLOCKS")) c->sha.pin_blocks_per_cu = atoi(e) != 0;
    if (const char* e = getenv("MI_SHA_ROLES")) c->sha.roles = atoi(e) != 0;
    if (const char* e = getenv("MI_SHA_PRIO")) c->sha.prio = atoi(e) != 0;
    if (const char* e = getenv("MI_SHA_WAVE_STATS")) (void)mi_debug_sha_wave_stats(c, e);     // diagnostics, read per ctx here
    if (const char* e = getenv("MI_SHA_LONG_SHIFT")) {
        const int v = atoi(e);
        if (v >= 0 && v <= 16) c->sha.long_shift = v;
    }
    if (cfg->sha_sched & MI_SHA_SCHED_FLAT) c->sha.roles = false;
    if ((cfg->sha_sched >> 8) & 0x1Fu) c->sha.long_shift = (int)((cfg->sha_sched >> 8) & 0x1Fu) - 1;
    if (cfg->sha_blocks_per_cu >= 1 && cfg->sha_blocks_per_cu <= 8) c->sha.blocks_per_cu = (int)cfg->sha_blocks_per_cu;
    if (cfg->sha_coop_min_gib) c->sha.coop_min_bytes = (u64)cfg->sha_coop_min_gib << 30;
    if (cfg->sha_load_scheme == MI_SHA_LOADS_LANE) c->sha.coop_min_bytes = c->sha.coop_min_bytes_pieces = ~0ull;
    if (cfg->sha_load_scheme == MI_SHA_LOADS_COOP) c->sha.coop_min_bytes = c->sha.coop_min_bytes_pieces = 0;
    if (cfg->sha_coop_blocks_per_cu >= 1 && cfg->sha_coop_blocks_per_cu <= 3)
        c->sha.coop_blocks_per_cu = (int)cfg->sha_coop_blocks_per_cu;
    memset(&c->stats, 0, sizeof c->stats);
#undef CREATE_CHK
    *out = c;
    return MI_OK;
}

// What a process's first content-aware commit would otherwise pay (tools/first_commit_probe.py: 0.07 s instead of 0.011 for a
// 100-file tree): every reader thread with its pinned slab and stream, and the code objects of the scan's kernels (a four-file
// synthetic batch through the whole pipeline).  Blocking, on the caller's thread: a host calls it beside whatever it does
// between creating the ctx and its first commit (the shim: a goroutine next to the Dockerfile's parsing and the base image's
// pull), and joins it before the ctx's next call -- a ctx is not re-entrant, this call included.
int mi_ctx_warm(mi_ctx* c) {
    if (!c) return MI_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_stager(c);
    if (rc) return rc;
    (void)stager_ready_all(c->stager);
    const mi_stats keep = c->stats;                           // (the caller's next mi_get_stats is about ITS batches)
    mi_batch* b = nullptr;
    rc = mi_batch_begin(c, 4, 0, &b);
    if (rc) return rc;
    const uint64_t sizes[4] = {65536, 200000, 1, 4097};
    rc = mi_batch_add_synthetic(b, 4, sizes, nullptr, c->cfg.gear_seed);
    if (rc == MI_OK) rc = mi_batch_run(b);
    (void)mi_batch_free(b);
    c->stats = keep;
    return rc;
}

int mi_ctx_destroy(mi_ctx* c) {
    if (!c) return MI_OK;
    if (c->live_children > 0)
        // batches and indexes hold a pointer to their ctx: destroying it under them would leave
        // dangling handles (a Go finalizer order can do exactly that).  Refuse; the ctx stays usable.
        return fail(c, MI_ERR_STATE, "mi_ctx_destroy: %d batch(es)/index(es) of this ctx are still alive",
                    c->live_children);
    (void)hipSetDevice(c->device);
    (void)mi_comm_destroy(c);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    stager_destroy(c->stager);
    c->stager = nullptr;
    for (auto e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->sha_done) (void)hipEventDestroy(c->sha_done);
    if (c->h_word) (void)hipHostFree(c->h_word);
    c->gear_table.release(); c->heads.release(); c->crc_consts.release();
    c->dd_table.release(); c->dd_slot.release(); c->dd_nuniq.release(); c->dd_tag.release();
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return MI_OK;
}

int mi_get_stats(mi_ctx* c, mi_stats* out) {
    if (!c || !out) return MI_ERR_INVALID;
    *out = c->stats;
    return MI_OK;
}

int mi_sha_valu_roof(mi_ctx* c, uint32_t waves_per_simd, uint32_t blocks, double* bytes_per_second) {
    if (!c || !bytes_per_second) return MI_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (waves_per_simd == 0) waves_per_simd = 8;
    if (blocks == 0) blocks = 512;
    if (waves_per_simd > 8 || blocks > (1u << 20)) return fail(c, MI_ERR_INVALID, "mi_sha_valu_roof: waves_per_simd <= 8, blocks <= 2^20");
    DevBuf scratch;
    HIPCHK(c, scratch.ensure((size_t)c->prop.multiProcessorCount * waves_per_simd * kShaWG * 4));
    *bytes_per_second = measure_sha_valu_roof(c->prop.multiProcessorCount, (int)waves_per_simd, blocks,
                                              scratch.as<u32>(), c->stream, c->ev[0], c->ev[1]);
    const hipError_t e = hipGetLastError();
    scratch.release();
    if (e != hipSuccess || *bytes_per_second <= 0) return fail(c, MI_ERR_HIP, "mi_sha_valu_roof: %s", hipGetErrorString(e));
    return MI_OK;
}

int mi_device_info(mi_ctx* c, int32_t* n_cu, int32_t* clock_mhz, uint64_t* hbm_bytes, char* name,
                   size_t name_cap) {
    if (!c) return MI_ERR_INVALID;
    if (n_cu) *n_cu = c->prop.multiProcessorCount;
    if (clock_mhz) *clock_mhz = c->prop.clockRate / 1000;
    if (hbm_bytes) *hbm_bytes = c->prop.totalGlobalMem;
    if (name && name_cap) snprintf(name, name_cap, "%s (%s)", c->prop.name, c->prop.gcnArchName);
    return MI_OK;
}

int mi_batch_begin(mi_ctx* c, uint64_t n_files_hint, uint64_t bytes_hint, mi_batch** out) {
    if (!c || !out) return MI_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    mi_batch* b = new mi_batch();
    b->ctx = c;
    memset(&b->stats, 0, sizeof b->stats);
    memset(&b->stage_stats, 0, sizeof b->stage_stats);
    b->files.reserve(n_files_hint);
    b->keep_sums = c->file_sums;
    hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    for (auto& ev : b->ev) if (e == hipSuccess) e = hipEventCreate(&ev);
    if (e == hipSuccess) e = hipHostMalloc((void**)&b->h_counts, 16, hipHostMallocDefault);
    ++c->live_children;                                 // mi_batch_free undoes it (error paths included)
    if (e != hipSuccess) {
        int rc = fail(c, MI_ERR_HIP, "mi_batch_begin: %s", hipGetErrorString(e));
        mi_batch_free(b);
        return rc;
    }
    if (bytes_hint) {
        int rc = arena_reserve(b, bytes_hint + n_files_hint * kFileAlign, true);
        if (rc) { mi_batch_free(b); return rc; }
    }
    *out = b;
    return MI_OK;
}

// host buffers below this size are copied inline by the calling thread; larger ones are split
// over the reader threads (a single memcpy stream tops out far below PCIe Gen5)
static const u64 kInlineBytes = 1ull << 20;

int mi_batch_add_bytes(mi_batch* b, const void* data, uint64_t len, uint64_t user_tag) {
    if (!b || (!data && len)) return MI_ERR_INVALID;
    mi_ctx* c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    u64 at;
    int rc = batch_add_common(b, len, user_tag, &at);
    if (rc) return rc;
    if (len == 0) return MI_OK;
    const auto t0 = std::chrono::steady_clock::now();
    mi_sum::FileSum* sums = b->files.back().sums;
    if (len < kInlineBytes) {
        if (sums) mi_sum::row_add(data, len, 0, sums);           // (the caller's buffer: where these bytes were last known good)
        rc = staging_append(b, at, (const u8*)data, len);
    } else {
        rc = ensure_stager(c);
        if (!rc) rc = stager_put_bytes(c->stager, b, at, data, len, sums);   // returns when `data` has been consumed
        b->staged_any = true;
    }
    b->ms_h2d += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

// mi_batch_add_path / _range: the file is opened and checked NOW (a missing or short file is the
// caller's error to see at this call, like the reference's open + CopyN at lib/tario/write.go:37-45);
// its bytes are read by the reader threads, so a file that shrinks later fails the batch at run time.
// origin: the row is a PART of a file whose staged bytes begin at file offset `origin` (= offset); its sums lie on the file's grid
static int add_file_range(mi_batch* b, const char* path, uint64_t offset, uint64_t size, uint64_t user_tag, uint64_t origin = 0) {
    if (!b || !path) return MI_ERR_INVALID;
    mi_ctx* c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    // The descriptor travels with the file's pieces until the reader threads have read them: a caller that adds files faster
    // than they are read can exhaust a small RLIMIT_NOFILE with descriptors of its own queue.  That is not the caller's error:
    // wait for the readers and try again (only a table that stays full for seconds is reported).
    int fd = -1;
    for (int tries = 0;; ++tries) {
        fd = open(path, O_RDONLY | O_CLOEXEC);
        if (fd >= 0 || (errno != EMFILE && errno != ENFILE) || !c->stager || tries >= 400) break;
        const int keep = errno;
        stager_pause(c->stager, 5);
        errno = keep;
    }
    if (fd < 0) return fail(c, MI_ERR_IO, "open %s: %s", path, strerror(errno));
    mi_io::content_opens.fetch_add(1, std::memory_order_relaxed);
    struct stat sb;
    if (fstat(fd, &sb) != 0) {
        int rc = fail(c, MI_ERR_IO, "stat %s: %s", path, strerror(errno));
        close(fd);
        return rc;
    }
    if (S_ISREG(sb.st_mode) && (offset > (u64)sb.st_size || size > (u64)sb.st_size - offset)) {
        close(fd);
        return fail(c, MI_ERR_IO, "read %s: file shorter than the size given", path);
    }
    u64 at;
    int rc = batch_add_common(b, size, user_tag, &at, origin);
    if (rc == MI_OK) rc = ensure_stager(c);
    if (rc) { close(fd); return rc; }
    const auto t0 = std::chrono::steady_clock::now();
    rc = stager_put_file(c->stager, b, at, fd, offset, size, path, b->files.back().sums, origin % mi_sum::kChunk);   // owns fd from here on
    b->staged_any = true;
    b->ms_h2d += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

int mi_batch_add_path(mi_batch* b, const char* path, uint64_t size, uint64_t user_tag) {
    return add_file_range(b, path, 0, size, user_tag);
}

// Bulk form with DEFERRED opens: nothing is opened or read in this call -- the reader threads open,
// read and close the files, several at a time, and consecutive small files share one PCIe transfer.
// What it costs per file here is a table row (a tree of 100 000 4 KiB files: 9 us per file through
// mi_batch_add_path -- open, fstat, a queue hand-over each -- against well under 1 us).
int mi_batch_add_paths(mi_batch* b, uint64_t n, const char* const* paths, const uint64_t* sizes,
                       const uint64_t* user_tags) {
    if (!b || (n && (!paths || !sizes))) return MI_ERR_INVALID;
    if (is_group(b)) {
        // each file to the member with the fewest bytes so far (the streaming form of longest-processing-time-first: the walk
        // hands files over as it finds them); a member's files keep the walk's order among themselves
        const size_t nm = b->members.size();
        std::vector<std::vector<const char*>> mp(nm);
        std::vector<std::vector<u64>> ms(nm), mt(nm);
        static const u64 split_min = [] {
            const char* e = getenv("MI_COMMIT_SPLIT_MIB");
            const long v = e && *e ? atol(e) : 256;
            return v <= 0 ? ~0ull : (u64)v << 20;
        }();
        for (u64 i = 0; i < n; ++i) {
            if (sizes[i] >= split_min && sizes[i] >= 2 * mi_sum::kChunk) {
                // A file of 256 MiB and more is SPLIT: one part per member, at most -- each part to the member with
                // the fewest bytes so far, staged over that GPU's own link behind its halo (mi_batch_add_path_part).  Parts begin
                // on MiB boundaries of the file, so every 1 MiB chunk the tar writer checks lies in ONE part's own range.  The
                // parts' owners agree on the boundary cuts when the group runs (group_resolve_parts).
                u64 np = std::min<u64>(nm, sizes[i] / (split_min / 2 ? split_min / 2 : 1));
                if (np < 2) np = 2;
                const u64 step = (sizes[i] / np + mi_sum::kChunk - 1) / mi_sum::kChunk * mi_sum::kChunk;
                mi_batch::Split sp;
                sp.size = sizes[i];
                for (u64 begin = 0; begin < sizes[i]; begin += step) {
                    const u64 end = std::min(begin + step, (u64)sizes[i]);
                    const size_t k = group_least_loaded(b);
                    // (the member's earlier pending files first: a part is added at once, its row must follow theirs)
                    if (!mp[k].empty()) {
                        const int rc0 = mi_batch_add_paths(b->members[k], mp[k].size(), mp[k].data(), ms[k].data(), mt[k].data());
                        if (rc0) return group_fail(b, k, rc0);
                        mp[k].clear(); ms[k].clear(); mt[k].clear();
                    }
                    const u64 row = b->members[k]->files.size();
                    const int rc = mi_batch_add_path_part(b->members[k], paths[i], sizes[i], begin, end, user_tags ? user_tags[i] : 0);
                    if (rc) return group_fail(b, k, rc);
                    b->member_bytes[k] += end - begin;
                    sp.parts.push_back({(u32)k, row, begin, end});
                }
                b->row_member.push_back(kGroupSplit);
                b->row_row.push_back(b->splits.size());
                b->splits.push_back(std::move(sp));
                b->total_bytes += sizes[i];
                continue;
            }
            const size_t k = group_least_loaded(b);
            b->member_bytes[k] += sizes[i] + 4096;                 // (a file costs something even when it is empty: rows, a descriptor)
            b->row_member.push_back((u32)k);
            b->row_row.push_back(b->members[k]->files.size() + mp[k].size());
            mp[k].push_back(paths[i]);
            ms[k].push_back(sizes[i]);
            mt[k].push_back(user_tags ? user_tags[i] : 0);
            b->total_bytes += sizes[i];
        }
        for (size_t k = 0; k < nm; ++k) {
            if (mp[k].empty()) continue;
            const int rc = mi_batch_add_paths(b->members[k], mp[k].size(), mp[k].data(), ms[k].data(), mt[k].data());
            if (rc) return group_fail(b, k, rc);
        }
        return MI_OK;
    }
    mi_ctx* c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (n == 0) return MI_OK;
    if (b->staged) return fail(c, MI_ERR_STATE, "batch already ran; begin a new batch");
    for (u64 i = 0; i < n; ++i)
        if (!paths[i]) return fail(c, MI_ERR_INVALID, "mi_batch_add_paths: path %llu is NULL", (unsigned long long)i);
    int rc = staging_flush(b);                    // the inline window may hold bytes of earlier small adds
    if (rc) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    // one reservation for all of them (arena_reserve may move the arena: nothing may be in flight into it)
    u64 end = b->arena_used;
    for (u64 i = 0; i < n; ++i) end = align_up(end, kFileAlign) + sizes[i];
    rc = arena_reserve(b, align_up(end, kFileAlign));
    if (rc == MI_OK) rc = ensure_stager(c);
    if (rc) return rc;
    std::vector<u64> at(n);
    std::vector<mi_sum::FileSum*> sums(b->keep_sums ? n : 0);
    for (u64 i = 0; i < n; ++i) {
        at[i] = align_up(b->arena_used, kFileAlign);
        b->files.push_back({at[i], sizes[i], user_tags ? user_tags[i] : 0});
        if (b->keep_sums) b->files.back().sums = sums[i] = b->sum_pool.take(mi_sum::chunks_of(sizes[i]));
        b->arena_used = at[i] + sizes[i];
        b->total_bytes += sizes[i];
    }
    rc = stager_put_paths(c->stager, b, n, paths, at.data(), sizes, b->keep_sums ? sums.data() : nullptr);
    b->staged_any = true;
    b->ms_h2d += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

// For the tree walk (mi_tree.hip), which reads small files where it lists them: a block of host memory that holds files
// as they are to lie in the arena goes in as ONE piece of the arena (*at_out = where), copied by the reader threads;
// release(release_arg) is called when the block is no longer needed.  The files themselves are then table rows
// (mi_batch_add_placed): where a file lies in the arena has nothing to do with its index.
extern "C" int mi_batch_add_block(mi_batch* b, const void* src, uint64_t len, void (*release)(void*), void* release_arg,
                                  uint64_t* at_out) {
    if (b && is_group(b)) {                          // the whole block -- a directory's small files -- to ONE member
        const size_t k = group_least_loaded(b);
        uint64_t at = 0;
        const int rc = mi_batch_add_block(b->members[k], src, len, release, release_arg, &at);
        if (rc) return group_fail(b, k, rc);
        b->member_bytes[k] += len;
        if (at_out) *at_out = ((u64)k << kGroupAtShift) | at;
        return MI_OK;
    }
    std::shared_ptr<void> keep(release_arg, release ? release : +[](void*) {});
    if (!b || !src || !at_out) return MI_ERR_INVALID;
    mi_ctx* c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (b->staged) return fail(c, MI_ERR_STATE, "batch already ran; begin a new batch");
    int rc = staging_flush(b);                    // the inline window may hold bytes of earlier small adds
    if (rc) return rc;
    const u64 at = align_up(b->arena_used, kFileAlign);
    rc = arena_reserve(b, at + align_up(len, kFileAlign));
    if (rc == MI_OK) rc = ensure_stager(c);
    if (rc) return rc;
    b->arena_used = at + len;
    *at_out = at;
    rc = stager_put_block(c->stager, b, at, src, len, std::move(keep));
    b->staged_any = true;
    return rc;
}
// sums (optional; a batch that keeps sums): 2 words per file -- the (a, b) of mi_filesum.h over the file's bytes as its reader
// took them out of the file; a placed file is at most one chunk long
extern "C" int mi_batch_add_placed(mi_batch* b, uint64_t n, const uint64_t* arena_off, const uint64_t* sizes,
                                   const uint64_t* tags, const uint64_t* sums) {
    if (!b || (n && (!arena_off || !sizes))) return MI_ERR_INVALID;
    if (is_group(b)) {                               // rows of blocks that went to different members, in the walk's order
        const size_t nm = b->members.size();
        std::vector<std::vector<u64>> mo(nm), ms(nm), mt(nm), mq(nm);
        for (u64 i = 0; i < n; ++i) {
            const size_t k = (size_t)(arena_off[i] >> kGroupAtShift);
            if (k >= nm) return fail(b->ctx, MI_ERR_INVALID, "mi_batch_add_placed: no such member");
            b->row_member.push_back((u32)k);
            b->row_row.push_back(b->members[k]->files.size() + mo[k].size());
            mo[k].push_back(arena_off[i] & kGroupAtMask);
            ms[k].push_back(sizes[i]);
            mt[k].push_back(tags ? tags[i] : 0);
            if (sums) { mq[k].push_back(sums[2 * i]); mq[k].push_back(sums[2 * i + 1]); }
            b->total_bytes += sizes[i];
        }
        for (size_t k = 0; k < nm; ++k) {
            if (mo[k].empty()) continue;
            const int rc = mi_batch_add_placed(b->members[k], mo[k].size(), mo[k].data(), ms[k].data(), mt[k].data(), sums ? mq[k].data() : nullptr);
            if (rc) return group_fail(b, k, rc);
        }
        return MI_OK;
    }
    if (b->staged) return fail(b->ctx, MI_ERR_STATE, "batch already ran; begin a new batch");
    for (u64 i = 0; i < n; ++i) {
        if (sizes[i] > b->arena_used || arena_off[i] > b->arena_used - sizes[i])          // (no sum: it could wrap)
            return fail(b->ctx, MI_ERR_INVALID, "mi_batch_add_placed: outside the arena");
        b->files.push_back({arena_off[i], sizes[i], tags ? tags[i] : 0});
        if (b->keep_sums && sums && sizes[i] <= mi_sum::kChunk) {
            mi_sum::FileSum* fsum = b->sum_pool.take(1);
            fsum->a.store(sums[2 * i], std::memory_order_relaxed);
            fsum->b.store(sums[2 * i + 1], std::memory_order_relaxed);
            b->files.back().sums = fsum;
        }
        b->total_bytes += sizes[i];
    }
    return MI_OK;
}
extern "C" int mi_batch_keeps_sums(mi_batch* b) { return b && (is_group(b) ? b->members[0]->keep_sums : b->keep_sums) ? 1 : 0; }
extern "C" void mi_batch_keep_sums(mi_batch* b, int on) {
    if (b && is_group(b)) { if (b->row_member.empty()) for (mi_batch* m : b->members) mi_batch_keep_sums(m, on); return; }
    if (b && b->files.empty()) b->keep_sums = on != 0;
}

// Room for what the caller knows is coming: the arena grows ONCE, now, instead of in steps under way -- every growth has
// to drain the reader threads first (copies in flight target the old arena) and moves what the arena already holds.
static int batch_reserve(mi_batch* b, uint64_t more_files, uint64_t more_bytes, bool ahead);
int mi_batch_reserve(mi_batch* b, uint64_t more_files, uint64_t more_bytes) { return batch_reserve(b, more_files, more_bytes, false); }
// the walk's form (and mi_memfs_reserve_device's): what is coming is known only roughly and keeps growing -- a piecewise arena
int mi_batch_reserve_ahead(mi_batch* b, uint64_t more_files, uint64_t more_bytes) { return batch_reserve(b, more_files, more_bytes, true); }
static int batch_reserve(mi_batch* b, uint64_t more_files, uint64_t more_bytes, bool ahead) {
    if (!b) return MI_ERR_INVALID;
    if (is_group(b)) {                               // every member its share and a quarter (the split is by bytes, not exact)
        const u64 nm = b->members.size();
        for (size_t k = 0; k < nm; ++k) {
            const int rc = batch_reserve(b->members[k], more_files / nm + 1, more_bytes / nm + more_bytes / (4 * nm), ahead);
            if (rc) return group_fail(b, k, rc);
        }
        return MI_OK;
    }
    mi_ctx* c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (b->staged) return fail(c, MI_ERR_STATE, "batch already ran; begin a new batch");
    if (more_files == 0 && more_bytes == 0) return MI_OK;
    b->files.reserve(b->files.size() + more_files);
    u64 want = align_up(b->arena_used, kFileAlign) + more_bytes + (more_files + 1) * kFileAlign;
    if (guard_alloc()) want = align_up(b->arena_used, kFileAlign) + more_bytes;     // the audit: exactly what was said
    else if ((b->arena.p ? b->arena_plain : !ahead) && want + 4096 > b->arena.bytes) {
        // it has to grow: then by a step worth the drain -- at least twice what there is, at least 64 MiB (a walk that
        // reserves as it enumerates would otherwise grow a 400 MB layer fifteen times)
        const u64 step = 2 * b->arena.bytes > (64ull << 20) ? 2 * b->arena.bytes : (64ull << 20);
        if (want < step) want = step;
    }
    return arena_reserve(b, want, true, ahead);
}

// A file that is a byte range of another file: a member of an uncompressed layer tar
// (mi_tar_entries gives the ranges).
int mi_batch_add_path_range(mi_batch* b, const char* path, uint64_t offset, uint64_t size, uint64_t user_tag) {
    return add_file_range(b, path, offset, size, user_tag);
}

int mi_batch_add_synthetic(mi_batch* b, uint64_t n_files, const uint64_t* sizes,
                           const uint64_t* content_ids, uint64_t seed) {
    if (!b || (!sizes && n_files)) return MI_ERR_INVALID;
    mi_ctx* c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (b->staged) return fail(c, MI_ERR_STATE, "batch already ran; begin a new batch");
    // synthetic files are generated on the device at run time; flush any host window first
    int rc = staging_flush(b);
    if (rc) return rc;
    SynthSpec sp;
    sp.f0 = b->files.size();
    sp.n = n_files;
    sp.seed = seed;
    sp.cids.resize(n_files);
    u64 end = b->arena_used;
    for (u64 i = 0; i < n_files; ++i) {
        const u64 at = align_up(end, kFileAlign);
        const u64 cid = content_ids ? content_ids[i] : sp.f0 + i;
        sp.cids[i] = cid;
        b->files.push_back({at, sizes[i], cid});
        end = at + sizes[i];
        b->total_bytes += sizes[i];
    }
    rc = arena_reserve(b, align_up(end, kFileAlign));
    if (rc) return rc;
    b->arena_used = end;
    b->synth.push_back(std::move(sp));
    return MI_OK;
}

// Stages everything that was added (flush of the pinned ring, file tables, synthetic
// generation).  Done once per batch, before its first submit.
static int stage_batch(mi_batch* b);

// ---- parts: one file split across batches / GPUs ------------------------------
static int part_geometry(mi_batch* b, uint64_t file_size, uint64_t begin, uint64_t end, PartRec* pr) {
    mi_ctx* c = b->ctx;
    if (begin >= end || end > file_size)
        return fail(c, MI_ERR_INVALID, "part [%llu, %llu) of a %llu-byte file", (unsigned long long)begin,
                    (unsigned long long)end, (unsigned long long)file_size);
    if (begin % MI_PART_ALIGN || (end % MI_PART_ALIGN && end != file_size))
        return fail(c, MI_ERR_INVALID, "part bounds must be multiples of MI_PART_ALIGN (the end may be the file's)");
    u64 hg = 0;
    if (begin) {
        hg = (c->cfg.max_size + kGroupBytes - 1) / kGroupBytes;      // >= max_size bytes of halo
        if (hg == 0) hg = 1;
        if (hg * kGroupBytes > begin) hg = begin / kGroupBytes;      // the file starts inside it
    }
    pr->file_index = b->files.size();
    pr->file_size = file_size;
    pr->begin = begin;
    pr->end = end;
    pr->halo_groups = (u32)hg;
    pr->halo_bytes = hg * kGroupBytes;
    pr->confirmed = hg == 0;
    return MI_OK;
}

int mi_batch_add_path_part(mi_batch* b, const char* path, uint64_t file_size, uint64_t begin, uint64_t end,
                           uint64_t user_tag) {
    if (!b || !path) return MI_ERR_INVALID;
    PartRec pr;
    int rc = part_geometry(b, file_size, begin, end, &pr);
    if (rc) return rc;
    rc = add_file_range(b, path, begin - pr.halo_bytes, pr.halo_bytes + (end - begin), user_tag, begin - pr.halo_bytes);
    if (rc) return rc;
    b->files.back().part = (int)b->parts.size();
    b->parts.push_back(pr);
    return MI_OK;
}

int mi_batch_add_synthetic_part(mi_batch* b, uint64_t file_size, uint64_t content_id, uint64_t seed,
                                uint64_t begin, uint64_t end) {
    if (!b) return MI_ERR_INVALID;
    PartRec pr;
    int rc = part_geometry(b, file_size, begin, end, &pr);
    if (rc) return rc;
    const uint64_t size = pr.halo_bytes + (end - begin);
    rc = mi_batch_add_synthetic(b, 1, &size, &content_id, seed);
    if (rc) return rc;
    b->synth.back().unit0 = (begin - pr.halo_bytes) / 16;
    b->files.back().part = (int)b->parts.size();
    b->parts.push_back(pr);
    return MI_OK;
}

int mi_batch_scan_cuts(mi_batch* b) {
    if (!b) return MI_ERR_INVALID;
    mi_ctx* c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (b->in_flight) return fail(c, MI_ERR_STATE, "batch is in flight; mi_batch_wait first");
    int rc = stage_batch(b);
    if (rc) return rc;
    if (b->files.empty()) return MI_OK;
    b->cuts_ready = false;
    if ((rc = enqueue_cuts(b))) return rc;
    if ((rc = read_part_states(b))) return rc;
    HIPCHK(c, hipGetLastError());
    b->cuts_ready = true;
    return MI_OK;
}

int mi_batch_parts(mi_batch* b, mi_part_state* out, uint64_t cap, uint64_t* n_parts) {
    if (!b || (!out && cap)) return MI_ERR_INVALID;
    if (n_parts) *n_parts = b->parts.size();
    for (size_t i = 0; i < b->parts.size() && i < cap; ++i) {
        const PartRec& p = b->parts[i];
        const u64 base = p.begin - p.halo_bytes;
        mi_part_state& o = out[i];
        memset(&o, 0, sizeof o);
        o.file_index = p.file_index;
        o.file_size = p.file_size;
        o.begin = p.begin;
        o.end = p.end;
        o.entry = base + (p.set_entry_rel != ~0ull ? p.set_entry_rel : p.entry_rel);
        o.exit = base + p.exit_rel;
        o.entry_confirmed = p.confirmed ? 1u : 0u;
        o.cuts_current = (p.set_entry_rel == ~0ull || p.set_entry_rel == p.entry_rel) ? 1u : 0u;
    }
    return MI_OK;
}

int mi_batch_set_part_entry(mi_batch* b, uint64_t file_index, uint64_t entry) {
    if (!b) return MI_ERR_INVALID;
    mi_ctx* c = b->ctx;
    if (b->in_flight) return fail(c, MI_ERR_STATE, "batch is in flight; mi_batch_wait first");
    if (file_index >= b->files.size() || b->files[file_index].part < 0)
        return fail(c, MI_ERR_INVALID, "file %llu is not a part", (unsigned long long)file_index);
    PartRec& p = b->parts[b->files[file_index].part];
    if (p.halo_groups == 0) {
        if (entry != 0) return fail(c, MI_ERR_INVALID, "a file's first part starts at its first byte");
        return MI_OK;
    }
    const u64 base = p.begin - p.halo_bytes;
    if (entry > p.begin || entry < base || p.begin - entry > c->cfg.max_size)
        return fail(c, MI_ERR_INVALID, "entry cut %llu cannot precede a part that begins at %llu (max chunk %u)",
                    (unsigned long long)entry, (unsigned long long)p.begin, c->cfg.max_size);
    p.set_entry_rel = entry - base;
    p.confirmed = true;
    if (p.set_entry_rel != p.entry_rel || !b->cuts_ready) b->parts_dirty = true;
    b->results_valid = false;
    return MI_OK;
}

int mi_batch_fix_cuts(mi_batch* b) {
    if (!b) return MI_ERR_INVALID;
    mi_ctx* c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (b->in_flight) return fail(c, MI_ERR_STATE, "batch is in flight; mi_batch_wait first");
    if (!b->cuts_ready) return fail(c, MI_ERR_STATE, "mi_batch_fix_cuts without mi_batch_scan_cuts");
    if (!b->parts_dirty) return MI_OK;
    int rc = enqueue_cuts(b);                   // cuts_ready: only the parts' entry override + fix-up
    if (rc) return rc;
    if ((rc = read_part_states(b))) return rc;
    HIPCHK(c, hipGetLastError());
    b->cuts_ready = true;
    return MI_OK;
}

static int stage_batch(mi_batch* b) {
    mi_ctx* c = b->ctx;
    if (b->staged) return MI_OK;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = staging_flush(b);
    if (rc) return rc;
    rc = staging_sync(b);                       // everything the reader threads hold has landed
    if (rc) return rc;
    rc = arena_wait_mapped(c, &b->arena, b->arena.bytes);   // ... and the kernels may touch the slack behind the last file
    if (rc) return rc;
    if (!b->stage_err.empty())                  // sticky (a failed final verification, a batch without readers)
        return fail(c, MI_ERR_IO, "%s", b->stage_err.c_str());
    if (c->verify_staging && (rc = stage_verify_final(b))) return rc;
    if (b->staged_any)          // host->device staging time: ring memcpy/pread + waits + final drain
        b->ms_h2d += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    const u64 nf = b->files.size();
    // Segment tables (gear_cdc.hip): files in add order, a large file's 256 KiB groups in place.
    std::vector<u64> off(nf), size(nf), seg0(nf + 1), seg_slot;
    std::vector<u32> small, seg_file, seg_group, gfile, gindex, large, large_g0;
    const u64 region = gear_group_region(c->cfg.min_size);
    u64 cap_chunks = 0, ends_total = 0, mx = 0;
    bool any_group = false;
    for (u64 f = 0; f < nf; ++f)
        if (b->files[f].size > (u64)kGearTile || b->files[f].part >= 0) { any_group = true; break; }
    seg_file.reserve(nf);
    seg_slot.reserve(nf);
    for (u64 f = 0; f < nf; ++f) {
        off[f] = b->files[f].off;
        size[f] = b->files[f].size;
        mx = size[f] > mx ? size[f] : mx;
        seg0[f] = seg_file.size();
        cap_chunks += size[f] / c->cfg.min_size + 2;          // upper bound of the file's chunk count
        if (size[f] <= (u64)kGearTile && b->files[f].part < 0) {
            small.push_back((u32)seg_file.size());
            seg_file.push_back((u32)f);
            seg_slot.push_back(ends_total);
            if (any_group) seg_group.push_back(0xFFFFFFFFu);
            ends_total += size[f] / c->cfg.min_size + 2;
        } else {
            const u64 ng = gear_large_groups(size[f]);
            if (gfile.size() + ng >= 0xFFFFFFFFull || seg_file.size() + ng >= 0xFFFFFFFFull)
                return fail(c, MI_ERR_INVALID, "batch too large: more than 2^32 tile groups");
            large.push_back((u32)f);
            large_g0.push_back((u32)gfile.size());
            if (b->files[f].part >= 0) {
                PartRec& pr = b->parts[b->files[f].part];
                pr.group0 = (u32)gfile.size();
                pr.n_groups = (u32)ng;
            }
            for (u64 gi = 0; gi < ng; ++gi) {
                seg_group.push_back((u32)gfile.size());
                gfile.push_back((u32)f);
                gindex.push_back((u32)gi);
                seg_file.push_back((u32)f);
                seg_slot.push_back(ends_total);
                ends_total += 2 * region;                     // speculative list + prefix
            }
        }
    }
    seg0[nf] = seg_file.size();
    if (seg_file.size() >= 0xFFFFFFFFull)
        return fail(c, MI_ERR_INVALID, "batch too large: more than 2^32 segments");
    {
        u64 nodes = mx / c->cfg.min_size + 2;              // upper bound of a file's chunk count
        b->root_passes = 0;
        while (nodes > kChunkRootFanout) { nodes = (nodes + kChunkRootFanout - 1) / kChunkRootFanout; ++b->root_passes; }
        if (b->root_passes > kMaxRootPasses) return fail(c, MI_ERR_INVALID, "file too large for the root tree");
    }
    b->total_slots = cap_chunks;
    b->ends_total = ends_total;
    b->n_segs = seg_file.size();
    b->n_small = (u32)small.size();
    b->n_groups = (u32)gfile.size();
    b->n_large = (u32)large.size();
    if ((rc = upload(c, b->small_list, small))) return rc;
    if ((rc = upload(c, b->seg_file, seg_file))) return rc;
    if ((rc = upload(c, b->seg_slot, seg_slot))) return rc;
    if ((rc = upload(c, b->seg_group, seg_group))) return rc;
    if ((rc = upload(c, b->file_seg0, seg0))) return rc;
    if ((rc = upload(c, b->group_file, gfile))) return rc;
    if ((rc = upload(c, b->group_index, gindex))) return rc;
    if ((rc = upload(c, b->large_list, large))) return rc;
    if ((rc = upload(c, b->large_group0, large_g0))) return rc;
    if ((rc = upload(c, b->file_off, off))) return rc;
    if ((rc = upload(c, b->file_size, size))) return rc;
    b->fsha_host.clear();
    std::vector<u64> fl;                                    // (alive until the uploads below have been waited for)
    if (c->cfg.flags & MI_FLAG_FILE_SHA256) {
        // whole-file digests: a file too long for a GPU lane goes to the reader threads' SHA-NI streams (route_long_strings);
        // the GPU pass sees it as an empty string.  (Parts have no whole-file values.)
        fl = size;
        for (u64 f = 0; f < nf; ++f) if (b->files[f].part >= 0) fl[f] = 0;
        route_long_strings(fl.data(), nf, c->stage_threads, false, &b->fsha_host);
        for (u32 f : b->fsha_host) fl[f] = 0;
        if ((rc = upload(c, b->fsha_len, fl))) return rc;
        if (!b->fsha_host.empty() && (rc = ensure_stager(c))) return rc;
    }
    std::vector<u32> fflags, pfile, pg0, phalo;
    if (!b->parts.empty()) {
        fflags.assign(nf, 0u);
        for (const PartRec& pr : b->parts) {
            if (pr.end < pr.file_size) fflags[pr.file_index] |= kFileOpenEnd;
            pfile.push_back((u32)pr.file_index);
            pg0.push_back(pr.group0);
            phalo.push_back(pr.halo_groups);
        }
        if ((rc = upload(c, b->file_flags, fflags))) return rc;
        if ((rc = upload(c, b->part_file, pfile))) return rc;
        if ((rc = upload(c, b->part_group0, pg0))) return rc;
        if ((rc = upload(c, b->part_halo, phalo))) return rc;
    }
    std::vector<u32> tile_file;
    std::vector<u64> first_tile;
    if (c->cfg.flags & MI_FLAG_FILE_CRC32) {
        first_tile.resize(nf);
        for (u64 f = 0; f < nf; ++f) {
            first_tile[f] = tile_file.size();
            const u64 nt = (size[f] + kGearTile - 1) / kGearTile;
            tile_file.insert(tile_file.end(), nt, (u32)f);
        }
        b->n_tiles = tile_file.size();
        if ((rc = upload(c, b->tile_file, tile_file))) return rc;
        if ((rc = upload(c, b->first_tile, first_tile))) return rc;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));    // the vectors above go out of scope
    for (const SynthSpec& sp : b->synth) {
        HIPCHK(c, b->cids.ensure(sp.n * 8 + 16));
        HIPCHK(c, hipMemcpyAsync(b->cids.p, sp.cids.data(), sp.n * 8, hipMemcpyHostToDevice,
                                 c->stream));
        launch_synth_fill(b->arena.as<u8>(), b->file_off.as<u64>() + sp.f0,
                          b->file_size.as<u64>() + sp.f0, b->cids.as<u64>(), sp.n, sp.seed, sp.unit0,
                          c->stream);
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    b->staged = true;
    return MI_OK;
}

int mi_batch_submit(mi_batch* b) {
    if (!b) return MI_ERR_INVALID;
    mi_ctx* c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (b->in_flight) return fail(c, MI_ERR_STATE, "batch is already in flight; mi_batch_wait first");
    int rc = stage_batch(b);
    if (rc) return rc;
    return submit_pipeline(b);
}

int mi_batch_wait(mi_batch* b) {
    if (!b) return MI_ERR_INVALID;
    HIPCHK(b->ctx, hipSetDevice(b->ctx->device));
    return wait_pipeline(b);
}

// The parts of the group's split files agree on their boundary cuts (the parts protocol of include/makisu_mi.h, all owners in
// this process): every member that holds parts makes its cuts under an assumed entry; then, round by round, every part but a
// file's first is told its predecessor's last cut and the members re-select where that differed -- until no exit moved (one
// round on ordinary data, at most parts-per-file).
static int group_resolve_parts(mi_batch* h) {
    const size_t nm = h->members.size();
    std::vector<char> holds(nm, 0);
    for (const mi_batch::Split& sp : h->splits) for (const mi_batch::SplitPart& pt : sp.parts) holds[pt.member] = 1;
    {
        std::vector<int> rcs(nm, MI_OK);
        std::vector<std::thread> th;
        for (size_t k = 0; k < nm; ++k) if (holds[k]) th.emplace_back([&, k] { rcs[k] = mi_batch_scan_cuts(h->members[k]); });
        for (auto& t : th) t.join();
        for (size_t k = 0; k < nm; ++k) if (rcs[k]) return group_fail(h, k, rcs[k]);
    }
    for (int round = 0; round < 70; ++round) {
        std::vector<std::vector<mi_part_state>> st(nm);
        for (size_t k = 0; k < nm; ++k) {
            if (!holds[k]) continue;
            uint64_t np = 0;
            mi_batch_parts(h->members[k], nullptr, 0, &np);
            st[k].resize(np ? np : 1);
            const int rc = mi_batch_parts(h->members[k], st[k].data(), np, &np);
            if (rc) return group_fail(h, k, rc);
            st[k].resize(np);
        }
        auto state_of = [&](const mi_batch::SplitPart& pt) -> const mi_part_state* {
            for (const mi_part_state& x : st[pt.member]) if (x.file_index == pt.row) return &x;
            return nullptr;
        };
        bool redo = false;
        for (const mi_batch::Split& sp : h->splits)
            for (size_t i = 1; i < sp.parts.size(); ++i) {
                const mi_part_state *prev = state_of(sp.parts[i - 1]), *cur = state_of(sp.parts[i]);
                if (!prev || !cur) return mi::fail(h->ctx, MI_ERR_STATE, "a split file's part is not among its member's parts");
                if (prev->exit != cur->entry) redo = true;
                if (prev->exit != cur->entry || !cur->entry_confirmed) {
                    const int rc = mi_batch_set_part_entry(h->members[sp.parts[i].member], sp.parts[i].row, prev->exit);
                    if (rc) return group_fail(h, sp.parts[i].member, rc);
                }
            }
        for (size_t k = 0; k < nm; ++k)
            if (holds[k]) { const int rc = mi_batch_fix_cuts(h->members[k]); if (rc) return group_fail(h, k, rc); }
        if (!redo) return MI_OK;
    }
    return mi::fail(h->ctx, MI_ERR_STATE, "the parts of a split file did not agree on their boundary cuts in 70 rounds");
}

int mi_batch_run(mi_batch* b) {
    if (!b) return MI_ERR_INVALID;
    if (is_group(b)) {                               // every member on a thread of its own: one GPU each
        const size_t nm = b->members.size();
        if (!b->splits.empty()) {
            const int rc = group_resolve_parts(b);
            if (rc) return rc;
        }
        std::vector<int> rcs(nm, MI_OK);
        std::vector<std::thread> th;
        for (size_t k = 1; k < nm; ++k) th.emplace_back([&, k] { rcs[k] = mi_batch_run(b->members[k]); });
        rcs[0] = mi_batch_run(b->members[0]);
        for (auto& t : th) t.join();
        b->n_chunks = 0;
        for (size_t k = 0; k < nm; ++k) {
            if (rcs[k]) return group_fail(b, k, rcs[k]);
            b->n_chunks += b->members[k]->n_chunks;
        }
        b->ran = true;
        return MI_OK;
    }
    if (b->ran || b->in_flight)
        return fail(b->ctx, MI_ERR_STATE, "batch already ran; use mi_batch_rerun");
    int rc = mi_batch_submit(b);
    if (rc) { b->in_flight = false; return rc; }
    return mi_batch_wait(b);
}

int mi_batch_rerun(mi_batch* b) {
    if (!b) return MI_ERR_INVALID;
    if (!b->ran) return fail(b->ctx, MI_ERR_STATE, "mi_batch_rerun before mi_batch_run");
    int rc = mi_batch_submit(b);
    if (rc) { b->in_flight = false; return rc; }
    return mi_batch_wait(b);
}

// Empties the batch for the next set of files; every device allocation (arena, tables) and the
// pinned window stay, so a host that scans layer after layer pays hipMalloc -- and the driver's
// clearing of fresh VRAM, which competes with the H2D copies for the SDMA engines -- once.
static void read_windows_drop(mi_batch* b);
int mi_batch_reset(mi_batch* b) {
    if (!b) return MI_ERR_INVALID;
    if (is_group(b)) {
        for (size_t k = 0; k < b->members.size(); ++k) {
            const int rc = mi_batch_reset(b->members[k]);
            if (rc) return group_fail(b, k, rc);
        }
        if (b->tree) { mi_batch_tree_free(b->tree); b->tree = nullptr; }
        b->row_member.clear();
        b->row_row.clear();
        b->splits.clear();
        b->member_bytes.assign(b->members.size(), 0);
        b->total_bytes = 0;
        b->n_chunks = 0;
        b->ran = false;
        return MI_OK;
    }
    mi_ctx* c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (b->in_flight) return fail(c, MI_ERR_STATE, "batch is in flight; mi_batch_wait first");
    int rc = staging_sync(b);                           // nothing may still be writing into the arena
    (void)rc;                                           // a sticky staging failure ends here
    if (b->fsha_latch) { (void)stager_hash_wait(c, b->fsha_latch); b->fsha_latch = nullptr; }
    b->fsha_host.clear();
    b->stage_err.clear();
    b->stage_note.clear();
    {
        std::lock_guard<std::mutex> g(b->span_mu);
        b->stage_spans.clear();
        memset(&b->stage_stats, 0, sizeof b->stage_stats);
    }
    if (b->tree) { mi_batch_tree_free(b->tree); b->tree = nullptr; }
    b->files.clear();
    b->sum_pool.clear();
    b->synth.clear();
    b->parts.clear();
    b->cuts_ready = b->parts_dirty = false;
    b->total_bytes = b->arena_used = 0;
    b->cur = 0;
    b->win_start = b->win_fill = 0;
    b->staged_any = false;
    b->ms_h2d = 0;
    b->staged = b->ran = b->results_valid = false;
    b->h_roots_valid = false;
    read_windows_drop(b);                               // the windows held bytes of the old arena contents
    b->stage_unordered = false;
    b->n_chunks = b->total_slots = 0;
    b->n_h_files = 0;
    memset(&b->stats, 0, sizeof b->stats);
    return MI_OK;
}

int mi_batch_stage_stats(mi_batch* b, mi_stage_stats* out) {
    if (!b || !out) return MI_ERR_INVALID;
    std::lock_guard<std::mutex> g(b->span_mu);
    *out = b->stage_stats;
    return MI_OK;
}

const char* mi_batch_stage_note(mi_batch* b) {
    if (!b) return "";
    std::lock_guard<std::mutex> g(b->span_mu);
    return b->stage_note.c_str();
}

int mi_batch_counts(mi_batch* b, uint64_t* n_files, uint64_t* n_chunks, uint64_t* n_bytes) {
    if (!b) return MI_ERR_INVALID;
    if (is_group(b)) {
        if (n_files) *n_files = b->row_member.size();
        if (n_chunks) *n_chunks = b->n_chunks;
        if (n_bytes) *n_bytes = b->total_bytes;
        return MI_OK;
    }
    if (n_files) *n_files = b->files.size();
    if (n_chunks) *n_chunks = b->n_chunks;
    if (n_bytes) *n_bytes = b->total_bytes;
    return MI_OK;
}

int mi_batch_files(mi_batch* b, mi_file_result* out, uint64_t cap) {
    if (!b || (!out && cap)) return MI_ERR_INVALID;
    HIPCHK(b->ctx, hipSetDevice(b->ctx->device));
    int rc = fetch_results(b);
    if (rc) return rc;
    if (cap < b->n_h_files)
        return fail(b->ctx, MI_ERR_CAPACITY, "file result buffer holds %llu rows, need %zu",
                    (unsigned long long)cap, b->n_h_files);
    if (b->n_h_files) memcpy(out, b->h_files, b->n_h_files * sizeof(mi_file_result));
    return MI_OK;
}

int mi_batch_files_view(mi_batch* b, const mi_file_result** rows, uint64_t* n_files) {
    if (!b || !rows) return MI_ERR_INVALID;
    HIPCHK(b->ctx, hipSetDevice(b->ctx->device));
    int rc = fetch_results(b);
    if (rc) return rc;
    *rows = b->n_h_files ? b->h_files : nullptr;
    if (n_files) *n_files = b->n_h_files;
    return MI_OK;
}

int mi_batch_chunks(mi_batch* b, mi_chunk_result* out, uint64_t cap) {
    if (!b || (!out && cap)) return MI_ERR_INVALID;
    HIPCHK(b->ctx, hipSetDevice(b->ctx->device));
    int rc = fetch_results(b);
    if (rc) return rc;
    if (cap < b->n_chunks)
        return fail(b->ctx, MI_ERR_CAPACITY, "chunk result buffer holds %llu rows, need %llu",
                    (unsigned long long)cap, (unsigned long long)b->n_chunks);
    if (b->n_chunks) memcpy(out, b->rows_h, b->n_chunks * sizeof(mi_chunk_result));
    return MI_OK;
}

int mi_batch_chunks_view(mi_batch* b, const mi_chunk_result** rows, uint64_t* n_chunks) {
    if (!b || !rows) return MI_ERR_INVALID;
    HIPCHK(b->ctx, hipSetDevice(b->ctx->device));
    int rc = fetch_results(b);
    if (rc) return rc;
    *rows = b->n_chunks ? (const mi_chunk_result*)b->rows_h : nullptr;
    if (n_chunks) *n_chunks = b->n_chunks;
    return MI_OK;
}

int mi_batch_device_digests(mi_batch* b, const void** d_digests, uint64_t* n_chunks) {
    if (!b || !d_digests || !n_chunks) return MI_ERR_INVALID;
    if (!b->ran) return fail(b->ctx, MI_ERR_STATE, "digests requested before mi_batch_run");
    *d_digests = b->digests.p;
    *n_chunks = b->n_chunks;
    return MI_OK;
}

int mi_batch_device_dup_of(mi_batch* b, const void** d_dup_of, uint64_t* n_chunks) {
    if (!b || !d_dup_of || !n_chunks) return MI_ERR_INVALID;
    if (!b->ran) return fail(b->ctx, MI_ERR_STATE, "dup_of requested before mi_batch_run");
    *d_dup_of = b->dup_of.p;
    *n_chunks = b->n_chunks;
    return MI_OK;
}

int mi_batch_read_back(mi_batch* b, void* out, uint64_t cap) {
    if (!b || (!out && cap)) return MI_ERR_INVALID;
    mi_ctx* c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (!b->ran) return fail(c, MI_ERR_STATE, "read back before mi_batch_run");
    if (cap < b->total_bytes) return fail(c, MI_ERR_CAPACITY, "read-back buffer too small");
    u8* dst = (u8*)out;
    for (const auto& f : b->files) {
        if (f.size) HIPCHK(c, hipMemcpy(dst, b->arena.as<u8>() + f.off, f.size, hipMemcpyDeviceToHost));
        dst += f.size;
    }
    return MI_OK;
}

// The per-file chunk roots alone, 32 bytes a file in add order: one device-to-host copy of n_files x 32 bytes, without
// the chunk rows mi_batch_files / _chunks bring along.
int mi_batch_roots(mi_batch* b, uint8_t* out, uint64_t cap) {
    if (!b || (!out && cap)) return MI_ERR_INVALID;
    if (is_group(b)) {                               // the members' roots, put back into the walk's order
        if (!b->ran) return fail(b->ctx, MI_ERR_STATE, "roots requested before mi_batch_run");
        const u64 nf = b->row_member.size();
        if (cap < nf) return fail(b->ctx, MI_ERR_CAPACITY, "root buffer holds %llu rows, need %llu", (unsigned long long)cap, (unsigned long long)nf);
        std::vector<std::vector<u8>> mr(b->members.size());
        for (size_t k = 0; k < b->members.size(); ++k) {
            const u64 n = b->members[k]->files.size();
            mr[k].resize(n * 32 + 32);
            const int rc = mi_batch_roots(b->members[k], mr[k].data(), n);
            if (rc) return group_fail(b, k, rc);
        }
        for (u64 g = 0; g < nf; ++g) {
            if (b->row_member[g] != kGroupSplit) { memcpy(out + 32 * g, mr[b->row_member[g]].data() + 32 * b->row_row[g], 32); continue; }
            // a split file's root: mi_chunk_root over its parts' chunk digests put end to end (include/makisu_mi.h, "parts")
            std::vector<u8> dg;
            for (const mi_batch::SplitPart& pt : b->splits[b->row_row[g]].parts) {
                const mi_file_result* fr = nullptr;
                const mi_chunk_result* cr = nullptr;
                uint64_t n1 = 0, n2 = 0;
                int rc = mi_batch_files_view(b->members[pt.member], &fr, &n1);
                if (!rc) rc = mi_batch_chunks_view(b->members[pt.member], &cr, &n2);
                if (rc) return group_fail(b, pt.member, rc);
                const mi_file_result& f = fr[pt.row];
                for (u64 j = 0; j < f.n_chunks; ++j) { const u8* d = cr[f.first_chunk + j].sha256; dg.insert(dg.end(), d, d + 32); }
            }
            const int rc = mi_chunk_root(dg.data(), dg.size() / 32, out + 32 * g);
            if (rc) return fail(b->ctx, rc, "the root of a split file");
        }
        return MI_OK;
    }
    mi_ctx* c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (!b->ran) return fail(c, MI_ERR_STATE, "roots requested before mi_batch_run");
    const u64 nf = b->files.size();
    if (cap < nf) return fail(c, MI_ERR_CAPACITY, "root buffer holds %llu rows, need %llu", (unsigned long long)cap, (unsigned long long)nf);
    if (!b->h_roots_valid) {
        b->h_roots.resize(nf * 32);
        if (nf) HIPCHK(c, hipMemcpy(b->h_roots.data(), b->roots.p, nf * 32, hipMemcpyDeviceToHost));
        b->h_roots_valid = true;
    }
    if (nf) memcpy(out, b->h_roots.data(), nf * 32);
    return MI_OK;
}

// Bytes [offset, offset + len) of file `file_index` as they lie in HBM, through pinned windows: a copy brings more than was
// asked for (files that were staged together lie together; a layer's files are asked for in nearly that order); its length
// doubles while reads continue where the last window ended -- 256 KiB when a layer picks single files out of a tree, 8 MiB when
// it streams -- and a reader that streams finds the NEXT range already on its way into the second window: the copy of one
// overlaps the consumption of the other (round 6: with one window every 1 MiB the tar writer asked for was a copy of 1 MiB it
// waited for -- 6 192 of them for 48 x 128 MiB -- and while eight reader threads kept PCIe busy each wait was long enough to
// leave the layer's SHA-256 thread without work: 0.03-0.11 s of a 2.7 s commit, profiles/r06_commit_large_before.txt).
constexpr u64 kReadWinBytes = 8ull << 20, kReadWinMin = 256ull << 10;
static int read_windows(mi_batch* b) {
    mi_ctx* c = b->ctx;
    if (b->rb[0].p) return MI_OK;
    HIPCHK(c, hipStreamCreateWithFlags(&b->rb_stream, hipStreamNonBlocking));
    for (auto& w : b->rb) {
        HIPCHK(c, hipHostMalloc(&w.p, kReadWinBytes, hipHostMallocDefault));
        HIPCHK(c, hipEventCreateWithFlags(&w.ev, hipEventDisableTiming));
        w.len = 0;
        w.pending = false;
    }
    b->rb_next = kReadWinMin;
    return MI_OK;
}
// a copy into a window has completed.  MI_STAGE_FAULT=readback:N[:K] (tests): the N-th .. N+K-1-th arrive with a byte flipped
static void window_arrived(mi_batch* b, void* p, u64 len) {
    mi_ctx* c = b->ctx;
    const long long k = (long long)b->rb_copies++;
    if (c->fault_readback >= 0 && k >= c->fault_readback && k < c->fault_readback + c->fault_readback_n && len) ((u8*)p)[len / 2] ^= 0x20;
}
// nothing of the windows is valid any more (the arena's contents changed); a copy under way is waited for
static void read_windows_drop(mi_batch* b) {
    for (auto& w : b->rb) {
        if (w.pending) (void)hipEventSynchronize(w.ev);
        w.pending = false;
        w.len = 0;
    }
    b->rb_next = kReadWinMin;
}
// while_staging: the caller is the pipelined commit (mi_memfs.hip) -- the batch is still being staged and scanned by another
// thread; the file's bytes are waited for (stager_wait_landed), nothing else of the batch's state is touched
// own_range: the row is a PART and `offset` a FILE offset inside the part's own range [begin, end) (a split file of a batch group)
static int read_file_impl(mi_batch* b, uint64_t file_index, uint64_t offset, void* dst, uint64_t len, bool while_staging, bool own_range = false) {
    if (!b || (!dst && len)) return MI_ERR_INVALID;
    if (is_group(b)) {                               // from the GPU that holds the file -- a split file: from the GPUs that hold its parts
        if (file_index >= b->row_member.size()) return fail(b->ctx, MI_ERR_INVALID, "mi_batch_read_file: no file %llu", (unsigned long long)file_index);
        if (b->row_member[file_index] == kGroupSplit) {
            const mi_batch::Split& sp = b->splits[b->row_row[file_index]];
            if (offset > sp.size || len > sp.size - offset) return fail(b->ctx, MI_ERR_INVALID, "mi_batch_read_file: outside split file %llu", (unsigned long long)file_index);
            u8* d = (u8*)dst;
            for (const mi_batch::SplitPart& pt : sp.parts) {
                if (!len) break;
                if (offset >= pt.end) continue;
                const u64 take = std::min(len, pt.end - offset);
                const int rc = read_file_impl(b->members[pt.member], pt.row, offset, d, take, while_staging, true);
                if (rc) return group_fail(b, pt.member, rc);
                d += take;
                offset += take;
                len -= take;
            }
            return MI_OK;
        }
        const size_t k = b->row_member[file_index];
        const int rc = read_file_impl(b->members[k], b->row_row[file_index], offset, dst, len, while_staging);
        return rc ? group_fail(b, k, rc) : MI_OK;
    }
    mi_ctx* c = b->ctx;
    if (!while_staging && (!b->staged || b->in_flight))
        return fail(c, MI_ERR_STATE, "mi_batch_read_file: the batch is not staged, or in flight");
    if (file_index >= b->files.size()) return fail(c, MI_ERR_INVALID, "mi_batch_read_file: no file %llu", (unsigned long long)file_index);
    const mi_batch::FileRec& f = b->files[file_index];
    if (f.part >= 0) {
        if (!own_range) return fail(c, MI_ERR_INVALID, "mi_batch_read_file: file %llu is a part", (unsigned long long)file_index);
        const PartRec& pr = b->parts[f.part];
        if (offset < pr.begin || offset > pr.end || len > pr.end - offset)
            return fail(c, MI_ERR_INVALID, "mi_batch_read_file: [%llu, +%llu) is outside the part [%llu, %llu)", (unsigned long long)offset,
                        (unsigned long long)len, (unsigned long long)pr.begin, (unsigned long long)pr.end);
        offset -= f.origin;                          // from here on: an offset inside the staged range
    } else if (offset > f.size || len > f.size - offset)
        return fail(c, MI_ERR_INVALID, "mi_batch_read_file: [%llu, +%llu) is outside file %llu of %llu bytes", (unsigned long long)offset,
                    (unsigned long long)len, (unsigned long long)file_index, (unsigned long long)f.size);
    if (!len) return MI_OK;
    HIPCHK(c, hipSetDevice(c->device));
    {
        const int rc = read_windows(b);
        if (rc) return rc;
    }
    const bool staging = while_staging && c->stager;
    // the range behind `from` on its way into window `w` (not waited for): as far as the batch's bytes have landed
    auto prefetch = [&](mi_batch::ReadWin& w, u64 from) -> int {
        w.len = 0;
        if (from >= b->arena_used) return MI_OK;
        u64 want = std::min(b->rb_next, b->arena_used - from);
        if (staging) {
            const u64 landed = stager_landed(c->stager, b);
            if (landed != ~0ull) {
                if (landed <= from) return MI_OK;
                want = std::min(want, landed - from);
            }
        }
        HIPCHK(c, hipMemcpyAsync(w.p, b->arena.as<u8>() + from, want, hipMemcpyDeviceToHost, b->rb_stream));
        HIPCHK(c, hipEventRecord(w.ev, b->rb_stream));
        w.start = from;
        w.len = want;
        w.pending = true;
        ++b->rb_fetches;
        b->rb_bytes += want;
        return MI_OK;
    };
    u64 at = f.off + offset;
    u8* d = (u8*)dst;
    while (len) {
        mi_batch::ReadWin* w = &b->rb[b->rb_cur];
        if (!(w->len && at >= w->start && at < w->start + w->len)) {
            mi_batch::ReadWin* nx = &b->rb[b->rb_cur ^ 1];
            const bool follows = w->len && at == w->start + w->len;     // the reader continues where the window ended: it streams
            if (nx->len && at >= nx->start && at < nx->start + nx->len) {
                if (nx->pending) {
                    const auto tf = std::chrono::steady_clock::now();
                    HIPCHK(c, hipEventSynchronize(nx->ev));
                    b->rb_fetch_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tf).count();
                    nx->pending = false;
                    window_arrived(b, nx->p, nx->len);
                }
                b->rb_cur ^= 1;
                if (follows) b->rb_next = std::min(b->rb_next * 2, kReadWinBytes);
                const int rc = prefetch(*w, nx->start + nx->len);        // the window just left takes what follows the new one
                if (rc) return rc;
                continue;
            }
            if (nx->pending) { HIPCHK(c, hipEventSynchronize(nx->ev)); nx->pending = false; }
            nx->len = 0;
            b->rb_next = follows ? std::min(b->rb_next * 2, kReadWinBytes) : kReadWinMin;
            u64 want = std::max(b->rb_next, std::min(len, kReadWinBytes));
            want = std::min(want, b->arena_used - at);
            if (staging) {
                // what was asked for is waited for; the window then takes what ELSE has landed behind it (the neighbours that
                // will be asked for next) and nothing that is still on its way
                const u64 need = std::min(len, std::min(kReadWinBytes, f.off + f.size - at));
                u64 landed = ~0ull;
                const auto tw = std::chrono::steady_clock::now();
                const int rc = stager_wait_landed(c->stager, b, at + need, &landed);
                b->rb_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
                if (rc) return rc;
                if (landed != ~0ull) want = std::min(want, std::max(need, landed > at ? landed - at : 0));
            }
            const auto tf = std::chrono::steady_clock::now();
            HIPCHK(c, hipMemcpyAsync(w->p, b->arena.as<u8>() + at, want, hipMemcpyDeviceToHost, b->rb_stream));
            HIPCHK(c, hipStreamSynchronize(b->rb_stream));
            b->rb_fetch_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tf).count();
            window_arrived(b, w->p, want);
            w->start = at;
            w->len = want;
            w->pending = false;
            ++b->rb_fetches;
            b->rb_bytes += want;
            if (follows || want < len) {                                 // streaming (or a read longer than a window): the next range
                const int rc = prefetch(*nx, at + want);                 // sets out while this one is consumed
                if (rc) return rc;
            }
        }
        const u64 take = std::min(len, w->start + w->len - at);
        memcpy(d, (const u8*)w->p + (at - w->start), take);
        d += take;
        at += take;
        len -= take;
    }
    return MI_OK;
}
int mi_batch_read_file(mi_batch* b, uint64_t file_index, uint64_t offset, void* dst, uint64_t len) {
    return read_file_impl(b, file_index, offset, dst, len, false);
}
int mi_batch_read_file_landed(mi_batch* b, uint64_t file_index, uint64_t offset, void* dst, uint64_t len) {   // (hidden: mi_local.h)
    return read_file_impl(b, file_index, offset, dst, len, true);
}

void mi_batch_read_stats(mi_batch* b, double* wait_s, double* fetch_s, uint64_t* fetches, uint64_t* bytes) {
    if (b && is_group(b)) {
        double w = 0, f = 0;
        uint64_t nf = 0, nb = 0;
        for (mi_batch* m : b->members) { w += m->rb_wait_s; f += m->rb_fetch_s; nf += m->rb_fetches; nb += m->rb_bytes; }
        if (wait_s) *wait_s = w;
        if (fetch_s) *fetch_s = f;
        if (fetches) *fetches = nf;
        if (bytes) *bytes = nb;
        return;
    }
    if (wait_s) *wait_s = b ? b->rb_wait_s : 0;
    if (fetch_s) *fetch_s = b ? b->rb_fetch_s : 0;
    if (fetches) *fetches = b ? b->rb_fetches : 0;
    if (bytes) *bytes = b ? b->rb_bytes : 0;
}
// the layer writer's check (mi_layer.hip): the sums of chunk k (1 MiB of the FILE) of a row as they were taken where the bytes were
// read; *has = 0: the batch keeps none for this row.  A split file's chunk lies in the own range of exactly one part (parts begin
// on MiB boundaries of the file).
static const mi_batch::SplitPart* split_part_of(const mi_batch::Split& sp, u64 file_off) {
    for (const mi_batch::SplitPart& pt : sp.parts) if (file_off >= pt.begin && file_off < pt.end) return &pt;
    return nullptr;
}
int mi_batch_chunk_sum(mi_batch* b, uint64_t file_index, uint64_t k, uint64_t* sum_a, uint64_t* sum_b, int* has) {
    if (!b || !has) return MI_ERR_INVALID;
    *has = 0;
    if (is_group(b)) {
        if (file_index >= b->row_member.size()) return MI_ERR_INVALID;
        if (b->row_member[file_index] == kGroupSplit) {
            const mi_batch::Split& sp = b->splits[b->row_row[file_index]];
            const mi_batch::SplitPart* pt = split_part_of(sp, k * mi_sum::kChunk);
            if (!pt) return sp.size == 0 ? MI_OK : MI_ERR_INVALID;
            return mi_batch_chunk_sum(b->members[pt->member], pt->row, k, sum_a, sum_b, has);
        }
        return mi_batch_chunk_sum(b->members[b->row_member[file_index]], b->row_row[file_index], k, sum_a, sum_b, has);
    }
    if (file_index >= b->files.size()) return MI_ERR_INVALID;
    const mi_batch::FileRec& f = b->files[file_index];
    if (!f.sums) return MI_OK;
    const u64 k0 = f.origin / mi_sum::kChunk;
    if (k < k0 || k - k0 >= mi_sum::chunks_of(f.origin % mi_sum::kChunk + f.size)) return MI_ERR_INVALID;
    *has = 1;
    if (sum_a) *sum_a = f.sums[k - k0].a.load(std::memory_order_relaxed);
    if (sum_b) *sum_b = f.sums[k - k0].b.load(std::memory_order_relaxed);
    return MI_OK;
}
// the read-back windows (two pinned 8 MiB buffers, a stream, two events) ahead of the first read: mi_memfs_reserve_device
int mi_batch_prepare_read(mi_batch* b) {
    if (!b) return MI_ERR_INVALID;
    if (is_group(b)) {
        for (size_t k = 0; k < b->members.size(); ++k) { const int rc = mi_batch_prepare_read(b->members[k]); if (rc) return group_fail(b, k, rc); }
        return MI_OK;
    }
    HIPCHK(b->ctx, hipSetDevice(b->ctx->device));
    return read_windows(b);
}
void mi_batch_drop_windows(mi_batch* b) {
    if (b && is_group(b)) { for (mi_batch* m : b->members) read_windows_drop(m); return; }
    if (b) read_windows_drop(b);
}
// A chunk that came back from HBM with other sums than it went with, twice: WHICH hop?  The chunk once more, by a plain copy
// into memory of this call's own (not the windows, not their stream): the same sums as at the source -- HBM holds the right
// bytes and the read-back windows delivered others; other sums -- the arena does not hold what the file had when it was read.
int mi_batch_explain_chunk(mi_batch* b, uint64_t file_index, uint64_t chunk, char* msg, uint64_t cap) {
    if (b && is_group(b)) {
        if (file_index >= b->row_member.size() || !msg || cap < 16) return MI_ERR_INVALID;
        u32 member = b->row_member[file_index];
        u64 row = b->row_row[file_index];
        if (member == kGroupSplit) {
            const mi_batch::SplitPart* pt = split_part_of(b->splits[row], chunk * mi_sum::kChunk);
            if (!pt) return MI_ERR_INVALID;
            member = pt->member;
            row = pt->row;
        }
        const int n = snprintf(msg, (size_t)cap, "gpu %u: ", member);
        return mi_batch_explain_chunk(b->members[member], row, chunk, msg + n, cap - (uint64_t)n);
    }
    if (!b || !msg || !cap || file_index >= b->files.size()) return MI_ERR_INVALID;
    mi_ctx* c = b->ctx;
    const mi_batch::FileRec& f = b->files[file_index];
    const u64 off = chunk * mi_sum::kChunk;                             // in the FILE; the row's bytes begin at f.origin
    const u64 k0 = f.origin / mi_sum::kChunk;
    if (!f.sums || off < f.origin - f.origin % mi_sum::kChunk || off >= f.origin + f.size) return MI_ERR_INVALID;
    const u64 from = off > f.origin ? off : f.origin;                   // (a part's first chunk may begin before its staged bytes)
    const u64 len = std::min(off + mi_sum::kChunk, f.origin + f.size) - from;
    std::vector<u8> again(len);
    (void)hipSetDevice(c->device);
    const hipError_t e = hipMemcpy(again.data(), b->arena.as<u8>() + f.off + (from - f.origin), len, hipMemcpyDeviceToHost);
    u64 a = 0, bb = 0;
    if (e == hipSuccess) mi_sum::chunk_add(again.data(), (size_t)len, (size_t)(from - off), &a, &bb);
    const u64 wa = f.sums[chunk - k0].a.load(), wb = f.sums[chunk - k0].b.load();
    snprintf(msg, (size_t)cap, "arena [%llu, +%llu) (file %llu, bytes [%llu, +%llu)): sums where the bytes were read %016llx/%016llx; %s",
             (unsigned long long)(f.off + (from - f.origin)), (unsigned long long)len, (unsigned long long)file_index, (unsigned long long)from, (unsigned long long)len,
             (unsigned long long)wa, (unsigned long long)wb,
             e != hipSuccess ? "a third copy failed" :
             a == wa && bb == wb ? "a plain copy out of HBM has them: the arena holds the file's bytes, the hop HBM -> pinned read-back window delivered others, twice"
                                 : "a plain copy out of HBM has others too: the arena does not hold what the file held when it was read (the hop pinned slab -> HBM, or HBM itself)");
    return MI_OK;
}
int mi_batch_file_size(mi_batch* b, uint64_t file_index, uint64_t* size) {       // (internal: mi_layer.hip)
    if (b && is_group(b)) {
        if (file_index >= b->row_member.size() || !size) return MI_ERR_INVALID;
        if (b->row_member[file_index] == kGroupSplit) { *size = b->splits[b->row_row[file_index]].size; return MI_OK; }
        return mi_batch_file_size(b->members[b->row_member[file_index]], b->row_row[file_index], size);
    }
    if (!b || !size || file_index >= b->files.size() || b->files[file_index].part >= 0) return MI_ERR_INVALID;
    *size = b->files[file_index].size;
    return MI_OK;
}
int mi_batch_arena_info(mi_batch* b, uint64_t* bytes, uint64_t* pieces, uint64_t* moves) {
    if (!b) return MI_ERR_INVALID;
    if (is_group(b)) {
        uint64_t tb = 0, tp = 0, tm = 0;
        for (mi_batch* m : b->members) { uint64_t x = 0, y = 0, z = 0; mi_batch_arena_info(m, &x, &y, &z); tb += x; tp += y; tm += z; }
        if (bytes) *bytes = tb;
        if (pieces) *pieces = tp;
        if (moves) *moves = tm;
        return MI_OK;
    }
    u64 mapped = 0, n = 0;
    arena_counts(&b->arena, &mapped, &n, nullptr);
    if (bytes) *bytes = b->arena.vm ? mapped : b->arena.bytes;
    if (pieces) *pieces = b->arena.vm ? n : (b->arena.p ? 1 : 0);
    if (moves) *moves = b->arena.moves;
    return MI_OK;
}
int mi_batch_arena_room(mi_batch* b, uint64_t* bytes) {
    if (!b || !bytes) return MI_ERR_INVALID;
    if (is_group(b)) {                               // what every member can count on: the smallest
        *bytes = ~0ull;
        for (mi_batch* m : b->members) if (m->arena.bytes < *bytes) *bytes = m->arena.bytes;
        return MI_OK;
    }
    *bytes = b->arena.bytes;
    return MI_OK;
}
const char* mi_last_error_of_batch(mi_batch* b) {                // (a copy of the caller's own: see mi::fail)
    static thread_local std::string mine;
    if (!b) return "";
    std::lock_guard<std::mutex> g(g_err_mu);
    mine = b->ctx->err;
    return mine.c_str();
}

void** mi_batch_tree_slot(mi_batch* b) { return &b->tree; }
void mi_set_error(mi_batch* b, const char* msg) {                // b NULL: the message mi_last_error(NULL) returns
    std::lock_guard<std::mutex> g(g_err_mu);
    if (b) b->ctx->err = msg; else g_create_err = msg;
}

// One handle over n batches, one per ctx (hidden: mi_memfs_commit_layer_n's batch).  The head belongs to ctxs[0] (its errors are
// reported there, "gpu k of n: ..." naming the member); it has no device state of its own.
int mi_batch_group_begin(mi_ctx* const* ctxs, uint32_t n, mi_batch** out) {
    if (!ctxs || n < 2 || n > 64 || !out) return MI_ERR_INVALID;
    for (uint32_t i = 0; i < n; ++i) {
        if (!ctxs[i]) return MI_ERR_INVALID;
        for (uint32_t j = 0; j < i; ++j) if (ctxs[j] == ctxs[i]) return fail(ctxs[0], MI_ERR_INVALID, "a batch group needs %u DIFFERENT ctxs", n);
    }
    mi_batch* h = new mi_batch();
    h->ctx = ctxs[0];
    memset(&h->stats, 0, sizeof h->stats);
    memset(&h->stage_stats, 0, sizeof h->stage_stats);
    ++h->ctx->live_children;
    for (uint32_t i = 0; i < n; ++i) {
        mi_batch* m = nullptr;
        const int rc = mi_batch_begin(ctxs[i], 0, 0, &m);
        if (rc) {
            std::string e;
            { std::lock_guard<std::mutex> g(g_err_mu); e = ctxs[i]->err; }
            for (mi_batch* x : h->members) mi_batch_free(x);
            h->members.clear();
            --h->ctx->live_children;
            delete h;
            return fail(ctxs[0], rc, "gpu %u of %u: %s", i, n, e.c_str());
        }
        h->members.push_back(m);
    }
    h->member_bytes.assign(n, 0);
    *out = h;
    return MI_OK;
}
uint64_t mi_batch_group_splits(mi_batch* b) { return b ? b->splits.size() : 0; }
int mi_batch_group_members(mi_batch* b, mi_batch* const** members, const uint64_t** bytes, uint64_t* n) {
    if (!b || !n) return MI_ERR_INVALID;
    *n = b->members.size();
    if (members) *members = b->members.data();
    if (bytes) *bytes = b->member_bytes.data();
    return MI_OK;
}

int mi_batch_free(mi_batch* b) {
    if (!b) return MI_ERR_INVALID;
    if (is_group(b)) {
        if (b->tree) { mi_batch_tree_free(b->tree); b->tree = nullptr; }
        for (mi_batch* m : b->members) mi_batch_free(m);
        --b->ctx->live_children;
        delete b;
        return MI_OK;
    }
    if (b->tree) { mi_batch_tree_free(b->tree); b->tree = nullptr; }
    mi_ctx* c = b->ctx;
    (void)hipSetDevice(c->device);
    if (b->in_flight) {                         // submitted and never waited for: the stream is drained below
        b->in_flight = false;
        if (c->batches_in_flight > 0) --c->batches_in_flight;
    }
    (void)staging_sync(b);                      // reader threads may still hold pieces of this batch
    if (b->fsha_latch) { (void)stager_hash_wait(c, b->fsha_latch); b->fsha_latch = nullptr; }
    b->fsha_len.release();
    for (int i = 0; i < 2; ++i) {
        if (b->ring_ev[i]) (void)hipEventDestroy(b->ring_ev[i]);
        if (b->ring[i]) (void)hipHostFree(b->ring[i]);
    }
    if (b->ring_stream) (void)hipStreamDestroy(b->ring_stream);
    (void)hipStreamSynchronize(c->stream);
    --c->live_children;
    if (b->stream) { (void)hipStreamSynchronize(b->stream); (void)hipStreamDestroy(b->stream); }
    for (auto e : b->ev) if (e) (void)hipEventDestroy(e);
    if (b->h_counts) (void)hipHostFree(b->h_counts);
    if (b->rows_h) (void)hipHostFree(b->rows_h);
    read_windows_drop(b);
    if (b->rb_stream) { (void)hipStreamSynchronize(b->rb_stream); (void)hipStreamDestroy(b->rb_stream); }   // (before its events go)
    for (auto& w : b->rb) {
        if (w.p) (void)hipHostFree(w.p);
        if (w.ev) (void)hipEventDestroy(w.ev);
    }
    if (b->h_files) (void)hipHostFree(b->h_files);
    DevBuf* bufs[] = {&b->root_addr, &b->root_cnt, &b->rseg_cnt, &b->rseg_first, &b->rseg_total,
                      &b->root_items_off, &b->root_items_len, &b->root_level[0], &b->root_level[1],
                      &b->root_level[2], &b->root_level[3], &b->root_level[4], &b->root_addr2, &b->root_cnt2,
                      &b->group_file, &b->group_index, &b->group_recs, &b->tile_lists, &b->tile_fast, &b->large_list,
                      &b->large_group0, &b->seg_file, &b->seg_slot, &b->seg_n, &b->seg_first, &b->seg_group,
                      &b->file_seg0, &b->ends32, &b->tile_file, &b->first_tile, &b->tile_raw, &b->crc_d, &b->ctl, &b->dd_table, &b->dd_slot,
                      &b->q_off, &b->q_len, &b->q_id, &b->small_list, &b->file_off, &b->file_size, &b->cids,
                      &b->n_chunks_d, &b->first, &b->scratch,
                      &b->chunk_off, &b->chunk_len, &b->chunk_file, &b->chunk_start, &b->digests, &b->item_off, &b->item_len, &b->roots,
                      &b->file_sha, &b->dup_of, &b->file_flags, &b->part_file, &b->part_group0, &b->part_halo,
                      &b->part_entry, &b->rows_d, &b->file_rows_d, &b->file_base, &b->span_off, &b->span_len, &b->span_sums, &b->dense_list};
    for (DevBuf* d : bufs) d->release();
    arena_release(&b->arena);
    delete b;
    return MI_OK;
}

int mi_context_checksum(mi_batch* b, const void* prefix, uint64_t prefix_len,
                        const mi_ctx_entry* entries, uint64_t n, uint32_t* crc_out) {
    if (!b || !crc_out || (n && !entries) || (prefix_len && !prefix)) return MI_ERR_INVALID;
    mi_ctx* c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (!(c->cfg.flags & MI_FLAG_FILE_CRC32))
        return fail(c, MI_ERR_STATE, "mi_context_checksum needs a ctx created with MI_FLAG_FILE_CRC32");
    int rc = fetch_results(b);
    if (rc) return rc;
    // one running CRC32-IEEE over: prefix, then per walked path its relpath and
    // (symlink) the link target or (regular file) all file bytes -- the byte stream
    // checksumPathContents writes (lib/builder/step/add_copy_step.go:194-238).  File bytes
    // were reduced on the GPU; they are spliced in with crc(A||B) = crc(A)*x^(8|B|) + crc(B).
    u32 crc = crc32_host_bytes(0, prefix, prefix_len);
    for (u64 i = 0; i < n; ++i) {
        const mi_ctx_entry& e = entries[i];
        if (!e.relpath) return fail(c, MI_ERR_INVALID, "entry %llu has no relpath", (unsigned long long)i);
        crc = crc32_host_bytes(crc, e.relpath, strlen(e.relpath));
        if (e.link_target) {
            crc = crc32_host_bytes(crc, e.link_target, strlen(e.link_target));
        } else if (e.file_index >= 0) {
            if ((u64)e.file_index >= b->n_h_files)
                return fail(c, MI_ERR_INVALID, "entry %llu: file index %lld out of range",
                            (unsigned long long)i, (long long)e.file_index);
            const mi_file_result& fr = b->h_files[(size_t)e.file_index];
            crc = crc32_host_combine(crc, fr.crc32, fr.size);
        }
    }
    *crc_out = crc;
    return MI_OK;
}

int mi_dedup_mark(mi_ctx* c, const void* d_digests, uint64_t n, void* d_dup_of, uint64_t* n_unique) {
    if (!c || (n && (!d_digests || !d_dup_of))) return MI_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (n >= 0xFFFFFFFFull) return fail(c, MI_ERR_INVALID, "dedup set too large");
    u64 cap = 1024;
    while (cap < 2 * n) cap <<= 1;
    HIPCHK(c, c->dd_table.ensure(cap * 8));
    HIPCHK(c, c->dd_slot.ensure(n * 4 + 16));
    HIPCHK(c, c->dd_nuniq.ensure(8));
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    launch_dedup_mark((const u8*)d_digests, n, nullptr, c->dd_table.as<u32>(), c->dd_slot.as<u32>(), cap,
                      (i64*)d_dup_of, c->dd_nuniq.as<u64>(), true, c->stream);
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_word, c->dd_nuniq.p, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    const u64 nu = *c->h_word;
    c->stats.ms_dedup = ev_ms(c->ev[0], c->ev[1]);
    c->stats.n_unique = nu;
    if (n_unique) *n_unique = nu;
    return MI_OK;
}

// enqueue only (ctx stream): the rows' dup_of and, in c->dd_nuniq (a device word), how many of them are
// job-wide first occurrences -- mi_comm.hip all-gathers that word without a host round trip
int mi_dedup_mark_range_enqueue(mi_ctx* c, const void* d_digests, uint64_t n_total, uint64_t own_first,
                                uint64_t own_n, void* d_dup_of_own) {
    if (!c || own_first > n_total || own_n > n_total - own_first || (own_n && (!d_digests || !d_dup_of_own)))
        return MI_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (n_total >= 0xFFFFFFFFull) return fail(c, MI_ERR_INVALID, "dedup set too large");
    u64 cap = 1024;
    while (cap < 2 * own_n) cap <<= 1;
    HIPCHK(c, c->dd_table.ensure(cap * 12));
    HIPCHK(c, c->dd_tag.ensure(cap * 8));
    HIPCHK(c, c->dd_slot.ensure(own_n * 4 + 16));
    HIPCHK(c, c->dd_nuniq.ensure(8));
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    launch_dedup_mark_range((const u8*)d_digests, own_first, own_n, c->dd_table.as<u32>(), c->dd_tag.as<u64>(),
                            c->dd_slot.as<u32>(), cap, (i64*)d_dup_of_own, c->dd_nuniq.as<u64>(), c->stream);
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    return MI_OK;
}

int mi_dedup_mark_range(mi_ctx* c, const void* d_digests, uint64_t n_total, uint64_t own_first,
                        uint64_t own_n, void* d_dup_of_own, uint64_t* n_own_first) {
    int rc = mi_dedup_mark_range_enqueue(c, d_digests, n_total, own_first, own_n, d_dup_of_own);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->h_word, c->dd_nuniq.p, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    const u64 nf = *c->h_word;
    c->stats.ms_dedup = ev_ms(c->ev[0], c->ev[1]);
    if (n_own_first) *n_own_first = nf;
    return MI_OK;
}

int mi_batch_mark_global(mi_batch* b, const void* d_digests_all, uint64_t n_total, uint64_t own_first,
                         uint64_t* n_own_first) {
    if (!b) return MI_ERR_INVALID;
    mi_ctx* c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (!b->ran || b->in_flight) return fail(c, MI_ERR_STATE, "the batch must have run (and been waited for)");
    HIPCHK(c, b->dup_of.ensure(b->n_chunks * 8 + 16));      // absent when the ctx has MI_FLAG_NO_DEDUP
    int rc = mi_dedup_mark_range(c, d_digests_all, n_total, own_first, b->n_chunks, b->dup_of.p, n_own_first);
    b->results_valid = false;
    return rc;
}

int mi_batch_set_global_dedup(mi_batch* b, const void* d_dup_of_global, uint64_t first_global) {
    if (!b || !d_dup_of_global) return MI_ERR_INVALID;
    mi_ctx* c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (!b->ran) return fail(c, MI_ERR_STATE, "global dedup before mi_batch_run");
    if (b->n_chunks)
        HIPCHK(c, hipMemcpy(b->dup_of.p, (const i64*)d_dup_of_global + first_global,
                            b->n_chunks * 8, hipMemcpyDeviceToDevice));
    b->results_valid = false;
    return MI_OK;
}

int mi_sha256_many(mi_ctx* c, const void* data, const uint64_t* offsets, const uint64_t* lens,
                   uint64_t n, uint8_t* out) {
    if (!c || (n && (!offsets || !lens || !out))) return MI_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (n == 0) return MI_OK;
    if (n >= 0xFFFFFFFFull) return fail(c, MI_ERR_INVALID, "too many strings");
    u64 span = 0;
    for (u64 i = 0; i < n; ++i) if (offsets[i] + lens[i] > span) span = offsets[i] + lens[i];
    if (span && !data) return MI_ERR_INVALID;
    // A single SHA-256 stream is serial: one GPU lane does 13.5 MB/s, one host core with SHA-NI 2.4 GB/s.
    // Strings that would hold the launch up -- a `docker save` layer tar handed to `makisu push` (bin/makisu/cmd/push.go:207,230) --
    // are hashed HERE, one stream per host thread, straight from the caller's memory, while the GPU takes the many short ones
    // (route_long_strings decides; 8 x 128 MiB: 10 s on eight lanes, 0.06 s on eight cores).  This is not a fallback: it is where
    // a long Merkle-Damgard stream belongs on this machine.
    std::vector<u32> to_host;
    unsigned hw = std::thread::hardware_concurrency();
    if (const char* e = getenv("MI_SHA_HOST_THREADS")) { const int v = atoi(e); if (v >= 0 && v <= 256) hw = (unsigned)v; }
    if (hw > 16) hw = 16;
    route_long_strings(lens, n, hw, true, &to_host);
    std::vector<std::thread> pool;
    std::atomic<size_t> next{0};
    if (!to_host.empty()) {
        const unsigned nt = hw < to_host.size() ? hw : (unsigned)to_host.size();
        for (unsigned t = 0; t < nt; ++t)
            pool.emplace_back([&] {
                for (;;) {
                    const size_t k = next.fetch_add(1);
                    if (k >= to_host.size()) return;
                    const u32 i = to_host[k];
                    mi_host::Sha256 sha;
                    sha.update((const u8*)data + offsets[i], (size_t)lens[i]);
                    sha.final(out + 32 * (size_t)i);
                }
            });
    }
    struct Join { std::vector<std::thread>& p; ~Join() { for (auto& t : p) if (t.joinable()) t.join(); } } join{pool};
    // the rest: compacted (only what the GPU hashes crosses PCIe), one lane per string
    std::vector<u8> is_host(to_host.empty() ? 0 : n, 0);
    for (u32 i : to_host) is_host[i] = 1;
    std::vector<u64> g_off, g_len, g_idx;
    u64 g_bytes = 0;
    if (to_host.empty()) {
        g_bytes = span;
    } else {
        for (u64 i = 0; i < n; ++i) if (!is_host[i]) { g_off.push_back(g_bytes); g_len.push_back(lens[i]); g_idx.push_back(i); g_bytes += (lens[i] + 15) & ~15ull; }
    }
    const u64 ng = to_host.empty() ? n : g_idx.size();
    if (ng == 0) return MI_OK;
    DevBuf d_data, d_off, d_len, d_out;
    int rc = MI_OK;
    hipError_t e;
    std::vector<u8> packed, g_out;
    if (!to_host.empty()) {
        packed.resize(g_bytes);
        for (u64 k = 0; k < ng; ++k) memcpy(packed.data() + g_off[k], (const u8*)data + offsets[g_idx[k]], (size_t)g_len[k]);
        g_out.resize(ng * 32);
    }
    const void* src = to_host.empty() ? data : (const void*)packed.data();
    const u64* src_off = to_host.empty() ? offsets : g_off.data();
    const u64* src_len = to_host.empty() ? lens : g_len.data();
    u8* dst = to_host.empty() ? out : g_out.data();
    if ((e = d_data.ensure(g_bytes + 64)) != hipSuccess || (e = d_off.ensure(ng * 8)) != hipSuccess ||
        (e = d_len.ensure(ng * 8)) != hipSuccess || (e = d_out.ensure(ng * 32)) != hipSuccess) {
        rc = fail(c, MI_ERR_NOMEM, "mi_sha256_many: %s", hipGetErrorString(e));
    } else {
        if (g_bytes) e = hipMemcpy(d_data.p, src, g_bytes, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_off.p, src_off, ng * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_len.p, src_len, ng * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            launch_sha256_items(kShaBlobs, d_data.as<u8>(), d_off.as<u64>(), d_len.as<u64>(), nullptr, (u32)ng,
                                nullptr, c->heads.as<u32>(), nullptr, true, d_out.as<u8>(), c->sha,      // blobs in arrival order: flat sharing
                                c->prop.multiProcessorCount, g_bytes, c->stream);
            e = hipStreamSynchronize(c->stream);
        }
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(dst, d_out.p, ng * 32, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(c, MI_ERR_HIP, "mi_sha256_many: %s", hipGetErrorString(e));
    }
    d_data.release(); d_off.release(); d_len.release(); d_out.release();
    if (!rc && !to_host.empty())
        for (u64 k = 0; k < ng; ++k) memcpy(out + 32 * g_idx[k], g_out.data() + 32 * k, 32);
    return rc;
}

}  // extern "C"
