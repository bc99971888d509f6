// mi_common.h -- shared declarations for the gfx950 kernels and the host pipeline.
// Product code: never includes anything under oracle/.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace mi {

typedef uint8_t  u8;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int64_t  i64;

// ---- Gear-CDC tile geometry (gear_cdc.hip) ------------------------------------
constexpr int kGearWG     = 256;                 // threads per workgroup of the kernels that keep bitmaps and of the
                                                 // group kernels (4 waves = the 4 tiles of a group); the bitmap-free
                                                 // marking kernels run 512 threads around one 64 KiB table (gear_cdc.hip)
constexpr int kGearTile   = 65536;               // bytes of file one wave marks at a time
constexpr int kGearHalo   = 64;                  // Gear window: h depends on <= 64 bytes
constexpr int kGearTableCopies = 8;              // LDS replicas of the Gear table in the kernels that keep bitmaps (32 in
                                                 // the marking kernels)

// ---- SHA-256 work queues (sha256.hip) ----------------------------------------
constexpr int kShaQueues  = 8;                   // one head word per XCD (block b runs on XCD b % 8)
constexpr int kShaWG      = 256;

// chunk_root tree fan-out (tables.hip "per-file chunk roots"; the oracle's mi_ref_chunk_root)
constexpr u32 kChunkRootFanout = 64;
constexpr int kMaxRootPasses = 5;                // 64^6 nodes: more than a batch can hold rows

// file placement in the device arena: every file starts on this boundary so tile
// loads are 16-byte aligned and coalesced
constexpr u64 kFileAlign  = 256;

struct CdcParams {
    u32 thresh_m1;     // candidate iff hi32(h) <= thresh_m1  (top mask_bits bits zero)
    u32 min_size;
    u32 max_size;
    u32 pad;
};

// splitmix64 finalizer (Steele, Lea, Flood 2014) -- synthetic data + Gear table
__host__ __device__ inline u64 splitmix64_mix(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
constexpr u64 kSmGamma = 0x9E3779B97F4A7C15ULL;

// ---- kernel launchers (each defined next to its kernels) ----------------------
// gear_cdc.hip.  The unit the CDC pass writes is a SEGMENT: a file of <= kGearTile bytes, or one
// 256 KiB group of a larger file (segments are numbered in file order, a large file's groups in
// place: group gi of file f is segment file_seg0[f] + gi).  A segment's chunk ends are u32
// offsets relative to the segment start at ends32[seg_slot[s] ..]; seg_n[s] = its chunk count.
// A group segment owns 2 x gear_group_region(min_size) entries (speculative list, prefix) and a
// GroupRec; see gear_cdc.hip "large files".
struct GearLaunch {
    const u8*  data;
    const u64* file_off;
    const u64* file_size;
    const u64* file_seg0;      // first segment of every file (n_files + 1 entries)
    const u32* seg_file;
    const u64* seg_slot;
    u32*       ends32;
    u32*       seg_n;
    const u32* small_list;     // segments that are small files
    u32        n_small;
    const u32* group_file;     // per group: file, index inside the file
    const u32* group_index;
    u32        n_groups;
    const u32* large_list;     // per large file: file index, its first group
    const u32* large_group0;
    u32        n_large;
    void*      group_recs;     // n_groups x gear_group_rec_bytes()
    u32*       tile_lists;     // n_groups x 4 tiles x 64 candidates
    u32*       tile_fast;      // n_groups x 4: 1 = the tile's list is complete, 0 = dense tile
    const u32* file_flags;     // per file, kFile* bits; nullptr when the batch holds no parts
    const u64* gear_table;
    u32*       dense_list;     // n_small words + one count word: small files the bitmap-free kernel
    u32*       dense_count;    // ... hands to the bitmap kernel (more than 64 candidates in the file)
};
constexpr u32 kFileOpenEnd = 1u;     // a part that is not its file's last: no cut at its last byte
struct GroupRec {
    u64 spec_exit;      // E_g: last cut of the speculative selection (a cut assumed at the group start)
    u64 final_exit;     // last cut of the final list, i.e. under `entry`
    u64 entry;          // the previous cut the group's final list starts from
    u32 spec_n, pcnt, sidx, flags;   // final list = prefix[0, pcnt) ++ spec[sidx, spec_n)
};
constexpr u32 kGroupDense = 1u;      // some tile has more than 64 candidates: no list
constexpr u32 kGroupValid = 2u;      // pcnt / sidx / final_exit are set for `entry`
constexpr u64 kGroupBytes = (u64)kGearTile * (kGearWG / 64);
u64    gear_large_groups(u64 size);
u64    gear_group_region(u32 min_size);
size_t gear_group_rec_bytes();
void   launch_gear_cdc(const GearLaunch& a, CdcParams p, int n_cu, hipStream_t s);
// parts (split files): entry override + fix-up over the parts (refix), then the halo groups' chunk
// counts are cleared; d_part_* are per part: file index, first group, halo groups, entry (relative
// to the item's first byte, ~0 = keep the halo's own exit)
void   launch_gear_parts(const GearLaunch& a, const u32* d_part_file, const u32* d_part_group0,
                         const u32* d_part_halo, const u64* d_part_entry, u32 n_parts, bool refix,
                         CdcParams p, hipStream_t s);

// sha256.hip : n independent byte strings -> n digests.  Queue position p holds the string
// base[off[p] .. +len[p]) whose digest goes to out[32 * (ids ? ids[p] : p)]; positions are
// consumed in order, so put the longest strings first.  heads = kShaHeadWords words (two ranges of
// kShaQueues queue heads); roles = kShaRoleWords words (one arrival counter per SIMD of the device) or
// nullptr: without it every wave is equal and there is one range (strings of one length: the root passes).
constexpr int kShaHeadWords = 2 * kShaQueues;
constexpr int kShaRoleWords = 8 * 256 * 4;       // XCC x (SE SH CU of HW_ID) x SIMD
enum ShaPass { kShaChunks = 0, kShaRoots = 1, kShaFiles = 2, kShaBlobs = 3 };
// per-ctx tuning of the hashing launches (mi_config.sha_*)
struct ShaTune {
    int blocks_per_cu = 2;                   // workgroups per CU
    u64 coop_min_bytes = 9ull << 30;         // footprint from which the quad-cooperative loads are used
                                             // (0 = always, ~0 = never)
    u64 coop_min_bytes_pieces = 1ull << 30;  // the same for an arena of SMALL PIECES (mi_arena.hip, pieces under 256 MiB): the
                                             // lane-owned loads touch 64 pages per wave instruction and run at the speed of the
                                             // page-table fragments behind them (C2 chunk pass on 32 MiB pieces 5.0-5.3 ms against
                                             // 4.13 on one allocation), the cooperative loads 16 (4.2-4.4 ms on the same pieces:
                                             // profiles/r06_arena_ab.txt); below 1 GiB the lane-owned loads win on either kind
                                             // (profiles/r06_pieces_scheme_ab.txt)
    int coop_blocks_per_cu = 0;              // 0 = 3 from 24 GiB up, else blocks_per_cu
    bool pin_blocks_per_cu = true;           // pad every workgroup's LDS request so that NO CU can take more
                                             // than blocks_per_cu of them (sha256.hip launch_sha256_items)
    bool roles = true;                       // false: every wave equal, one range (the scheme before round 3)
    bool prio = false;                       // true: the first wave on a SIMD also raises s_setprio (3, the others 1).
                                             // The issue arbiter prefers the older wave anyway (3.5 vs 12 us per
                                             // iteration with or without it); with two batches in flight the raised
                                             // priorities cost 2 % (5.72 vs 5.61 ms per C2 step): off.
    int long_shift = 2;                      // the first n >> long_shift strings (the longest) go to the wave that
                                             // arrived first on each SIMD and runs at priority (sha256.hip roles)
    const char* wave_stats_path = nullptr;   // diagnostics (mi_debug_sha_wave_stats): per-wave records of every chunk pass
                                             // of this ctx are appended to this file; the ctx owns the string
};
// n = string count (or its upper bound when d_n, a device word holding the real count, is given)
// d_heads and d_roles must be zero on entry unless zero_heads (then the launcher clears them first)
void launch_sha256_items(ShaPass pass, const u8* d_base, const u64* d_off, const u64* d_len,
                         const u32* d_ids, u32 n, const u64* d_n, u32* d_heads, u32* d_roles, bool zero_heads,
                         u8* d_out, const ShaTune& tune, int n_cu, u64 footprint_bytes, hipStream_t s);
// footprint_bytes: the span of memory the strings lie in (picks the load scheme, sha256.hip kCoop)
// d_scratch: n_cu * waves_per_simd * kShaWG words
double measure_sha_valu_roof(int n_cu, int waves_per_simd, u32 blocks, u32* d_scratch, hipStream_t s,
                             hipEvent_t e0, hipEvent_t e1);

// tables.hip
// unit0: the files' first byte is byte 16 * unit0 of their content stream (0 except for parts)
void launch_synth_fill(u8* d_data, const u64* d_file_off, const u64* d_file_size,
                       const u64* d_content_id, u64 n_files, u64 seed, u64 unit0, hipStream_t s);
// n_chunks[f] -> first_chunk[f] (exclusive scan); d_total receives the sum
void launch_scan_counts(const u32* d_counts, u64* d_first, u64* d_total, u64 n,
                        u64* d_scratch, hipStream_t s);
u64  scan_scratch_elems(u64 n);
// chunk rows from the segments' end lists (scan of seg_n in d_seg_first, total in d_n); also the
// per-file first row / row count.  d_seg_group (segment -> group index, ~0u for small files),
// d_group_recs and region are only read when the batch has groups (d_seg_group != nullptr).
void launch_compact_chunks(const u64* d_file_off, const u64* d_file_seg0, const u32* d_seg_file,
                           const u64* d_seg_slot, const u32* d_ends32, const u64* d_seg_first,
                           const u32* d_seg_group, const void* d_group_recs, u32 region,
                           u64 n_files, u64 n_segs, u64 n_max, const u64* d_n, u64* d_chunk_off,
                           u64* d_chunk_len, u32* d_chunk_file, u64* d_chunk_start, u64* d_first,
                           u32* d_n_chunks, u32* d_hist, u32 n_bins, u32 bin_shift, const u8* d_digests,
                           u64* d_item_off, u64* d_item_len, hipStream_t s);
// the chunk table as mi_chunk_result rows (64 B each: file, offset (+ d_file_base[file] when given),
// length, dup_of, digest)
void launch_pack_file_rows(u64 n, const u64* d_file_size, const u64* d_first, const u32* d_n_chunks, const u32* d_crc,
                           const u8* d_roots, const u8* d_file_sha, void* d_rows, hipStream_t s);
void launch_pack_chunk_rows(u64 n, const u32* d_chunk_file, const u64* d_chunk_start, const u64* d_chunk_len,
                            const i64* d_dup_of, const u8* d_digests, const u64* d_file_base, void* d_rows,
                            hipStream_t s);
// d_hist (compaction) and d_cursor (binning) must be zero on entry; d_item_off/len (optional): the
// root pass's item list when no file needs a reduction pass
// queue descriptors in processing order (longest first): s_off/s_len/s_id[pos]
void launch_bin_order(const u64* d_off, const u64* d_len, u32 n, const u64* d_n, u32* d_hist,
                      u32* d_cursor, u32 n_bins, u32 bin_shift, u64* d_s_off, u64* d_s_len,
                      u32* d_s_id, hipStream_t s);
// chunk roots: node lists as (absolute device address, digest count) per file; one
// launch_root_level per reduction pass (files with > 1024 nodes), then the final items
void launch_root_init(const u8* d_digests, const u64* d_first, const u32* d_n_chunks, u64 n_files,
                      u64* d_cur_addr, u32* d_cur_cnt, hipStream_t s);
// one reduction pass: reads cur_*, writes next_* (the caller swaps); n_nodes_ub bounds the nodes it makes
void launch_root_level(u64 n_files, u64 n_nodes_ub, const u64* d_cur_addr, const u32* d_cur_cnt,
                       u64* d_next_addr, u32* d_next_cnt, u32* d_seg_cnt, u64* d_seg_first,
                       u64* d_seg_total, u64* d_scratch, u8* d_level_out, u64* d_item_off,
                       u64* d_item_len, hipStream_t s);
void launch_root_final_items(const u64* d_cur_addr, const u32* d_cur_cnt, u64 n_files, u64* d_off,
                             u64* d_len, hipStream_t s);
// crc32.hip: constant block layout (u32 words) + launcher + host helpers for path strings
constexpr u32 kCrcPow1kOff = 1024, kCrcPowBytesOff = 1024 + 65, kCrcPowTileOff = 1024 + 65 + 1025,
              kCrcConstWords = kCrcPowTileOff + 40;      // + x^(8 * 65536 * 2^j), j = 0..39
void crc32_build_tables(u32* out /* kCrcConstWords */);
void launch_crc32_files(const u8* d_data, const u64* d_file_off, const u64* d_file_size,
                        const u32* d_tile_file, const u64* d_first_tile, u64 n_tiles, u64 n_files,
                        const u32* d_consts, u32* d_tile_raw, u32* d_crc, hipStream_t s);
u32 crc32_host_bytes(u32 crc, const void* data, size_t len);
u32 crc32_host_combine(u32 crc1, u32 crc2, u64 len2);
// d_table: 2 x cap_pow2 words (rep | complemented minima), cleared here with one memset
void launch_dedup_mark(const u8* d_digests, u64 n, const u64* d_n, u32* d_table, u32* d_slot_of,
                       u64 cap_pow2, i64* d_dup_of, u64* d_n_unique, bool zero_count, hipStream_t s);
// d_table: 3 x cap_pow2 words (rep | minima | foreign minima)
void launch_dedup_mark_range(const u8* d_all, u64 own_first, u64 own_n, u32* d_table, u64* d_tag,
                             u32* d_slot_of, u64 cap_pow2, i64* d_dup_own, u64* d_n_first,
                             hipStream_t s);

}  // namespace mi
