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
constexpr int kGearWG     = 256;                 // threads per workgroup (4 waves, 3 workgroups per CU)
constexpr int kGearTile   = 65536;               // bytes of file one wave marks at a time
constexpr int kGearHalo   = 64;                  // Gear window: h depends on <= 64 bytes
constexpr int kGearTableCopies = 8;              // LDS replicas of the Gear table

// ---- SHA-256 work queues (sha256.hip) ----------------------------------------
constexpr int kShaQueues  = 8;                   // one head word per XCD (block b runs on XCD b % 8)
constexpr int kShaWG      = 256;

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
// gear_cdc.hip
// small_list: indices of files of <= kGearTile bytes (one wave each).  Larger files are
// cut into groups of gear_large_groups(size) x 256 KiB, listed file-major in
// group_file[]/group_index[]/group_prev[] (file index, group index inside the file, ticket of
// the file's previous group); d_ticket (2 x u32: counter, chain-error flag) and d_tokens
// (16 B per group) are scratch the launcher zeroes.
u64  gear_large_groups(u64 size);
void launch_gear_cdc(const u8* d_data, const u64* d_file_off, const u64* d_file_size,
                     const u64* d_slot_base, u64* d_slot_ends, u32* d_n_chunks,
                     const u32* d_small_list, u32 n_small, const u32* d_group_file,
                     const u32* d_group_index, const u32* d_group_prev, u32 n_groups,
                     u32* d_ticket, void* d_tokens, const u64* d_gear_table, CdcParams p, int n_cu,
                     hipStream_t s);

// sha256.hip : n independent byte strings -> n digests.  Queue position p holds the string
// base[off[p] .. +len[p]) whose digest goes to out[32 * (ids ? ids[p] : p)]; positions are
// consumed in order, so put the longest strings first.  heads = kShaQueues words.
enum ShaPass { kShaChunks = 0, kShaRoots = 1, kShaFiles = 2, kShaBlobs = 3 };
// n = string count (or its upper bound when d_n, a device word holding the real count, is given)
void launch_sha256_items(ShaPass pass, const u8* d_base, const u64* d_off, const u64* d_len,
                         const u32* d_ids, u32 n, const u64* d_n, u32* d_heads, u8* d_out,
                         int blocks_per_cu, int n_cu, hipStream_t s);

// tables.hip
void launch_synth_fill(u8* d_data, const u64* d_file_off, const u64* d_file_size,
                       const u64* d_content_id, u64 n_files, u64 seed, hipStream_t s);
// n_chunks[f] -> first_chunk[f] (exclusive scan); d_total receives the sum
void launch_scan_counts(const u32* d_counts, u64* d_first, u64* d_total, u64 n,
                        u64* d_scratch, hipStream_t s);
u64  scan_scratch_elems(u64 n);
void launch_compact_chunks(const u64* d_file_off, const u64* d_slot_base, const u64* d_slot_ends,
                           const u32* d_n_chunks, const u64* d_first, u64 n_files, u64 n_max,
                           const u64* d_n, u64* d_chunk_off, u64* d_chunk_len, u32* d_chunk_file,
                           u64* d_chunk_start, u32* d_hist, u32 n_bins, u32 bin_shift,
                           hipStream_t s);
// queue descriptors in processing order (longest first): s_off/s_len/s_id[pos]
void launch_bin_order(const u64* d_off, const u64* d_len, u32 n, const u64* d_n, u32* d_hist,
                      u32* d_cursor, u32 n_bins, u32 bin_shift, u64* d_s_off, u64* d_s_len,
                      u32* d_s_id, hipStream_t s);
// chunk roots: node lists as (absolute device address, digest count) per file; one
// launch_root_level per reduction pass (files with > 1024 nodes), then the final items
void launch_root_init(const u8* d_digests, const u64* d_first, const u32* d_n_chunks, u64 n_files,
                      u64* d_cur_addr, u32* d_cur_cnt, hipStream_t s);
void launch_root_level(u64 n_files, u64* d_cur_addr, u32* d_cur_cnt, u32* d_seg_cnt, u64* d_seg_first,
                       u64* d_seg_total, u64* d_scratch, u8* d_level_out, u64* d_item_off,
                       u64* d_item_len, hipStream_t s);
void launch_root_final_items(const u64* d_cur_addr, const u32* d_cur_cnt, u64 n_files, u64* d_off,
                             u64* d_len, hipStream_t s);
// crc32.hip: constant block layout (u32 words) + launcher + host helpers for path strings
constexpr u32 kCrcPow1kOff = 1024, kCrcPowBytesOff = 1024 + 65, kCrcConstWords = 1024 + 65 + 1025;
void crc32_build_tables(u32* out /* kCrcConstWords */);
void launch_crc32_files(const u8* d_data, const u64* d_file_off, const u64* d_file_size,
                        const u32* d_tile_file, const u64* d_first_tile, u64 n_tiles, u64 n_files,
                        const u32* d_consts, u32* d_tile_raw, u32* d_crc, hipStream_t s);
u32 crc32_host_bytes(u32 crc, const void* data, size_t len);
u32 crc32_host_combine(u32 crc1, u32 crc2, u64 len2);
void launch_dedup_mark(const u8* d_digests, u64 n, const u64* d_n, u32* d_rep, u32* d_minid,
                       u32* d_slot_of, u64 cap_pow2, i64* d_dup_of, u64* d_n_unique,
                       hipStream_t s);
void launch_dedup_mark_range(const u8* d_all, u64 own_first, u64 own_n, u32* d_rep, u32* d_minid,
                             u64* d_tag, u32* d_fmin, u32* d_slot_of, u64 cap_pow2, i64* d_dup_own,
                             u64* d_n_first, hipStream_t s);

}  // namespace mi
