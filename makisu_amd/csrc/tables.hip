// tables.hip -- the small kernels around the two hot ones: synthetic file
// generation, chunk-table compaction, longest-first binning for the SHA queues,
// per-file root items, and duplicate marking over a digest set.
//
// None of these has a reference counterpart except duplicate marking, which is the
// chunk-granular analogue of the reference's content-addressed layer dedup
// (lib/builder/step/common.go:88-91; lib/registry/client.go:123-131,177-185).
#include "mi_common.h"

namespace mi {

typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

// ---- synthetic content: BASELINE.md section 3 / oracle mi_ref_synth_fill -------
// word w of content c under seed s = mix(mix(s + (c+1)*G) + (w+1)*G); bytes little-endian.
__global__ __launch_bounds__(256)
void synth_fill_kernel(u8* __restrict__ data, const u64* __restrict__ file_off,
                       const u64* __restrict__ file_size, const u64* __restrict__ content_id,
                       u64 n_files, u64 seed, u64 unit0) {
    // blockIdx.x = file, blockIdx.y strides the file in 64 KiB pieces
    const u64 f = blockIdx.x;
    const u64 size = file_size[f];
    const u64 cid = content_id ? content_id[f] : f;
    const u64 base = splitmix64_mix(seed + (cid + 1) * kSmGamma);
    u64x2* dst = (u64x2*)(data + file_off[f]);
    const u64 n16 = (size + 15) / 16;
    for (u64 u = (u64)blockIdx.y * blockDim.x + threadIdx.x; u < n16;
         u += (u64)gridDim.y * blockDim.x) {
        u64x2 v;
        v.x = splitmix64_mix(base + (2 * (u + unit0) + 1) * kSmGamma);
        v.y = splitmix64_mix(base + (2 * (u + unit0) + 2) * kSmGamma);
        dst[u] = v;
    }
}

void launch_synth_fill(u8* d_data, const u64* d_file_off, const u64* d_file_size,
                       const u64* d_content_id, u64 n_files, u64 seed, u64 unit0, hipStream_t s) {
    if (n_files == 0) return;
    // y-dimension: enough pieces that a handful of huge files still fill the chip
    u32 gy = n_files >= 4096 ? 1 : (u32)(4096 / n_files);
    if (gy > 1024) gy = 1024;
    hipLaunchKernelGGL(synth_fill_kernel, dim3((u32)n_files, gy), dim3(256), 0, s, d_data,
                       d_file_off, d_file_size, d_content_id, n_files, seed, unit0);
}

// ---- exclusive scan of per-file chunk counts ------------------------------------
constexpr int kScanBlock = 256;
constexpr int kScanPer   = 8;                      // elements per thread
constexpr int kScanTile  = kScanBlock * kScanPer;  // 2048 per block

__device__ __forceinline__ u64 block_exclusive_scan(u64 v, u64* total, u64* lds /*>=8*/) {
    // wave inclusive scan with DPP-free shuffles, then across the 4 waves through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u64 x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 lo = __shfl_up((u32)x, d), hi = __shfl_up((u32)(x >> 32), d);
        const u64 y = ((u64)hi << 32) | lo;
        if (lane >= d) x += y;
    }
    if (lane == 63) lds[wave] = x;
    __syncthreads();
    u64 wave_off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kScanBlock / 64; ++w) {
        const u64 s = lds[w];
        if (w < wave) wave_off += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return wave_off + x - v;
}

__global__ __launch_bounds__(kScanBlock)
void scan_block_sums_kernel(const u32* __restrict__ counts, u64 n, u64* __restrict__ block_sums) {
    __shared__ u64 lds[8];
    __builtin_amdgcn_s_setprio(3);     // tiny kernels between the two long ones: do not queue behind them
    const u64 base = (u64)blockIdx.x * kScanTile + (u64)threadIdx.x * kScanPer;
    u64 s = 0;
#pragma unroll
    for (int k = 0; k < kScanPer; ++k) if (base + k < n) s += counts[base + k];
    u64 tot;
    (void)block_exclusive_scan(s, &tot, lds);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kScanBlock)
void scan_block_offsets_kernel(u64* __restrict__ block_sums, u64 n_blocks, u64* __restrict__ total) {
    // single block: exclusive scan of block_sums in place
    __shared__ u64 lds[8];
    u64 carry = 0;
    for (u64 b0 = 0; b0 < n_blocks; b0 += kScanBlock) {
        const u64 i = b0 + threadIdx.x;
        const u64 v = i < n_blocks ? block_sums[i] : 0;
        u64 tot;
        const u64 ex = block_exclusive_scan(v, &tot, lds);
        if (i < n_blocks) block_sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(kScanBlock)
void scan_final_kernel(const u32* __restrict__ counts, u64 n, const u64* __restrict__ block_off,
                       u64* __restrict__ first) {
    __shared__ u64 lds[8];
    const u64 base = (u64)blockIdx.x * kScanTile + (u64)threadIdx.x * kScanPer;
    u32 c[kScanPer];
    u64 s = 0;
#pragma unroll
    for (int k = 0; k < kScanPer; ++k) { c[k] = base + k < n ? counts[base + k] : 0; s += c[k]; }
    u64 tot;
    u64 ex = block_exclusive_scan(s, &tot, lds) + block_off[blockIdx.x];
#pragma unroll
    for (int k = 0; k < kScanPer; ++k) {
        if (base + k < n) first[base + k] = ex;
        ex += c[k];
    }
}

// offsets + final in one launch: every block sums the block totals before it (a few hundred
// L2-resident loads) instead of waiting for a single-block scan kernel in between
__global__ __launch_bounds__(kScanBlock)
void scan_final_fused_kernel(const u32* __restrict__ counts, u64 n, const u64* __restrict__ block_sums,
                             u64 n_blocks, u64* __restrict__ first, u64* __restrict__ total) {
    __shared__ u64 lds[8];
    __shared__ u64 s_before, s_all;
    __builtin_amdgcn_s_setprio(3);
    u64 before = 0, all = 0;
    for (u64 i = threadIdx.x; i < n_blocks; i += kScanBlock) {
        const u64 v = block_sums[i];
        all += v;
        if (i < blockIdx.x) before += v;
    }
    u64 t0, t1;
    (void)block_exclusive_scan(before, &t0, lds);
    (void)block_exclusive_scan(all, &t1, lds);
    if (threadIdx.x == 0) { s_before = t0; s_all = t1; }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) *total = s_all;
    const u64 base = (u64)blockIdx.x * kScanTile + (u64)threadIdx.x * kScanPer;
    u32 c[kScanPer];
    u64 sum = 0;
#pragma unroll
    for (int k = 0; k < kScanPer; ++k) { c[k] = base + k < n ? counts[base + k] : 0; sum += c[k]; }
    u64 tot;
    u64 ex = block_exclusive_scan(sum, &tot, lds) + s_before;
#pragma unroll
    for (int k = 0; k < kScanPer; ++k) {
        if (base + k < n) first[base + k] = ex;
        ex += c[k];
    }
}

u64 scan_scratch_elems(u64 n) { return (n + kScanTile - 1) / kScanTile + 1; }

void launch_scan_counts(const u32* d_counts, u64* d_first, u64* d_total, u64 n, u64* d_scratch,
                        hipStream_t s) {
    if (n == 0) { (void)hipMemsetAsync(d_total, 0, sizeof(u64), s); return; }
    const u64 nb = (n + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3((u32)nb), dim3(kScanBlock), 0, s, d_counts, n,
                       d_scratch);
    if (nb <= 4096) {                                     // two launches
        hipLaunchKernelGGL(scan_final_fused_kernel, dim3((u32)nb), dim3(kScanBlock), 0, s, d_counts, n,
                           d_scratch, nb, d_first, d_total);
        return;
    }
    hipLaunchKernelGGL(scan_block_offsets_kernel, dim3(1), dim3(kScanBlock), 0, s, d_scratch, nb,
                       d_total);
    hipLaunchKernelGGL(scan_final_kernel, dim3((u32)nb), dim3(kScanBlock), 0, s, d_counts, n,
                       d_scratch, d_first);
}

// ---- chunk-table compaction + length histogram ----------------------------------
__device__ __forceinline__ u32 sha_blocks_of(u64 len) {
    // 64-byte compressions SHA-256 needs for len bytes (data + 0x80 + 8-byte length)
    return (u32)(len >> 6) + 1u + ((len & 63) > 55 ? 1u : 0u);
}
__device__ __forceinline__ u32 bin_of(u64 len, u32 n_bins, u32 bin_shift) {
    const u32 b = sha_blocks_of(len) >> bin_shift;
    return b < n_bins ? b : n_bins - 1;
}

constexpr int kMaxBins = 1024;          // LDS-private histogram size (bins are block counts >> shift)
// End offset (file-relative) of chunk k of segment s.  Small-file segments hold one list; a group
// segment's final list is prefix[0, pcnt) ++ spec[sidx, spec_n) (gear_cdc.hip "large files").
__device__ __forceinline__ u64 seg_chunk_end(const u32* __restrict__ ends, u64 k, u64 seg_start,
                                             const GroupRec* r, u32 region) {
    if (!r) return ends[k];
    const u32 rel = k < r->pcnt ? ends[region + k] : ends[r->sidx + (k - r->pcnt)];
    return seg_start + rel;
}

// One thread per chunk row: row g belongs to the segment s with seg_first[s] <= g < seg_first[s+1]
// (binary search over the scanned counts -- a handful of L2-resident probes), so a batch of
// four 4 GiB files is compacted as fast as one of 100 000 small ones.  Length bins are counted
// in LDS and flushed once per workgroup (no hot global atomics).
__global__ __launch_bounds__(256)
void compact_chunks_kernel(const u64* __restrict__ file_off, const u64* __restrict__ file_seg0,
                           const u32* __restrict__ seg_file, const u64* __restrict__ seg_slot,
                           const u32* __restrict__ ends32, const u64* __restrict__ seg_first,
                           const u32* __restrict__ seg_group, const GroupRec* __restrict__ recs,
                           u32 region, u64 n_segs, u64 n_max, const u64* __restrict__ n_ptr,
                           u64* __restrict__ chunk_off, u64* __restrict__ chunk_len,
                           u32* __restrict__ chunk_file, u64* __restrict__ chunk_start,
                           u32* __restrict__ hist, u32 n_bins, u32 bin_shift) {
    __shared__ u32 lh[kMaxBins];
    __builtin_amdgcn_s_setprio(3);
    for (u32 i = threadIdx.x; i < n_bins; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    const u64 n = n_ptr ? *n_ptr : n_max;
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (u64)gridDim.x * blockDim.x) {
        u64 lo = 0, hi = n_segs;                             // last s with seg_first[s] <= g
        while (hi - lo > 1) {
            const u64 mid = (lo + hi) >> 1;
            if (seg_first[mid] <= g) lo = mid; else hi = mid;
        }
        const u64 s = lo;                                    // (empty segments never win: the search
        const u64 k = g - seg_first[s];                      //  keeps the LAST index with first <= g)
        const u32 f = seg_file[s];
        const u32* ends = ends32 + seg_slot[s];
        const GroupRec* r = nullptr;
        u64 seg_start = 0, entry = 0;
        if (seg_group) {
            const u32 gr = seg_group[s];
            if (gr != 0xFFFFFFFFu) {
                r = recs + gr;
                seg_start = (s - file_seg0[f]) * kGroupBytes;
                entry = r->entry;
            }
        }
        const u64 start = k ? seg_chunk_end(ends, k - 1, seg_start, r, region) : entry;
        const u64 len = seg_chunk_end(ends, k, seg_start, r, region) - start;
        chunk_off[g] = file_off[f] + start;
        chunk_len[g] = len;
        chunk_file[g] = f;
        chunk_start[g] = start;
        atomicAdd(&lh[bin_of(len, n_bins, bin_shift)], 1u);
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < n_bins; i += blockDim.x)
        if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// first chunk row / row count of every file, from the scanned segment counts; when no file of the
// batch needs a root reduction pass (all <= 64 chunks) also the root pass's item list: file f's
// string is its digest run (what root_init + root_final_items produce otherwise)
__global__ __launch_bounds__(256)
void file_rows_kernel(const u64* __restrict__ file_seg0, const u64* __restrict__ seg_first,
                      u64 n_files, u64 n_segs, const u64* __restrict__ total,
                      u64* __restrict__ first, u32* __restrict__ n_chunks, const u8* __restrict__ digests,
                      u64* __restrict__ item_off, u64* __restrict__ item_len) {
    __builtin_amdgcn_s_setprio(3);
    const u64 f = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_files) return;
    const u64 s0 = file_seg0[f], s1 = file_seg0[f + 1];
    const u64 a = s0 < n_segs ? seg_first[s0] : *total;
    const u64 b = s1 < n_segs ? seg_first[s1] : *total;
    first[f] = a;
    n_chunks[f] = (u32)(b - a);
    if (item_off) {
        item_off[f] = (u64)(uintptr_t)digests + a * 32;
        item_len[f] = (b - a) * 32;
    }
}

// ---- result rows: the chunk table as the ABI's mi_chunk_result rows (64 B each), packed on the
// device so that the host needs ONE copy instead of five column copies and a repacking loop
__global__ __launch_bounds__(256)
void pack_chunk_rows_kernel(u64 n, const u32* __restrict__ chunk_file, const u64* __restrict__ chunk_start,
                            const u64* __restrict__ chunk_len, const i64* __restrict__ dup_of,
                            const u8* __restrict__ digests, const u64* __restrict__ file_base,
                            u32x4* __restrict__ rows) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 f = chunk_file[i];
    const u64 off = chunk_start[i] + (file_base ? file_base[f] : 0ull);
    const i64 dup = dup_of[i];
    const u32x4* dg = (const u32x4*)(digests + 32 * i);
    u32x4 q0, q1;
    q0.x = f; q0.y = 0; q0.z = (u32)off; q0.w = (u32)(off >> 32);
    q1.x = (u32)chunk_len[i]; q1.y = 0; q1.z = (u32)(u64)dup; q1.w = (u32)((u64)dup >> 32);
    rows[4 * i] = q0;
    rows[4 * i + 1] = q1;
    rows[4 * i + 2] = dg[0];
    rows[4 * i + 3] = dg[1];
}

// ... and the file table as mi_file_result rows (96 B each): size, first chunk, chunk count, CRC, chunk root, whole-file
// digest; the user tag (host knowledge) is left zero for the host to fill in
__global__ __launch_bounds__(256)
void pack_file_rows_kernel(u64 n, const u64* __restrict__ file_size, const u64* __restrict__ first,
                           const u32* __restrict__ n_chunks, const u32* __restrict__ crc, const u8* __restrict__ roots,
                           const u8* __restrict__ file_sha, u32x4* __restrict__ rows) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 sz = file_size[i], fc = first[i];
    u32x4 q0, q1, z;
    z.x = z.y = z.z = z.w = 0;
    q0.x = 0; q0.y = 0; q0.z = (u32)sz; q0.w = (u32)(sz >> 32);
    q1.x = (u32)fc; q1.y = (u32)(fc >> 32); q1.z = n_chunks[i]; q1.w = crc ? crc[i] : 0u;
    const u32x4* rt = (const u32x4*)(roots + 32 * i);
    rows[6 * i] = q0;
    rows[6 * i + 1] = q1;
    rows[6 * i + 2] = rt[0];
    rows[6 * i + 3] = rt[1];
    if (file_sha) {
        const u32x4* fs = (const u32x4*)(file_sha + 32 * i);
        rows[6 * i + 4] = fs[0];
        rows[6 * i + 5] = fs[1];
    } else {
        rows[6 * i + 4] = z;
        rows[6 * i + 5] = z;
    }
}

void launch_pack_file_rows(u64 n, const u64* d_file_size, const u64* d_first, const u32* d_n_chunks, const u32* d_crc,
                           const u8* d_roots, const u8* d_file_sha, void* d_rows, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(pack_file_rows_kernel, dim3((u32)((n + 255) / 256)), dim3(256), 0, s, n, d_file_size, d_first,
                       d_n_chunks, d_crc, d_roots, d_file_sha, (u32x4*)d_rows);
}

void launch_pack_chunk_rows(u64 n, const u32* d_chunk_file, const u64* d_chunk_start, const u64* d_chunk_len,
                            const i64* d_dup_of, const u8* d_digests, const u64* d_file_base, void* d_rows,
                            hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(pack_chunk_rows_kernel, dim3((u32)((n + 255) / 256)), dim3(256), 0, s, n, d_chunk_file,
                       d_chunk_start, d_chunk_len, d_dup_of, d_digests, d_file_base, (u32x4*)d_rows);
}

void launch_compact_chunks(const u64* d_file_off, const u64* d_file_seg0, const u32* d_seg_file,
                           const u64* d_seg_slot, const u32* d_ends32, const u64* d_seg_first,
                           const u32* d_seg_group, const void* d_group_recs, u32 region,
                           u64 n_files, u64 n_segs, u64 n_max, const u64* d_n, u64* d_chunk_off,
                           u64* d_chunk_len, u32* d_chunk_file, u64* d_chunk_start, u64* d_first,
                           u32* d_n_chunks, u32* d_hist, u32 n_bins, u32 bin_shift, const u8* d_digests,
                           u64* d_item_off, u64* d_item_len, hipStream_t s) {
    if (n_files == 0) return;                             // d_hist: zeroed by the caller
    hipLaunchKernelGGL(file_rows_kernel, dim3((u32)((n_files + 255) / 256)), dim3(256), 0, s, d_file_seg0,
                       d_seg_first, n_files, n_segs, d_n, d_first, d_n_chunks, d_digests, d_item_off,
                       d_item_len);
    u64 want = (n_max + 255) / 256;
    const u32 grid = (u32)(want < 2048 ? (want ? want : 1) : 2048);
    hipLaunchKernelGGL(compact_chunks_kernel, dim3(grid), dim3(256), 0, s, d_file_off, d_file_seg0,
                       d_seg_file, d_seg_slot, d_ends32, d_seg_first, d_seg_group,
                       (const GroupRec*)d_group_recs, region, n_segs, n_max, d_n, d_chunk_off,
                       d_chunk_len, d_chunk_file, d_chunk_start, d_hist, n_bins, bin_shift);
}

// ---- longest-first processing order (counting sort by block-count bin) -----------
// Two-level scatter: a workgroup ranks its kScatterPer*256 items per bin in LDS, reserves one
// range per non-empty bin with a single global atomic, then every item writes its queue
// descriptor at range start + local rank.  The range starts are the descending exclusive scan of
// the length histogram; every workgroup recomputes it in LDS (<= 1024 bins) instead of waiting
// for a single-block kernel, `cursor` (zeroed by the caller) only counts what has been handed out.
constexpr int kScatterPer = 8;
__global__ __launch_bounds__(256)
void bin_scatter_kernel(const u64* __restrict__ off, const u64* __restrict__ len, u32 n_max,
                        const u64* __restrict__ n_ptr, const u32* __restrict__ hist,
                        u32* __restrict__ cursor, u32 n_bins, u32 bin_shift,
                        u64* __restrict__ s_off, u64* __restrict__ s_len, u32* __restrict__ s_id) {
    __shared__ u32 cnt[kMaxBins];
    __shared__ u32 start[kMaxBins];
    __shared__ u64 lds[8];
    __builtin_amdgcn_s_setprio(3);
    const u32 n = n_ptr ? (u32)*n_ptr : n_max;
    const u32 base = blockIdx.x * (256 * kScatterPer);
    if (base >= n) return;
    {   // start[bin] = number of items in bins > bin
        u64 carry = 0;
        for (u32 b0 = 0; b0 < n_bins; b0 += 256) {
            const u32 i = b0 + threadIdx.x;                     // position from the top
            const u32 bin = n_bins - 1 - i;
            const u64 v = i < n_bins ? hist[bin] : 0;
            u64 tot;
            const u64 ex = block_exclusive_scan(v, &tot, lds);
            if (i < n_bins) start[bin] = (u32)(carry + ex);
            carry += tot;
        }
    }
    for (u32 i = threadIdx.x; i < n_bins; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    u32 bin[kScatterPer], rank[kScatterPer];
    u64 l[kScatterPer];
#pragma unroll
    for (int k = 0; k < kScatterPer; ++k) {
        const u32 g = base + k * 256 + threadIdx.x;
        if (g < n) {
            l[k] = len[g];
            bin[k] = bin_of(l[k], n_bins, bin_shift);
            rank[k] = atomicAdd(&cnt[bin[k]], 1u);
        }
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < n_bins; i += blockDim.x) {
        const u32 c = cnt[i];
        if (c) cnt[i] = start[i] + atomicAdd(&cursor[i], c);   // now the range start of bin i
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kScatterPer; ++k) {
        const u32 g = base + k * 256 + threadIdx.x;
        if (g < n) {
            const u32 pos = cnt[bin[k]] + rank[k];          // queue position, longest first
            s_off[pos] = off[g];
            s_len[pos] = l[k];
            s_id[pos] = g;
        }
    }
}

void launch_bin_order(const u64* d_off, const u64* d_len, u32 n, const u64* d_n, u32* d_hist,
                      u32* d_cursor, u32 n_bins, u32 bin_shift, u64* d_s_off, u64* d_s_len,
                      u32* d_s_id, hipStream_t s) {
    if (n == 0) return;                                   // d_cursor: zeroed by the caller
    const u32 per = 256 * kScatterPer;
    hipLaunchKernelGGL(bin_scatter_kernel, dim3((n + per - 1) / per), dim3(256), 0, s, d_off, d_len,
                       n, d_n, d_hist, d_cursor, n_bins, bin_shift, d_s_off, d_s_len, d_s_id);
}

// ---- per-file chunk roots -----------------------------------------------------------
// chunk_root(f): SHA-256 over the file's concatenated chunk digests when it has <= 64 chunks;
// beyond that a fan-out-64 tree: REDUCTION passes hash runs of 64 child digests
// (2 KiB strings, 33 compressions) into node digests until <= 64 nodes are left, the FINAL pass
// hashes those.  (Round 1 used a fan-out of 1024: every pass was then as long as ONE 32 KiB
// string -- 1.5-2.5 ms of a single lane -- whatever the batch held; 3-5 ms per batch with files
// above 8 MiB.)  A file's current node list is (cur_addr, cur_cnt): absolute device address +
// digest count; a pass reads cur_* and writes next_* (ping-pong).
constexpr u32 kRootFanout = kChunkRootFanout;

__global__ __launch_bounds__(256)
void root_init_kernel(const u8* __restrict__ digests, const u64* __restrict__ first,
                      const u32* __restrict__ n_chunks, u64 n_files, u64* __restrict__ cur_addr,
                      u32* __restrict__ cur_cnt) {
    const u64 f = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_files) return;
    cur_addr[f] = (u64)(uintptr_t)digests + first[f] * 32;
    cur_cnt[f] = n_chunks[f];
}

// nodes a file contributes to this pass (0 = it already fits the final pass: carried over)
__global__ __launch_bounds__(256)
void root_level_counts_kernel(const u64* __restrict__ cur_addr, const u32* __restrict__ cur_cnt, u64 n_files,
                              u32* __restrict__ seg_cnt, u64* __restrict__ next_addr,
                              u32* __restrict__ next_cnt) {
    const u64 f = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_files) return;
    const u32 c = cur_cnt[f];
    seg_cnt[f] = c > kRootFanout ? (c + kRootFanout - 1) / kRootFanout : 0;
    next_addr[f] = cur_addr[f];
    next_cnt[f] = c;
}

// items of one reduction pass (absolute addresses): one thread per node, its file found by binary
// search over the scanned node counts (a 4 GiB file has 6 500 nodes in its first pass)
__global__ __launch_bounds__(256)
void root_level_items_kernel(const u64* __restrict__ seg_first, const u64* __restrict__ seg_total,
                             u64 n_files, u8* __restrict__ level_out, const u64* __restrict__ cur_addr,
                             const u32* __restrict__ cur_cnt, u64* __restrict__ next_addr,
                             u32* __restrict__ next_cnt, u64* __restrict__ item_off,
                             u64* __restrict__ item_len) {
    const u64 n = *seg_total;
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (u64)gridDim.x * blockDim.x) {
        u64 lo = 0, hi = n_files;                            // last f with seg_first[f] <= g
        while (hi - lo > 1) {
            const u64 mid = (lo + hi) >> 1;
            if (seg_first[mid] <= g) lo = mid; else hi = mid;
        }
        const u64 f = lo, s0 = seg_first[f];
        const u32 j = (u32)(g - s0), cnt = cur_cnt[f];
        const u32 left = cnt - j * kRootFanout;
        item_off[g] = cur_addr[f] + (u64)j * kRootFanout * 32;
        item_len[g] = (u64)(left < kRootFanout ? left : kRootFanout) * 32;
        if (j == 0) {                                        // the file moves on to its node list
            next_addr[f] = (u64)(uintptr_t)level_out + s0 * 32;
            next_cnt[f] = (cnt + kRootFanout - 1) / kRootFanout;
        }
    }
}

__global__ __launch_bounds__(256)
void root_final_items_kernel(const u64* __restrict__ cur_addr, const u32* __restrict__ cur_cnt,
                             u64 n_files, u64* __restrict__ off, u64* __restrict__ len) {
    const u64 f = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_files) return;
    off[f] = cur_addr[f];
    len[f] = (u64)cur_cnt[f] * 32;
}

void launch_root_init(const u8* d_digests, const u64* d_first, const u32* d_n_chunks, u64 n_files,
                      u64* d_cur_addr, u32* d_cur_cnt, hipStream_t s) {
    if (n_files == 0) return;
    hipLaunchKernelGGL(root_init_kernel, dim3((u32)((n_files + 255) / 256)), dim3(256), 0, s, d_digests,
                       d_first, d_n_chunks, n_files, d_cur_addr, d_cur_cnt);
}

void launch_root_level(u64 n_files, u64 n_nodes_ub, const u64* d_cur_addr, const u32* d_cur_cnt,
                       u64* d_next_addr, u32* d_next_cnt, u32* d_seg_cnt, u64* d_seg_first,
                       u64* d_seg_total, u64* d_scratch, u8* d_level_out, u64* d_item_off,
                       u64* d_item_len, hipStream_t s) {
    if (n_files == 0) return;
    const u32 grid = (u32)((n_files + 255) / 256);
    hipLaunchKernelGGL(root_level_counts_kernel, dim3(grid), dim3(256), 0, s, d_cur_addr, d_cur_cnt, n_files,
                       d_seg_cnt, d_next_addr, d_next_cnt);
    launch_scan_counts(d_seg_cnt, d_seg_first, d_seg_total, n_files, d_scratch, s);
    u64 want = (n_nodes_ub + 255) / 256;
    const u32 igrid = (u32)(want < 2048 ? (want ? want : 1) : 2048);
    hipLaunchKernelGGL(root_level_items_kernel, dim3(igrid), dim3(256), 0, s, d_seg_first, d_seg_total, n_files,
                       d_level_out, d_cur_addr, d_cur_cnt, d_next_addr, d_next_cnt, d_item_off, d_item_len);
}

void launch_root_final_items(const u64* d_cur_addr, const u32* d_cur_cnt, u64 n_files, u64* d_off,
                             u64* d_len, hipStream_t s) {
    if (n_files == 0) return;
    hipLaunchKernelGGL(root_final_items_kernel, dim3((u32)((n_files + 255) / 256)), dim3(256), 0, s,
                       d_cur_addr, d_cur_cnt, n_files, d_off, d_len);
}

// ---- duplicate marking over a digest set -----------------------------------------
// Open-addressing table keyed by the first 8 digest bytes (already uniform).  A slot
// holds a representative row (rep) and the minimum row index among equal digests
// (minid); equality is always checked on all 32 bytes, so a 64-bit key collision only
// costs an extra probe.  Output is deterministic: dup_of = smallest equal row, or -1.
// Minima are kept COMPLEMENTED (atomicMax of ~index, 0 = none) so that the whole table --
// rep | minid [| fmin], contiguous -- is cleared by ONE memset.
__device__ __forceinline__ bool digest_eq(const u8* a, const u8* b) {
    const u32x4 a0 = ((const u32x4*)a)[0], a1 = ((const u32x4*)a)[1];
    const u32x4 b0 = ((const u32x4*)b)[0], b1 = ((const u32x4*)b)[1];
    const u32x4 d0 = a0 ^ b0, d1 = a1 ^ b1;
    return (d0.x | d0.y | d0.z | d0.w | d1.x | d1.y | d1.z | d1.w) == 0;
}

__global__ __launch_bounds__(256)
void dedup_insert_kernel(const u8* __restrict__ digests, u64 n_max, const u64* __restrict__ n_ptr,
                         u32* __restrict__ rep, u32* __restrict__ minid,
                         u32* __restrict__ slot_of, u64 mask) {
    __builtin_amdgcn_s_setprio(3);
    const u64 n = n_ptr ? *n_ptr : n_max;
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u8* mine = digests + 32 * i;
    u64 slot = *(const u64*)mine & mask;
    for (;;) {
        u32 r = atomicCAS(&rep[slot], 0u, (u32)i + 1u);
        if (r == 0u || r == (u32)i + 1u || digest_eq(digests + 32ull * (r - 1u), mine)) break;
        slot = (slot + 1) & mask;
    }
    atomicMax(&minid[slot], ~(u32)i);
    slot_of[i] = (u32)slot;
}

__global__ __launch_bounds__(256)
void dedup_finish_kernel(u64 n_max, const u64* __restrict__ n_ptr, const u32* __restrict__ minid,
                         const u32* __restrict__ slot_of, i64* __restrict__ dup_of,
                         u64* __restrict__ n_unique) {
    // grid-stride; uniques are counted per wave (ballot) and per workgroup (LDS) so the one
    // global counter sees a few hundred atomics, not one per wave (same-address atomics
    // retire at ~90 per microsecond)
    __shared__ u32 wg_count;
    __builtin_amdgcn_s_setprio(3);
    if (threadIdx.x == 0) wg_count = 0;
    __syncthreads();
    const u64 n = n_ptr ? *n_ptr : n_max;
    u32 mine = 0;
    for (u64 i0 = (u64)blockIdx.x * blockDim.x; i0 < n; i0 += (u64)gridDim.x * blockDim.x) {
        const u64 i = i0 + threadIdx.x;
        bool first = false;
        if (i < n) {
            const u32 m = ~minid[slot_of[i]];
            first = (m == (u32)i);
            dup_of[i] = first ? -1 : (i64)m;
        }
        mine += (u32)__popcll(__ballot(first));
    }
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&wg_count, mine);
    __syncthreads();
    if (threadIdx.x == 0 && wg_count) atomicAdd((unsigned long long*)n_unique, (unsigned long long)wg_count);
}

void launch_dedup_mark(const u8* d_digests, u64 n, const u64* d_n, u32* d_table, u32* d_slot_of,
                       u64 cap_pow2, i64* d_dup_of, u64* d_n_unique, bool zero_count, hipStream_t s) {
    if (zero_count) (void)hipMemsetAsync(d_n_unique, 0, sizeof(u64), s);
    if (n == 0) return;
    u32* d_rep = d_table;
    u32* d_minid = d_table + cap_pow2;
    (void)hipMemsetAsync(d_table, 0, sizeof(u32) * 2 * cap_pow2, s);
    const u32 grid = (u32)((n + 255) / 256);
    hipLaunchKernelGGL(dedup_insert_kernel, dim3(grid), dim3(256), 0, s, d_digests, n, d_n, d_rep,
                       d_minid, d_slot_of, cap_pow2 - 1);
    hipLaunchKernelGGL(dedup_finish_kernel, dim3(grid < 1024 ? grid : 1024), dim3(256), 0, s, n, d_n,
                       d_minid, d_slot_of, d_dup_of, d_n_unique);
}

// ---- marking ONE rank's rows against the job-wide set --------------------------------
// After the digest all-gather every rank holds all n_total rows (rank-major), but it only has to
// answer for its OWN rows [own_first, own_first + own_n): dup_of = the smallest global index
// holding the same digest, or -1.  Rows of later ranks can never be that minimum, so:
//   1. the own rows build the table (same insert kernel as the in-batch marking; a tag kernel
//      then stores the first 8 digest bytes next to every occupied slot),
//   2. the rows of EARLIER ranks only probe it (read-only walk: tag, then the full 32 bytes on a
//      tag match) and atomicMin their global index into the slot they hit,
//   3. an own row takes the foreign minimum if there is one, else its in-rank first occurrence.
// Cost on rank r of R: own_n inserts + r * own_n probes into a table that stays cache-resident,
// instead of R * own_n inserts into a table R times the size.
__global__ __launch_bounds__(256)
void dedup_tag_kernel(const u8* __restrict__ own, const u32* __restrict__ rep, u64* __restrict__ tag,
                      u64 cap) {
    const u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= cap) return;
    const u32 r = rep[s];
    if (r) tag[s] = *(const u64*)(own + 32ull * (r - 1u));
}

__global__ __launch_bounds__(256)
void dedup_probe_kernel(const u8* __restrict__ all, u64 n_foreign, const u8* __restrict__ own,
                        const u32* __restrict__ rep, const u64* __restrict__ tag,
                        u32* __restrict__ fmin, u64 mask) {
    const u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_foreign) return;
    const u8* mine = all + 32 * g;
    const u64 key = *(const u64*)mine;
    u64 slot = key & mask;
    for (;;) {
        const u32 r = rep[slot];
        if (r == 0u) return;                                 // not among the own rows
        if (tag[slot] == key && digest_eq(own + 32ull * (r - 1u), mine)) {
            atomicMax(&fmin[slot], ~(u32)g);
            return;
        }
        slot = (slot + 1) & mask;
    }
}

__global__ __launch_bounds__(256)
void dedup_finish_range_kernel(u64 own_n, u64 own_first, const u32* __restrict__ minid,
                               const u32* __restrict__ fmin, const u32* __restrict__ slot_of,
                               i64* __restrict__ dup_own, u64* __restrict__ n_first) {
    __shared__ u32 wg_count;
    if (threadIdx.x == 0) wg_count = 0;
    __syncthreads();
    u32 mine = 0;
    for (u64 i0 = (u64)blockIdx.x * blockDim.x; i0 < own_n; i0 += (u64)gridDim.x * blockDim.x) {
        const u64 i = i0 + threadIdx.x;
        bool first = false;
        if (i < own_n) {
            const u32 s = slot_of[i];
            const u32 f = ~fmin[s], m = ~minid[s];
            if (f != 0xFFFFFFFFu) dup_own[i] = (i64)f;                   // an earlier rank has it
            else if (m != (u32)i) dup_own[i] = (i64)(own_first + m);     // earlier in this rank
            else { dup_own[i] = -1; first = true; }
        }
        mine += (u32)__popcll(__ballot(first));
    }
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&wg_count, mine);
    __syncthreads();
    if (threadIdx.x == 0 && wg_count) atomicAdd((unsigned long long*)n_first, (unsigned long long)wg_count);
}

void launch_dedup_mark_range(const u8* d_all, u64 own_first, u64 own_n, u32* d_table, u64* d_tag,
                             u32* d_slot_of, u64 cap_pow2, i64* d_dup_own, u64* d_n_first,
                             hipStream_t s) {
    (void)hipMemsetAsync(d_n_first, 0, sizeof(u64), s);
    if (own_n == 0) return;
    u32* d_rep = d_table;
    u32* d_minid = d_table + cap_pow2;
    u32* d_fmin = d_table + 2 * cap_pow2;
    (void)hipMemsetAsync(d_table, 0, sizeof(u32) * 3 * cap_pow2, s);
    const u8* own = d_all + 32 * own_first;
    const u32 grid = (u32)((own_n + 255) / 256);
    hipLaunchKernelGGL(dedup_insert_kernel, dim3(grid), dim3(256), 0, s, own, own_n, (const u64*)nullptr,
                       d_rep, d_minid, d_slot_of, cap_pow2 - 1);
    if (own_first) {
        hipLaunchKernelGGL(dedup_tag_kernel, dim3((u32)((cap_pow2 + 255) / 256)), dim3(256), 0, s, own,
                           d_rep, d_tag, cap_pow2);
        hipLaunchKernelGGL(dedup_probe_kernel, dim3((u32)((own_first + 255) / 256)), dim3(256), 0, s, d_all,
                           own_first, own, d_rep, d_tag, d_fmin, cap_pow2 - 1);
    }
    hipLaunchKernelGGL(dedup_finish_range_kernel, dim3(grid < 1024 ? grid : 1024), dim3(256), 0, s, own_n,
                       own_first, d_minid, d_fmin, d_slot_of, d_dup_own, d_n_first);
}

}  // namespace mi
