// gear_cdc.hip -- Gear rolling-hash content-defined chunking on gfx950.
//
// No reference counterpart (uber/makisu has no CDC -- SURVEY.md section 0); the
// spec is DESIGN.md "Gear-CDC spec" and the parity oracle is oracle/mi_oracle.c
// (mi_ref_cdc_two_phase / mi_ref_cdc_classic).  This kernel adds a content scan
// at the seam where the reference only compares tar headers
// (lib/snapshot/mem_fs.go:487-503 -> lib/tario/compare.go:104-120) and where it
// streams file bytes into the layer tar (lib/tario/write.go:28-52).
//
// Spec recap: h_i = sum_{k<64} G[b_{i-k}] << k  (mod 2^64) -- a pure function of
// the <=64 bytes ending at i, which is what makes marking embarrassingly parallel;
// position e = i+1 is a CANDIDATE cut iff the top mask_bits bits of h_i are zero;
// cuts are then SELECTED sequentially: skip candidates closer than min_size to
// the previous cut, force a cut at max_size, the file end always cuts.
//
// Mapping (v2): one WAVE marks one 64 KiB tile; every lane owns a contiguous 1 KiB
// run of it.
//   1. a lane streams its run straight from HBM in 128-byte pieces (8 x 16 B loads
//      = exactly one cache line, touched once), the next piece in flight while the
//      current one is hashed; it warms h over the 64 bytes before its run (6 % extra
//      reads, L2 hits: they are the previous lane's last line);
//   2. the Gear table lives in LDS, replicated kCopies times and interleaved so lanes
//      that differ in (lane % kCopies) never share a bank: the byte-indexed
//      ds_read_b64 lookups -- the dominant LDS traffic, 8 B per input byte -- stay
//      near conflict-free.  No file bytes are staged in LDS, so the LDS budget goes
//      to the table copies and the per-wave candidate bitmaps;
//   3. h rolls with one v_lshl_add_u64 per byte; the candidate test costs half a VALU
//      op per byte (v_min3_u32 over the high words of 16 consecutive hashes, a slow
//      path only when the minimum passes the mask); hits set bits in the wave's LDS
//      bitmap (one bit per byte of the tile);
//   4. cuts are selected per tile, carrying last_cut across tiles: normally from a sorted
//      64-entry candidate list the wave compacts out of its lanes' packed candidates
//      (__ballot + popcount prefix sums) -- one ballot per cut --, and from the bitmap with
//      wave-wide find-first-set (64 lanes x 64 bits per step, __ballot + ctz) when a tile
//      has too many candidates for the list; chunk ends go to the file's slot region in HBM.
// Small files (<= one tile): one wave per file, four files per workgroup, no
// workgroup barrier at all.  Large files: chained groups of four tiles -- persistent
// workgroups mark groups in parallel (even within ONE file) and pass the cut state from
// group to group (see gear_cdc_large_kernel).
// HBM traffic: every file byte read once (+6 % warm-up, mostly L2 hits), 8 B written
// per chunk.  Bound: HBM bandwidth / LDS lookup rate.
#include "mi_common.h"

namespace mi {

typedef u32 u32x4 __attribute__((ext_vector_type(4)));

constexpr int kCopies      = kGearTableCopies;          // Gear table replicas in LDS
constexpr int kWavesPerWG  = kGearWG / 64;              // 4
constexpr int kLaneRun     = kGearTile / 64;            // 1 KiB per lane
constexpr int kPiece       = 128;                       // bytes per load group (one cache line)
constexpr int kBitmapWords = kGearTile / 32;            // u32 words per wave bitmap (8 KiB)
constexpr int kTableBytes  = 256 * 8 * kCopies;
// LDS: table | one bitmap per wave | one 64-entry candidate list per wave | fast flags
constexpr int kLdsListOff  = kTableBytes + kWavesPerWG * kBitmapWords * 4;
constexpr int kLdsFastOff  = kLdsListOff + kWavesPerWG * 64 * 4;
constexpr int kGearLdsBytes = kLdsFastOff + 16;
constexpr u32 kNoCand = 0xFFFFFFFFu;

// ---- fast path for cut selection: a tile's candidates as ONE sorted 64-entry list ----------
// While marking, a lane also packs up to three candidates of its run into one VGPR (10-bit
// run offsets, count in bits 30..31).  If no lane overflowed and the tile has <= 64
// candidates (mask 13 bits: ~8 expected), the wave compacts them -- ballot + popcount prefix
// sums over the 2-bit counts -- into a sorted list in LDS; selection then needs one ballot per
// cut instead of a bitmap search.  Otherwise the bitmap (exact for any density) is used.
__device__ __forceinline__ void cand_push(u32& pk, bool& ovf, u32 o) {
    const u32 cnt = pk >> 30;
    if (cnt < 3) pk = ((cnt + 1) << 30) | (((pk & 0x3FFFFFFFu) << 10) | o);
    else ovf = true;
}

// returns true (wave-uniform) when `list` holds the tile's candidates (tile-relative byte
// indices, ascending, padded with kNoCand)
__device__ __forceinline__ bool cand_compact(u32 pk, bool ovf, u32 run0, int lane, u32* list) {
    const u32 cnt = pk >> 30;
    const u64 b0 = __ballot(cnt & 1u), b1 = __ballot(cnt & 2u);
    const u32 total = (u32)__popcll(b0) + 2u * (u32)__popcll(b1);
    if (__ballot(ovf) || total > 64u) return false;
    const u64 below = (1ull << lane) - 1ull;
    const u32 first = (u32)__popcll(b0 & below) + 2u * (u32)__popcll(b1 & below);
    list[lane] = kNoCand;
#pragma unroll
    for (u32 k = 0; k < 3; ++k)                          // field cnt-1-k holds my k-th (ascending) one
        if (k < cnt) list[first + k] = run0 + ((pk >> (10 * (cnt - 1 - k))) & 1023u);
    return true;
}

// first set bit of an LDS bitmap within [lo, hi] (bit indices, inclusive), -1 if none.
// Executed by one full wave; all lanes return the same value.
__device__ __forceinline__ int bitmap_find_first(const u64* bm, int lo, int hi, int lane) {
    const int w_lo = lo >> 6, w_hi = hi >> 6;
    for (int w0 = w_lo; w0 <= w_hi; w0 += 64) {
        const int w = w0 + lane;
        u64 v = 0;
        if (w <= w_hi) v = bm[w];
        if (w == w_lo) v &= ~0ull << (lo & 63);
        if (w == w_hi) v &= ~0ull >> (63 - (hi & 63));
        const u64 bal = __ballot(v != 0);
        if (bal) {
            const int src = __ffsll((unsigned long long)bal) - 1;
            const u32 vlo = __shfl((u32)v, src), vhi = __shfl((u32)(v >> 32), src);
            const u64 vv = ((u64)vhi << 32) | vlo;
            return (w0 + src) * 64 + (__ffsll((unsigned long long)vv) - 1);
        }
    }
    return -1;
}

// 16 more bytes into the rolling hash; hh[k] = high word after byte k.
// The table sits at the start of the workgroup's LDS (16 KiB-aligned), entry b of copy c at
// byte b * 64 + c * 8: the lookup address of byte k of a dword is
// ((w >> (8k - 6)) & 0x3FC0) | (c * 8) -- one full-rate shift and one v_bitop3_b32
// ((a & b) | c) instead of the half-rate v_bfe_u32 + v_lshl_add_u32 pair.
typedef __attribute__((address_space(3))) const u64 lds_cu64;
static_assert(kCopies == 8, "lookup address arithmetic assumes a 64-byte entry stride");
__device__ __forceinline__ void roll16(u64& h, const u32x4 v, u32 lane_tab, u32 (&hh)[16]) {
    const u32 wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const u32 w = wv[k >> 2];
        const int sh = 8 * (k & 3) - 6;
        const u32 t = sh < 0 ? w << 6 : w >> sh;
        const u32 addr = __builtin_amdgcn_bitop3_b32(t, 0x3FC0u, lane_tab, 0xEA);   // (t & 0x3FC0) | lane_tab
        h = (h << 1) + *(lds_cu64*)(size_t)addr;
        hh[k] = (u32)(h >> 32);
    }
}

// LDS byte address of this lane's table copy (entry 0); traps if the table is not where the
// OR-based lookup arithmetic needs it (start of LDS, 16 KiB-aligned).
__device__ __forceinline__ u32 lds_lane_table(const u64* table, int lane) {
    const u32 base = (u32)(size_t)(__attribute__((address_space(3))) const u64*)table;
    if (base & (u32)(kTableBytes - 1)) __builtin_trap();
    return base | ((u32)(lane % kCopies) * 8u);
}

__device__ __forceinline__ void load_piece(const u8* p, u32x4 (&d)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = *(const u32x4*)(p + 16 * i);
}

// One wave marks the candidates of tile [ts, ts+tlen) of a file into `bitmap`
// (bit p <-> cut end ts + p + 1).  fptr is 16-byte aligned, ts a multiple of kGearTile.
// pk/ovf: the lane's packed candidates for the fast selection path (cand_push / cand_compact).
__device__ __forceinline__ void mark_tile(const u8* __restrict__ fptr, u64 ts, u32 tlen,
                                          u32* bitmap, u32 tab, u32 thresh_m1, int lane,
                                          u32& pk, bool& ovf) {
    {   // clear the bitmap: 32 words per lane
        u32x4* bz = (u32x4*)bitmap;
        const u32x4 z = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < kBitmapWords / 4 / 64; ++i) bz[i * 64 + lane] = z;
    }
    pk = 0;
    ovf = false;
    const u32 run0 = (u32)lane * kLaneRun;               // tile-relative start of my run
    if (run0 >= tlen) return;
    const u8* p = fptr + ts + run0;
    const u32 run_len = tlen - run0 < (u32)kLaneRun ? tlen - run0 : (u32)kLaneRun;
    const int n_pieces = (int)((run_len + kPiece - 1) / kPiece);
    u32x4 cur[8], nxt[8];
    load_piece(p, cur);
    u64 h = 0;
    u32 hh[16];
    if (ts + run0 != 0) {                                // warm the window: 64 bytes before my run
        u32x4 wq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wq[i] = *(const u32x4*)(p - 64 + 16 * i);
#pragma unroll
        for (int i = 0; i < 4; ++i) roll16(h, wq[i], tab, hh);
    }
    for (int pc = 0; pc < n_pieces; ++pc) {
        if (pc + 1 < n_pieces) load_piece(p + (pc + 1) * kPiece, nxt);   // in flight while hashing
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            roll16(h, cur[g], tab, hh);
            u32 m = 0xFFFFFFFFu;
#pragma unroll
            for (int k = 0; k < 16; k += 2) m = min(m, min(hh[k], hh[k + 1]));   // v_min3_u32
            if (m <= thresh_m1) {                        // rare: a candidate among these 16 bytes
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const u32 pos = run0 + pc * kPiece + g * 16 + k;            // byte index in tile
                    if (hh[k] <= thresh_m1 && pos < tlen) {
                        atomicOr(&bitmap[pos >> 5], 1u << (pos & 31));
                        cand_push(pk, ovf, pos - run0);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
    }
}

// Wave-uniform cut selection over one marked tile; appends chunk ends, updates last/n_out.
// `list` != nullptr: the tile's sorted candidate list (cand_compact succeeded) -- one ballot per
// cut; otherwise the bitmap is searched.
__device__ __forceinline__ void select_tile(const u32* bitmap, const u32* list, u64 ts, u32 tlen,
                                            const CdcParams& p, u64& last, u32& n_out,
                                            u64* __restrict__ ends, int lane) {
    const u64 te = ts + tlen;                            // ends in this tile: (ts, te]
    const u32 cand = list ? list[lane] : kNoCand;        // lane i = i-th candidate (ascending)
    for (;;) {
        u64 lo = last + p.min_size;
        if (lo < ts + 1) lo = ts + 1;
        u64 hi = last + p.max_size;
        if (hi > te) hi = te;
        if (lo <= hi) {
            int b;
            if (list) {
                const u32 lo_rel = (u32)(lo - ts - 1), hi_rel = (u32)(hi - ts - 1);
                const u64 bal = __ballot(cand >= lo_rel && cand <= hi_rel);
                b = bal ? (int)__shfl(cand, __ffsll((unsigned long long)bal) - 1) : -1;
            } else {
                b = bitmap_find_first((const u64*)bitmap, (int)(lo - ts - 1), (int)(hi - ts - 1), lane);
            }
            if (b >= 0) {
                last = ts + (u64)b + 1;
                if (lane == 0) ends[n_out] = last;
                ++n_out;
                continue;
            }
        }
        if (last + p.max_size <= te) {                   // forced cut at max_size
            last += p.max_size;
            if (lane == 0) ends[n_out] = last;
            ++n_out;
            continue;
        }
        break;
    }
}

__device__ __forceinline__ void load_table(u64* table, const u64* __restrict__ gear_table, int tid) {
    // table[b * kCopies + c] = G[b] for every copy c
    for (int i = tid; i < 256 * kCopies; i += kGearWG) table[i] = gear_table[i / kCopies];
}

// ---- small files: one wave per file (size <= kGearTile) ----------------------------------
__global__ __launch_bounds__(kGearWG)
void gear_cdc_small_kernel(const u8* __restrict__ data, const u64* __restrict__ file_off,
                           const u64* __restrict__ file_size, const u64* __restrict__ slot_base,
                           u64* __restrict__ slot_ends, u32* __restrict__ n_chunks,
                           const u32* __restrict__ list, u32 n_list,
                           const u64* __restrict__ gear_table, CdcParams p) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    u64* table = (u64*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32* bitmap = (u32*)(smem + kTableBytes) + wave * kBitmapWords;
    u32* cand_list = (u32*)(smem + kLdsListOff) + wave * 64;
    load_table(table, gear_table, tid);
    __syncthreads();
    const u32 lane_tab = lds_lane_table(table, lane);
    const u32 li = blockIdx.x * kWavesPerWG + wave;
    if (li >= n_list) return;
    const u32 f = list[li];
    const u64 size = file_size[f];
    u64* ends = slot_ends + slot_base[f];
    u64 last = 0;
    u32 n_out = 0;
    if (size) {
        u32 pk;
        bool ovf;
        mark_tile(data + file_off[f], 0, (u32)size, bitmap, lane_tab, p.thresh_m1, lane,
                  pk, ovf);
        const bool fast = cand_compact(pk, ovf, (u32)lane * kLaneRun, lane, cand_list);
        // the wave's own LDS writes are ordered for the wave itself after the waitcnt the
        // compiler inserts; no other wave touches this bitmap / list
        __builtin_amdgcn_wave_barrier();
        select_tile(bitmap, fast ? cand_list : nullptr, 0, (u32)size, p, last, n_out, ends, lane);
    }
    if (lane == 0) {
        if (size > last) { ends[n_out] = size; ++n_out; }   // the file end always cuts
        n_chunks[f] = n_out;
    }
}

// ---- large files: chained groups ------------------------------------------------------------
// A large file is cut into GROUPS of kWavesPerWG tiles (256 KiB).  Persistent workgroups draw
// group tickets from one atomic counter in file-major order; a workgroup marks its group's
// tiles independently of everybody else (the expensive part), then wave 0 waits for the CUT
// STATE (last cut, chunks emitted) handed over by the previous group of the same file,
// selects this group's cuts and hands the state on.  So even ONE huge file keeps the whole
// chip busy; only the cheap selection is serial.
// Hand-over (MI355X_MICROARCH.md "R2: the data IS the flag"): two 8-byte granules per group,
// {tag:32 | n_out:32} and {tag:16 | last:48}, each written by ONE relaxed agent-scope store and
// polled with relaxed agent-scope loads -- no fences; granules are zeroed before every launch.
// Deadlock-free: tickets are drawn in order by running workgroups, so the predecessor of any
// waiting group already holds a ticket and never waits on its successors.
struct GroupToken { u64 a, b; };
typedef __attribute__((address_space(1))) unsigned long long gu64;

__global__ __launch_bounds__(kGearWG)
void gear_cdc_large_kernel(const u8* __restrict__ data, const u64* __restrict__ file_off,
                           const u64* __restrict__ file_size, const u64* __restrict__ slot_base,
                           u64* __restrict__ slot_ends, u32* __restrict__ n_chunks,
                           const u32* __restrict__ group_file, const u32* __restrict__ group_index,
                           const u32* __restrict__ group_prev, u32 n_groups,
                           u32* __restrict__ ticket_counter,
                           GroupToken* __restrict__ tokens, const u64* __restrict__ gear_table,
                           CdcParams p) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    u64* table = (u64*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32* bitmaps = (u32*)(smem + kTableBytes);
    u32* cand_lists = (u32*)(smem + kLdsListOff);
    volatile u32* fast_flags = (volatile u32*)(smem + kLdsFastOff);
    // (the ticket word lives in the dynamic region: a static __shared__ would shift its base)
    volatile u32* s_ticket = (volatile u32*)(smem + kGearLdsBytes);
    load_table(table, gear_table, tid);
    __syncthreads();
    const u32 lane_tab = lds_lane_table(table, lane);
    constexpr u64 kGroupBytes = (u64)kGearTile * kWavesPerWG;
    for (;;) {
        if (tid == 0) *s_ticket = atomicAdd(ticket_counter, 1u);
        __syncthreads();
        const u32 g = *s_ticket;
        if (g >= n_groups) break;
        const u32 f = group_file[g], gi = group_index[g];
        const u64 size = file_size[f];
        const u8* fptr = data + file_off[f];
        const u64 g0 = (u64)gi * kGroupBytes;
        const u64 ts = g0 + (u64)wave * kGearTile;
        if (ts < size) {
            const u32 tlen = (u32)((size - ts < (u64)kGearTile) ? (size - ts) : (u64)kGearTile);
            u32 pk;
            bool ovf;
            mark_tile(fptr, ts, tlen, bitmaps + wave * kBitmapWords, lane_tab,
                      p.thresh_m1, lane, pk, ovf);
            const bool fast = cand_compact(pk, ovf, (u32)lane * kLaneRun, lane, cand_lists + wave * 64);
            if (lane == 0) fast_flags[wave] = fast ? 1u : 0u;
        }
        __syncthreads();
        if (wave == 0) {
            u64 last = 0;
            u32 n_out = 0;
            if (gi > 0) {                                 // cut state from the file's previous group
                const u32 prev = group_prev[g];           // its ticket (always < g)
                gu64* ta = (gu64*)&tokens[prev].a;
                gu64* tb = (gu64*)&tokens[prev].b;
                u64 a = 0, b = 0;
                if (lane == 0) {
                    // bounded: a broken chain must surface as an error, not as a hung GPU
                    for (u32 spins = 0;; ++spins) {
                        a = __hip_atomic_load(ta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        b = __hip_atomic_load(tb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((a >> 32) == 1u && (b >> 48) == 1u) break;
                        if (spins > (1u << 24)) { atomicExch(ticket_counter + 1, 1u); a = b = 0; break; }
                        __builtin_amdgcn_s_sleep(4);
                    }
                }
                const u32 alo = __shfl((u32)a, 0);
                const u32 blo = __shfl((u32)b, 0), bhi = __shfl((u32)(b >> 32), 0);
                n_out = alo;
                last = (((u64)bhi << 32) | blo) & 0xFFFFFFFFFFFFull;
            }
            u64* ends = slot_ends + slot_base[f];
            for (int t = 0; t < kWavesPerWG; ++t) {
                const u64 tts = g0 + (u64)t * kGearTile;
                if (tts >= size) break;
                const u32 tlen = (u32)((size - tts < (u64)kGearTile) ? (size - tts) : (u64)kGearTile);
                select_tile(bitmaps + t * kBitmapWords, fast_flags[t] ? cand_lists + t * 64 : nullptr,
                            tts, tlen, p, last, n_out, ends, lane);
            }
            if (lane == 0) {
                if (g0 + kGroupBytes >= size) {           // the file's last group
                    if (size > last) { ends[n_out] = size; ++n_out; }
                    n_chunks[f] = n_out;
                } else {
                    __hip_atomic_store((gu64*)&tokens[g].a, ((u64)1 << 32) | n_out, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store((gu64*)&tokens[g].b, ((u64)1 << 48) | last, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        __syncthreads();                                  // bitmaps and s_ticket are reused
    }
}

u64 gear_large_groups(u64 size) {
    const u64 gb = (u64)kGearTile * kWavesPerWG;
    return (size + gb - 1) / gb;
}

void launch_gear_cdc(const u8* d_data, const u64* d_file_off, const u64* d_file_size,
                     const u64* d_slot_base, u64* d_slot_ends, u32* d_n_chunks,
                     const u32* d_small_list, u32 n_small, const u32* d_group_file,
                     const u32* d_group_index, const u32* d_group_prev, u32 n_groups,
                     u32* d_ticket, void* d_tokens, const u64* d_gear_table, CdcParams p, int n_cu,
                     hipStream_t s) {
    (void)hipFuncSetAttribute((const void*)gear_cdc_small_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, kGearLdsBytes);
    (void)hipFuncSetAttribute((const void*)gear_cdc_large_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, kGearLdsBytes + 16);
    if (n_small)
        hipLaunchKernelGGL(gear_cdc_small_kernel, dim3((n_small + kWavesPerWG - 1) / kWavesPerWG),
                           dim3(kGearWG), kGearLdsBytes, s, d_data, d_file_off, d_file_size,
                           d_slot_base, d_slot_ends, d_n_chunks, d_small_list, n_small,
                           d_gear_table, p);
    if (n_groups) {
        (void)hipMemsetAsync(d_ticket, 0, 2 * sizeof(u32), s);   // [0] ticket counter, [1] chain error
        (void)hipMemsetAsync(d_tokens, 0, sizeof(GroupToken) * (size_t)n_groups, s);
        u32 grid = (u32)n_cu * 3;                         // 3 workgroups per CU fit (LDS, VGPRs)
        if (grid > n_groups) grid = n_groups;
        hipLaunchKernelGGL(gear_cdc_large_kernel, dim3(grid), dim3(kGearWG), kGearLdsBytes + 16, s,
                           d_data, d_file_off, d_file_size, d_slot_base, d_slot_ends, d_n_chunks,
                           d_group_file, d_group_index, d_group_prev, n_groups, d_ticket,
                           (GroupToken*)d_tokens, d_gear_table, p);
    }
}

}  // namespace mi
