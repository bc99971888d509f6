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
//   2. the Gear table lives in LDS, replicated and interleaved so lanes that differ in
//      (lane % copies) never share a bank: the byte-indexed ds_read_b64 lookups -- the
//      dominant LDS traffic, 8 B per input byte -- are conflict-free with the 32 copies of
//      the marking kernels (entry stride 256 B: the data byte is byte 1 of the address, one
//      v_perm_b32) and near it with the 8 copies of the kernels that also keep per-wave
//      candidate bitmaps.  No file bytes are staged in LDS;
//   3. h rolls with one v_lshl_add_u64 per byte; the candidate test costs half a VALU
//      op per byte (v_min3_u32 over the high words of 16 consecutive hashes, a slow
//      path only when the minimum passes the mask); hits set bits in the wave's LDS
//      bitmap (one bit per byte of the tile);
//   4. cuts are selected per tile, carrying last_cut across tiles: normally from a sorted
//      64-entry candidate list the wave compacts out of its lanes' packed candidates
//      (__ballot + popcount prefix sums) -- one ballot per cut --, and from the bitmap with
//      wave-wide find-first-set (64 lanes x 64 bits per step, __ballot + ctz) when a tile
//      has too many candidates for the list; chunk ends go to the segment's u32 list in HBM.
// Small files (<= one tile): one wave per file, eight files per workgroup, no
// workgroup barrier behind the table load.  Large files: GROUPS of four tiles (256 KiB) that are marked AND
// cut in parallel -- every group selects speculatively as if a cut fell on its first byte,
// a second pass re-selects from the previous group's speculative exit until it meets the
// speculative cut list again (Gear + min/max re-synchronises within a few chunks), and a
// per-file pass only walks the groups whose assumption failed (see "large files" below).
// HBM traffic: every file byte read once (+6 % warm-up, mostly L2 hits), 4 B written
// per chunk (+ 256 B of candidates per tile and a 40-byte record per group for large files).  Bound: HBM bandwidth / LDS lookup rate.
#include "mi_common.h"

namespace mi {

typedef u32 u32x4 __attribute__((ext_vector_type(4)));

constexpr int kCopies      = kGearTableCopies;          // Gear table replicas in LDS (kernels that keep bitmaps)
// 32 table copies: a 64 KiB table shared by a 512-thread workgroup, two workgroups per CU (16 waves per CU).  The variants that
// were measured against this one -- 16 copies with 256-thread workgroups, a VGPR budget for more waves per SIMD, 64- and 32-byte
// pieces, the marking through an LDS exchange of coalesced loads (mark_tile_coal), raised wave priority, a v_min3 tree, the
// warm-up line taken from the lane's own run (wrong cuts; an upper bound of what a free warm-up could save) -- live as a patch
// under tools/experiments/gear_cdc_experiments.patch with their numbers (profiles/r03_gear_ab.txt, r04_gear_ab.txt): the
// shipped source has no switch that changes a cut.
constexpr int kFastCopies  = 32;                        // ... in the bitmap-free marking kernels (see below)
constexpr int kFastWG      = 512;
constexpr int kFastWaves   = kFastWG / 64;

constexpr int kWavesPerWG  = kGearWG / 64;              // 4
constexpr int kLaneRun     = kGearTile / 64;            // 1 KiB per lane
constexpr int kPiece       = 128;                       // bytes per load group (128 = one cache line)
constexpr int kPieceUnits  = kPiece / 16;
constexpr int kBitmapWords = kGearTile / 32;            // u32 words per wave bitmap (8 KiB)
constexpr int kTableBytes  = 256 * 8 * kCopies;
// The bitmap-free kernels (round 3): a tile's candidates live ONLY in its lanes' packed registers (up to
// six per lane) and the 64-entry list; the 8 KiB-per-wave LDS bitmap -- two thirds of the old LDS budget,
// there for tiles with more than 64 candidates -- is gone from the fast path, and such tiles (none on
// random data at the default mask) are redone by the bitmap kernel.  The freed LDS holds SIXTEEN table
// copies (entry stride 128 B).  CDNA4 serves a ds_read_b64 in two groups of 32 lanes over 64 banks
// (bank = dword address mod 64): with 8 copies four lanes of a group share a copy and collide whenever
// their bytes agree mod 4 positions of the row; with 16 copies two lanes share one and collide when
// their bytes have equal parity -- SQ_LDS_BANK_CONFLICT falls from 62 % to 49 % of the LDS-array cycles
// and those cycles by a third (profiles/r03_sq_counters.txt).  THIRTY-TWO copies (entry stride 256 B, 64 KiB) are
// conflict-free -- and the data byte then IS byte 1 of the lookup address, so the address is ONE v_perm_b32
// instead of a shift and a v_bitop3_b32 (3.7 instead of 4.7 VALU/LDS instructions per byte).  Four 256-thread
// workgroups per CU cannot hold 64 KiB each; two 512-thread workgroups can, with the same 16 waves per CU (the
// marking kernels have no barrier behind the table load, a wave is on its own): C2 marking 1.38-1.42 -> 1.33-1.34 ms,
// step 5.97-6.00 -> 5.80-5.85 ms on one box (profiles/r03_gear_ab.txt).
constexpr int kFastTableBytes = 256 * 8 * kFastCopies;  // 64 KiB (32 KiB with 16 copies)
constexpr int kFastListOff = kFastTableBytes;                                     // table | lists
constexpr int kFastLdsBytes = kFastListOff + kFastWaves * 64 * 4;
// LDS: table | one bitmap per wave | one 64-entry candidate list per wave | fast flags
constexpr int kLdsListOff  = kTableBytes + kWavesPerWG * kBitmapWords * 4;
constexpr int kLdsFastOff  = kLdsListOff + kWavesPerWG * 64 * 4;
constexpr int kGearLdsBytes = kLdsFastOff + 16;
constexpr u32 kNoCand = 0xFFFFFFFFu;

// value of `v` in lane `src` for a WAVE-UNIFORM src: v_readlane_b32 (a few cycles) instead of the
// ds_bpermute round trip __shfl costs -- cut selection is one dependent chain of these
__device__ __forceinline__ u32 lane_value(u32 v, int src) { return (u32)__builtin_amdgcn_readlane((int)v, src); }

// ---- fast path for cut selection: a tile's candidates as ONE sorted 64-entry list ----------
// While marking, a lane also packs up to three candidates of its run into one VGPR (10-bit
// run offsets, count in bits 30..31).  If no lane overflowed and the tile has <= 64
// candidates (mask 13 bits: ~8 expected), the wave compacts them -- ballot + popcount prefix
// sums over the 2-bit counts -- into a sorted list in LDS; selection then needs one ballot per
// cut instead of a bitmap search.  Otherwise the bitmap (exact for any density) is used.
struct CandPack { u32 a, b; };                           // candidates 0..2 in a, 3..5 in b (10-bit run offsets, count in bits 30..31)
__device__ __forceinline__ void cand_push(CandPack& pk, bool& ovf, u32 o) {
    const u32 ca = pk.a >> 30;
    if (ca < 3) { pk.a = ((ca + 1) << 30) | (((pk.a & 0x3FFFFFFFu) << 10) | o); return; }
    const u32 cb = pk.b >> 30;
    if (cb < 3) pk.b = ((cb + 1) << 30) | (((pk.b & 0x3FFFFFFFu) << 10) | o);
    else ovf = true;
}

// returns true (wave-uniform) when `list` holds the tile's candidates (tile-relative byte
// indices, ascending, padded with kNoCand)
__device__ __forceinline__ bool cand_compact(CandPack pk, bool ovf, u32 run0, int lane, u32* list) {
    const u32 ca = pk.a >> 30, cb = pk.b >> 30;
    const u32 cnt = ca + cb;                             // 0..6
    const u64 b0 = __ballot(cnt & 1u), b1 = __ballot(cnt & 2u), b2 = __ballot(cnt & 4u);
    const u32 total = (u32)__popcll(b0) + 2u * (u32)__popcll(b1) + 4u * (u32)__popcll(b2);
    if (__ballot(ovf) || total > 64u) return false;
    const u64 below = (1ull << lane) - 1ull;
    const u32 first = (u32)__popcll(b0 & below) + 2u * (u32)__popcll(b1 & below) + 4u * (u32)__popcll(b2 & below);
    list[lane] = kNoCand;
#pragma unroll
    for (u32 k = 0; k < 3; ++k)                          // field c-1-k of a register holds its k-th (ascending) one
        if (k < ca) list[first + k] = run0 + ((pk.a >> (10 * (ca - 1 - k))) & 1023u);
    if (cb) {                                            // rare: a fourth..sixth candidate in one 1 KiB run
#pragma unroll
        for (u32 k = 0; k < 3; ++k)
            if (k < cb) list[first + 3 + k] = run0 + ((pk.b >> (10 * (cb - 1 - k))) & 1023u);
    }
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
            const u32 vlo = lane_value((u32)v, src), vhi = lane_value((u32)(v >> 32), src);
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
// kC copies: entry b of copy c at byte b * (8 kC) + c * 8 -- a 64-byte stride for 8 copies, 128 for 16
template <int kC>
__device__ __forceinline__ void roll16(u64& h, const u32x4 v, u32 lane_tab, u32 (&hh)[16]) {
    static_assert(kC == 8 || kC == 16 || kC == 32, "lookup address arithmetic: 64-, 128- or 256-byte entry stride");
    constexpr int kShift = kC == 8 ? 6 : 7;
    constexpr u32 kMask = 0xFFu << kShift;
    const u32 wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const u32 w = wv[k >> 2];
        u32 addr;
        if constexpr (kC == 32) {
            // entry stride 256 B: the data byte IS byte 1 of the address -- ONE v_perm_b32
            // {0, lane_tab.byte2 (table base / 64 KiB), w.byte k, lane_tab.byte0 (copy * 8)}
            addr = __builtin_amdgcn_perm(w, lane_tab, 0x0C020000u | ((4u + (u32)(k & 3)) << 8));
        } else {
            const int sh = 8 * (k & 3) - kShift;
            const u32 t = sh < 0 ? w << -sh : w >> sh;
            addr = __builtin_amdgcn_bitop3_b32(t, kMask, lane_tab, 0xEA);   // (t & mask) | lane_tab
        }
        h = (h << 1) + *(lds_cu64*)(size_t)addr;
        hh[k] = (u32)(h >> 32);
    }
}

// LDS byte address of this lane's table copy (entry 0); traps if the table is not where the
// OR-based lookup arithmetic needs it (start of LDS, aligned to its own size).
template <int kC>
__device__ __forceinline__ u32 lds_lane_table(const u64* table, int lane) {
    const u32 base = (u32)(size_t)(__attribute__((address_space(3))) const u64*)table;
    if (base & (u32)(256 * 8 * kC - 1)) __builtin_trap();
    return base | ((u32)(lane % kC) * 8u);
}

__device__ __forceinline__ void load_piece(const u8* p, u32x4 (&d)[kPieceUnits]) {
#pragma unroll
    for (int i = 0; i < kPieceUnits; ++i) d[i] = *(const u32x4*)(p + 16 * i);
}

// 16 more bytes of a lane's run (run-relative offset `base`): roll, test, record the candidates.
template <int kC, bool kBitmap>
__device__ __forceinline__ void hash16(u64& h, const u32x4 v, u32 tab, u32 thresh_m1, u32 run0, u32 base,
                                       u32* bitmap, CandPack& pk, bool& ovf) {
    u32 hh[16];
    roll16<kC>(h, v, tab, hh);
    // (a v_min3_u32 chain would be 8 ops instead of the 11 hipcc emits, but it is one dependent
    // chain: A/B on one box, 1.50 ms against 1.46 ms for the compiler's tree)
    u32 m = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < 16; k += 2) m = min(m, min(hh[k], hh[k + 1]));
    if (m <= thresh_m1) {                                // rare: a candidate among these 16 bytes
        // (positions at or past the file end are not filtered here: selection never looks
        // beyond the tile's last byte, and at most one lane hashes up to 127 slack bytes)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (hh[k] <= thresh_m1) {
                const u32 pos = run0 + base + (u32)k;    // byte index in tile
                if (kBitmap) atomicOr(&bitmap[pos >> 5], 1u << (pos & 31));
                cand_push(pk, ovf, base + (u32)k);
            }
        }
    }
}

// One wave marks the candidates of tile [ts, ts+tlen) of a file into `bitmap`
// (bit p <-> cut end ts + p + 1).  fptr is 16-byte aligned, ts a multiple of kGearTile.
// pk/ovf: the lane's packed candidates for the fast selection path (cand_push / cand_compact).
// kBitmap = false: no bitmap at all (bitmap may be nullptr); the candidates are in pk only.
template <int kC, bool kBitmap>
__device__ __forceinline__ void mark_tile(const u8* __restrict__ fptr, u64 ts, u32 tlen,
                                          u32* bitmap, u32 tab, u32 thresh_m1, int lane,
                                          CandPack& pk, bool& ovf) {
    if (kBitmap) {   // clear the bitmap: 32 words per lane
        u32x4* bz = (u32x4*)bitmap;
        const u32x4 z = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < kBitmapWords / 4 / 64; ++i) bz[i * 64 + lane] = z;
    }
    pk.a = pk.b = 0;
    ovf = false;
    const u32 run0 = (u32)lane * kLaneRun;               // tile-relative start of my run
    if (run0 >= tlen) return;
    const u8* p = fptr + ts + run0;
    const u32 run_len = tlen - run0 < (u32)kLaneRun ? tlen - run0 : (u32)kLaneRun;
    const int n_pieces = (int)((run_len + kPiece - 1) / kPiece);
    u32x4 cur[kPieceUnits];
    load_piece(p, cur);
    u64 h = 0;
    u32 hh[16];
    if (ts + run0 != 0) {                                // warm the window: 64 bytes before my run
        u32x4 wq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wq[i] = *(const u32x4*)(p - 64 + 16 * i);
#pragma unroll
        for (int i = 0; i < 4; ++i) roll16<kC>(h, wq[i], tab, hh);
    }
    u32x4 nxt[kPieceUnits];
    for (int pc = 0; pc < n_pieces; ++pc) {
        if (pc + 1 < n_pieces) load_piece(p + (pc + 1) * kPiece, nxt);   // in flight while hashing
#pragma unroll
        for (int g = 0; g < kPieceUnits; ++g)
            hash16<kC, kBitmap>(h, cur[g], tab, thresh_m1, run0, (u32)(pc * kPiece + g * 16), bitmap, pk, ovf);
#pragma unroll
        for (int i = 0; i < kPieceUnits; ++i) cur[i] = nxt[i];
    }
    // (Round 3 tried one piece of registers instead of two -- each 16-byte unit of the next piece requested
    // into the registers of the unit just hashed, 84 VGPRs, five or six waves per SIMD: 2.15-2.28 ms instead
    // of 1.34, profiles/r03_gear_ab.txt; a load per unit means a wait per unit.)
}

// Wave-uniform cut selection over one marked tile.  Every cut is handed to emit(cut) (wave-uniform
// call, absolute END offset inside the file); emit returns true to stop the selection at once
// (the validation pass stops where it meets the speculative cut list again).  Returns true when
// stopped.  `list` != nullptr: the tile's sorted candidate list (cand_compact / list_from_bitmap
// succeeded) -- one ballot per cut; otherwise the bitmap is searched.
template <typename Emit>
__device__ __forceinline__ bool select_tile(const u32* bitmap, const u32* list, u64 ts, u32 tlen,
                                            const CdcParams& p, u64& last, int lane, Emit&& emit) {
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
                b = bal ? (int)lane_value(cand, __ffsll((unsigned long long)bal) - 1) : -1;
            } else {
                b = bitmap_find_first((const u64*)bitmap, (int)(lo - ts - 1), (int)(hi - ts - 1), lane);
            }
            if (b >= 0) {
                last = ts + (u64)b + 1;
                if (emit(last)) return true;
                continue;
            }
        }
        if (last + p.max_size <= te) {                   // forced cut at max_size
            last += p.max_size;
            if (emit(last)) return true;
            continue;
        }
        break;
    }
    return false;
}

// Slow path of the list construction: a tile whose lanes overflowed their 3 packed candidates
// (about 6 in 10 000 tiles on random data) still has few candidates in total; rebuild the sorted
// list from the bitmap -- lane i owns words [32 i, 32 i + 32) = its own 1 KiB run.  False
// (wave-uniform) when the tile really holds more than 64 candidates.
__device__ __forceinline__ bool list_from_bitmap(const u32* bitmap, int lane, u32* list) {
    u32 cnt = 0;
#pragma unroll 4
    for (int w = 0; w < 32; ++w) cnt += (u32)__popc(bitmap[lane * 32 + w]);
    u32 incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 y = __shfl_up(incl, d);
        if (lane >= d) incl += y;
    }
    const u32 total = lane_value(incl, 63);
    if (total > 64u) return false;
    list[lane] = kNoCand;
    u32 pos = incl - cnt;
    if (cnt) {
        for (int w = 0; w < 32; ++w) {
            u32 bits = bitmap[lane * 32 + w];
            while (bits) {
                const int b = __ffs(bits) - 1;
                bits &= bits - 1;
                list[pos++] = (u32)lane * kLaneRun + (u32)w * 32u + (u32)b;
            }
        }
    }
    return true;
}

template <int kC, int kWG = kGearWG>
__device__ __forceinline__ void load_table(u64* table, const u64* __restrict__ gear_table, int tid) {
    // table[b * kC + c] = G[b] for every copy c
    for (int i = tid; i < 256 * kC; i += kWG) table[i] = gear_table[i / kC];
}

// ---- small files: one wave per file (size <= kGearTile) ----------------------------------
// A small file is one SEGMENT: its chunk ends go to ends32[seg_slot[s] ..] (u32, file-relative),
// its chunk count to seg_n[s].
// gear_cdc_small_fast_kernel: the bitmap-free form (16 table copies, candidates in registers + the
// 64-entry list).  A file with more than 64 candidates (or a lane with more than six) is appended to
// dense_list and left to gear_cdc_small_kernel, the round-1/2 form with its exact per-wave bitmap,
// which runs over that list afterwards (n_list_dev: the list's length, known on the device only).
__global__ __launch_bounds__(kFastWG)
void gear_cdc_small_fast_kernel(const u8* __restrict__ data, const u64* __restrict__ file_off,
                                const u64* __restrict__ file_size, const u32* __restrict__ seg_file,
                                const u64* __restrict__ seg_slot, u32* __restrict__ ends32,
                                u32* __restrict__ seg_n, const u32* __restrict__ list, u32 n_list,
                                const u64* __restrict__ gear_table, CdcParams p,
                                u32* __restrict__ dense_list, u32* __restrict__ dense_count) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    u64* table = (u64*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32* cand_list = (u32*)(smem + kFastListOff) + wave * 64;
    load_table<kFastCopies, kFastWG>(table, gear_table, tid);
    __syncthreads();
    const u32 lane_tab = lds_lane_table<kFastCopies>(table, lane);
    const u32 li = blockIdx.x * kFastWaves + wave;
    if (li >= n_list) return;
    const u32 s = list[li];
    const u32 f = seg_file[s];
    const u64 size = file_size[f];
    u32* ends = ends32 + seg_slot[s];
    u64 last = 0;
    u32 n_out = 0;
    if (size) {
        CandPack pk;
        bool ovf;
        mark_tile<kFastCopies, false>(data + file_off[f], 0, (u32)size, nullptr, lane_tab, p.thresh_m1, lane, pk, ovf);
        if (!cand_compact(pk, ovf, (u32)lane * kLaneRun, lane, cand_list)) {      // wave-uniform
            if (lane == 0) dense_list[atomicAdd(dense_count, 1u)] = s;
            return;
        }
        // the wave's own LDS writes are ordered for the wave itself after the waitcnt the
        // compiler inserts; no other wave touches this list
        __builtin_amdgcn_wave_barrier();
        select_tile(nullptr, cand_list, 0, (u32)size, p, last, lane,
                    [&](u64 c) { if (lane == 0) ends[n_out] = (u32)c; ++n_out; return false; });
    }
    if (lane == 0) {
        if (size > last) { ends[n_out] = (u32)size; ++n_out; }   // the file end always cuts
        seg_n[s] = n_out;
    }
}

__global__ __launch_bounds__(kGearWG)
void gear_cdc_small_kernel(const u8* __restrict__ data, const u64* __restrict__ file_off,
                           const u64* __restrict__ file_size, const u32* __restrict__ seg_file,
                           const u64* __restrict__ seg_slot, u32* __restrict__ ends32,
                           u32* __restrict__ seg_n, const u32* __restrict__ list, u32 n_list,
                           const u32* __restrict__ n_list_dev,
                           const u64* __restrict__ gear_table, CdcParams p) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    u64* table = (u64*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (n_list_dev) {                                    // the fast kernel's leftovers: usually none
        n_list = *n_list_dev;
        if (blockIdx.x * kWavesPerWG >= n_list) return;
    }
    u32* bitmap = (u32*)(smem + kTableBytes) + wave * kBitmapWords;
    u32* cand_list = (u32*)(smem + kLdsListOff) + wave * 64;
    load_table<kCopies>(table, gear_table, tid);
    __syncthreads();
    const u32 lane_tab = lds_lane_table<kCopies>(table, lane);
    const u32 li = blockIdx.x * kWavesPerWG + wave;
    if (li >= n_list) return;
    const u32 s = list[li];
    const u32 f = seg_file[s];
    const u64 size = file_size[f];
    u32* ends = ends32 + seg_slot[s];
    u64 last = 0;
    u32 n_out = 0;
    if (size) {
        CandPack pk;
        bool ovf;
        mark_tile<kCopies, true>(data + file_off[f], 0, (u32)size, bitmap, lane_tab, p.thresh_m1, lane,
                                 pk, ovf);
        const bool fast = cand_compact(pk, ovf, (u32)lane * kLaneRun, lane, cand_list);
        // the wave's own LDS writes are ordered for the wave itself after the waitcnt the
        // compiler inserts; no other wave touches this bitmap / list
        __builtin_amdgcn_wave_barrier();
        select_tile(bitmap, fast ? cand_list : nullptr, 0, (u32)size, p, last, lane,
                    [&](u64 c) { if (lane == 0) ends[n_out] = (u32)c; ++n_out; return false; });
    }
    if (lane == 0) {
        if (size > last) { ends[n_out] = (u32)size; ++n_out; }   // the file end always cuts
        seg_n[s] = n_out;
    }
}

// ---- large files: speculative groups, validation, per-file fix-up ---------------------------
// A large file is cut into GROUPS of kWavesPerWG tiles (256 KiB); group gi of file f is segment
// file_seg0[f] + gi.  Cut selection is a sequential recurrence over the file (the next cut depends
// on the previous one), but it forgets its start quickly: from ANY previous cut the chosen sequence
// joins the true one as soon as both pick the same candidate, and then stays on it.  So:
//   A1 gear_tile_mark_kernel (one wave per tile, all tiles in parallel, no workgroup barrier): mark,
//      keep the tile's sorted candidate list (<= 64 entries, u32) in HBM.
//   A2 gear_group_spec_kernel (one wave per group): select SPECULATIVELY from the four lists as if a
//      cut fell on the group's first byte -> spec list (u32, relative to the group start), spec exit
//      E_g = its last cut.  Group 0 of a file starts at a true cut: its spec list is final.
//      (Until round 2's last change A1 + A2 were one kernel, a workgroup per group: three waves
//      waited while wave 0 selected -- 15-20 % more wave-cycles for the same instructions.)
//   B  gear_group_validate_kernel (all groups gi > 0 in parallel, one wave each): re-select from
//      the ASSUMED entry E_{g-1} until a cut is also a spec cut of this group (index sidx): the
//      group's cuts are then prefix[0, pcnt) ++ spec[sidx, spec_n) and its exit is E_g again.  If it
//      never meets the spec list the whole re-selection is the prefix and the exit is its own.
//   C  gear_file_fix_kernel (one workgroup per large file): walks the file's groups 64 at a time
//      checking entry_g == exit_{g-1}; where that fails (the previous group did not come back to
//      its spec exit, or the group is DENSE) the group is re-selected from the true entry, which
//      usually re-synchronises inside that same group.  Data that never re-synchronises (e.g.
//      forced cuts only, with max_size not dividing the group size) degenerates to one sequential
//      re-selection per group -- correct, as slow as a serial chunker.
// DENSE tile: more than 64 candidates (mask_bits far below the default): no list, its group gets no
// speculation; B skips such groups and their successors, C re-marks the tiles to get the bitmaps back
// and selects from the true entry (a file of dense groups is cut at the pace of a serial chunker).
// Result per group: GroupRec + two u32 regions of R = kGroupBytes / min_size + 2 entries at
// ends32[seg_slot[s]]: [0, R) spec list, [R, 2R) prefix.

// Re-selection of one group by one wave from `entry` (<= g0), meeting the spec list if it can.
// lists: LDS, kWavesPerWG x 64 candidates (tile-relative); bitmaps (LDS) are used for tiles whose
// bit in fast_mask is clear.  Writes prefix cuts (relative to g0) and fills rec's final fields.
__device__ __forceinline__ u64 reselect_group(const u32* lists, const u32* bitmaps, u32 fast_mask,
                                               u64 g0, u64 size, u64 entry, const u32* __restrict__ spec,
                                               u32 spec_n, u64 spec_exit, u32* __restrict__ prefix,
                                               const CdcParams& p, int lane, GroupRec* rec, u32* seg_n_out,
                                               bool open_end) {
    u64 last = entry;
    u32 pcnt = 0, sidx = spec_n;
    u32 sbase = 0;
    u32 sreg = lane < (int)spec_n ? spec[lane] : kNoCand;      // spec[sbase + lane]
    bool synced = false;
    auto emit = [&](u64 c) -> bool {
        const u32 rel = (u32)(c - g0);
        for (;;) {                                             // first spec cut >= rel
            const u64 bal = __ballot(sreg >= rel);
            if (bal) {
                const int i = __ffsll((unsigned long long)bal) - 1;
                if (lane_value(sreg, i) == rel && sbase + (u32)i < spec_n) { sidx = sbase + (u32)i; synced = true; }
                break;
            }
            if (sbase + 64u >= spec_n) break;
            sbase += 64u;
            sreg = sbase + (u32)lane < spec_n ? spec[sbase + lane] : kNoCand;
        }
        if (synced) return true;
        if (lane == 0) prefix[pcnt] = rel;
        ++pcnt;
        return false;
    };
    for (int t = 0; t < kWavesPerWG && !synced; ++t) {
        const u64 tts = g0 + (u64)t * kGearTile;
        if (tts >= size) break;
        const u32 tlen = (u32)((size - tts < (u64)kGearTile) ? (size - tts) : (u64)kGearTile);
        select_tile(bitmaps + t * kBitmapWords, (fast_mask >> t) & 1u ? lists + t * 64 : nullptr, tts, tlen,
                    p, last, lane, emit);
    }
    if (!synced && !open_end && g0 + kGroupBytes >= size && size > last) {  // the file's last group: the end cuts
        if (!emit(size)) last = size;
    }
    const u64 exit = synced ? spec_exit : last;
    if (lane == 0) {
        rec->entry = entry;
        rec->pcnt = pcnt;
        rec->sidx = sidx;
        rec->final_exit = exit;
        rec->flags = (rec->flags & kGroupDense) | kGroupValid;
        *seg_n_out = pcnt + (spec_n - sidx);
    }
    return exit;                                               // wave-uniform
}

// A1: one wave per TILE of every large file -- the small-file kernel's marking without its selection: a
// workgroup per group (four tiles), no workgroup barrier behind the table load, no loop (a persistent
// form needs 168+ VGPRs where this one, like the small-file kernel, takes 146: three workgroups per CU).  Per tile: the sorted candidate list (64 x u32, HBM) and
// tile_fast = 1, or tile_fast = 0 for a DENSE tile (more than 64 candidates: no list).
__global__ __launch_bounds__(kFastWG)
void gear_tile_mark_kernel(const u8* __restrict__ data, const u64* __restrict__ file_off,
                           const u64* __restrict__ file_size, const u32* __restrict__ group_file,
                           const u32* __restrict__ group_index, u32 n_groups,
                           u32* __restrict__ tile_lists, u32* __restrict__ tile_fast,
                           const u64* __restrict__ gear_table, CdcParams p) {
    // bitmap-free like gear_cdc_small_fast_kernel (16 table copies): a tile with more than 64 candidates
    // -- or a lane with more than six -- is DENSE (tile_fast = 0) and re-marked, with bitmaps, by C
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    u64* table = (u64*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32* cl = (u32*)(smem + kFastListOff) + wave * 64;
    load_table<kFastCopies, kFastWG>(table, gear_table, tid);
    __syncthreads();
    const u32 lane_tab = lds_lane_table<kFastCopies>(table, lane);
    // one wave per tile, four tiles per group, whatever the workgroup's size
    const u64 gw = (u64)blockIdx.x * kFastWaves + (u64)wave;
    const u32 g = (u32)(gw / kWavesPerWG);
    if (g >= n_groups) return;
    const int tw = (int)(gw % kWavesPerWG);
    const u64 t = (u64)g * kWavesPerWG + tw;
    const u32 f = group_file[g];
    const u64 size = file_size[f];
    const u64 ts = (u64)group_index[g] * kGroupBytes + (u64)tw * kGearTile;
    bool fast = true;                                         // tiles past the end count as listed
    if (ts < size) {
        const u32 tlen = (u32)((size - ts < (u64)kGearTile) ? (size - ts) : (u64)kGearTile);
        CandPack pk;
        bool ovf;
        mark_tile<kFastCopies, false>(data + file_off[f], ts, tlen, nullptr, lane_tab, p.thresh_m1, lane, pk, ovf);
        fast = cand_compact(pk, ovf, (u32)lane * kLaneRun, lane, cl);
        __builtin_amdgcn_wave_barrier();
        tile_lists[t * 64 + lane] = fast ? cl[lane] : kNoCand;
    }
    if (lane == 0) tile_fast[t] = fast ? 1u : 0u;
}

// A2: one wave per group: the SPECULATIVE selection (a cut assumed at the group's first byte) from the
// tiles' candidate lists.  A group with a dense tile gets no speculation: kGroupDense, left to C.
__global__ __launch_bounds__(kGearWG)
void gear_group_spec_kernel(const u64* __restrict__ file_size, const u64* __restrict__ file_seg0,
                            const u64* __restrict__ seg_slot, u32* __restrict__ ends32,
                            u32* __restrict__ seg_n, const u32* __restrict__ group_file,
                            const u32* __restrict__ group_index, u32 n_groups,
                            GroupRec* __restrict__ recs, const u32* __restrict__ tile_lists,
                            const u32* __restrict__ tile_fast, const u32* __restrict__ file_flags,
                            CdcParams p) {
    __shared__ u32 lists[kWavesPerWG][kWavesPerWG * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 g = blockIdx.x * kWavesPerWG + wave;
    if (g >= n_groups) return;
    const u32 f = group_file[g], gi = group_index[g];
    const u64 size = file_size[f];
    const bool open_end = file_flags && (file_flags[f] & kFileOpenEnd);
    const u64 g0 = (u64)gi * kGroupBytes;
    const u64 s = file_seg0[f] + gi;
    u32 fast_mask = 0;
#pragma unroll
    for (int t = 0; t < kWavesPerWG; ++t) {
        lists[wave][t * 64 + lane] = tile_lists[((u64)g * kWavesPerWG + t) * 64 + lane];
        fast_mask |= (tile_fast[(u64)g * kWavesPerWG + t] ? 1u : 0u) << t;
    }
    __builtin_amdgcn_wave_barrier();
    GroupRec r;
    r.pcnt = 0;
    r.sidx = 0;
    r.entry = gi == 0 ? 0ull : ~0ull;
    if (fast_mask != (1u << kWavesPerWG) - 1u) {              // dense: C re-marks it and selects from the bitmaps
        r.spec_exit = r.final_exit = ~0ull;
        r.spec_n = 0;
        r.flags = kGroupDense;
        if (lane == 0) { recs[g] = r; seg_n[s] = 0; }
        return;
    }
    u32* spec = ends32 + seg_slot[s];
    u64 last = g0;                                            // speculation: a cut at the group start
    u32 n_out = 0;
    for (int t = 0; t < kWavesPerWG; ++t) {
        const u64 tts = g0 + (u64)t * kGearTile;
        if (tts >= size) break;
        const u32 tlen = (u32)((size - tts < (u64)kGearTile) ? (size - tts) : (u64)kGearTile);
        select_tile(nullptr, lists[wave] + t * 64, tts, tlen, p, last, lane,
                    [&](u64 c) { if (lane == 0) spec[n_out] = (u32)(c - g0); ++n_out; return false; });
    }
    if (lane == 0) {
        if (!open_end && g0 + kGroupBytes >= size && size > last) {   // the file's last group: the end cuts
            spec[n_out] = (u32)(size - g0); ++n_out; last = size;
        }
        r.spec_exit = last;
        r.spec_n = n_out;
        r.final_exit = last;
        r.flags = 0;
        if (gi == 0) { r.flags |= kGroupValid; seg_n[s] = n_out; }   // starts at a true cut: final
        recs[g] = r;
    }
}

// B: one wave per group.  Groups of one file are consecutive in g, so g - 1 is the previous group
// of the same file whenever gi > 0.
__global__ __launch_bounds__(kGearWG)
void gear_group_validate_kernel(const u64* __restrict__ file_size, const u64* __restrict__ file_seg0,
                                const u64* __restrict__ seg_slot, u32* __restrict__ ends32,
                                u32* __restrict__ seg_n, const u32* __restrict__ group_file,
                                const u32* __restrict__ group_index, u32 n_groups,
                                GroupRec* __restrict__ recs, const u32* __restrict__ tile_lists,
                                const u32* __restrict__ file_flags, u32 region, CdcParams p) {
    __shared__ u32 lists[kWavesPerWG][kWavesPerWG * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 g = blockIdx.x * kWavesPerWG + wave;
    if (g >= n_groups) return;
    const u32 gi = group_index[g];
    if (gi == 0) return;
    GroupRec* rec = recs + g;
    if ((rec->flags | recs[g - 1].flags) & kGroupDense) return;   // own or previous group dense: left to the per-file pass
    const u32 f = group_file[g];
    const u64 s = file_seg0[f] + gi;
#pragma unroll
    for (int t = 0; t < kWavesPerWG; ++t)
        lists[wave][t * 64 + lane] = tile_lists[((u64)g * kWavesPerWG + t) * 64 + lane];
    __builtin_amdgcn_wave_barrier();
    u32* spec = ends32 + seg_slot[s];
    (void)reselect_group(lists[wave], nullptr, (1u << kWavesPerWG) - 1u, (u64)gi * kGroupBytes, file_size[f],
                         recs[g - 1].spec_exit, spec, rec->spec_n, rec->spec_exit, spec + region, p, lane,
                         rec, seg_n + s, file_flags && (file_flags[f] & kFileOpenEnd));
}

// C: one workgroup per large file.
__global__ __launch_bounds__(kGearWG)
void gear_file_fix_kernel(const u8* __restrict__ data, const u64* __restrict__ file_off,
                          const u64* __restrict__ file_size, const u64* __restrict__ file_seg0,
                          const u64* __restrict__ seg_slot, u32* __restrict__ ends32,
                          u32* __restrict__ seg_n, const u32* __restrict__ large_list,
                          const u32* __restrict__ large_group0, GroupRec* __restrict__ recs,
                          const u32* __restrict__ tile_lists, const u32* __restrict__ file_flags,
                          u32 region, const u64* __restrict__ gear_table, CdcParams p) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    u64* table = (u64*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32* bitmaps = (u32*)(smem + kTableBytes);
    u32* cand_lists = (u32*)(smem + kLdsListOff);
    volatile u32* fast_flags = (volatile u32*)(smem + kLdsFastOff);
    volatile u32* s_next = (volatile u32*)(smem + kGearLdsBytes);     // [0] group to redo or ~0, [1] dense?
    volatile u64* s_entry = (volatile u64*)(smem + kGearLdsBytes + 8);
    const u32 f = large_list[blockIdx.x];
    const u64 size = file_size[f];
    const u8* fptr = data + file_off[f];
    const u32 gb = large_group0[blockIdx.x];
    const bool open_end = file_flags && (file_flags[f] & kFileOpenEnd);
    const u32 ng = (u32)((size + kGroupBytes - 1) / kGroupBytes);
    u32 gi = 0;                                               // next group to check
    u64 prev_exit = 0;                                        // true exit of group gi - 1 (wave 0); the file starts at a cut
    bool have_table = false;                                  // loaded only if some group has to be redone
    u32 lane_tab = 0;
    for (;;) {
        // wave 0: skip ahead over groups whose assumption holds
        if (wave == 0) {
            u32 redo = 0xFFFFFFFFu, dense = 0;
            while (gi < ng) {
                // 256 groups per round: four records per lane, all loads issued before the first use
                // (a 16 GiB file has 65 536 groups; one record per lane and round cost 0.6 ms)
                bool bad[4];
                u64 ex[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const u32 my = gi + (u32)(k * 64 + lane);
                    bad[k] = false;
                    ex[k] = 0;
                    if (my < ng) {
                        const GroupRec r = recs[gb + my];
                        const u64 pe = my == gi ? prev_exit : recs[gb + my - 1].final_exit;
                        ex[k] = r.final_exit;
                        bad[k] = !(r.flags & kGroupValid) || r.entry != pe;
                    }
                }
                bool stop = false;
#pragma unroll
                for (int k = 0; k < 4 && !stop; ++k) {
                    if (gi >= ng) break;
                    const u64 bal = __ballot(bad[k]);
                    if (!bal) {                               // up to 64 good groups: their exits are true
                        const u32 n = ng - gi < 64u ? ng - gi : 64u;
                        const u32 lo = lane_value((u32)ex[k], (int)n - 1), hi = lane_value((u32)(ex[k] >> 32), (int)n - 1);
                        prev_exit = ((u64)hi << 32) | lo;
                        gi += n;
                        continue;
                    }
                    const int j = __ffsll((unsigned long long)bal) - 1;
                    if (j > 0) {                              // groups before j are good
                        const u32 lo = lane_value((u32)ex[k], j - 1), hi = lane_value((u32)(ex[k] >> 32), j - 1);
                        prev_exit = ((u64)hi << 32) | lo;
                    }
                    gi += (u32)j;
                    redo = gi;
                    dense = recs[gb + gi].flags & kGroupDense;
                    stop = true;
                }
                if (stop) break;
            }
            if (lane == 0) { s_next[0] = redo; s_next[1] = dense; *s_entry = prev_exit; }
        }
        __syncthreads();
        const u32 redo = s_next[0];
        if (redo == 0xFFFFFFFFu) break;
        if (!have_table) {                                    // first group to redo: now the table is needed
            load_table<kCopies>(table, gear_table, tid);
            __syncthreads();
            lane_tab = lds_lane_table<kCopies>(table, lane);
            have_table = true;
        }
        const u32 g = gb + redo;
        const u64 g0 = (u64)redo * kGroupBytes;
        if (s_next[1]) {                                      // dense: the bitmaps have to be rebuilt
            const u64 ts = g0 + (u64)wave * kGearTile;
            if (lane == 0) fast_flags[wave] = 1u;
            if (ts < size) {
                const u32 tlen = (u32)((size - ts < (u64)kGearTile) ? (size - ts) : (u64)kGearTile);
                CandPack pk;
                bool ovf;
                u32* bm = bitmaps + wave * kBitmapWords;
                u32* cl = cand_lists + wave * 64;
                mark_tile<kCopies, true>(fptr, ts, tlen, bm, lane_tab, p.thresh_m1, lane, pk, ovf);
                bool fast = cand_compact(pk, ovf, (u32)lane * kLaneRun, lane, cl);
                if (!fast) {
                    __builtin_amdgcn_wave_barrier();
                    fast = list_from_bitmap(bm, lane, cl);
                }
                if (lane == 0) fast_flags[wave] = fast ? 1u : 0u;
            }
        } else {
            cand_lists[wave * 64 + lane] = tile_lists[((u64)g * kWavesPerWG + wave) * 64 + lane];
            if (lane == 0) fast_flags[wave] = 1u;
        }
        __syncthreads();
        if (wave == 0) {
            u32 fast_mask = 0;
            for (int t = 0; t < kWavesPerWG; ++t) fast_mask |= (fast_flags[t] ? 1u : 0u) << t;
            const u64 s = file_seg0[f] + redo;
            u32* spec = ends32 + seg_slot[s];
            GroupRec* rec = recs + g;
            const u64 entry = *s_entry;
            prev_exit = reselect_group(cand_lists, bitmaps, fast_mask, g0, size, entry, spec, rec->spec_n,
                                       rec->spec_exit, spec + region, p, lane, rec, seg_n + s, open_end);
            gi = redo + 1;
        }
        __syncthreads();
    }
}

u64 gear_large_groups(u64 size) { return (size + kGroupBytes - 1) / kGroupBytes; }
u64 gear_group_region(u32 min_size) { return kGroupBytes / min_size + 2; }
size_t gear_group_rec_bytes() { return sizeof(GroupRec); }

// more than 64 KiB of dynamic LDS needs the attribute; per device (several ctxs may use several GPUs)
static void gear_lds_attributes() {
    static thread_local int done_for = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (done_for == dev) return;
    (void)hipFuncSetAttribute((const void*)gear_cdc_small_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, kGearLdsBytes);
    (void)hipFuncSetAttribute((const void*)gear_cdc_small_fast_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, kFastLdsBytes);
    (void)hipFuncSetAttribute((const void*)gear_tile_mark_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, kFastLdsBytes);
    (void)hipFuncSetAttribute((const void*)gear_file_fix_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, kGearLdsBytes + 16);
    done_for = dev;
}

void launch_gear_cdc(const GearLaunch& a, CdcParams p, int n_cu, hipStream_t s) {
    (void)n_cu;
    gear_lds_attributes();
    if (a.n_small) {
        const dim3 grid((a.n_small + kWavesPerWG - 1) / kWavesPerWG);
        const dim3 fast_grid((a.n_small + kFastWaves - 1) / kFastWaves);
        // expected candidates per 64 KiB tile from the mask alone: with a dozen or more the 64-entry list
        // overflows too often for the bitmap-free kernel to be worth its pass
        const bool try_fast = a.dense_list && p.thresh_m1 <= 0x003FFFFFu;         // mask_bits >= 10
        if (try_fast) {
            (void)hipMemsetAsync(a.dense_count, 0, 4, s);
            hipLaunchKernelGGL(gear_cdc_small_fast_kernel, fast_grid, dim3(kFastWG), kFastLdsBytes, s, a.data,
                               a.file_off, a.file_size, a.seg_file, a.seg_slot, a.ends32, a.seg_n, a.small_list,
                               a.n_small, a.gear_table, p, a.dense_list, a.dense_count);
            hipLaunchKernelGGL(gear_cdc_small_kernel, grid, dim3(kGearWG), kGearLdsBytes, s, a.data, a.file_off,
                               a.file_size, a.seg_file, a.seg_slot, a.ends32, a.seg_n, a.dense_list, 0u,
                               (const u32*)a.dense_count, a.gear_table, p);
        } else {
            hipLaunchKernelGGL(gear_cdc_small_kernel, grid, dim3(kGearWG), kGearLdsBytes, s, a.data, a.file_off,
                               a.file_size, a.seg_file, a.seg_slot, a.ends32, a.seg_n, a.small_list, a.n_small,
                               (const u32*)nullptr, a.gear_table, p);
        }
    }
    if (a.n_groups) {
        const u32 region = (u32)gear_group_region(p.min_size);
        GroupRec* recs = (GroupRec*)a.group_recs;
        const u64 n_tiles = (u64)a.n_groups * kWavesPerWG;
        hipLaunchKernelGGL(gear_tile_mark_kernel, dim3((u32)((n_tiles + kFastWaves - 1) / kFastWaves)), dim3(kFastWG), kFastLdsBytes, s, a.data,
                           a.file_off, a.file_size, a.group_file, a.group_index, a.n_groups, a.tile_lists,
                           a.tile_fast, a.gear_table, p);
        const dim3 per_group((a.n_groups + kWavesPerWG - 1) / kWavesPerWG);
        hipLaunchKernelGGL(gear_group_spec_kernel, per_group, dim3(kGearWG), 0, s, a.file_size, a.file_seg0,
                           a.seg_slot, a.ends32, a.seg_n, a.group_file, a.group_index, a.n_groups, recs,
                           a.tile_lists, a.tile_fast, a.file_flags, p);
        if (a.n_groups > a.n_large)                       // some file has more than one group
            hipLaunchKernelGGL(gear_group_validate_kernel, per_group, dim3(kGearWG), 0, s, a.file_size,
                               a.file_seg0, a.seg_slot, a.ends32, a.seg_n, a.group_file, a.group_index,
                               a.n_groups, recs, a.tile_lists, a.file_flags, region, p);
        // always: a dense group (even a file's only one) is selected here; a file with nothing to redo
        // costs one record read
        hipLaunchKernelGGL(gear_file_fix_kernel, dim3(a.n_large), dim3(kGearWG), kGearLdsBytes + 16, s,
                           a.data, a.file_off, a.file_size, a.file_seg0, a.seg_slot, a.ends32, a.seg_n,
                           a.large_list, a.large_group0, recs, a.tile_lists, a.file_flags, region,
                           a.gear_table, p);
    }
}

// ---- parts: a byte range of a file that is split across batches / GPUs ------------------------
// A part is staged with a HALO of whole groups in front of its own range (>= max_size bytes, so the
// chunk that straddles the part's start lies inside the item) and is chunked like any large file;
// then: (1) part_apply puts the TRUE entry of the first own group -- the previous part's last cut,
// known only after the parts' owners have talked -- where the fix-up pass reads the halo's exit, and
// that pass runs again over the parts (a no-op walk when the halo had re-synchronised, the usual
// case); (2) the halo groups' chunk counts are cleared: their cuts belong to the previous part.
// A part that is not the file's last ends OPEN (kFileOpenEnd): no cut at its last byte.
__global__ __launch_bounds__(64)
void part_apply_kernel(const u32* __restrict__ part_group0, const u32* __restrict__ part_halo,
                       const u64* __restrict__ part_entry, u32 n_parts, GroupRec* __restrict__ recs) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_parts || part_halo[i] == 0 || part_entry[i] == ~0ull) return;
    recs[part_group0[i] + part_halo[i] - 1].final_exit = part_entry[i];
}

__global__ __launch_bounds__(64)
void part_halo_clear_kernel(const u32* __restrict__ part_file, const u32* __restrict__ part_halo,
                            u32 n_parts, const u64* __restrict__ file_seg0, u32* __restrict__ seg_n) {
    const u32 i = blockIdx.x;
    if (i >= n_parts) return;
    const u64 s0 = file_seg0[part_file[i]];
    for (u32 g = threadIdx.x; g < part_halo[i]; g += blockDim.x) seg_n[s0 + g] = 0;
}

void launch_gear_parts(const GearLaunch& a, const u32* d_part_file, const u32* d_part_group0,
                       const u32* d_part_halo, const u64* d_part_entry, u32 n_parts, bool refix,
                       CdcParams p, hipStream_t s) {
    if (n_parts == 0) return;
    GroupRec* recs = (GroupRec*)a.group_recs;
    if (refix) {
        const u32 region = (u32)gear_group_region(p.min_size);
        hipLaunchKernelGGL(part_apply_kernel, dim3((n_parts + 63) / 64), dim3(64), 0, s, d_part_group0,
                           d_part_halo, d_part_entry, n_parts, recs);
        hipLaunchKernelGGL(gear_file_fix_kernel, dim3(n_parts), dim3(kGearWG), kGearLdsBytes + 16, s,
                           a.data, a.file_off, a.file_size, a.file_seg0, a.seg_slot, a.ends32, a.seg_n,
                           d_part_file, d_part_group0, recs, a.tile_lists, a.file_flags, region,
                           a.gear_table, p);
    }
    hipLaunchKernelGGL(part_halo_clear_kernel, dim3(n_parts), dim3(64), 0, s, d_part_file, d_part_halo,
                       n_parts, a.file_seg0, a.seg_n);
}

}  // namespace mi
