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
// Mapping: one 256-thread workgroup per file, walking the file in 64 KiB tiles.
//   1. tile (+64 B halo) -> LDS with coalesced 16 B/lane global loads; rows of 256 B
//      are padded by 16 B so the per-lane ds_read_b128 below are conflict-free;
//   2. every lane owns a 256 B run: warms h over the 64 bytes before it, then rolls
//      over its run (Gear table = 2 KiB in LDS), tracking min(hi32(h)) per 16 bytes
//      with v_min3_u32 so the candidate test costs 1/2 VALU op per byte; hits set
//      bits in an LDS bitmap (one bit per byte of the tile);
//   3. wave 0 selects cuts from the bitmap with wave-wide find-first-set
//      (64 lanes x 64 bits per step, __ballot + ctz), carrying last_cut across
//      tiles, and appends chunk ends to the file's slot region in HBM.
// HBM traffic: every file byte read once (+64 B halo per tile), 8 B written per
// chunk.  Bound: HBM / LDS-lookup rate.
#include "mi_common.h"

namespace mi {

typedef u32 u32x4 __attribute__((ext_vector_type(4)));

constexpr int kRowPad     = 16;
constexpr int kTileLogical = kGearHalo + kGearTile;                 // halo + tile bytes
constexpr int kTileRows   = kTileLogical / kGearRun + 1;            // 257 rows of 256 B
constexpr int kTileLds    = kTileLogical + kRowPad * kTileRows;     // padded bytes
constexpr int kBitmapWords = kGearTile / 32;                        // u32 words
constexpr int kGearLdsBytes = ((kTileLds + 15) / 16) * 16 + kBitmapWords * 4 + 256 * 8;

__device__ __forceinline__ u32 lds_phys(u32 x) { return x + ((x >> 8) << 4); }

// first set bit of the LDS bitmap within [lo, hi] (bit indices, inclusive), -1 if none.
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

__global__ __launch_bounds__(kGearWG)
void gear_cdc_files_kernel(const u8* __restrict__ data, const u64* __restrict__ file_off,
                           const u64* __restrict__ file_size, const u64* __restrict__ slot_base,
                           u64* __restrict__ slot_ends, u32* __restrict__ n_chunks,
                           const u64* __restrict__ gear_table, CdcParams p) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    u8*  tile   = smem;                                                  // kTileLds bytes
    u32* bitmap = (u32*)(smem + ((kTileLds + 15) / 16) * 16);            // kBitmapWords
    u64* table  = (u64*)(bitmap + kBitmapWords);                         // 256 x u64

    const int tid = threadIdx.x, lane = tid & 63;
    const u64 f = blockIdx.x;
    const u64 size = file_size[f];
    const u8* fptr = data + file_off[f];
    u64* ends = slot_ends + slot_base[f];

    table[tid] = gear_table[tid];                                        // kGearWG == 256

    u64 last = 0;          // wave-0 uniform: previous cut
    u32 n_out = 0;         // wave-0 uniform: chunks emitted

    for (u64 ts = 0; ts < size; ts += kGearTile) {
        const u32 tlen = (u32)((size - ts < (u64)kGearTile) ? (size - ts) : (u64)kGearTile);
        // ---- 1. stage halo + tile into LDS -----------------------------------------
        // logical byte x of the staging space = file byte ts - 64 + x
        const u32 n16 = (kGearHalo + tlen + 15) / 16;
        const u32 u_first = (ts == 0) ? kGearHalo / 16 : 0;              // no halo before byte 0
#pragma unroll 4
        for (u32 u = u_first + tid; u < n16; u += kGearWG) {
            const u32 x = u * 16;
            const u32x4 v = *(const u32x4*)(fptr + ts + x - kGearHalo);
            *(u32x4*)(tile + lds_phys(x)) = v;
        }
        for (u32 i = tid; i < (u32)kBitmapWords; i += kGearWG) bitmap[i] = 0;
        __syncthreads();

        // ---- 2. mark candidates -----------------------------------------------------
        {
            const u32 run0 = (u32)tid * kGearRun;        // tile-relative first byte of my run
            if (run0 < tlen) {
                u64 h = 0;
                const u32 xb = run0;                     // logical x of the warm-up start
#pragma unroll
                for (int j = 0; j < kGearHalo / 16; ++j) {
                    const u32x4 v = *(const u32x4*)(tile + lds_phys(xb + 16 * j));
                    const u32 wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        h = (h << 1) + table[(wv[k >> 2] >> (8 * (k & 3))) & 0xFF];
                }
                if (ts == 0 && tid == 0) h = 0;          // file start: window starts empty
#pragma unroll 2
                for (int j = 0; j < kGearRun / 16; ++j) {
                    const u32x4 v = *(const u32x4*)(tile + lds_phys(xb + kGearHalo + 16 * j));
                    const u32 wv[4] = {v.x, v.y, v.z, v.w};
                    u32 hh[16];
                    u32 m = 0xFFFFFFFFu;
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        h = (h << 1) + table[(wv[k >> 2] >> (8 * (k & 3))) & 0xFF];
                        hh[k] = (u32)(h >> 32);
                    }
#pragma unroll
                    for (int k = 0; k < 16; k += 2) m = min(m, min(hh[k], hh[k + 1]));   // v_min3_u32
                    if (m <= p.thresh_m1) {              // rare: ~1 lane in 512 per step
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            const u32 pos = run0 + 16 * j + k;           // byte index in tile
                            if (hh[k] <= p.thresh_m1 && pos < tlen)
                                atomicOr(&bitmap[pos >> 5], 1u << (pos & 31));
                        }
                    }
                }
            }
        }
        __syncthreads();

        // ---- 3. select cuts (wave 0) -------------------------------------------------
        if (tid < 64) {
            const u64 te = ts + tlen;                    // ends in this tile: (ts, te]
            for (;;) {
                u64 lo = last + p.min_size;
                if (lo < ts + 1) lo = ts + 1;
                u64 hi = last + p.max_size;
                if (hi > te) hi = te;
                if (lo <= hi) {
                    const int b = bitmap_find_first((const u64*)bitmap, (int)(lo - ts - 1),
                                                    (int)(hi - ts - 1), lane);
                    if (b >= 0) {
                        last = ts + (u64)b + 1;
                        if (lane == 0) ends[n_out] = last;
                        ++n_out;
                        continue;
                    }
                }
                if (last + p.max_size <= te) {           // forced cut at max_size
                    last += p.max_size;
                    if (lane == 0) ends[n_out] = last;
                    ++n_out;
                    continue;
                }
                break;
            }
        }
        __syncthreads();                                 // bitmap + tile are reused
    }
    if (tid == 0) {
        if (size > last) { ends[n_out] = size; ++n_out; }   // the file end always cuts
        n_chunks[f] = n_out;
    }
}

void launch_gear_cdc_files(const u8* d_data, const u64* d_file_off, const u64* d_file_size,
                           const u64* d_slot_base, u64* d_slot_ends, u32* d_n_chunks,
                           u64 n_files, const u64* d_gear_table, CdcParams p, hipStream_t s) {
    if (n_files == 0) return;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gear_cdc_files_kernel,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kGearLdsBytes);
        attr_set = true;
    }
    hipLaunchKernelGGL(gear_cdc_files_kernel, dim3((u32)n_files), dim3(kGearWG), kGearLdsBytes, s,
                       d_data, d_file_off, d_file_size, d_slot_base, d_slot_ends, d_n_chunks,
                       d_gear_table, p);
}

}  // namespace mi
