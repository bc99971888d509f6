// crc32.hip -- CRC32-IEEE of every file of a batch on gfx950.
//
// What it serves: the per-file term of the reference's COPY/ADD cache ID --
// checksumPathContents (lib/builder/step/add_copy_step.go:194-238) feeds ONE running
// hash/crc32 (IEEE) with relpath, then all file bytes, for every walked path
// (:153-169), and SetCacheID prints it with "%x" (:102-122).  A running CRC is
// serial, but CRC is linear over GF(2):  crc(A||B) = crc(A) * x^(8|B|) + crc(B)
// (mod P), so file bytes are reduced in parallel here and the host splices the path
// strings in between with the same identity (mi_context_checksum in mi_api.hip).
//
// Mapping: one wave per 64 KiB tile, one lane per 1 KiB run (the Gear kernel's
// streaming pattern).  A lane runs slicing-by-4 (four 1 KiB tables in LDS) from state 0
// over its run; the wave folds the 64 lane remainders with per-lane multiplications by
// x^(8192*k) and one by x^(8*len_of_last_run) (tables of powers precomputed on the host),
// then every tile's term of its file's CRC is folded in parallel (crc32_fold_kernel: one lane per
// tile, x^(8 * distance) by square-and-multiply) and one lane per file finishes: crc = ~(acc ^ ~0 * x^(8 size)).
// Bytes: 1 B read per file byte, 4 B written per tile.  This pass is optional
// (MI_FLAG_FILE_CRC32) and not on the benchmarked path.
#include "mi_common.h"

namespace mi {

typedef u32 u32x4 __attribute__((ext_vector_type(4)));

constexpr u32 kCrcPoly = 0xEDB88320u;        // IEEE 802.3, reflected

// a(x) * b(x) mod P(x), reflected bit order (bit 31 = x^0)
__host__ __device__ inline u32 crc_mulmod(u32 a, u32 b) {
    u32 prod = 0;
    for (int i = 0; i < 32; ++i) {
        if (a & (0x80000000u >> i)) prod ^= b;
        b = (b & 1u) ? (b >> 1) ^ kCrcPoly : b >> 1;
    }
    return prod;
}

// Host-built constant block uploaded once per ctx (layout shared with mi_api.hip):
//   [0 .. 1024)        slicing tables T0..T3 (256 words each)
//   [1024 .. 1024+65)  pow1k[k]  = x^(8*1024*k) mod P, k = 0..64
//   [1089 .. 1089+1025) powb[j]  = x^(8*j) mod P,      j = 0..1024
//   [2114 .. 2114+40)  powt[j]   = x^(8*65536*2^j) mod P, j = 0..39 (tile distances of files up to 2^55 bytes)
void crc32_build_tables(u32* out) {
    u32* T = out;
    for (u32 i = 0; i < 256; ++i) {
        u32 c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ kCrcPoly : c >> 1;
        T[i] = c;
    }
    for (int t = 1; t < 4; ++t)
        for (u32 i = 0; i < 256; ++i) T[t * 256 + i] = (T[(t - 1) * 256 + i] >> 8) ^ T[T[(t - 1) * 256 + i] & 0xFF];
    u32* powb = out + kCrcPowBytesOff;
    powb[0] = 0x80000000u;                   // x^0
    const u32 x8 = 0x00800000u;              // x^8
    for (int j = 1; j <= 1024; ++j) powb[j] = crc_mulmod(powb[j - 1], x8);
    u32* pow1k = out + kCrcPow1kOff;
    pow1k[0] = 0x80000000u;
    for (int k = 1; k <= 64; ++k) pow1k[k] = crc_mulmod(pow1k[k - 1], powb[1024]);
    u32* powt = out + kCrcPowTileOff;
    powt[0] = pow1k[64];
    for (int j = 1; j < 40; ++j) powt[j] = crc_mulmod(powt[j - 1], powt[j - 1]);
}

__device__ __forceinline__ u32 crc_word(u32 c, u32 w, const u32* T) {
    c ^= w;
    return T[768 + (c & 0xFF)] ^ T[512 + ((c >> 8) & 0xFF)] ^ T[256 + ((c >> 16) & 0xFF)] ^ T[c >> 24];
}

// raw (state-0) remainder of tile `tile` of file `f`; one wave per tile
__global__ __launch_bounds__(256)
void crc32_tiles_kernel(const u8* __restrict__ data, const u64* __restrict__ file_off,
                        const u64* __restrict__ file_size, const u32* __restrict__ tile_file,
                        const u64* __restrict__ first_tile, u64 n_tiles,
                        const u32* __restrict__ consts, u32* __restrict__ tile_raw) {
    __shared__ u32 T[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) T[i] = consts[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const u64 tile = (u64)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    const u32 f = tile_file[tile];
    const u64 ts = (tile - first_tile[f]) * (u64)kGearTile;
    const u64 size = file_size[f];
    const u32 tlen = (u32)((size - ts < (u64)kGearTile) ? (size - ts) : (u64)kGearTile);
    const u32 run0 = (u32)lane * 1024u;
    u32 c = 0, run_len = 0;
    if (run0 < tlen) {
        run_len = tlen - run0 < 1024u ? tlen - run0 : 1024u;
        const u8* p = data + file_off[f] + ts + run0;
        u32 done = 0;
        for (; done + 128 <= run_len; done += 128) {          // one cache line at a time
            u32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = *(const u32x4*)(p + done + 16 * i);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                c = crc_word(c, v[i].x, T); c = crc_word(c, v[i].y, T);
                c = crc_word(c, v[i].z, T); c = crc_word(c, v[i].w, T);
            }
        }
        for (; done + 4 <= run_len; done += 4) c = crc_word(c, *(const u32*)(p + done), T);
        for (; done < run_len; ++done) c = (c >> 8) ^ T[(c ^ p[done]) & 0xFF];
    }
    // fold the 64 lane remainders:  raw = (XOR_{l<last} c_l * x^(8192*(last-1-l))) * x^(8*len_last) ^ c_last
    const u32 n_runs = (tlen + 1023u) / 1024u;                // >= 1
    const u32 last = n_runs - 1;
    u32 part = 0;
    if ((u32)lane < last) part = crc_mulmod(c, consts[kCrcPow1kOff + (last - 1 - (u32)lane)]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part ^= __shfl_xor(part, d);
    const u32 c_last = __shfl(c, (int)last);
    const u32 len_last = tlen - last * 1024u;
    if (lane == 0) tile_raw[tile] = crc_mulmod(part, consts[kCrcPowBytesOff + len_last]) ^ c_last;
}

// x^(8 * 65536 * k) mod P by square-and-multiply over the precomputed squares
__device__ __forceinline__ u32 crc_pow_tiles(u64 k, const u32* __restrict__ consts) {
    u32 r = 0x80000000u;                                       // x^0
    for (int j = 0; k; ++j, k >>= 1)
        if (k & 1) r = crc_mulmod(r, consts[kCrcPowTileOff + j]);
    return r;
}
__device__ __forceinline__ u32 crc_pow_tail(u32 len, const u32* __restrict__ consts) {   // x^(8 len), len <= 65536
    return crc_mulmod(consts[kCrcPow1kOff + (len >> 10)], consts[kCrcPowBytesOff + (len & 1023u)]);
}

// A file's CRC is linear in its tiles:  ~crc = 0xFFFFFFFF * x^(8 size)  ^  XOR_t raw_t * x^(8 * bytes behind tile t),
// so every tile's term is independent -- one LANE per tile here (a 16 GiB file has 262 144 of them; the first
// form of this pass walked them one after the other on one lane per file: 48 ms for four 4 GiB files).  The terms
// of a file are XORed into acc[f] (zeroed by the launcher): one atomic per wave when the wave's 64 tiles belong to
// one file, one per lane otherwise (small files: one tile each, no contention).
__global__ __launch_bounds__(256)
void crc32_fold_kernel(const u64* __restrict__ file_size, const u32* __restrict__ tile_file,
                       const u64* __restrict__ first_tile, u64 n_tiles, const u32* __restrict__ consts,
                       const u32* __restrict__ tile_raw, u32* __restrict__ acc) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = t < n_tiles;
    u32 f = 0xFFFFFFFFu, term = 0;
    if (live) {
        f = tile_file[t];
        const u64 size = file_size[f];
        const u64 lt = t - first_tile[f];
        const u64 nt = (size + kGearTile - 1) / kGearTile;
        term = tile_raw[t];
        if (lt + 1 < nt) {                                     // bytes behind me: nt-2-lt whole tiles + the last one
            const u32 len_last = (u32)(size - (nt - 1) * (u64)kGearTile);
            term = crc_mulmod(term, crc_mulmod(crc_pow_tiles(nt - 2 - lt, consts), crc_pow_tail(len_last, consts)));
        }
    }
    const u32 f0 = (u32)__builtin_amdgcn_readfirstlane((int)f);
    if (!__ballot(f != f0)) {                                  // the whole wave in one file (or all dead)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) term ^= __shfl_xor(term, d);
        if ((threadIdx.x & 63) == 0 && live) atomicXor(acc + f, term);
    } else if (live) {
        atomicXor(acc + f, term);
    }
}

// one lane per file: crc = ~(acc ^ 0xFFFFFFFF * x^(8 size))
__global__ __launch_bounds__(256)
void crc32_files_kernel(const u64* __restrict__ file_size, u64 n_files, const u32* __restrict__ consts,
                        u32* __restrict__ crc) {
    const u64 f = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_files) return;
    const u64 size = file_size[f];
    const u64 whole = size / kGearTile;
    const u32 m = crc_mulmod(crc_pow_tiles(whole, consts), crc_pow_tail((u32)(size - whole * (u64)kGearTile), consts));
    crc[f] = ~(crc[f] ^ crc_mulmod(0xFFFFFFFFu, m));
}

void launch_crc32_files(const u8* d_data, const u64* d_file_off, const u64* d_file_size,
                        const u32* d_tile_file, const u64* d_first_tile, u64 n_tiles, u64 n_files,
                        const u32* d_consts, u32* d_tile_raw, u32* d_crc, hipStream_t s) {
    if (n_files == 0) return;
    (void)hipMemsetAsync(d_crc, 0, n_files * sizeof(u32), s);          // the accumulators of the fold
    if (n_tiles) {
        hipLaunchKernelGGL(crc32_tiles_kernel, dim3((u32)((n_tiles + 3) / 4)), dim3(256), 0, s, d_data,
                           d_file_off, d_file_size, d_tile_file, d_first_tile, n_tiles, d_consts,
                           d_tile_raw);
        hipLaunchKernelGGL(crc32_fold_kernel, dim3((u32)((n_tiles + 255) / 256)), dim3(256), 0, s, d_file_size,
                           d_tile_file, d_first_tile, n_tiles, d_consts, d_tile_raw, d_crc);
    }
    hipLaunchKernelGGL(crc32_files_kernel, dim3((u32)((n_files + 255) / 256)), dim3(256), 0, s,
                       d_file_size, n_files, d_consts, d_crc);
}

// ---- host-side helpers for the context-checksum splice (strings only) -----------------
u32 crc32_host_bytes(u32 crc, const void* data, size_t len) {
    // several ctxs may compute checksums on different threads at once: the table is built by a
    // thread-safe function-local static (C++11 "magic static"), not by a hand-rolled init flag
    struct Table {
        u32 t[256];
        Table() {
            for (u32 i = 0; i < 256; ++i) {
                u32 c = i;
                for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ kCrcPoly : c >> 1;
                t[i] = c;
            }
        }
    };
    static const Table table;
    const u32* T = table.t;
    const u8* p = (const u8*)data;
    u32 c = ~crc;
    while (len--) c = (c >> 8) ^ T[(c ^ *p++) & 0xFF];
    return ~c;
}

u32 crc32_host_combine(u32 crc1, u32 crc2, u64 len2) {
    u32 result = 0x80000000u, sq = 0x00800000u;              // x^0, x^8
    for (u64 n = len2; n; n >>= 1) {
        if (n & 1) result = crc_mulmod(sq, result);
        sq = crc_mulmod(sq, sq);
    }
    return crc_mulmod(result, crc1) ^ crc2;
}

}  // namespace mi
