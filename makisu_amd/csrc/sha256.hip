// sha256.hip -- batched SHA-256 on gfx950: one lane per independent byte string.
//
// What it replaces: the per-stream crypto/sha256 hashers of the reference
// (lib/builder/step/common.go:44-45; lib/docker/image/digester.go:33-60), applied
// per chunk / per file instead of per layer.  A single SHA-256 stream is serial
// (Merkle-Damgard), so the parallelism is ACROSS strings: every lane owns one
// string, keeps its 8-word state and 16-word schedule in VGPRs and runs the 64
// rounds with v_alignbit_b32 (rotates), v_bitop3_b32 (xor3 / Ch / Maj in one op) and v_add3_u32.
//
// Roofline: this kernel is VALU-integer bound, not HBM bound: ~1650 VALU ops per
// 64-byte block against 256 CU x 128 lanes/clk.  HBM traffic is
// 1 byte read per byte hashed + 32 bytes written per string.
//
// Scheduling: strings have very different lengths (2 KiB..64 KiB chunks), so lanes
// pull work from kShaQueues global queues holding the items longest-first (LPT):
// a lane that finishes its string takes the next one instead of idling until the
// slowest lane of its wave is done.  Dequeues are aggregated per wave (one atomic
// for all lanes that finished in the same iteration).
#include "mi_common.h"

namespace mi {

typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_unaligned __attribute__((aligned(1)));
typedef u32 u32_unaligned __attribute__((aligned(1)));

__device__ __forceinline__ u32 rotr(u32 x, u32 n) { return __builtin_amdgcn_alignbit(x, x, n); }
// gfx950 v_bitop3_b32: any 3-input boolean in ONE VALU op (truth table: a=0xF0, b=0xCC, c=0xAA)
__device__ __forceinline__ u32 xor3(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ u32 ch3(u32 e, u32 f, u32 g)  { return __builtin_amdgcn_bitop3_b32(e, f, g, 0xCA); }
__device__ __forceinline__ u32 maj3(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }

#define MI_SHA_K(i) kK256[i]
__device__ constexpr u32 kK256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4,
    0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe,
    0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f,
    0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7,
    0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc,
    0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b,
    0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116,
    0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7,
    0xc67178f2};

// One compression: st += F(st, w).  Fully unrolled so a..h and the 16-entry
// schedule ring live in fixed VGPRs and K[i] folds into literals.
__device__ __forceinline__ void sha256_compress(u32 (&st)[8], u32 (&w)[16]) {
    u32 a = st[0], b = st[1], c = st[2], d = st[3];
    u32 e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        u32 wi;
        if (i < 16) {
            wi = w[i];
        } else {
            const u32 w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
            const u32 s0 = xor3(rotr(w15, 7), rotr(w15, 18), w15 >> 3);
            const u32 s1 = xor3(rotr(w2, 17), rotr(w2, 19), w2 >> 10);
            wi = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
            w[i & 15] = wi;
        }
        const u32 S1 = xor3(rotr(e, 6), rotr(e, 11), rotr(e, 25));
        const u32 t1 = (h + ch3(e, f, g) + (wi + kK256[i])) + S1;
        const u32 S0 = xor3(rotr(a, 2), rotr(a, 13), rotr(a, 22));
        const u32 t2 = S0 + maj3(a, b, c);
        h = g; g = f; f = e; e = d + t1;
        d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d;
    st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

__device__ __forceinline__ void sha256_iv(u32 (&st)[8]) {
    st[0] = 0x6a09e667; st[1] = 0xbb67ae85; st[2] = 0x3c6ef372; st[3] = 0xa54ff53a;
    st[4] = 0x510e527f; st[5] = 0x9b05688c; st[6] = 0x1f83d9ab; st[7] = 0x5be0cd19;
}

// Final partial block: rem (<64) data bytes, 0x80, zeros; exact reads only (no
// byte past the string is touched).
__device__ __forceinline__ void load_tail_block(const u8* p, u32 rem, u32 (&w)[16]) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        u32 v = 0;
        const u32 o = 4u * k;
        if (o + 4 <= rem) {
            v = __builtin_bswap32(*(const u32_unaligned*)(p + o));
        } else if (o <= rem) {
            const u32 r = rem - o;                     // 0..3 data bytes in this word
            if (r > 0) v |= (u32)p[o] << 24;
            if (r > 1) v |= (u32)p[o + 1] << 16;
            if (r > 2) v |= (u32)p[o + 2] << 8;
            v |= 0x80000000u >> (8 * r);
        }
        w[k] = v;
    }
}

__global__ __launch_bounds__(kShaWG)
void sha256_items_kernel(const u8* __restrict__ base, const u64* __restrict__ off,
                         const u64* __restrict__ len, const u32* __restrict__ order, u32 n,
                         u32* __restrict__ heads, u8* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int q0 = blockIdx.x % kShaQueues;
    // items of queue q: order[q], order[q + Q], ...  (order is longest-first)
    const u8* ptr = nullptr;
    u64 rem = 0, total = 0;
    u32 item = 0;
    u32 st[8];
    bool active = false, exhausted = false, pad_block = false;

    for (;;) {
        // ---- refill: lanes without a string dequeue one (wave-aggregated) ----------
        if (__ballot(!active && !exhausted)) {
            for (int t = 0; t < kShaQueues; ++t) {
                const bool need = !active && !exhausted;
                const u64 m = __ballot(need);
                if (!m) break;
                const int q = (q0 + t) % kShaQueues;
                const int leader = __ffsll((unsigned long long)m) - 1;
                u32 first = 0;
                if (lane == leader) first = atomicAdd(&heads[q], (u32)__popcll(m));
                first = __shfl(first, leader);
                const u32 mine = first + (u32)__popcll(m & ((1ull << lane) - 1ull));
                const u64 idx = (u64)mine * kShaQueues + (u32)q;
                if (need && idx < n) {
                    item = order ? order[idx] : (u32)idx;
                    ptr = base + off[item];
                    total = rem = len[item];
                    sha256_iv(st);
                    pad_block = false;
                    active = true;
                }
            }
            if (!active) exhausted = true;
        }
        if (!__ballot(active)) break;

        if (active) {
            u32 w[16];
            bool last = false;
            if (rem >= 64) {
                const u32x4 v0 = *(const u32x4_unaligned*)(ptr);
                const u32x4 v1 = *(const u32x4_unaligned*)(ptr + 16);
                const u32x4 v2 = *(const u32x4_unaligned*)(ptr + 32);
                const u32x4 v3 = *(const u32x4_unaligned*)(ptr + 48);
                w[0] = __builtin_bswap32(v0.x); w[1] = __builtin_bswap32(v0.y);
                w[2] = __builtin_bswap32(v0.z); w[3] = __builtin_bswap32(v0.w);
                w[4] = __builtin_bswap32(v1.x); w[5] = __builtin_bswap32(v1.y);
                w[6] = __builtin_bswap32(v1.z); w[7] = __builtin_bswap32(v1.w);
                w[8] = __builtin_bswap32(v2.x); w[9] = __builtin_bswap32(v2.y);
                w[10] = __builtin_bswap32(v2.z); w[11] = __builtin_bswap32(v2.w);
                w[12] = __builtin_bswap32(v3.x); w[13] = __builtin_bswap32(v3.y);
                w[14] = __builtin_bswap32(v3.z); w[15] = __builtin_bswap32(v3.w);
                ptr += 64;
                rem -= 64;
            } else {
                const u64 bits = total * 8;
                if (!pad_block) {
                    load_tail_block(ptr, (u32)rem, w);
                    if (rem <= 55) {
                        w[14] = (u32)(bits >> 32); w[15] = (u32)bits;
                        last = true;
                    } else {
                        pad_block = true;             // length goes into one more block
                    }
                    rem = 0;
                } else {
#pragma unroll
                    for (int k = 0; k < 14; ++k) w[k] = 0;
                    w[14] = (u32)(bits >> 32); w[15] = (u32)bits;
                    last = true;
                }
            }
            sha256_compress(st, w);
            if (last) {
                u32x4* o = (u32x4*)(out + 32ull * item);
                u32x4 d0, d1;
                d0.x = __builtin_bswap32(st[0]); d0.y = __builtin_bswap32(st[1]);
                d0.z = __builtin_bswap32(st[2]); d0.w = __builtin_bswap32(st[3]);
                d1.x = __builtin_bswap32(st[4]); d1.y = __builtin_bswap32(st[5]);
                d1.z = __builtin_bswap32(st[6]); d1.w = __builtin_bswap32(st[7]);
                o[0] = d0; o[1] = d1;
                active = false;
            }
        }
    }
}

void launch_sha256_items(const u8* d_base, const u64* d_off, const u64* d_len, const u32* d_order,
                         u32 n, u32* d_heads, u8* d_out, int blocks_per_cu, int n_cu,
                         hipStream_t s) {
    if (n == 0) return;
    (void)hipMemsetAsync(d_heads, 0, sizeof(u32) * kShaQueues, s);
    u64 want = ((u64)n + kShaWG - 1) / kShaWG;
    u64 cap = (u64)blocks_per_cu * (u64)n_cu;
    u32 grid = (u32)(want < cap ? want : cap);
    // keep the grid a multiple of the queue count so every queue has the same number of pullers
    if (grid >= (u32)kShaQueues) grid -= grid % kShaQueues;
    if (grid == 0) grid = 1;
    hipLaunchKernelGGL(sha256_items_kernel, dim3(grid), dim3(kShaWG), 0, s, d_base, d_off, d_len,
                       d_order, n, d_heads, d_out);
}

}  // namespace mi
