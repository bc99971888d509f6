// sha256.hip -- batched SHA-256 on gfx950: one lane per independent byte string.
//
// What it replaces: the per-stream crypto/sha256 hashers of the reference
// (lib/builder/step/common.go:44-45; lib/docker/image/digester.go:33-60), applied
// per chunk / per file instead of per layer.  A single SHA-256 stream is serial
// (Merkle-Damgard), so the parallelism is ACROSS strings: every lane owns one
// string, keeps its 8-word state and 16-word schedule in VGPRs and runs the 64
// rounds with v_alignbit_b32 (rotates), v_bitop3_b32 (xor3 / Ch / Maj in one op) and v_add3_u32.
//
// Roofline: this kernel is VALU-integer bound, not HBM bound: 1 460 VALU instructions per
// 64-byte block (1 399 of them the compression), and every one of them costs an issue slot of
// its SIMD whatever its "rate" -- DESIGN.md 4.2, profiles/r03_ubench_mix.txt.  HBM traffic is
// 1 byte read per byte hashed + 32 bytes written per string.
//
// Scheduling: strings have very different lengths (2 KiB..64 KiB chunks), so lanes
// pull work from global queues holding the items longest-first (LPT): a lane that
// finishes its string takes the next one instead of idling until the slowest lane of
// its wave is done.  Dequeues are aggregated per wave (one atomic for all lanes that
// finished in the same iteration).  Two waves share every SIMD and the issue arbiter
// prefers the older one (3.5 against 12 us per block): the wave that ARRIVES FIRST on a
// SIMD takes the longest quarter of the strings, the other(s) the rest (two ranges of
// kShaQueues queues; whoever runs dry continues in the other range).
#include "mi_common.h"

#include <stdio.h>
#include <stdlib.h>
#include <vector>

namespace mi {

typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_unaligned __attribute__((aligned(1)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));

// kCoop: how a lane's next 64-byte block comes in.
//   false  four byte-aligned 16-byte loads by the lane itself (64 lanes -> 64 pages per instruction,
//      and the texture addresser splits every unaligned dwordx4);
//   true   QUAD-COOPERATIVE: the four lanes of a quad fetch ONE owner's block with one instruction (lane
//      l&3 takes 16-byte piece l&3: 64 contiguous, dword-aligned bytes per quad), four instructions
//      for the quad's four owners; the pieces are transposed back to their owners through LDS
//      (ds_write_b128 / ds_read_b128: no VALU work).  The block window is [q + 4 + 64 j, +64) with
//      q = start & ~3 plus one carried dword in front, and the byte realignment rides on the
//      big-endian v_perm_b32 every word needs anyway.
// Cooperative loads cost ~2 % more VALU work and save TLB misses: they lose 2 % on a 6.5 GB arena
// and win 14 % on a 32 GB one (profiles/r02_sha_utcl1_and_load_alignment.txt), so the launcher
// picks by the footprint of the strings.
__device__ __forceinline__ u32 be_word(u32 hi, u32 lo, u32 sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
template <int kM>
__device__ __forceinline__ u32 quad_bcast(u32 v) {            // value of lane (lane & ~3) + kM
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, kM * 0x55, 0xF, 0xF, true);
}
constexpr int kXRow = 20;                                      // LDS row: 64 B of block + 16 B pad (dwords)

__device__ __forceinline__ u32 rotr(u32 x, u32 n) { return __builtin_amdgcn_alignbit(x, x, n); }
// gfx950 v_bitop3_b32: any 3-input boolean in ONE VALU op (truth table: a=0xF0, b=0xCC, c=0xAA)
__device__ __forceinline__ u32 xor3(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ u32 ch3(u32 e, u32 f, u32 g)  { return __builtin_amdgcn_bitop3_b32(e, f, g, 0xCA); }
__device__ __forceinline__ u32 maj3(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }

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

// Turns the big-endian words of a 64-byte load into the final partial block of a
// string that has r (<64) bytes left: data bytes, 0x80, zeros.  Pure register work.
__device__ __forceinline__ void mask_tail_block(u32 (&w)[16], u32 r) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const u32 o = 4u * k;
        u32 v = w[k];
        if (r < o + 4) {
            if (r <= o) {
                v = (r == o) ? 0x80000000u : 0u;
            } else {
                const u32 nb = r - o;                  // 1..3 data bytes stay
                v = (v & (0xFFFFFFFFu << (32 - 8 * nb))) | (0x80000000u >> (8 * nb));
            }
        }
        w[k] = v;
    }
}


// the 16 big-endian words of the block in nx* (lane-owned loads: the block starts at nx0.x)
__device__ __forceinline__ void block_words_lane(u32 (&w)[16], const u32x4& nx0, const u32x4& nx1,
                                                 const u32x4& nx2, const u32x4& nx3) {
    w[0] = __builtin_bswap32(nx0.x); w[1] = __builtin_bswap32(nx0.y);
    w[2] = __builtin_bswap32(nx0.z); w[3] = __builtin_bswap32(nx0.w);
    w[4] = __builtin_bswap32(nx1.x); w[5] = __builtin_bswap32(nx1.y);
    w[6] = __builtin_bswap32(nx1.z); w[7] = __builtin_bswap32(nx1.w);
    w[8] = __builtin_bswap32(nx2.x); w[9] = __builtin_bswap32(nx2.y);
    w[10] = __builtin_bswap32(nx2.z); w[11] = __builtin_bswap32(nx2.w);
    w[12] = __builtin_bswap32(nx3.x); w[13] = __builtin_bswap32(nx3.y);
    w[14] = __builtin_bswap32(nx3.z); w[15] = __builtin_bswap32(nx3.w);
}
// cooperative loads: the window is dword-aligned, `carry` is the dword in front of it and `sel` realigns
__device__ __forceinline__ void block_words_coop(u32 (&w)[16], const u32x4& nx0, const u32x4& nx1,
                                                 const u32x4& nx2, const u32x4& nx3, u32& carry, u32 sel) {
    w[0] = be_word(nx0.x, carry, sel); w[1] = be_word(nx0.y, nx0.x, sel);
    w[2] = be_word(nx0.z, nx0.y, sel); w[3] = be_word(nx0.w, nx0.z, sel);
    w[4] = be_word(nx1.x, nx0.w, sel); w[5] = be_word(nx1.y, nx1.x, sel);
    w[6] = be_word(nx1.z, nx1.y, sel); w[7] = be_word(nx1.w, nx1.z, sel);
    w[8] = be_word(nx2.x, nx1.w, sel); w[9] = be_word(nx2.y, nx2.x, sel);
    w[10] = be_word(nx2.z, nx2.y, sel); w[11] = be_word(nx2.w, nx2.z, sel);
    w[12] = be_word(nx3.x, nx2.w, sel); w[13] = be_word(nx3.y, nx3.x, sel);
    w[14] = be_word(nx3.z, nx3.y, sel); w[15] = be_word(nx3.w, nx3.z, sel);
    carry = nx3.w;
}
// every quad fetches the next blocks of its owners that want one (kAll: all four do), owner m's 64
// bytes with ONE instruction; piece `sub` of owner qbase + m lands in g_m
template <bool kAll>
__device__ __forceinline__ void coop_fetch(u32x4& g0, u32x4& g1, u32x4& g2, u32x4& g3, const u8* ptr, bool want, int sub) {
    const u32 wf = want ? 1u : 0u;
    const u32 plo = (u32)(size_t)ptr, phi = (u32)((size_t)ptr >> 32);
    const u32 l0 = quad_bcast<0>(plo), l1 = quad_bcast<1>(plo), l2 = quad_bcast<2>(plo), l3 = quad_bcast<3>(plo);
    const u32 h0 = quad_bcast<0>(phi), h1 = quad_bcast<1>(phi), h2 = quad_bcast<2>(phi), h3 = quad_bcast<3>(phi);
    const u64 mine = 16u * (u32)sub;                 // my 16-byte piece of every owner's block
    if (kAll || quad_bcast<0>(wf)) g0 = *(const u32x4_a4*)(size_t)((((u64)h0 << 32) | l0) + mine);
    if (kAll || quad_bcast<1>(wf)) g1 = *(const u32x4_a4*)(size_t)((((u64)h1 << 32) | l1) + mine);
    if (kAll || quad_bcast<2>(wf)) g2 = *(const u32x4_a4*)(size_t)((((u64)h2 << 32) | l2) + mine);
    if (kAll || quad_bcast<3>(wf)) g3 = *(const u32x4_a4*)(size_t)((((u64)h3 << 32) | l3) + mine);
}

// Lane pipeline (one iteration = one 64-byte compression per lane):
//   cur  : the string being hashed: ptr/rem/total/slot, state st[8], and nx* = its NEXT
//          64 bytes, loaded one iteration ahead so HBM latency hides under 64 rounds;
//   next : the lane's next string, acquired kLook iterations before cur ends through a
//          4-stage pipeline  (1) wave-aggregated atomic dequeue, (1b) resolve it to a queue
//          position, (2) load off/len/slot of that position, (3) load its first 64 bytes.
//          Each stage is consumed one iteration after it was issued, so a lane switches
//          strings without ever waiting on memory.
// Reads may run up to 63 bytes past a string's end (67 with kCoop, plus 3 bytes in front of a
// string that does not start on a dword) -- never used; every buffer this is launched on carries
// that slack (arena: 4 KiB; digest arrays and mi_sha256_many staging: DevBuf adds 256 bytes) and no
// string starts unaligned at a buffer's first byte.
constexpr u32 kLook = 5;

// kPass only names the instantiation (chunk pass / root pass / ...) so profiles tell them apart.
// kStats: the per-wave record of MI_SHA_WAVE_STATS (a second instantiation: its two counters cost two
// instructions per iteration).
template <int kPass, bool kCoop, bool kStats = false>
__global__ __launch_bounds__(kShaWG)
void sha256_items_kernel(const u8* __restrict__ base, const u64* __restrict__ off,
                         const u64* __restrict__ len, const u32* __restrict__ ids, u32 n_max,
                         const u64* __restrict__ n_ptr, u32* __restrict__ heads,
                         u32* __restrict__ roles, u32 long_shift,
                         u8* __restrict__ out, u32* __restrict__ wave_stats) {
    // the string count may only be known on the device (no host sync between pipeline stages)
    const u32 n = n_ptr ? (u32)*n_ptr : n_max;
    // diagnostics (MI_SHA_WAVE_STATS, see the launcher): where this wave ran, when, and how much it hashed
    const u64 ws_t0 = kStats ? wall_clock64() : 0;
    u32 ws_iters = 0, ws_lane_blocks = 0, ws_full = 0, ws_wait = 0;
    const int lane = threadIdx.x & 63;
    const int q0 = blockIdx.x % kShaQueues;
    // ---- the wave's role on its SIMD ---------------------------------------------------------------
    // Two (or three) waves of this grid share every SIMD, and one string is a serial chain: a 64 KiB chunk
    // is 1025 compressions, 5.4 ms at the pace two equal waves leave each other but 3.6 ms for a wave that
    // is preferred by the issue arbiter over its neighbour -- and the whole launch has 4.2 ms.  So the strings are split
    // at position L (the array is sorted longest-first): the FIRST wave to arrive on a SIMD -- the one the issue
    // arbiter prefers from then on, being the older -- takes the long ones [0, L), the others take [L, n) with
    // what the first leaves them (~1/3 of its pace); whoever runs dry continues in the other range.  Roles come from an atomic per
    // SIMD (key = XCC | SE SH CU | SIMD of HW_ID), not from the dequeue order: with priorities by dequeue
    // rank two long-string waves could land on one SIMD, halve each other's pace and finish ~1 ms after
    // everybody else (profiles/r03_sha_wave_stats.txt: the 5.2 ms launches).
    u32 role = 0;
    if (roles) {
        const u32 hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));        // HW_REG_HW_ID
        const u32 xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11)) & 7u; // HW_REG_XCC_ID
        const u32 key = (xcc << 10) | (((hw >> 8) & 0xFFu) << 2) | ((hw >> 4) & 3u);
        u32 r = 0;
        if (lane == 0) r = atomicAdd(roles + key, 1u);
        role = (u32)__builtin_amdgcn_readfirstlane((int)r);
        // s_setprio only on request (ShaTune.prio): age alone gives the same 3.5 / 12 us per iteration, and raised
        // priorities starve the other batch's passes when two are in flight
        if (long_shift & 0x100u)  __builtin_amdgcn_s_setprio(0);
        else if (role == 0)       __builtin_amdgcn_s_setprio(3);
        else                      __builtin_amdgcn_s_setprio(1);
    }
    const u32 long_n = roles ? ((n >> (long_shift & 31u)) & ~(u32)(kShaQueues - 1)) : 0u;   // L, a multiple of the queue count
    const int my_set = role == 0 ? 0 : 1;            // range A = [0, L) | range B = [L, n), kShaQueues queues each
    const int qtry_end = (role == 0 || long_n) ? 2 * kShaQueues : kShaQueues;
    // current string
    const u8* ptr = nullptr;
    u64 rem = 0, total = 0;
    u32 slot = 0;
    u32 st[8];
    u32x4 nx0, nx1, nx2, nx3;
    // kCoop only:
    __shared__ __attribute__((aligned(16))) u32 xpose[kCoop ? kShaWG / 64 : 1][kCoop ? 64 * kXRow : 4];
    u32* xw = xpose[kCoop ? threadIdx.x >> 6 : 0];
    const int sub = lane & 3, qbase = lane & ~3;
    u32x4 g0 = {0, 0, 0, 0}, g1 = g0, g2 = g0, g3 = g0;   // piece `sub` of the blocks of owners qbase + 0..3
    bool loaded = false;                                   // my next block is among the pieces in flight
    u32 carry = 0, sel = 0x00010203u;   // dword in front of nx*, byte selector of the string's misalignment
    u32 fc = 0;
    bool active = false, pad_block = false;
    // next string
    enum : u32 { kNone = 0, kReq = 1, kPos = 2, kDesc = 3, kReady = 4, kDry = 5 };
    u32 nstate = kNone, npos = 0, nslot = 0, areq = 0, req_rank = 0;
    int req_q = 0, req_set = 0, req_leader = 0;     // wave-uniform: the dequeue in flight
    u64 noff = 0, nlen = 0;
    u32x4 f0, f1, f2, f3;
    // queues already found empty by this wave (uniform): my range's first, then the other range's
    int qtry = (role == 0 && long_n == 0) ? kShaQueues : 0;

    for (;;) {
        // (A lean inner loop for iterations in which all 64 lanes are mid-string -- 89 % of them on C2 -- was
        // tried in round 3: ~75 instructions fewer per iteration, 175 instead of 144 VGPRs, and the same
        // 4.5 ms: the general iteration below already averages 1 460 VALU instructions per block against the
        // compression's 1 399 + 16 byte swaps, profiles/r03_sq_counters.txt.)
        // Everything issued in the previous iteration (first-block loads, descriptors, the
        // dequeue atomic, the nx prefetch) has had a whole compression to land.  Consume it
        // all HERE, before this iteration issues anything new, so no wait below can stall on
        // a freshly issued request (hipcc's s_waitcnt for a divergent region is vmcnt(0)).
        u64 ws_w0 = 0;
        if constexpr (kStats) ws_w0 = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // incl. the hand-issued dequeue atomic
        if constexpr (kStats) ws_wait += (u32)(wall_clock64() - ws_w0);
        asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(nx0), "+v"(nx1),
                          "+v"(nx2), "+v"(nx3));
        asm volatile("" : "+v"(noff), "+v"(nlen), "+v"(nslot), "+v"(areq));
        if constexpr (kCoop) asm volatile("" : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3), "+v"(fc));
        if (kCoop && __ballot(loaded)) {
            // pieces -> owners: piece `sub` of owner qbase + m lies in g_m
            *(u32x4*)&xw[(qbase + 0) * kXRow + 4 * sub] = g0;
            *(u32x4*)&xw[(qbase + 1) * kXRow + 4 * sub] = g1;
            *(u32x4*)&xw[(qbase + 2) * kXRow + 4 * sub] = g2;
            *(u32x4*)&xw[(qbase + 3) * kXRow + 4 * sub] = g3;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (loaded) {
                nx0 = *(const u32x4*)&xw[lane * kXRow];
                nx1 = *(const u32x4*)&xw[lane * kXRow + 4];
                nx2 = *(const u32x4*)&xw[lane * kXRow + 8];
                nx3 = *(const u32x4*)&xw[lane * kXRow + 12];
                loaded = false;
            }
            __builtin_amdgcn_wave_barrier();               // the rows are rewritten next iteration
        }
        // ---- stage 1b: resolve last iteration's dequeue --------------------------------
        if (__ballot(nstate == kReq)) {
            const u32 first = __shfl(areq, req_leader);
            bool missed = false;
            if (nstate == kReq) {
                const u64 pos = (u64)(first + req_rank) * kShaQueues + (u32)req_q + (req_set ? long_n : 0u);
                if (pos < (req_set ? n : long_n)) { npos = (u32)pos; nstate = kPos; }
                else { nstate = kNone; missed = true; }
            }
            if (__ballot(missed)) ++qtry;              // a position past the end: that queue is dry
        }
        // ---- switch to the next string (its first block arrived an iteration ago) -------
        if (!active && nstate == kReady) {
            if constexpr (kCoop) {
                const u8* p = base + noff;
                const u32 mis = (u32)(size_t)p & 3u;
                ptr = p - mis + 4;                     // the aligned window behind the carried dword
                sel = 0x00010203u + mis * 0x01010101u;
                carry = fc;
            } else {
                ptr = base + noff;
            }
            total = rem = nlen;
            slot = nslot;
            nx0 = f0; nx1 = f1; nx2 = f2; nx3 = f3;
            sha256_iv(st);
            pad_block = false;
            active = true;
            nstate = kNone;
        }
        // ---- stage 3: descriptor known -> fetch the first block -----------------------
        if (nstate == kDesc) {
            const u8* p = base + noff;
            if (nlen) {
              if constexpr (kCoop) {
                // a string's FIRST block is fetched by its own lane (once per string)
                const u8* q = p - ((size_t)p & 3u);
                // (the dword in front comes through a laundered pointer: seeing q and q + 4 together,
                // hipcc merges the five loads into overlapping ones and shuffles registers right
                // behind them -- a full memory wait inside the iteration)
                typedef __attribute__((address_space(1))) const u32 glob_cu32;   // stays a global_load
                glob_cu32* qc = (glob_cu32*)q;
                asm volatile("" : "+v"(qc));
                fc = *qc;
                f0 = *(const u32x4_a4*)(q + 4);
                f1 = *(const u32x4_a4*)(q + 20);
                f2 = *(const u32x4_a4*)(q + 36);
                f3 = *(const u32x4_a4*)(q + 52);
              } else {
                f0 = *(const u32x4_unaligned*)(p);
                f1 = *(const u32x4_unaligned*)(p + 16);
                f2 = *(const u32x4_unaligned*)(p + 32);
                f3 = *(const u32x4_unaligned*)(p + 48);
              }
            }
            nstate = kReady;
        }
        // ---- stage 2: position known -> load its descriptor ---------------------------
        if (nstate == kPos) {
            noff = off[npos];
            nlen = len[npos];
            nslot = ids ? ids[npos] : npos;
            nstate = kDesc;
        }
        // ---- stage 1: reserve a position for lanes about to run dry -------------------
        // Only ISSUES the wave-aggregated atomic; its result is resolved at the top of the
        // next iteration (stage 1b), so the wave never waits for the atomic round trip.
        {
            const bool want = nstate == kNone && (!active || rem < 64ull * kLook);
            const u64 m = __ballot(want);
            u64 leader_mask = 0;                           // wave-uniform: the lane that performs the atomic
            if (m) {
                if (qtry >= qtry_end) {
                    if (want) nstate = kDry;
                } else {
                    req_q = (q0 + qtry) % kShaQueues;
                    req_set = qtry < kShaQueues ? my_set : 1 - my_set;
                    req_leader = __ffsll((unsigned long long)m) - 1;
                    leader_mask = 1ull << req_leader;
                    if (want) {
                        req_rank = (u32)__popcll(m & ((1ull << lane) - 1ull));
                        nstate = kReq;
                    }
                }
            }
            // The atomic itself, in straight-line code and by hand.  Written as
            // `if (lane == leader) areq = atomicAdd(...)`, the merge of areq's two values behind the
            // branch is a v_mov that needs the RESULT: hipcc put an s_waitcnt vmcnt(0) there -- a
            // memory round trip, and a wait for every load this iteration had issued, in each
            // iteration that dequeues.  EXEC is the leader lane alone (or empty: no request); the
            // compiler does not know this load, the wait for it is the explicit one at the loop top.
            {
                u64 saved;
                const u32 cnt = (u32)__popcll(m);
                const u32* head = heads + req_set * kShaQueues + req_q;
                asm volatile("s_mov_b64 %[sv], exec\n\t"
                             "s_mov_b64 exec, %[mk]\n\t"
                             "global_atomic_add %[ret], %[hd], %[val], off sc0\n\t"
                             "s_mov_b64 exec, %[sv]"
                             : [ret] "+v"(areq), [sv] "=&s"(saved)
                             : [mk] "s"(leader_mask), [val] "v"(cnt), [hd] "v"(head)
                             : "memory");
            }
        }
        if (!__ballot(active || nstate != kDry)) break;
        if constexpr (kStats) {
            ++ws_iters;
            ws_lane_blocks += (u32)__popcll(__ballot(active));
            // iterations in which every lane is mid-string: nothing pending, no request due
            if (!__ballot(!(active && nstate == kNone && rem >= 64ull * kLook))) ++ws_full;
        }

        u32 w[16];
        bool last = false;
        bool want = false;                                 // kCoop: my string has another block to fetch
        if (active) {
            if (!pad_block) {
              if constexpr (kCoop) block_words_coop(w, nx0, nx1, nx2, nx3, carry, sel);
              else                 block_words_lane(w, nx0, nx1, nx2, nx3);
                if (rem >= 64) {
                    ptr += 64;
                    rem -= 64;
                    if (rem) {                         // in flight during this block's 64 rounds
                        if constexpr (kCoop) {
                            want = true;
                        } else {
                            nx0 = *(const u32x4_unaligned*)(ptr);
                            nx1 = *(const u32x4_unaligned*)(ptr + 16);
                            nx2 = *(const u32x4_unaligned*)(ptr + 32);
                            nx3 = *(const u32x4_unaligned*)(ptr + 48);
                        }
                    }
                } else {
                    mask_tail_block(w, (u32)rem);
                    if (rem <= 55) {
                        const u64 bits = total * 8;
                        w[14] = (u32)(bits >> 32); w[15] = (u32)bits;
                        last = true;
                    } else {
                        pad_block = true;             // the length needs one more block
                    }
                    rem = 0;
                }
            } else {
                const u64 bits = total * 8;
#pragma unroll
                for (int k = 0; k < 14; ++k) w[k] = 0;
                w[14] = (u32)(bits >> 32); w[15] = (u32)bits;
                last = true;
            }
        }
        // kCoop.  The whole wave is here (no lane leaves the loop alone): every quad fetches the next blocks
        // of its owners that want one, owner m's 64 bytes with ONE instruction.
        if (kCoop && __ballot(want)) {
            coop_fetch<false>(g0, g1, g2, g3, ptr, want, sub);
            loaded = want;
        }
        if (active) {
            sha256_compress(st, w);
            if (last) {
                u32x4* o = (u32x4*)(out + 32ull * slot);
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
    if (kStats && wave_stats && lane == 0) {
        u32* d = wave_stats + 8u * (blockIdx.x * (kShaWG / 64) + (threadIdx.x >> 6));
        const u64 t1 = wall_clock64();
        d[0] = __builtin_amdgcn_s_getreg(4 | (31 << 11));        // HW_REG_HW_ID: wave 3:0, SIMD 5:4, CU 11:8, SH 12, SE 15:13
        d[1] = (__builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xFu) | (role << 8);   // HW_REG_XCC_ID | role on the SIMD
        d[2] = (u32)ws_t0; d[3] = ws_wait;                        // start (low word) | ticks spent in the loop-top s_waitcnt
        d[4] = (u32)(t1 - ws_t0);                                 // 100 MHz ticks
        d[5] = ws_iters;
        d[6] = ws_lane_blocks;
        d[7] = ws_full;
    }
}

// Diagnostics (mi_debug_sha_wave_stats / MI_SHA_WAVE_STATS=<file> at ctx creation): the chunk pass with the per-wave
// record -- appended to `path` per launch: {grid, waves per workgroup, coop, n}, then 8 words per wave: HW_ID, XCC_ID |
// role << 8, start on the 100 MHz wall clock (low word), ticks in the loop-top s_waitcnt, duration, loop iterations,
// lane-blocks hashed, iterations with all 64 lanes mid-string (tools/sha_wave_stats.py reads it).  The launch is followed
// by a stream synchronize and a read-back: a ctx that records is not a ctx to time anything else on.
static void launch_sha256_chunks_recorded(bool coop, u32 grid, size_t lds_pad, const u8* d_base, const u64* d_off,
                                          const u64* d_len, const u32* d_order, u32 n, const u64* d_n, u32* d_heads,
                                          u32* d_roles, u32 shift_flags, u8* d_out, const char* path, hipStream_t s) {
    u32* d_stats = nullptr;
    const size_t stats_words = (size_t)grid * (kShaWG / 64) * 8;
    if (hipMalloc((void**)&d_stats, stats_words * sizeof(u32)) != hipSuccess) d_stats = nullptr;
    else (void)hipMemsetAsync(d_stats, 0, stats_words * sizeof(u32), s);
    if (coop)
        hipLaunchKernelGGL((sha256_items_kernel<kShaChunks, true, true>), dim3(grid), dim3(kShaWG), lds_pad, s, d_base, d_off,
                           d_len, d_order, n, d_n, d_heads, d_roles, shift_flags, d_out, d_stats);
    else
        hipLaunchKernelGGL((sha256_items_kernel<kShaChunks, false, true>), dim3(grid), dim3(kShaWG), lds_pad, s, d_base, d_off,
                           d_len, d_order, n, d_n, d_heads, d_roles, shift_flags, d_out, d_stats);
    if (!d_stats) return;
    std::vector<u32> h(stats_words);
    if (hipStreamSynchronize(s) == hipSuccess &&
        hipMemcpy(h.data(), d_stats, stats_words * sizeof(u32), hipMemcpyDeviceToHost) == hipSuccess) {
        if (FILE* f = fopen(path, "ab")) {
            const u32 hdr[4] = {grid, (u32)(kShaWG / 64), coop ? 1u : 0u, n};
            fwrite(hdr, sizeof(u32), 4, f);
            fwrite(h.data(), sizeof(u32), stats_words, f);
            fclose(f);
        }
    }
    (void)hipFree(d_stats);
}

// With the TLB out of the way (cooperative loads) a third workgroup per CU pays (161 VGPRs: three
// waves per SIMD fit): 26 GB arena 1.43 (byte loads, 2/CU) -> 1.57 (cooperative, 2/CU) -> 1.63 TB/s
// (cooperative, 3/CU); on 6.5 GB three are slower with either scheme (coarser tail).
void launch_sha256_items(ShaPass pass, const u8* d_base, const u64* d_off, const u64* d_len,
                         const u32* d_order, u32 n, const u64* d_n, u32* d_heads, u32* d_roles, bool zero_heads,
                         u8* d_out, const ShaTune& tune, int n_cu, u64 footprint_bytes, hipStream_t s) {
    if (n == 0) return;
    if (!tune.roles) d_roles = nullptr;
    if (zero_heads) {
        (void)hipMemsetAsync(d_heads, 0, sizeof(u32) * kShaHeadWords, s);
        if (d_roles) (void)hipMemsetAsync(d_roles, 0, sizeof(u32) * kShaRoleWords, s);
    }
    const bool coop = pass != kShaRoots && footprint_bytes >= tune.coop_min_bytes;
    int blocks_per_cu = tune.blocks_per_cu;
    if (coop) blocks_per_cu = tune.coop_blocks_per_cu ? tune.coop_blocks_per_cu
                            : footprint_bytes >= (24ull << 30) ? 3 : blocks_per_cu;   // enough work per lane for a third
    // The grid is blocks_per_cu x n_cu persistent workgroups -- but the dispatcher places by free
    // resources, and at 136 VGPRs a CU has room for THREE: some CUs take three workgroups and others one,
    // the waves of a crowded CU run at two thirds of the pace, and the launch ends with them (measured on
    // one box, same clocks: 4.13 ... 5.48 ms from launch to launch).  An LDS request of just over
    // 160 KiB / (blocks_per_cu + 1) per workgroup -- unused memory -- makes the intended placement the
    // only possible one.
    size_t lds_pad = 0;
    // (Only for the lane-owned scheme: the cooperative kernel's 161 VGPRs already cap a CU at three
    // workgroups, and pinned to two it ran SLOWER on a 6.5 GB arena -- 5.4 against 4.5 ms,
    // profiles/r03_sha_placement.txt -- so it is left to the dispatcher.)
    if (tune.pin_blocks_per_cu && !coop && blocks_per_cu >= 1 && blocks_per_cu <= 3) {
        static thread_local int attr_dev = -1;
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (attr_dev != dev) {
#define MI_SHA_ATTR(P, C) (void)hipFuncSetAttribute((const void*)sha256_items_kernel<P, C>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)
            MI_SHA_ATTR(kShaChunks, false); MI_SHA_ATTR(kShaChunks, true); MI_SHA_ATTR(kShaRoots, false);
            (void)hipFuncSetAttribute((const void*)sha256_items_kernel<kShaChunks, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            MI_SHA_ATTR(kShaFiles, false); MI_SHA_ATTR(kShaFiles, true); MI_SHA_ATTR(kShaBlobs, false); MI_SHA_ATTR(kShaBlobs, true);
#undef MI_SHA_ATTR
            attr_dev = dev;
        }
        const size_t per_wg = (160u * 1024u) / (size_t)(blocks_per_cu + 1) + 1024u;
        const size_t fixed = coop ? (size_t)(kShaWG / 64) * 64 * kXRow * 4 : 16;     // the kernel's own static LDS
        lds_pad = per_wg > fixed ? per_wg - fixed : 0;
    }
    u64 want = ((u64)n + kShaWG - 1) / kShaWG;
    u64 cap = (u64)blocks_per_cu * (u64)n_cu;
    u32 grid = (u32)(want < cap ? want : cap);
    // keep the grid a multiple of the queue count so every queue has the same number of pullers
    if (grid >= (u32)kShaQueues) grid -= grid % kShaQueues;
    if (grid == 0) grid = 1;
#define MI_SHA_LAUNCH(P, C)                                                                   \
    hipLaunchKernelGGL((sha256_items_kernel<P, C>), dim3(grid), dim3(kShaWG), lds_pad, s, d_base, d_off, \
                       d_len, d_order, n, d_n, d_heads, d_roles, (u32)tune.long_shift | (tune.prio ? 0u : 0x100u), d_out, nullptr)
    switch (pass) {
        case kShaChunks:
            if (tune.wave_stats_path) {      // diagnostics, per ctx (mi_debug_sha_wave_stats): the recording instantiation
                launch_sha256_chunks_recorded(coop, grid, lds_pad, d_base, d_off, d_len, d_order, n, d_n, d_heads, d_roles,
                                              (u32)tune.long_shift | (tune.prio ? 0u : 0x100u), d_out, tune.wave_stats_path, s);
                break;
            }
            if (coop) MI_SHA_LAUNCH(kShaChunks, true); else MI_SHA_LAUNCH(kShaChunks, false);
            break;
        case kShaRoots:  MI_SHA_LAUNCH(kShaRoots, false); break;
        case kShaFiles:  if (coop) MI_SHA_LAUNCH(kShaFiles, true); else MI_SHA_LAUNCH(kShaFiles, false); break;
        default:         if (coop) MI_SHA_LAUNCH(kShaBlobs, true); else MI_SHA_LAUNCH(kShaBlobs, false); break;
    }
#undef MI_SHA_LAUNCH
}

// ---- the VALU roof of this file's compression, measured on the device it runs on ------------------
// 64 rounds per lane and iteration over register data: no memory traffic, no queues, no tails.  What it
// times is sha256_compress itself -- the same code the item kernels inline -- so the rate it gives is
// the ceiling of ANY one-lane-per-string form of them on this chip at this moment's clocks.
__global__ __launch_bounds__(kShaWG)
void sha256_roof_kernel(u32* __restrict__ out, u32 blocks) {
    u32 st[8], w[16];
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) st[i] = t * 0x9E3779B9u + (u32)i;
    u32 x = t * 0x85EBCA6Bu + 1u;
    for (u32 b = 0; b < blocks; ++b) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { x = x * 1664525u + 1013904223u; w[i] = x ^ st[i & 7]; }
        sha256_compress(st, w);
    }
    u32 r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r ^= st[i];
    out[t] = r;
}

// bytes "hashed" per second by n_cu * waves_per_simd workgroups running `blocks` compressions per lane
double measure_sha_valu_roof(int n_cu, int waves_per_simd, u32 blocks, u32* d_scratch, hipStream_t s,
                             hipEvent_t e0, hipEvent_t e1) {
    const u32 grid = (u32)(n_cu * waves_per_simd);                 // a workgroup = one wave on each of the CU's 4 SIMDs
    hipLaunchKernelGGL(sha256_roof_kernel, dim3(grid), dim3(kShaWG), 0, s, d_scratch, blocks / 8 + 1);   // clocks up
    double best = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, s);
        hipLaunchKernelGGL(sha256_roof_kernel, dim3(grid), dim3(kShaWG), 0, s, d_scratch, blocks);
        (void)hipEventRecord(e1, s);
        if (hipStreamSynchronize(s) != hipSuccess) return 0;
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double rate = ms > 0 ? (double)grid * kShaWG * blocks * 64.0 / (ms * 1e-3) : 0;
        best = rate > best ? rate : best;
    }
    return best;
}

}  // namespace mi
