#!/usr/bin/env python3
"""Generates tools/ubench_sha_runs.hip: the 64-round SHA-256 compression (no memory traffic) as straight-line `asm volatile`
statements, one per VALU instruction, in an order this script chooses -- the compiler allocates registers and cannot reorder.

The question (profiles/r03_ubench_mix.txt): gfx950 issues "2-pass" VALU ops (add, xor, shift, bitop3 on
distinct banks) at 1.0 ns per wave instruction and SIMD and "4-pass" ops (alignbit, add3) at 1.74 -- but a fine mix of the two
runs ALL of them at ~1.75, and only runs of >= 8 of one class recover some of it (1.63).  SHA-256's round is a fine mix.
Does the same arithmetic, ordered into one 4-pass run and one 2-pass run per round, hash faster?

  order "natural": per round the dependency order a compiler would emit (rotates, xor3, ch, maj, adds), schedule beside it
  order "runs":    per round [4-pass: 3 add3 of the previous round's tail + 6 (+4 schedule) alignbit] [2-pass: xor3 x2, ch,
                   maj, K+W, S0+maj (+ two shifts, two xor3, one add of the schedule)] -- 32 more instructions per block
                   (1 440 against 1 407) for runs of 14 and 11.
"""
import sys

K = [0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
     0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
     0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
     0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
     0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
     0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
     0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]


class Gen:
    def __init__(self):
        self.lines = []
        self.n = 0

    def tmp(self):
        self.n += 1
        return "t%d" % self.n

    def op(self, text, dst, srcs, imm=None):
        ins = ", ".join('"v"(%s)' % s for s in srcs)
        if imm is not None:
            ins += (", " if ins else "") + '"n"(%d)' % imm
        self.lines.append('    u32 %s; asm volatile("%s" : "=v"(%s) : %s);' % (dst, text, dst, ins))
        return dst

    def rotr(self, x, n):
        return self.op("v_alignbit_b32 %0, %1, %1, %2", self.tmp(), [x], n)

    def shr(self, x, n):
        return self.op("v_lshrrev_b32 %0, %2, %1", self.tmp(), [x], n)

    def bitop(self, a, b, c, tt):
        return self.op("v_bitop3_b32 %%0, %%1, %%2, %%3 bitop3:0x%x" % tt, self.tmp(), [a, b, c])

    def add(self, a, b):
        return self.op("v_add_u32 %0, %1, %2", self.tmp(), [a, b])

    def addk(self, a, k):
        return self.op("v_add_u32 %0, %2, %1", self.tmp(), [a], k)

    def add3(self, a, b, c):
        return self.op("v_add3_u32 %0, %1, %2, %3", self.tmp(), [a, b, c])


def natural(g, st, w):
    a, b, c, d, e, f, gg, h = st
    w = list(w)
    for i in range(64):
        if i >= 16:
            w15, w2 = w[(i + 1) & 15], w[(i + 14) & 15]
            P = g.bitop(g.rotr(w15, 7), g.rotr(w15, 18), g.shr(w15, 3), 0x96)
            Q = g.bitop(g.rotr(w2, 17), g.rotr(w2, 19), g.shr(w2, 10), 0x96)
            w[i & 15] = g.add(g.add3(w[i & 15], P, Q), w[(i + 9) & 15])
        wi = w[i & 15]
        S1 = g.bitop(g.rotr(e, 6), g.rotr(e, 11), g.rotr(e, 25), 0x96)
        C = g.bitop(e, f, gg, 0xCA)
        kw = g.addk(wi, K[i])
        t = g.add3(h, kw, S1)
        S0 = g.bitop(g.rotr(a, 2), g.rotr(a, 13), g.rotr(a, 22), 0x96)
        M = g.bitop(a, b, c, 0xE8)
        T1 = g.add(t, C)
        ne = g.add(d, T1)
        na = g.add3(T1, S0, M)
        a, b, c, d, e, f, gg, h = na, a, b, c, ne, e, f, gg
    return [a, b, c, d, e, f, gg, h]


def runs(g, st, w):
    a, b, c, d, e, f, gg, h = st
    w = list(w)
    pending = None                                   # the previous round's tail: (t-inputs...) emitted at the head of this A run
    for i in range(64):
        # ---- 4-pass run: the previous round's three add3 were emitted at the end of its iteration (they head this run)
        r1, r2, r3 = g.rotr(e, 6), g.rotr(e, 11), g.rotr(e, 25)
        s1, s2, s3 = g.rotr(a, 2), g.rotr(a, 13), g.rotr(a, 22)
        j = i + 1                                    # the schedule word the NEXT round needs
        sched = 16 <= j < 64
        if sched:
            w15, w2 = w[(j + 1) & 15], w[(j + 14) & 15]
            p1, p2, q1, q2 = g.rotr(w15, 7), g.rotr(w15, 18), g.rotr(w2, 17), g.rotr(w2, 19)
        # ---- 2-pass run
        S1 = g.bitop(r1, r2, r3, 0x96)
        S0 = g.bitop(s1, s2, s3, 0x96)
        C = g.bitop(e, f, gg, 0xCA)
        M = g.bitop(a, b, c, 0xE8)
        kw = g.addk(w[i & 15], K[i])
        uM = g.add(S0, M)
        if sched:
            p3, q3 = g.shr(w15, 3), g.shr(w2, 10)
            P = g.bitop(p1, p2, p3, 0x96)
            Q = g.bitop(q1, q2, q3, 0x96)
            ww = g.add(w[j & 15], w[(j + 9) & 15])
        # ---- 4-pass run (continues into the next round's rotates)
        t = g.add3(h, kw, S1)
        ne = g.add3(t, C, d)
        na = g.add3(t, C, uM)
        if sched:
            w[j & 15] = g.add3(ww, P, Q)
        a, b, c, d, e, f, gg, h = na, a, b, c, ne, e, f, gg
    return [a, b, c, d, e, f, gg, h]


def emit(name, order):
    g = Gen()
    st = ["st[%d]" % i for i in range(8)]
    w = ["w[%d]" % i for i in range(16)]
    out = order(g, st, w)
    body = "\n".join(g.lines)
    tail = "\n".join("    st[%d] += %s;" % (i, out[i]) for i in range(8))
    return "__device__ __forceinline__ void compress_%s(u32 (&st)[8], u32 (&w)[16]) {\n%s\n%s\n}\n" % (name, body, tail), g.n


HEAD = r'''// GENERATED by tools/gen_ubench_sha_runs.py -- do not edit.  See that script for the question this answers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
typedef uint32_t u32;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ u32 rotr(u32 x, u32 n) { return __builtin_amdgcn_alignbit(x, x, n); }
__device__ __forceinline__ u32 xor3(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ u32 ch3(u32 e, u32 f, u32 g)  { return __builtin_amdgcn_bitop3_b32(e, f, g, 0xCA); }
__device__ __forceinline__ u32 maj3(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }
__device__ constexpr u32 kK[64] = {@K@};
// the compiler's own order (tools/ubench_sha.hip "unrolled"): the baseline, and the checker of the two generated orders
__device__ __forceinline__ void compress_compiler(u32 (&st)[8], u32 (&w)[16]) {
    u32 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        u32 wi;
        if (i < 16) wi = w[i];
        else {
            const u32 w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
            wi = w[i & 15] + xor3(rotr(w15, 7), rotr(w15, 18), w15 >> 3) + w[(i + 9) & 15] + xor3(rotr(w2, 17), rotr(w2, 19), w2 >> 10);
            w[i & 15] = wi;
        }
        const u32 t1 = (h + ch3(e, f, g) + (wi + kK[i])) + xor3(rotr(e, 6), rotr(e, 11), rotr(e, 25));
        const u32 t2 = xor3(rotr(a, 2), rotr(a, 13), rotr(a, 22)) + maj3(a, b, c);
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}
'''

TAIL = r'''
template <int MODE>
__global__ __launch_bounds__(256) void sha_loop(u32* out, int blocks) {
    u32 st[8], w[16];
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < 8; ++i) st[i] = t * 0x9E3779B9u + i;
    u32 x = t * 0x85EBCA6Bu + 1;
    for (int b = 0; b < blocks; ++b) {
        for (int i = 0; i < 16; ++i) { x = x * 1664525u + 1013904223u; w[i] = x ^ st[i & 7]; }
        if (MODE == 0) compress_compiler(st, w); else if (MODE == 1) compress_natural(st, w); else compress_runs(st, w);
    }
    u32 r = 0;
    for (int i = 0; i < 8; ++i) r ^= st[i];
    out[t] = r;
}
template <int MODE> static void launch(int grid, u32* out, int blocks) { hipLaunchKernelGGL(sha_loop<MODE>, dim3(grid), dim3(256), 0, 0, out, blocks); }
static void launch_mode(int mode, int grid, u32* out, int blocks) {
    if (mode == 0) launch<0>(grid, out, blocks); else if (mode == 1) launch<1>(grid, out, blocks); else launch<2>(grid, out, blocks);
}

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const size_t n = (size_t)256 * ncu * 8;
    u32 *out, *h0 = (u32*)malloc(n * 4), *h1 = (u32*)malloc(n * 4);
    CHK(hipMalloc(&out, n * 4));
    const char* names[3] = {"compiler", "natural", "runs"};
    // the three orders compute the same function
    for (int mode = 0; mode < 3; ++mode) {
        launch_mode(mode, ncu, out, 5);
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(mode ? h1 : h0, out, (size_t)256 * ncu * 4, hipMemcpyDeviceToHost));
        if (mode && memcmp(h0, h1, (size_t)256 * ncu * 4)) { printf("order %s computes something else\n", names[mode]); return 1; }
    }
    printf("the three orders agree on %d lanes x 5 blocks\n", 256 * ncu);
    for (int r = 0; r < 10; ++r) launch_mode(0, ncu * 8, out, 512);
    CHK(hipDeviceSynchronize());
    printf("order     W   us/wave-block/SIMD   chip TB/s hashed   (best of 3)\n");
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode) for (int W = 1; W <= 8; W = (W < 4 ? W + 1 : W * 2)) {
        const int blocks = 2048;
        float best = 1e30f;
        for (int k = 0; k < 3; ++k) {
            hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
            CHK(hipEventRecord(a, 0));
            launch_mode(mode, ncu * W, out, blocks);
            CHK(hipEventRecord(b, 0));
            CHK(hipDeviceSynchronize());
            float ms; CHK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        const double per_simd = best * 1e3 / blocks / W;
        const double tbs = (double)ncu * W * 256 * blocks * 64.0 / (best * 1e-3) / 1e12;
        printf("%-9s %2d   %8.3f             %6.3f\n", names[mode], W, per_simd, tbs);
    }
    return 0;
}
'''


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "tools/ubench_sha_runs.hip"
    nat, n_nat = emit("natural", natural)
    run, n_run = emit("runs", runs)
    with open(out, "w") as f:
        f.write(HEAD.replace("@K@", ", ".join("0x%08xu" % k for k in K)))
        f.write("// %d instructions\n" % n_nat + nat)
        f.write("// %d instructions\n" % n_run + run)
        f.write(TAIL)
    print("natural %d instructions, runs %d" % (n_nat, n_run))


if __name__ == "__main__":
    main()
