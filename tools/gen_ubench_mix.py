"""Generates tools/ubench_mix.hip: how do 2-pass ("full-rate") and 4-pass ("half-rate") VALU instructions
share a SIMD when they are MIXED in one instruction stream, as the SHA-256 compression mixes them?

Round 1's per-instruction table (profiles/r01_ubench_valu.txt) prices the compression at ~2.05 us per
wave-block per SIMD; the compression alone measures 2.37.  This benchmark runs explicit-register streams
(8 independent chains unless said otherwise) of a pattern over
    A  v_alignbit_b32 (4-pass)      B  v_xor_b32 (2-pass)     P  v_add_u32 (2-pass)
    T  v_bitop3_b32 (2-pass, 3 src) D  v_add3_u32 (4-pass)    L  v_add_u32 with a 32-bit literal
    S  v_add3_u32 with an SGPR src  R  v_lshrrev_b32
at 1..8 waves per SIMD and prints ns per wave64 instruction per SIMD next to the sum of the single-type rates.

    python tools/gen_ubench_mix.py > tools/ubench_mix.hip
    hipcc --offload-arch=gfx950 -O2 tools/ubench_mix.hip -o tools/bin/ubench_mix
"""
import sys

CH = [64 + i for i in range(8)]          # chain registers v64..v71
B_, C_ = 72, 73                          # plain sources


def ins(op, c, b=B_, cc=C_):
    if op == "A":
        return f"v_alignbit_b32 v{c}, v{c}, v{b}, 7"
    if op == "a":                         # self-rotate, as SHA uses it
        return f"v_alignbit_b32 v{c}, v{c}, v{c}, 7"
    if op == "B":
        return f"v_xor_b32_e32 v{c}, v{b}, v{c}"
    if op == "P":
        return f"v_add_u32_e32 v{c}, v{b}, v{c}"
    if op == "T":
        return f"v_bitop3_b32 v{c}, v{c}, v{b}, v{cc} bitop3:0x96"
    if op == "D":
        return f"v_add3_u32 v{c}, v{c}, v{b}, v{cc}"
    if op == "L":
        return f"v_add_u32_e32 v{c}, 0x9e3779b9, v{c}"
    if op == "S":
        return f"v_add3_u32 v{c}, v{c}, s21, v{cc}"
    if op == "R":
        return f"v_lshrrev_b32_e32 v{c}, 1, v{c}"
    raise ValueError(op)


def expand(pattern, n=64, chains=CH, b=B_, cc=C_):
    out = []
    for k in range(n):
        out.append(ins(pattern[k % len(pattern)], chains[k % len(chains)], b, cc))
    return out


def sigma_block(n_groups=16):
    """SHA's shape: three rotates of one value into temporaries, one xor3 of them back (dependent at
    distance 1), two interleaved values."""
    out = []
    for g in range(n_groups):
        x = 64 + (g % 4)
        out += [f"v_alignbit_b32 v80, v{x}, v{x}, 6", f"v_alignbit_b32 v81, v{x}, v{x}, 11",
                f"v_alignbit_b32 v82, v{x}, v{x}, 25", f"v_bitop3_b32 v{x}, v80, v81, v82 bitop3:0x96"]
    return out


def sigma_grouped(n_groups=16):
    """the same work, rotates of four values first (12 x 4-pass), then the four xor3 (4 x 2-pass)"""
    out = []
    for g in range(n_groups // 4):
        for x in range(4):
            t = 80 + 3 * x
            out += [f"v_alignbit_b32 v{t}, v{64 + x}, v{64 + x}, 6", f"v_alignbit_b32 v{t + 1}, v{64 + x}, v{64 + x}, 11",
                    f"v_alignbit_b32 v{t + 2}, v{64 + x}, v{64 + x}, 25"]
        for x in range(4):
            t = 80 + 3 * x
            out += [f"v_bitop3_b32 v{64 + x}, v{t}, v{t + 1}, v{t + 2} bitop3:0x96"]
    return out


PATTERNS = [
    ("B", "xor only", expand("B")),
    ("P", "add only", expand("P")),
    ("T", "bitop3 only", expand("T")),
    ("A", "alignbit only", expand("A")),
    ("a", "alignbit x,x (self)", expand("a")),
    ("D", "add3 only", expand("D")),
    ("L", "add literal only", expand("L")),
    ("S", "add3 with sgpr", expand("S")),
    ("R", "lshr only", expand("R")),
    ("AB", "alignbit,xor alternating", expand("AB")),
    ("AABB", "", expand("AABB")),
    ("A4B4", "", expand("AAAABBBB")),
    ("A8B8", "", expand("A" * 8 + "B" * 8)),
    ("A16B16", "", expand("A" * 16 + "B" * 16)),
    ("A32B32", "", expand("A" * 32 + "B" * 32)),
    ("AAB", "2:1", expand("AAB", 63)),
    ("ABB", "1:2", expand("ABB", 63)),
    ("A4B3", "SHA's ratio", expand("AAAABBB", 63)),
    ("BP", "xor,add alternating (two 2-pass kinds)", expand("BP")),
    ("BT", "xor,bitop3 alternating", expand("BT")),
    ("BPTR", "four 2-pass kinds", expand("BPTR")),
    ("AD", "alignbit,add3 alternating (two 4-pass kinds)", expand("AD")),
    ("DB", "add3,xor alternating", expand("DB")),
    ("DT", "add3,bitop3 alternating", expand("DT")),
    ("aT", "self-rotate,bitop3 alternating", expand("aT")),
    ("aaaT", "3 self-rotates + bitop3, independent", expand("aaaT")),
    ("sigma", "3 rotates -> xor3, dependent at distance 1 (SHA's shape)", sigma_block()),
    ("sigmaG", "the same work, 12 rotates then 4 xor3", sigma_grouped()),
    # register banks (bank = index mod 4): every operand of the instruction in ONE bank / in different banks
    ("Bsame", "xor, dst/src all bank 0", expand("B", 64, [64, 68, 76, 80, 84, 88, 92, 96], 72)),
    ("Bdiff", "xor, chain bank 1, src bank 0", expand("B", 64, [65, 69, 77, 81, 85, 89, 93, 97], 72)),
    ("Tsame", "bitop3, all three in bank 0", expand("T", 64, [64, 68, 80, 84, 88, 92, 96, 100], 72, 76)),
    ("Tdiff", "bitop3, banks 1,2,3", expand("T", 64, [65, 69, 77, 81, 85, 89, 93, 97], 74, 75)),
    ("T001", "bitop3, dst/src0 bank 0, src1 bank 0, src2 bank 1", expand("T", 64, [64, 68, 80, 84, 88, 92, 96, 100], 72, 73)),
    ("T011", "bitop3, src0 bank 0, src1 and src2 bank 1", expand("T", 64, [64, 68, 80, 84, 88, 92, 96, 100], 73, 77)),
    ("T012", "bitop3, banks 0,1,2", expand("T", 64, [64, 68, 80, 84, 88, 92, 96, 100], 73, 74)),
    ("Dsame", "add3, all three in bank 0", expand("D", 64, [64, 68, 80, 84, 88, 92, 96, 100], 72, 76)),
    ("Ddiff", "add3, banks 1,2,3", expand("D", 64, [65, 69, 77, 81, 85, 89, 93, 97], 74, 75)),
]

HEAD = r'''// ubench_mix.hip -- GENERATED by tools/gen_ubench_mix.py (edit that).  Mixed 2-pass / 4-pass VALU streams on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define CLOB "s20", "s21", "scc", "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", \
    "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103"
#define INIT \
    "v_mov_b32 v64, %1\n\t" "s_mov_b32 s21, 0x12345\n\t" \
    "v_add_u32_e32 v65, 0x1234567, v64\n\t v_add_u32_e32 v66, 0x2345678, v65\n\t v_add_u32_e32 v67, 0x3456789, v66\n\t" \
    "v_add_u32_e32 v68, 0x1234567, v67\n\t v_add_u32_e32 v69, 0x2345678, v68\n\t v_add_u32_e32 v70, 0x3456789, v69\n\t" \
    "v_add_u32_e32 v71, 0x1234567, v70\n\t v_add_u32_e32 v72, 0x2345678, v71\n\t v_add_u32_e32 v73, 0x3456789, v72\n\t" \
    "v_add_u32_e32 v74, 0x1234567, v73\n\t v_add_u32_e32 v75, 0x2345678, v74\n\t v_add_u32_e32 v76, 0x3456789, v75\n\t" \
    "v_add_u32_e32 v77, 0x1234567, v76\n\t v_add_u32_e32 v78, 0x2345678, v77\n\t v_add_u32_e32 v79, 0x3456789, v78\n\t" \
    "v_add_u32_e32 v80, 0x1234567, v79\n\t v_add_u32_e32 v81, 0x2345678, v80\n\t v_add_u32_e32 v82, 0x3456789, v81\n\t" \
    "v_add_u32_e32 v83, 0x1234567, v82\n\t v_add_u32_e32 v84, 0x2345678, v83\n\t v_add_u32_e32 v85, 0x3456789, v84\n\t" \
    "v_add_u32_e32 v86, 0x1234567, v85\n\t v_add_u32_e32 v87, 0x2345678, v86\n\t v_add_u32_e32 v88, 0x3456789, v87\n\t" \
    "v_add_u32_e32 v89, 0x1234567, v88\n\t v_add_u32_e32 v90, 0x2345678, v89\n\t v_add_u32_e32 v91, 0x3456789, v90\n\t" \
    "v_add_u32_e32 v92, 0x1234567, v91\n\t v_add_u32_e32 v93, 0x2345678, v92\n\t v_add_u32_e32 v94, 0x3456789, v93\n\t" \
    "v_add_u32_e32 v95, 0x1234567, v94\n\t v_add_u32_e32 v96, 0x2345678, v95\n\t v_add_u32_e32 v97, 0x3456789, v96\n\t" \
    "v_add_u32_e32 v98, 0x1234567, v97\n\t v_add_u32_e32 v99, 0x2345678, v98\n\t v_add_u32_e32 v100, 0x3456789, v99\n\t" \
    "v_add_u32_e32 v101, 0x1234567, v100\n\t v_add_u32_e32 v102, 0x2345678, v101\n\t v_add_u32_e32 v103, 0x3456789, v102\n\t"
#define FINI \
    "v_xor_b32_e32 v64, v65, v64\n\t v_xor_b32_e32 v64, v66, v64\n\t v_xor_b32_e32 v64, v67, v64\n\t v_xor_b32_e32 v64, v68, v64\n\t" \
    "v_xor_b32_e32 v64, v69, v64\n\t v_xor_b32_e32 v64, v70, v64\n\t v_xor_b32_e32 v64, v71, v64\n\t v_xor_b32_e32 v64, v76, v64\n\t" \
    "v_xor_b32_e32 v64, v77, v64\n\t v_xor_b32_e32 v64, v80, v64\n\t v_xor_b32_e32 v64, v81, v64\n\t v_xor_b32_e32 v64, v84, v64\n\t" \
    "v_xor_b32_e32 v64, v85, v64\n\t v_xor_b32_e32 v64, v88, v64\n\t v_xor_b32_e32 v64, v89, v64\n\t v_xor_b32_e32 v64, v92, v64\n\t" \
    "v_xor_b32_e32 v64, v93, v64\n\t v_xor_b32_e32 v64, v96, v64\n\t v_xor_b32_e32 v64, v97, v64\n\t v_xor_b32_e32 v64, v100, v64\n\t" \
    "v_mov_b32 %0, v64\n\t"
'''

MAIN = r'''
struct Entry { const char* name; const char* note; void (*fn)(uint32_t*, int); int n; int n4; };
int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("# device %s CUs %d;  WALL ns per wave64 instruction per SIMD, W waves per SIMD (one workgroup = one wave per SIMD,\n"
           "# W workgroups per CU held there by an LDS request);  n4 = 4-pass instructions of the block's n\n", prop.gcnArchName, ncu);
    uint32_t* out;
    CHK(hipMalloc(&out, sizeof(uint32_t) * 256 * ncu * 8));
    const int iters = argc > 1 ? atoi(argv[1]) : 16384;
    const int Ws[] = {1, 2, 3, 4, 8};
    for (auto& en : es) CHK(hipFuncSetAttribute((const void*)en.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    for (int r = 0; r < 30; ++r) hipLaunchKernelGGL(es[0].fn, dim3(ncu * 8), dim3(256), 0, 0, out, iters);   // clocks up
    CHK(hipDeviceSynchronize());
    printf("%-8s %3s %3s |", "pattern", "n", "n4");
    for (int W : Ws) printf("  W=%d  ", W);
    printf("| note\n");
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    for (auto& e : es) {
        printf("%-8s %3d %3d |", e.name, e.n, e.n4);
        for (int W : Ws) {
            const size_t lds = (160u * 1024u) / (size_t)(W + 1) + 1024u;
            hipLaunchKernelGGL(e.fn, dim3(ncu * W), dim3(256), lds, 0, out, 64);
            float best = 1e30f;
            for (int rep = 0; rep < 2; ++rep) {
                CHK(hipEventRecord(a, 0));
                hipLaunchKernelGGL(e.fn, dim3(ncu * W), dim3(256), lds, 0, out, iters);
                CHK(hipEventRecord(b, 0));
                CHK(hipDeviceSynchronize());
                float ms; CHK(hipEventElapsedTime(&ms, a, b));
                if (ms < best) best = ms;
            }
            printf(" %6.3f", (double)best * 1e6 / ((double)iters * e.n * W));
        }
        printf(" | %s\n", e.note);
    }
    return 0;
}
'''


SYNC_PATTERNS = ["B", "A", "AB", "AABB", "A4B4", "A8B8", "A16B16", "A32B32", "A4B3", "aT", "sigma", "sigmaG", "DT"]

SYNC_MAIN = r'''
__global__ __launch_bounds__(512) void k_where(uint32_t* out) {
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg(4 | (31 << 11));   // HW_REG_HW_ID
}
struct Entry { const char* name; const char* note; void (*fn[6])(uint32_t*, int); int n; int n4; int rep[6]; };
int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    uint32_t* out;
    CHK(hipMalloc(&out, sizeof(uint32_t) * 512 * ncu * 4));
    hipLaunchKernelGGL(k_where, dim3(4), dim3(512), 0, 0, out);
    uint32_t hw[32];
    CHK(hipMemcpy(hw, out, sizeof(hw), hipMemcpyDeviceToHost));
    printf("# SIMD of the 8 waves of a 512-thread workgroup (HW_ID bits 5:4), four workgroups:");
    for (int i = 0; i < 32; ++i) printf("%s%u", i % 8 ? "" : "  ", (hw[i] >> 4) & 3);
    printf("\n# 512-thread workgroups (two waves per SIMD each), G per CU held by an LDS request; WALL ns per wave64 instruction per SIMD\n"
           "# nosync: free-running;  sync/N: s_barrier every N instructions\n");
    const int iters = argc > 1 ? atoi(argv[1]) : 16384;
    for (auto& en : es) for (int v = 0; v < 6; ++v) CHK(hipFuncSetAttribute((const void*)en.fn[v], hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    for (int r = 0; r < 30; ++r) hipLaunchKernelGGL(es[0].fn[0], dim3(ncu * 2), dim3(512), 0, 0, out, iters);   // clocks up
    CHK(hipDeviceSynchronize());
    printf("%-8s %3s %3s | G=1 (W=2): nosync sync/64 /128 /256 /512 /1024 | G=2 (W=4): the same | note\n", "pattern", "n", "n4");
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    for (auto& e : es) {
        printf("%-8s %3d %3d |", e.name, e.n, e.n4);
        for (int G = 1; G <= 2; ++G) {
            printf("           ");
            for (int v = 0; v < 6; ++v) {
                const size_t lds = (160u * 1024u) / (size_t)(G + 1) + 1024u;
                const int it = iters / e.rep[v];
                hipLaunchKernelGGL(e.fn[v], dim3(ncu * G), dim3(512), lds, 0, out, 4);
                float best = 1e30f;
                for (int rep = 0; rep < 2; ++rep) {
                    CHK(hipEventRecord(a, 0));
                    hipLaunchKernelGGL(e.fn[v], dim3(ncu * G), dim3(512), lds, 0, out, it);
                    CHK(hipEventRecord(b, 0));
                    CHK(hipDeviceSynchronize());
                    float ms; CHK(hipEventElapsedTime(&ms, a, b));
                    if (ms < best) best = ms;
                }
                printf(" %6.3f  ", (double)best * 1e6 / ((double)it * e.rep[v] * e.n * 2 * G));
            }
            printf("|");
        }
        printf(" %s\n", e.note);
    }
    return 0;
}
'''


def main_sync():
    w = sys.stdout.write
    w(HEAD.replace("ubench_mix.hip", "ubench_mix_sync.hip"))
    entries = []
    pats = {name: (note, block) for name, note, block in PATTERNS}
    for name in SYNC_PATTERNS:
        note, block = pats[name]
        fns = []
        for tag, rep, bar in (("n", 1, False), ("s1", 1, True), ("s2", 2, True), ("s4", 4, True), ("s8", 8, True), ("s16", 16, True)):
            body = "".join(f'        "{line}\\n\\t"\n' for line in block) * rep
            barrier = '        "s_barrier\\n\\t"\n' if bar else ""
            w(f'''
__global__ __launch_bounds__(512) void k_{name}_{tag}(uint32_t* out, int iters) {{
    extern __shared__ uint8_t pad[];
    uint32_t r;
    asm volatile(INIT
        "s_mov_b32 s20, %2\\n\\t"
        ".Lmix_{name}_{tag}_%=:\\n\\t"
{barrier}{body}        "s_sub_u32 s20, s20, 1\\n\\t"
        "s_cmp_lg_u32 s20, 0\\n\\t"
        "s_cbranch_scc1 .Lmix_{name}_{tag}_%=\\n\\t"
        FINI
        : "=v"(r) : "v"(threadIdx.x + blockIdx.x * 512u), "s"(iters) : CLOB);
    out[blockIdx.x * 512u + threadIdx.x] = r;
    if (iters == -1) pad[threadIdx.x] = 0;
}}
''')
            fns.append(f"k_{name}_{tag}")
        n4 = sum(1 for line in block if line.startswith(("v_alignbit", "v_add3")))
        entries.append(f'    {{"{name}", "{note}", {{{", ".join(fns)}}}, {len(block)}, {n4}, {{1, 1, 2, 4, 8, 16}}}},\n')
    w(SYNC_MAIN.replace("int main(", "static Entry es[] = {\n" + "".join(entries) + "};\nint main(", 1))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "sync":
        return main_sync()
    w = sys.stdout.write
    w(HEAD)
    entries = []
    for name, note, block in PATTERNS:
        body = "".join(f'        "{line}\\n\\t"\n' for line in block)
        w(f'''
__global__ __launch_bounds__(256) void k_{name}(uint32_t* out, int iters) {{
    extern __shared__ uint8_t pad[];
    uint32_t r;
    asm volatile(INIT
        "s_mov_b32 s20, %2\\n\\t"
        ".Lmix_{name}_%=:\\n\\t"
{body}        "s_sub_u32 s20, s20, 1\\n\\t"
        "s_cmp_lg_u32 s20, 0\\n\\t"
        "s_cbranch_scc1 .Lmix_{name}_%=\\n\\t"
        FINI
        : "=v"(r) : "v"(threadIdx.x + blockIdx.x * 256u), "s"(iters) : CLOB);
    out[blockIdx.x * 256u + threadIdx.x] = r;
    if (iters == -1) pad[threadIdx.x] = 0;
}}
''')
        n4 = sum(1 for line in block if line.startswith(("v_alignbit", "v_add3")))
        entries.append(f'    {{"{name}", "{note}", k_{name}, {len(block)}, {n4}}},\n')
    main_txt = MAIN.replace("int main(", "static Entry es[] = {\n" + "".join(entries) + "};\nint main(", 1)
    w(main_txt)


if __name__ == "__main__":
    main()
