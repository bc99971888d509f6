"""What the compiler made of the kernels, checked without a GPU (hipcc cross-compiles gfx950): registers, spills, scratch
and static LDS of every kernel from -Rpass-analysis=kernel-resource-usage, against the budgets the launch geometry in
DESIGN.md 4.1-4.3 relies on.  A compiler or source change that pushes a hot kernel over its occupancy step, or makes any
kernel spill, fails HERE instead of showing up as a slower bench line a round later."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "makisu_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FILES = {"sha256.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"], "gear_cdc.hip": [], "tables.hip": [], "crc32.hip": [],
         "mi_stage.hip": []}

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")


def _usage(src, extra, tmp):
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "--cuda-device-only",
                          "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", os.path.join(tmp, "dev.o")] + extra,
                         capture_output=True, text=True, check=True).stderr
    kernels, cur = {}, None
    for line in out.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
            cur = cur[5:] if cur.startswith("void ") else cur
            kernels[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z][A-Za-z /\[\]]*?): (\d+) \[-Rpass", line)
        if m and cur:
            kernels[cur][m.group(1).strip()] = int(m.group(2))
    return kernels


@pytest.fixture(scope="module")
def usage(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("kres"))
    allk = {}
    for src, extra in FILES.items():
        for name, u in _usage(src, extra, tmp).items():
            allk[name] = u
    return allk


def test_no_kernel_spills_or_uses_scratch(usage):
    assert len(usage) >= 35
    for name, u in usage.items():
        assert u["ScratchSize [bytes/lane]"] == 0 and u["VGPRs Spill"] == 0, (name, u)
        # (SGPRs may spill into VGPR lanes without touching memory.  Two kernels do: the per-wave-record build of the chunk
        # pass <pass, loads, kStats = true>, a measuring tool, and the per-file fix pass of multi-GiB files -- 227 VGPRs,
        # one launch of a few hundred workgroups per batch, 0.5 ms for a 16 GiB file)
        assert u["SGPRs Spill"] == 0 or name.endswith(", true>") or name == "mi::gear_file_fix_kernel", (name, u)
        assert u["AGPRs"] == 0, name                              # integer code: nothing parked in accumulation registers


def test_hot_kernels_keep_their_occupancy_step(usage):
    def of(prefix):
        hits = {k: v for k, v in usage.items() if k.startswith(prefix)}
        assert hits, prefix
        return hits
    # SHA-256 over strings: 2 workgroups of 256 threads per CU by LDS pin, a third must still FIT in registers
    # when two batches overlap: <= 168 VGPRs (3 waves per SIMD of the 512 per lane); the cooperative-load form carries its
    # 20 KiB of LDS staging
    for name, u in of("mi::sha256_items_kernel<").items():
        assert u["VGPRs"] <= 168 and u["Occupancy [waves/SIMD]"] >= 3, (name, u)
        coop = re.search(r"<\d+, true", name) is not None
        assert u["LDS Size [bytes/block]"] == (20480 if coop else 0), (name, u)
    lane = [u["VGPRs"] for n, u in usage.items() if re.match(r"mi::sha256_items_kernel<\d+, false, false>", n)]
    assert lane and max(lane) <= 144                              # the lane-owned form: where it has been since round 3 (135)
    # Gear marking, bitmap-free kernels (4.1): 512-thread workgroups, two per CU = 4 waves per SIMD -> <= 128 VGPRs;
    # their 32 table copies are dynamic LDS (requested at launch), no static LDS
    for prefix in ("mi::gear_cdc_small_fast_kernel", "mi::gear_tile_mark_kernel"):
        for name, u in of(prefix).items():
            assert u["VGPRs"] <= 128 and u["Occupancy [waves/SIMD]"] >= 4 and u["LDS Size [bytes/block]"] == 0, (name, u)
    # the per-file CRC32 tiles (4.3b) and the small table kernels: light enough for full occupancy
    for prefix in ("mi::crc32_tiles_kernel", "mi::compact_chunks_kernel", "mi::bin_scatter_kernel", "mi::dedup_insert_kernel",
                   "mi::root_level_items_kernel", "mi::stage_sum_kernel"):
        for name, u in of(prefix).items():
            assert u["Occupancy [waves/SIMD]"] == 8, (name, u)


def test_sha256_instruction_count_is_at_its_floor(tmp_path):
    """the chunk pass is bound by the number of VALU instructions per 64-byte block times one issue slot.  The
    compression as compiled: 64 rounds x (6 rotates + 2 xor3 + Ch + Maj + 3 add3 + 1 add) + 48 schedule words x (4 rotates + 2
    shifts + 2 xor3 + add3 + add) + 8 state adds = 1 400 -- counted here in the disassembly of the roof microbenchmark (the
    compression alone) and of the chunk pass."""
    import collections
    obj, co = str(tmp_path / "sha.o"), str(tmp_path / "sha.co")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "--cuda-device-only", "-mllvm",
                    "-amdgpu-atomic-optimizer-strategy=None", "-c", os.path.join(CSRC, "sha256.hip"), "-o", obj], check=True)
    llvm = "/opt/rocm/lib/llvm/bin/"
    subprocess.run([llvm + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + obj,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
    asm = subprocess.run([llvm + "llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout
    mix = {}
    for f in re.split(r"\n(?=[0-9a-f]+ <)", asm):
        m = re.match(r"[0-9a-f]+ <(\S+)>:", f)
        if m:
            mix[m.group(1)] = collections.Counter(mm.group(1) for mm in re.finditer(r"^\s+([vs]_\w+)", f, re.M))
    roof = next(v for k, v in mix.items() if "sha256_roof_kernel" in k)
    chunk = next(v for k, v in mix.items() if "sha256_items_kernelILi0ELb0ELb0E" in k)
    valu = lambda c: sum(n for k, n in c.items() if k.startswith("v_"))                                   # noqa: E731
    for c in (roof, chunk):
        assert c["v_alignbit_b32"] in range(576, 584) and c["v_bitop3_b32"] == 352 and c["v_add3_u32"] in range(240, 244)
        assert c["v_lshrrev_b32_e32"] == 96
    assert valu(roof) <= 1460, valu(roof)                         # 1 450: the compression + the benchmark's loop
    assert valu(chunk) <= 1760, valu(chunk)                       # 1 734: + loads, byte swaps, padding, the dequeue, the store
    assert chunk["v_mov_b32_e32"] <= 90                           # no copying of state or schedule words around the block loop
