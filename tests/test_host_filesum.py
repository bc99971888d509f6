"""csrc/mi_filesum.h -- the sums the commit holds its bytes against -- checked against an independent statement in
numpy: per 1 MiB chunk of a file, the bytes as little-endian 64-bit words w_0..w_{n-1} (the last one zero-padded),
a = sum w_i, b = sum i * w_i (mod 2^64).  The header computes them piece by piece, in any order, from several threads; a piece's b
comes out of an add-add loop and two multiplications.  Also what the sums are FOR: a flipped bit, two words exchanged, a range that
reads as zeros all change them.  No GPU, no library: the header is compiled into a small program of this test's own."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHUNK = 1 << 20

PROGRAM = r"""
#include "mi_filesum.h"
#include <stdio.h>
#include <stdlib.h>
#include <thread>
#include <vector>
// usage: prog <file> <n_pieces> <off len>...  : the file's chunk sums, its pieces added by four threads in the order given
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb");
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> buf(size ? size : 1);
    if (size && fread(buf.data(), 1, size, f) != (size_t)size) return 1;
    fclose(f);
    const int n = atoi(argv[2]);
    std::vector<std::pair<unsigned long long, unsigned long long>> pieces;
    for (int i = 0; i < n; ++i) pieces.push_back({strtoull(argv[3 + 2 * i], 0, 10), strtoull(argv[4 + 2 * i], 0, 10)});
    const size_t nc = mi_sum::chunks_of((uint64_t)size);
    mi_sum::Pool pool;
    mi_sum::FileSum* sums = pool.take(nc);
    std::vector<std::thread> th;
    for (int t = 0; t < 4; ++t)
        th.emplace_back([&, t] { for (size_t i = t; i < pieces.size(); i += 4) mi_sum::row_add(buf.data() + pieces[i].first, pieces[i].second, pieces[i].first, sums); });
    for (auto& x : th) x.join();
    for (size_t k = 0; k < nc; ++k) printf("%llu %llu\n", (unsigned long long)sums[k].a.load(), (unsigned long long)sums[k].b.load());
    return 0;
}
"""


@pytest.fixture(scope="module")
def prog(tmp_path_factory):
    d = tmp_path_factory.mktemp("filesum")
    src = d / "filesum_check.cpp"
    src.write_text(PROGRAM)
    exe = str(d / "filesum_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "makisu_amd", "csrc"), str(src), "-o", exe, "-lpthread"])
    return exe


def naive(data):
    out = []
    for k in range(max(1, -(-len(data) // CHUNK))):
        c = data[k * CHUNK:(k + 1) * CHUNK]
        c = c + bytes(-len(c) % 8)
        w = np.frombuffer(c, dtype="<u8")
        a = int(w.sum(dtype=np.uint64)) if len(w) else 0
        b = 0
        for i0 in range(0, len(w), 1 << 16):                        # python ints: exact, then reduced
            seg = w[i0:i0 + (1 << 16)].astype(object)
            b += sum(int(x) * (i0 + j) for j, x in enumerate(seg))
        out.append((a % (1 << 64), b % (1 << 64)))
    return out


def run(prog, path, pieces):
    args = [prog, path, str(len(pieces))]
    for off, ln in pieces:
        args += [str(off), str(ln)]
    out = subprocess.run(args, capture_output=True, text=True, check=True).stdout
    return [tuple(int(x) for x in ln.split()) for ln in out.splitlines()]


@pytest.mark.parametrize("size", [0, 1, 7, 8, 9, 4096, CHUNK - 3, CHUNK, CHUNK + 1, 3 * CHUNK + 12345, 5 * CHUNK])
def test_the_headers_sums_are_the_definition_whatever_the_pieces(prog, tmp_path, size):
    rng = np.random.default_rng(size)
    data = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
    p = str(tmp_path / "f")
    open(p, "wb").write(data)
    want = naive(data)
    assert run(prog, p, [(0, size)] if size else []) == want                  # one piece
    for trial in range(3):                                                     # pieces on 8-byte boundaries, shuffled, threaded
        cuts = sorted(set(int(x) * 8 for x in rng.integers(0, size // 8 + 1, 9))) if size >= 8 else []
        bounds = [0] + [c for c in cuts if 0 < c < size] + [size]
        pieces = [(bounds[i], bounds[i + 1] - bounds[i]) for i in range(len(bounds) - 1) if bounds[i + 1] > bounds[i]]
        rng.shuffle(pieces)
        assert run(prog, p, [tuple(map(int, x)) for x in pieces]) == want, pieces


def test_what_the_sums_are_for(prog, tmp_path):
    rng = np.random.default_rng(5)
    data = bytearray(rng.integers(1, 256, 2 * CHUNK + 1000, dtype=np.uint8).tobytes())
    p = str(tmp_path / "f")

    def sums(b):
        open(p, "wb").write(bytes(b))
        return run(prog, p, [(0, len(b))])

    base = sums(data)
    flipped = bytearray(data); flipped[CHUNK + 77] ^= 0x01
    got = sums(flipped)
    assert got[0] == base[0] and got[1] != base[1] and got[2] == base[2]       # a bit: its chunk's sums, nobody else's
    swapped = bytearray(data); swapped[64:72], swapped[8000:8008] = data[8000:8008], data[64:72]
    got = sums(swapped)
    assert got[0][0] == base[0][0] and got[0][1] != base[0][1]                 # two words exchanged: a stays, b does not
    zeroed = bytearray(data); zeroed[2 * CHUNK + 100:2 * CHUNK + 600] = bytes(500)
    assert sums(zeroed)[2] != base[2]                                          # a range that reads as zeros (round 2's slab)
    moved = bytearray(data); moved[4096:8192], moved[8192:12288] = data[8192:12288], data[4096:8192]
    assert sums(moved)[0] != base[0]                                           # two pages exchanged
