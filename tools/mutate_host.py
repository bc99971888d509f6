"""Mutation testing of the host code behind the C ABI (no GPU): how much of a source file's logic do the CPU tests hold?

One mutant = one small change at one site of one source file (a comparison's strictness, == / !=, && / ||, + 1 / - 1,
an integer literal, true / false, a `continue` / `break` / `return x;` line removed).  The mutated file is compiled on
its own, linked with the unchanged objects of the normal build (makisu_amd/_obj) into a library of its own under
--work, and the given test files run against that library (MAKISU_MI_LIB).  A mutant the tests do not notice
SURVIVES: either the change is equivalent (diagnostics, a bound that cannot be reached) or a statement nothing checks.
The report lists the survivors with their line; what was made of them is in tools/experiments/README.md.

  python tools/mutate_host.py makisu_amd/csrc/mi_tar.hip --tests tests/test_host_tar.py tests/test_host_tar_fuzz.py \
      --n 60 --jobs 4 --seed 1 [--lines 100-300]
  a header: python tools/mutate_host.py makisu_amd/csrc/mi_memtree.h --unit makisu_amd/csrc/mi_memfs.hip --tests ...
      (the unit that includes it is compiled from a copy beside the mutated header, so that ITS include finds the mutant)
"""
import argparse
import os
import random
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from makisu_amd import build as B  # noqa: E402

# (name, regex, replacement function) -- applied to ONE match of ONE line
OPS = [
    ("<= -> <", re.compile(r"(?<![<>=!\-])<=(?!=)"), lambda m: "<"),
    (">= -> >", re.compile(r"(?<![<>=!\-])>=(?!=)"), lambda m: ">"),
    ("< -> <=", re.compile(r"(?<=[\w\)\]] )<(?= [\w\(\-!~\*&])"), lambda m: "<="),
    ("> -> >=", re.compile(r"(?<=[\w\)\]] )>(?= [\w\(\-!~\*&])"), lambda m: ">="),
    ("== -> !=", re.compile(r"(?<![=!<>])==(?!=)"), lambda m: "!="),
    ("!= -> ==", re.compile(r"!=(?!=)"), lambda m: "=="),
    ("&& -> ||", re.compile(r"&&"), lambda m: "||"),
    ("|| -> &&", re.compile(r"\|\|"), lambda m: "&&"),
    ("+ 1 -> - 1", re.compile(r"\+ 1\b(?!\.)"), lambda m: "- 1"),
    ("- 1 -> + 1", re.compile(r"(?<=[\w\)\]] )- 1\b(?!\.)"), lambda m: "+ 1"),
    ("n -> n + 1", re.compile(r"(?<![\w\.\"'\\x])([2-9]|[1-9]\d{1,5})(?![\w\.\"'])(?!u?l*\s*<<)"), lambda m: str(int(m) + 1)),
    ("true -> false", re.compile(r"\btrue\b"), lambda m: "false"),
    ("false -> true", re.compile(r"\bfalse\b"), lambda m: "true"),
    ("drop continue", re.compile(r"\bcontinue;"), lambda m: ";"),
    ("drop break", re.compile(r"\bbreak;"), lambda m: ";"),
    ("! dropped", re.compile(r"(?<=\()!(?=[\w\(])"), lambda m: ""),
    ("+= -> -=", re.compile(r"\+="), lambda m: "-="),
]


def strip_comment(line):
    i = line.find("//")
    return line if i < 0 else line[:i]


def sites(src_lines, lo, hi):
    out = []
    in_block = False
    for ln, line in enumerate(src_lines, 1):
        code = strip_comment(line)
        if "/*" in code:
            in_block = "*/" not in code
            continue
        if in_block:
            in_block = "*/" not in code
            continue
        if ln < lo or ln > hi:
            continue
        s = code.strip()
        if not s or s.startswith("#") or s.startswith("static_assert") or "getenv" in s or "fprintf(stderr" in s:
            continue
        # string literals are left alone: blank them for matching
        masked = re.sub(r'"(\\.|[^"\\])*"', lambda m: '"' + "_" * (len(m.group(0)) - 2) + '"', code)
        masked = re.sub(r"'(\\.|[^'\\])'", lambda m: "'" + "_" * (len(m.group(0)) - 2) + "'", masked)
        for oi, (name, rx, _) in enumerate(OPS):
            for m in rx.finditer(masked):
                out.append((ln, oi, m.start(), m.end()))
    return out


def run_mutant(k, args, src_lines, site, objs_other, flags):
    ln, oi, a, b = site
    name, rx, rep = OPS[oi]
    line = src_lines[ln - 1]
    new_line = line[:a] + rep(line[a:b]) + line[b:]
    work = os.path.join(args.work, "m%03d" % k)
    os.makedirs(work, exist_ok=True)
    base = os.path.basename(args.source)
    mutated = os.path.join(work, base)
    with open(mutated, "w") as f:
        f.writelines(src_lines[: ln - 1] + [new_line] + src_lines[ln:])
    unit = mutated
    if args.unit:                                              # a header: the including unit, copied beside the mutant
        unit = os.path.join(work, os.path.basename(args.unit))
        shutil.copy(args.unit, unit)
    obj = os.path.join(work, os.path.basename(unit).replace(".hip", ".o"))
    lib = os.path.join(work, "libmakisu_mi.so")
    tag = "%s:%d  [%s]  %s  ->  %s" % (base, ln, name, line.strip()[:110], new_line.strip()[:110])
    if args.oracle:                                           # the CPU oracle: plain C, one compile + link
        odir = os.path.join(ROOT, "oracle")
        lib = os.path.join(work, "libmi_oracle.so")
        cc = ["gcc", "-O1", "-fPIC", "-std=gnu11", "-pthread", "-w", "-shared", "-I", odir, "-I", os.path.join(ROOT, "include"),
              "-o", lib, mutated] + [os.path.join(odir, f) for f in ("mi_oracle.c", "mi_oracle_abi.c") if f != base] + ["-lpthread"]
        r = subprocess.run(cc, capture_output=True, text=True)
        if r.returncode != 0:
            shutil.rmtree(work, ignore_errors=True)
            return ("nocompile", tag, "")
        env = dict(os.environ, MI_ORACLE_LIB=lib, PYTHONDONTWRITEBYTECODE="1")
    else:
        cc = [B.HIPCC] + flags + ["-I", os.path.dirname(os.path.abspath(args.source)), "-c", unit, "-o", obj]
        r = subprocess.run(cc, capture_output=True, text=True)
        if r.returncode != 0:
            shutil.rmtree(work, ignore_errors=True)
            return ("nocompile", tag, "")
        link = [B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-Wl,--no-undefined", obj] + objs_other + \
               ["-ldl", "-lpthread", "-lz", "-o", lib]
        r = subprocess.run(link, capture_output=True, text=True)
        if r.returncode != 0:
            shutil.rmtree(work, ignore_errors=True)
            return ("nocompile", tag, "")
        env = dict(os.environ, MAKISU_MI_LIB=lib, PYTHONDONTWRITEBYTECODE="1")
    try:
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", args.marker, "-p", "no:cacheprovider"] + args.tests,
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=args.timeout)
        verdict = "killed" if r.returncode != 0 else "SURVIVED"
        detail = ""
        if r.returncode != 0:
            failed = re.findall(r"^(?:FAILED|ERROR) (\S+)", r.stdout, re.M)
            detail = failed[0] if failed else ("rc %d" % r.returncode)
    except subprocess.TimeoutExpired:
        verdict, detail = "killed", "timeout"
    shutil.rmtree(work, ignore_errors=True)
    return (verdict, tag, detail)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("--tests", nargs="+", required=True)
    ap.add_argument("--n", type=int, default=40)
    ap.add_argument("--jobs", type=int, default=4)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--lines", default="")
    ap.add_argument("--timeout", type=int, default=240)
    ap.add_argument("--work", default="/tmp/mi_mut")
    ap.add_argument("--out", default="")
    ap.add_argument("--oracle", action="store_true", help="the source is a file of oracle/ (gcc, MI_ORACLE_LIB)")
    ap.add_argument("--marker", default="not gpu", help="pytest -m: 'gpu' on a GPU box holds host code that only runs with real kernels "
                                                        "(the parts protocol of split files) against the GPU tests")
    ap.add_argument("--unit", default="", help="the source is a header: the .hip that includes it")
    args = ap.parse_args()
    B.build()
    with open(args.source) as f:
        src_lines = f.readlines()
    lo, hi = 1, len(src_lines)
    if args.lines:
        lo, hi = (int(x) for x in args.lines.split("-"))
    all_sites = sites(src_lines, lo, hi)
    rng = random.Random(args.seed)
    rng.shuffle(all_sites)
    # at most one mutant per (line, operator): spread over the file
    seen, chosen = set(), []
    for s in all_sites:
        if (s[0], s[1]) in seen:
            continue
        seen.add((s[0], s[1]))
        chosen.append(s)
        if len(chosen) >= args.n:
            break
    base = os.path.basename(args.unit or args.source)
    objs_other = [os.path.join(B.OBJ_DIR, s.replace(".hip", ".o")) for s in B.SOURCES if s != base]
    flags = ["--offload-arch=gfx950", "-O1", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-w"] + B.EXTRA.get(base, [])
    os.makedirs(args.work, exist_ok=True)
    print("%s: %d sites, %d mutants, tests: %s" % (base, len(all_sites), len(chosen), " ".join(args.tests)), flush=True)
    results = []
    with ThreadPoolExecutor(args.jobs) as ex:
        futs = [ex.submit(run_mutant, k, args, src_lines, s, objs_other, flags) for k, s in enumerate(chosen)]
        for f in futs:
            v, tag, detail = f.result()
            results.append((v, tag, detail))
            print("%-9s %s %s" % (v, tag, ("   <- " + detail) if detail else ""), flush=True)
    n_k = sum(1 for r in results if r[0] == "killed")
    n_s = sum(1 for r in results if r[0] == "SURVIVED")
    n_c = sum(1 for r in results if r[0] == "nocompile")
    summary = "%s: %d killed, %d survived, %d did not compile (of %d)" % (base, n_k, n_s, n_c, len(results))
    print(summary)
    if args.out:
        with open(args.out, "a") as f:
            f.write("# %s  seed %d  tests: %s\n" % (summary, args.seed, " ".join(args.tests)))
            for v, tag, detail in results:
                if v == "SURVIVED":
                    f.write("SURVIVED  %s\n" % tag)


if __name__ == "__main__":
    main()
