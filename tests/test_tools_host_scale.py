"""tools/host_scale_bench.py (the host rows at C2 / C4 entry counts, profiles/r04_host_scale.txt) keeps running: a small
count through the same code, every line there, the scan's layers of the sizes the generator implies."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_scale_bench_runs_and_reports_every_row(engine_lib):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_scale_bench.py"), "20000"], capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    out = p.stdout
    for row in ("mi_entries_commit_order, walk-ordered input", "mi_entries_commit_order, shuffled input",
                "mi_memfs_update_from_entries (layer merge)", "mi_memfs_add_layer_by_scan, 0.1 % changed",
                "mi_memfs_add_layer_by_scan, nothing changed", "mi_snapshot_diff (stateless twin)", "mi_layer_add, header-only members"):
        assert row in out, out
    assert "20200 entries (200 directories of 100 files)" in out
    assert re.search(r"layer merge\)\s+[\d.]+ s\s+[\d.]+ us / entry\s+\(20200 merged\)", out), out
    # 21 entries changed (every 1000th of 20 200): each with its directory carried along unless it IS a directory
    m = re.search(r"0\.1 % changed\s+[\d.]+ s\s+[\d.]+ us / entry\s+\(layer of (\d+)\)", out)
    assert m and 21 <= int(m.group(1)) <= 42, out
    assert re.search(r"nothing changed\s+[\d.]+ s\s+[\d.]+ us / entry\s+\(layer of 0\)", out), out


def test_the_gpu_side_shell_tools_parse():
    """tools/*.sh run on the GPU box where a syntax error costs a call: `bash -n` each of them here"""
    import glob
    scripts = sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh")))
    assert len(scripts) >= 10
    for s in scripts:
        p = subprocess.run(["bash", "-n", s], capture_output=True, text=True)
        assert p.returncode == 0, (s, p.stderr)
        assert os.access(s, os.X_OK), s + " is not executable"


def test_the_python_tools_compile():
    """... and the python ones (probes that drive the binding on the GPU box): a syntax error is found here"""
    import ast
    import glob
    tools = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    assert len(tools) >= 20
    for t in tools:
        ast.parse(open(t).read(), t)


def test_the_python_tools_name_things_the_binding_has(engine_lib):
    """a probe that calls `makisu_amd.<something>` that does not exist is 45 s of GPU time for a python exception
    (tools/experiments/README.md, round 4): every attribute the tools take from the binding's module exists"""
    import glob
    import importlib.util
    import makisu_amd
    missing = []
    for t in sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]:
        src = open(t).read()
        if "makisu_amd" not in src:
            continue
        for alias in set(re.findall(r"import makisu_amd as (\w+)", src)) | {"makisu_amd"}:
            for m in re.finditer(r"(?<![\w.])%s\.([A-Za-z_]\w*)" % re.escape(alias), src):
                name = m.group(1)
                if not hasattr(makisu_amd, name) and importlib.util.find_spec("makisu_amd." + name) is None:
                    missing.append((os.path.basename(t), name))
    assert not missing, sorted(set(missing))
