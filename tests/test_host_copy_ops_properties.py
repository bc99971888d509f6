"""Differential test of the copy-op layer (mi_memfs_add_layer_by_copy_ops / mi_snapshot_copy_ops, csrc/mi_memfs.hip) against
the line-by-line model of MemFS.addToLayer in tests/model_memfs.py: generated base trees (with files, symlinks and
directories in the way), a generated source tree really on disk, one or two COPY operations whose destinations are
existing paths, new paths, paths below files and below symlinks, with and without the trailing slash -- the layer's keys,
kinds, owners, permission bits and SOURCE paths must be the model's, the tree afterwards too, and where the model says
the reference fails ("missing intermediate directory", "symlink loop") so must the call."""
import os

import pytest
from hypothesis import event, given, settings, strategies as st

import makisu_amd as M
from model_memfs import ModelFS, ReferenceFails, abs_path
from test_host_diff_properties import tree_pairs
from test_host_memfs import _materialize


@st.composite
def copy_cases(draw):
    base, src_tree = draw(tree_pairs())                        # two related trees: one for the image, one for the context
    tops = [e["relpath"] for e in src_tree if "/" not in e["relpath"] and e["kind"] != M.KIND_SYMLINK]
    ops = []
    for _ in range(draw(st.integers(1, 2))):
        srcs = draw(st.lists(st.sampled_from(tops + ["."]), min_size=1, max_size=2, unique=True)) if tops else ["."]
        stems = ["/" + e["relpath"] for e in base] + ["/new", "/new/deeper", "/a/new", "/"]
        dst = draw(st.sampled_from(stems))
        if draw(st.booleans()):
            dst = dst.rstrip("/") + "/" + draw(st.sampled_from(["n1", "b", "n1/n2"]))
        if len(srcs) > 1 or draw(st.booleans()):
            dst = dst.rstrip("/") + "/"
        ops.append({"srcs": srcs, "dst": dst, "uid": draw(st.sampled_from([0, 7])), "gid": draw(st.sampled_from([0, 9]))})
    return base, src_tree, ops


@settings(max_examples=400, deadline=None, derandomize=True, database=None)
@given(copy_cases())
def test_copy_op_layer_equals_the_model(tmp_path_factory, case):
    base, src_tree, ops = case
    tmp = tmp_path_factory.mktemp("copyprop")
    root, ctx = str(tmp / "root"), str(tmp / "ctx")
    os.mkdir(root)
    os.mkdir(ctx)
    _materialize(ctx, src_tree)

    def walk_entries(src):
        out = []
        for e in M.tree_walk(src, src, (), M.TREE_SCAN, full=True):
            out.append((src if e["relpath"] == "." else src + "/" + e["relpath"], e))
        return out
    model = ModelFS()
    model.update_from_tar(base)
    model.layer = {}
    failed = None
    try:
        for op in ops:
            model.add_to_layer(dict(op, src_root=ctx), walk_entries, os.path.isdir)
    except ReferenceFails as e:
        failed = str(e)
    cops = [dict(op, src_root=ctx) for op in ops]
    with M.MemFS(root, now_sec=ModelFS.NOW) as fs:
        fs.update_from_entries(base)
        if failed:
            event("the reference fails: " + " ".join(failed.split(" ")[:2]))
            with pytest.raises(M.MiError) as ei:
                fs.add_layer_by_copy_ops(cops)
            assert failed[:120] in str(ei.value)
            return
        event("the reference fails: no")
        layer = fs.add_layer_by_copy_ops(cops)
        want = model.layer
        got = {}
        for e in layer:
            d, b = os.path.split("/" + e["relpath"])
            got[os.path.join(d, b[4:]) if b.startswith(".wh.") else "/" + e["relpath"]] = e
        assert sorted(got) == sorted(want), (sorted(got), sorted(want))
        assert ["/" + e["relpath"] for e in layer] == [w[1] if w[0] == "whiteout" else k for k, w in sorted(want.items())]
        strip = lambda s: s[len(root):] or "/" if s.startswith(root + "/") or s == root else s   # noqa: E731
        for k, (what, node) in want.items():
            if what == "whiteout":
                continue
            g, h = got[k], node.hdr
            assert (g["kind"], g["mode"] & 0o7777, g.get("uid", 0), g.get("gid", 0), g["size"], g["link_target"]) == \
                (h["kind"], h["mode"] & 0o7777, h.get("uid", 0), h.get("gid", 0), h["size"], h.get("link_target")), k
            assert strip(g["src"]) == strip(node.src), (k, g["src"], node.src)
            if node.made_up:
                assert g["mtime_sec"] == ModelFS.NOW
        tree_want = model.flat()
        tree_got = {"/" + e["relpath"]: e for e in fs.entries()}
        assert sorted(tree_got) == sorted(tree_want)
        for p, node in tree_want.items():
            assert tree_got[p]["kind"] == node.hdr["kind"] and strip(tree_got[p]["src"]) == strip(node.src), p
        event("layer entries: %s" % ("0" if not want else "1-4" if len(want) < 5 else "5+"))
        # the stateless call on the same lists: the same layer
        flat = M.copy_ops_layer(base, root, cops, now_sec=ModelFS.NOW)
        assert [(e["relpath"], e["kind"], e["uid"], e["gid"]) for e in flat] == [(e["relpath"], e["kind"], e["uid"], e["gid"]) for e in layer]
