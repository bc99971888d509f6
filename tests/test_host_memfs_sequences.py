"""The MemFS handle against the statement-by-statement model (tests/model_memfs.py) over SEQUENCES of a build's calls on
one tree: base layers merged (UpdateFromTarReader), COPY steps (AddLayerByCopyOps, the file system untouched), RUN steps
(the root really rewritten, then AddLayerByScan) in generated orders.  After every step: the layer (keys, kinds, order,
source paths) and the whole tree (created directories and every node's source included) are the model's.  What one call
leaves behind is what the next one starts from -- e.g. a copied file is "on disk" for the next scan while its SOURCE is."""
import os
import shutil

import pytest
from hypothesis import event, given, settings, strategies as st

import makisu_amd as M
from model_memfs import ModelFS, ReferenceFails, abs_path
from test_host_copy_ops_properties import copy_cases
from test_host_diff_properties import tree_pairs
from test_host_memfs import _materialize


@st.composite
def sequences(draw):
    base, ctx_tree, ops = draw(copy_cases())
    steps = []
    for _ in range(draw(st.integers(1, 4))):
        what = draw(st.sampled_from(["merge", "copy", "run", "run"]))
        if what == "merge":
            steps.append(("merge", draw(tree_pairs())[1]))
        elif what == "copy":
            steps.append(("copy", [draw(st.sampled_from(ops))]))
        else:
            steps.append(("run", draw(tree_pairs())[1]))
    return base, ctx_tree, steps


def _layer_keys(layer):
    out = []
    for e in layer:
        d, b = os.path.split("/" + e["relpath"])
        out.append((os.path.join(d, b[4:]) if b.startswith(".wh.") else "/" + e["relpath"], b.startswith(".wh.")))
    return out


@settings(max_examples=250, deadline=None, derandomize=True, database=None)
@given(sequences())
def test_sequences_of_merges_copies_and_scans(tmp_path_factory, seq):
    base, ctx_tree, steps = seq
    tmp = tmp_path_factory.mktemp("seq")
    root, ctx = str(tmp / "root"), str(tmp / "ctx")
    os.mkdir(root)
    os.mkdir(ctx)
    _materialize(ctx, ctx_tree)
    strip = lambda s: (s[len(root):] or "/") if s == root or s.startswith(root + "/") else s   # noqa: E731
    src_of = lambda p: root + p if p != "/" else root                                         # noqa: E731

    def walk_entries(src):
        return [(src if e["relpath"] == "." else src + "/" + e["relpath"], e) for e in M.tree_walk(src, src, (), M.TREE_SCAN, full=True)]
    root_hdr = M.tree_walk(root, root, (), M.TREE_SCAN, full=True)[0]
    model = ModelFS(dict(root_hdr, relpath=""))
    model.tree.src = root
    with M.MemFS(root, now_sec=ModelFS.NOW) as fs:
        model.update_from_tar(base, src_of)
        fs.update_from_entries(base)
        for k, (what, arg) in enumerate(steps):
            try:
                if what == "merge":
                    model.update_from_tar(arg, src_of)
                    n = fs.update_from_entries(arg)
                    assert n == len(model.layer)
                    layer = None
                elif what == "copy":
                    cops = [dict(op, src_root=ctx) for op in arg]
                    model.layer = {}
                    err = None
                    try:
                        for op in cops:
                            model.add_to_layer(op, walk_entries, os.path.isdir)
                    except ReferenceFails as e:
                        err = e
                    if err:
                        with pytest.raises(M.MiError):
                            fs.add_layer_by_copy_ops(cops)
                        raise err
                    layer = fs.add_layer_by_copy_ops(cops)
                else:
                    for name in os.listdir(root):                             # the RUN step: the root becomes `arg`
                        p = os.path.join(root, name)
                        shutil.rmtree(p) if os.path.isdir(p) and not os.path.islink(p) else os.unlink(p)
                    _materialize(root, arg)
                    os.utime(root, (root_hdr["mtime_sec"], root_hdr["mtime_sec"]))
                    walked = M.tree_walk(root, root, (), M.TREE_SCAN, full=True)
                    model.scan([(src_of(abs_path(e["relpath"])), abs_path(e["relpath"]), e) for e in walked], os.path.lexists)
                    layer = fs.add_layer_by_scan(walked)
            except ReferenceFails as e:
                event("the reference fails at a %s: %s" % (what, " ".join(str(e).split(" ")[:2])))
                if what != "copy":
                    with pytest.raises(M.MiError):
                        (fs.update_from_entries if what == "merge" else fs.add_layer_by_scan)(arg if what == "merge" else walked)
                return
            if layer is not None:
                want = sorted((key, w[0] == "whiteout") for key, w in model.layer.items())
                assert _layer_keys(layer) == want, (k, what)
                for e in layer:
                    w = model.layer.get("/" + e["relpath"])
                    if w and w[0] == "content":
                        assert e["kind"] == w[1].hdr["kind"] and strip(e["src"]) == strip(w[1].src), (k, e["relpath"])
            tree_want = model.flat()
            tree_got = {"/" + e["relpath"]: e for e in fs.entries()}
            assert sorted(tree_got) == sorted(tree_want), (k, what)
            for p, node in tree_want.items():
                assert tree_got[p]["kind"] == node.hdr["kind"] and strip(tree_got[p]["src"]) == strip(node.src), (k, what, p)
        event("steps: " + " ".join(w for w, _ in steps))
