"""CPU tests of the host-side walks (csrc/mi_tree.hip, no GPU needed): filepath.Walk order,
checksumPathContents' skip rule (lib/builder/step/add_copy_step.go:194-238) and the snapshot
walk's shouldSkip (lib/snapshot/utils.go:37-75), against an independent Python emulation.
The table cases for IsDescendantOfAny follow lib/pathutils/path.go:24-35."""
import os
import stat

import pytest


def _go_walk(root):
    st = os.lstat(root)
    yield root, st
    if stat.S_ISDIR(st.st_mode):
        for name in sorted(os.listdir(root), key=os.fsencode):
            yield from _go_walk(os.path.join(root, name))


def _special(st):
    return stat.S_ISFIFO(st.st_mode) or stat.S_ISSOCK(st.st_mode) or stat.S_ISCHR(st.st_mode) or stat.S_ISBLK(st.st_mode)


@pytest.fixture()
def tree(tmp_path):
    b = tmp_path / "ctx"
    os.makedirs(b / "a" / "deep" / "er")
    os.makedirs(b / "a-b")
    os.makedirs(b / "a.d")
    os.makedirs(b / "empty")
    os.makedirs(b / "z" / ".wh..wh.plnk")
    for rel, data in [("a/x.txt", b"x" * 10), ("a/deep/er/f", b"deep"), ("a-b/y", b""), ("a.d/q", b"q"),
                      ("a.txt", b"t"), ("z/last", b"l" * 3), ("z/.wh..wh.plnk/hidden", b"h"),
                      ("z/.wh.gone", b""), ("Z-upper", b"U"), ("\xc3\xa9-utf8".encode("latin1").decode("utf8"), b"e")]:
        (b / rel).write_bytes(data)
    os.symlink("x.txt", b / "a" / "lnk")
    os.symlink("/no/such", b / "dangling")
    os.symlink("a", b / "dirlink")                      # a symlink to a directory is NOT descended into
    os.mkfifo(b / "a" / "fifo")
    return b


def test_context_walk_matches_go_order(tree, engine_lib):
    import makisu_amd
    got = makisu_amd.tree_walk(str(tree))
    want = [(os.path.relpath(p, tree), st) for p, st in _go_walk(str(tree)) if not _special(st)]
    assert [g[0] for g in got] == [w[0] for w in want]
    assert got[0][0] == "."
    kinds = {g[0]: g[4] for g in got}
    assert kinds["a"] == 0 and kinds["a/x.txt"] == 1 and kinds["a/lnk"] == 2 and kinds["dirlink"] == 2
    assert "a/fifo" not in kinds                       # utils.IsSpecialFile
    assert not any(k.startswith("dirlink/") for k in kinds)
    links = {g[0]: g[1] for g in got if g[4] == 2}
    assert links == {"a/lnk": "x.txt", "dangling": "/no/such", "dirlink": "a"}
    # Walk order is per-directory lexical, NOT a sort of full paths: "a/..." comes before "a-b"
    names = [g[0] for g in got]
    assert names.index("a/x.txt") < names.index("a-b") < names.index("a.d") < names.index("a.txt")
    assert names != sorted(names)
    # regular files get running ordinals in visit order and stat sizes
    regs = [g for g in got if g[4] == 1]
    assert [g[2] for g in regs] == list(range(len(regs)))
    assert {g[0]: g[3] for g in regs}["a/x.txt"] == 10
    # context mode keeps whiteout-looking names: only special files are skipped
    assert "z/.wh..wh.plnk/hidden" in kinds and "z/.wh.gone" in kinds


def test_scan_walk_should_skip(tree, engine_lib):
    import makisu_amd
    bl = [str(tree / "a" / "deep"), str(tree / "a.d") + "/"]
    got = [g[0] for g in makisu_amd.tree_walk(str(tree), rel_base=str(tree.parent), blacklist=bl,
                                              mode=makisu_amd.TREE_SCAN)]
    assert got[0] == "ctx"
    assert "ctx/z/.wh.gone" in got                      # a plain whiteout marker is kept
    assert not any(".wh..wh." in g for g in got)        # AUFS metadata pruned with its subtree
    assert not any(g == "ctx/a/deep" or g.startswith("ctx/a/deep/") for g in got)
    assert not any(g == "ctx/a.d" or g.startswith("ctx/a.d/") for g in got)   # trailing "/" normalised (AbsPath)
    assert "ctx/a/x.txt" in got and "ctx/a/fifo" not in got
    # blacklisting "/" prunes everything (ancestor == "/" rule, path.go:29)
    assert makisu_amd.tree_walk(str(tree), blacklist=["/"], mode=makisu_amd.TREE_SCAN) == []
    # a path that only shares a name prefix with a blacklisted dir is NOT a descendant
    got2 = [g[0] for g in makisu_amd.tree_walk(str(tree), blacklist=[str(tree / "a")], mode=makisu_amd.TREE_SCAN)]
    assert "a-b" in got2 and "a.txt" in got2 and "a" not in got2 and "a/x.txt" not in got2


def test_walk_single_file_and_errors(tree, engine_lib):
    import makisu_amd
    # COPY of a single file: Walk visits just the file, relpath relative to the context dir
    one = makisu_amd.tree_walk(str(tree / "a" / "x.txt"), rel_base=str(tree))
    assert [(g[0], g[4]) for g in one] == [("a/x.txt", 1)]
    with pytest.raises(makisu_amd.MiError) as ei:
        makisu_amd.tree_walk(str(tree / "missing"))
    assert ei.value.code == -5
    with pytest.raises(makisu_amd.MiError) as ei:       # "write path is outside of context dir" (:205-209)
        makisu_amd.tree_walk(str(tree / "a"), rel_base=str(tree / "z"))
    assert ei.value.code == -1
