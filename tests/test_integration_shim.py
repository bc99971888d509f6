"""INTEGRATION.md's cgo fragments held against the headers they bind (the Go side cannot be
compiled here -- no `go` in the image -- so the text is checked the way cgo's first pass would): every `C.mi_*` call
names a function one of the two headers declares and passes as many arguments as the prototype takes, every `C.mi_*`
type is a type of the headers, every field selected on a `var x C.mi_<struct>` exists in that struct, and whatever
else is reached through `C.` is declared by a header the fragment's preamble includes.  No GPU, no library calls."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INCLUDE = os.path.join(ROOT, "include")
DOC = os.path.join(ROOT, "INTEGRATION.md")

CGO_BUILTINS = {"CString", "GoString", "GoStringN", "GoBytes", "CBytes"}
C_TYPES = {"int", "uint", "char", "uchar", "long", "ulong", "size_t", "int32_t", "uint32_t", "int64_t", "uint64_t",
           "uint8_t", "int8_t", "uint16_t", "int16_t"}
LIBC = {"free": "stdlib.h", "malloc": "stdlib.h", "memcpy": "string.h", "memset": "string.h"}


def _preprocessed_headers():
    src = '#include "makisu_mi.h"\n#include "makisu_mi_host.h"\n'
    out = subprocess.run(["gcc", "-E", "-P", "-I", INCLUDE, "-x", "c", "-"], input=src.encode(), stdout=subprocess.PIPE,
                         check=True).stdout.decode()
    return out


def _split_top_level(s, sep=","):
    """split at `sep` outside of (), [], {} and string / rune literals"""
    parts, depth, cur, i = [], 0, [], 0
    while i < len(s):
        ch = s[i]
        if ch in "\"'`":
            j = i + 1
            while j < len(s) and s[j] != ch:
                j += 2 if s[j] == "\\" and ch != "`" else 1
            cur.append(s[i:j + 1])
            i = j + 1
            continue
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            parts.append("".join(cur))
            cur = []
        else:
            cur.append(ch)
        i += 1
    parts.append("".join(cur))
    return [p.strip() for p in parts]


def _balanced(s, open_at):
    """the text between the parenthesis at s[open_at] and its partner"""
    depth, i = 0, open_at
    while i < len(s):
        ch = s[i]
        if ch in "\"'`":
            j = i + 1
            while j < len(s) and s[j] != ch:
                j += 2 if s[j] == "\\" and ch != "`" else 1
            i = j + 1
            continue
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
            if depth == 0:
                return s[open_at + 1:i]
        i += 1
    raise AssertionError("unbalanced call: %r" % s[open_at:open_at + 80])


def _prototypes(pre):
    """name -> number of parameters"""
    protos = {}
    for m in re.finditer(r"\b(mi_[a-z0-9_]+)\s*\(", pre):
        # a prototype, not a function-pointer member or a typedef'd callback: preceded by a return type at statement start
        head = pre[pre.rfind(";", 0, m.start()) + 1:m.start()]
        if "{" in head or "(" in head or "typedef" in head:
            continue
        args = _balanced(pre, m.end() - 1).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len(_split_top_level(args))
    return protos


def _structs(pre):
    """struct typedef name -> set of field names; also opaque / enum typedef names with an empty set"""
    types = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)?\s*\{(.*?)\}\s*(\w+)\s*;", pre, flags=re.S):
        fields = set()
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            fp = re.search(r"\(\s*\*\s*(\w+)\s*\)", decl)          # function-pointer member
            for piece in ([fp.group(1)] if fp else _split_top_level(decl)):
                name = re.search(r"(\w+)\s*(\[[^\]]*\]\s*)*$", piece)
                if name:
                    fields.add(name.group(1))
        types[m.group(3)] = fields
    for m in re.finditer(r"typedef\s+(?:struct|enum)\s+\w+\s+(\w+)\s*;", pre):
        types.setdefault(m.group(1), set())
    for m in re.finditer(r"typedef\s+enum\s*\w*\s*\{.*?\}\s*(\w+)\s*;", pre, flags=re.S):
        types.setdefault(m.group(1), set())
    return types


def _go_fragments():
    doc = open(DOC).read()
    return [m.group(1) for m in re.finditer(r"```go\n(.*?)```", doc, flags=re.S)]


def _strip_go_comments(src):
    out, i = [], 0
    while i < len(src):
        if src.startswith("//", i):
            i = src.find("\n", i) if "\n" in src[i:] else len(src)
            continue
        if src.startswith("/*", i):
            i = src.index("*/", i) + 2
            continue
        if src[i] in "\"`":
            j = i + 1
            while j < len(src) and src[j] != src[i]:
                j += 2 if src[j] == "\\" and src[i] == '"' else 1
            out.append(src[i:j + 1])
            i = j + 1
            continue
        out.append(src[i])
        i += 1
    return "".join(out)


def test_headers_parse_into_prototypes_and_structs():
    pre = _preprocessed_headers()
    protos, types = _prototypes(pre), _structs(pre)
    assert protos["mi_ctx_create"] == 2 and protos["mi_batch_add_path"] == 4 and protos["mi_abi_version"] == 0
    assert protos["mi_memfs_commit_layer"] == 9
    assert {"device", "gear_seed", "mask_bits", "min_size", "max_size", "struct_size"} <= types["mi_config"]
    assert {"tar_sha256", "gzip_sha256", "gzip_bytes"} <= types["mi_layer_result"]
    assert "mi_ctx" in types and "mi_batch" in types and types["mi_ctx"] == set()


def test_every_c_reference_of_the_cgo_fragments_resolves():
    pre = _preprocessed_headers()
    protos, types = _prototypes(pre), _structs(pre)
    frags = _go_fragments()
    assert len(frags) >= 6
    macros = set()
    for h in ("makisu_mi.h", "makisu_mi_host.h"):
        macros |= set(re.findall(r"^#define\s+(MI_[A-Z0-9_]+)\s+\(?-?(?:0x)?[0-9a-fA-F]+u?\)?\s*(?:/\*.*)?$",
                                 open(os.path.join(ROOT, "include", h)).read(), flags=re.M))
    assert "MI_MEMFS_TRUST_CTIME" in macros and "MI_GZIP_OFF" in macros and "MI_PART_ALIGN" in macros
    # the shim proper is the fragment with the cgo preamble; the later fragments are excerpts of the same file
    preamble = re.search(r"/\*(.*?)\*/\s*import \"C\"", frags[0], flags=re.S)
    assert preamble, "the first go fragment carries the cgo preamble"
    included = set(re.findall(r'#include\s+[<"]([\w./]+)[>"]', preamble.group(1)))
    assert "makisu_mi.h" in included
    calls = 0
    for frag in frags:
        src = _strip_go_comments(frag)
        for m in re.finditer(r"\bC\.(\w+)", src):
            name, after = m.group(1), src[m.end():m.end() + 1]
            if name in CGO_BUILTINS or name in C_TYPES:
                continue
            if name in LIBC:
                assert LIBC[name] in included, "C.%s needs <%s> in the cgo preamble" % (name, LIBC[name])
                continue
            if after == "(" and name in protos:
                args = _balanced(src, m.end()).strip()
                n = 0 if not args else len(_split_top_level(args))
                assert n == protos[name], "C.%s is called with %d argument(s); the header declares %d" % (name, n, protos[name])
                calls += 1
                continue
            if name in macros:                                  # an integer #define: cgo exposes it as a constant
                continue
            assert name in types or name in protos, "C.%s is declared by neither header" % name
            if name in protos and name not in types:
                raise AssertionError("C.%s is a function used as a type or value" % name)
        # host-helper functions need the second header in the preamble
        host = open(os.path.join(INCLUDE, "makisu_mi_host.h")).read()
        for m in re.finditer(r"\bC\.(mi_\w+)\s*\(", src):
            if re.search(r"\b%s\s*\(" % m.group(1), host) and not re.search(r"\b%s\s*\(" % m.group(1),
                                                                           open(os.path.join(INCLUDE, "makisu_mi.h")).read()):
                assert "makisu_mi_host.h" in included, "C.%s lives in makisu_mi_host.h, which the preamble does not include" % m.group(1)
    assert calls >= 30


def test_the_shim_proper_binds_core_calls_only():
    """the exports carry MI_CORE / MI_BLOCK / MI_DIAG; the cgo shim a maintainer adds first -- the fragment with
    the cgo preamble -- uses the core and nothing else (ctx, MemFS handle, the one-call commit and its options, cache strings,
    chunk index, push digests); the later fragments are the building blocks' and may use anything"""
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(INCLUDE, "makisu_mi.h")).read(), flags=re.S)
    tag = {m.group(2): m.group(1) for m in re.finditer(r"^(MI_CORE|MI_BLOCK|MI_DIAG)\s[^\n(;]*?\b(mi_[a-z0-9_]+)\s*\(", src, flags=re.M)}
    frags = _go_fragments()
    shim = _strip_go_comments(frags[0])
    used = sorted(set(re.findall(r"\bC\.(mi_\w+)\s*\(", shim)))
    assert len(used) >= 15 and "mi_memfs_commit_layer" in used and "mi_memfs_create" in used and "mi_sha256_many" in used, used
    not_core = [n for n in used if tag.get(n) != "MI_CORE"]
    assert not not_core, "the core shim calls %s" % not_core
    later = set(re.findall(r"\bC\.(mi_\w+)\s*\(", "".join(_strip_go_comments(f) for f in frags[1:])))
    assert any(tag.get(n) == "MI_BLOCK" for n in later)


def test_fields_selected_on_c_structs_exist():
    pre = _preprocessed_headers()
    types = _structs(pre)
    checked = 0
    for frag in _go_fragments():
        src = _strip_go_comments(frag)
        var_types = {}
        for m in re.finditer(r"\bvar\s+([\w, ]+?)\s+(\*?)C\.(mi_\w+)", src):
            for v in m.group(1).split(","):
                var_types[v.strip()] = m.group(3)
        for m in re.finditer(r"\b(\w+)\s*:?=\s*&?C\.(mi_\w+)\s*\{", src):
            var_types[m.group(1)] = m.group(2)
        for v, t in var_types.items():
            if not types.get(t):
                continue                                           # opaque handle: nothing to select
            for m in re.finditer(r"(?<![\w.])%s\.(\w+)" % re.escape(v), src):
                assert m.group(1) in types[t], "%s.%s: %s has no such field" % (v, m.group(1), t)
                checked += 1
    assert checked >= 10
