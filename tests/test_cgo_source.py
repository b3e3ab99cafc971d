"""cgo/ has never met a Go compiler (no toolchain in the image): everything that can be checked without one is checked here.
Every C function the Go files call exists in include/*.h with that many arguments; every struct field and constant they name
exists; the Go mirrors of C structs have the C layout; what boss_hip.go uses of package groothip exists there; and
cgo/ctest.c (built by `make -C cgo`, a -m gpu test in test_pipeline.py) replays groothip's call sequence in C.
The seam is theBoss.mapReads, src/pipeline/boss.go:108-242."""
import glob
import os
import re

from conftest import REPO

HEADERS = "\n".join(open(p).read() for p in sorted(glob.glob(os.path.join(REPO, "include", "*.h"))))
GO_FILES = sorted(glob.glob(os.path.join(REPO, "cgo", "**", "*.go"), recursive=True))


def strip_c_comments(src):
    return re.sub(r"/\*.*?\*/", " ", src, flags=re.S)


def c_prototypes():
    out = {}
    src = strip_c_comments(HEADERS)
    for m in re.finditer(r"\b(?:int|void|const char \*|uint32_t)\s*\**\s*(groot_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def c_structs():
    """struct name -> {field: type text}"""
    out = {}
    src = strip_c_comments(HEADERS)
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = {}
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            dm = re.match(r"(.*?)([\w\s,\*\[\]]+)$", decl, flags=re.S)
            names = re.split(r",", decl)
            first = names[0].strip()
            tm = re.match(r"(.+?[\s\*])(\w+)(\[\d*\])?$", first, flags=re.S)
            if not tm:
                continue
            ctype = tm.group(1).strip()
            fields[tm.group(2)] = ctype
            for extra in names[1:]:
                fields[extra.strip().lstrip("*").split("[")[0].strip()] = ctype
        out[m.group(3)] = fields
    return out


def call_arity(src, start):
    """number of top-level arguments of the call whose '(' is at src[start]"""
    depth, n, seen = 0, 0, False
    i = start
    while i < len(src):
        ch = src[i]
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
            if depth == 0:
                return n + 1 if seen else 0
        elif ch == "," and depth == 1:
            n += 1
        elif not ch.isspace() and depth >= 1:
            seen = True
        i += 1
    raise AssertionError("unbalanced call")


def go_source(path):
    src = open(path).read()
    src = re.sub(r"//[^\n]*", "", src)        # line comments (the cgo preamble is a /* */ block: kept out below)
    return src


def test_every_c_call_exists_with_that_arity():
    protos = c_prototypes()
    assert len(protos) > 40
    calls = 0
    for path in GO_FILES:
        src = go_source(path)
        for m in re.finditer(r"\bC\.(groot_\w+)\s*\(", src):
            name = m.group(1)
            if name not in protos:
                # a conversion to a C type, e.g. C.groot_params(...)?  none are used: every C.groot_x( is a call
                raise AssertionError(f"{os.path.basename(path)} calls C.{name}, which include/*.h does not declare")
            got = call_arity(src, m.end() - 1)
            assert got == protos[name], f"{os.path.basename(path)}: C.{name} called with {got} arguments, the header declares {protos[name]}"
            calls += 1
    assert calls >= 20


def test_every_c_type_field_and_constant_exists():
    structs = c_structs()
    hdr = strip_c_comments(HEADERS)
    for path in GO_FILES:
        src = go_source(path)
        for m in re.finditer(r"\bC\.(groot_\w+)\b(?!\s*\()", src):
            assert re.search(r"\b%s\b" % m.group(1), hdr), f"C.{m.group(1)} is not in include/*.h"
        for m in re.finditer(r"\bC\.(GROOT_\w+)\b", src):
            assert re.search(r"#define\s+%s\b|\b%s\s*=" % (m.group(1), m.group(1)), hdr), f"C.{m.group(1)} is not defined in include/*.h"
        # identifiers bound to a C struct: `var x C.T` (checked inside its function), struct members `x C.T` (reached as .x.field anywhere)
        def check(var_pat, ctype, text):
            for m in re.finditer(var_pat + r"\.(\w+)(?:\.(\w+))?", text):
                f1, f2 = m.group(1), m.group(2)
                assert f1 in structs[ctype], f"{os.path.basename(path)}: {m.group(0)}: {ctype} has no field {f1}"
                inner = structs[ctype][f1].replace("const", "").replace("struct", "").strip()
                if f2 and inner in structs:
                    assert f2 in structs[inner], f"{os.path.basename(path)}: {m.group(0)}: {inner} has no field {f2}"

        n_bound = 0
        for m in re.finditer(r"\bvar\s+(\w+)\s+C\.(groot_\w+)\s*\n", src):
            if m.group(2) in structs:
                end = src.find("\nfunc ", m.end())
                check(r"(?<![\.\w])" + re.escape(m.group(1)), m.group(2), src[m.end(): end if end > 0 else len(src)])
                n_bound += 1
        for m in re.finditer(r"^\s+(\w+)\s+C\.(groot_\w+)\s*$", src, flags=re.M):
            if m.group(2) in structs:
                check(r"\." + re.escape(m.group(1)), m.group(2), src)
                n_bound += 1
        assert n_bound >= 3 or "groothip.go" not in path


def test_go_mirrors_have_the_c_layout():
    structs = c_structs()
    src = go_source(os.path.join(REPO, "cgo", "groothip", "groothip.go"))
    size = {"uint8": 1, "uint16": 2, "uint32": 4, "uint64": 8, "uint8_t": 1, "uint16_t": 2, "uint32_t": 4, "uint64_t": 8}

    def go_fields(name):
        body = re.search(r"type %s struct \{(.*?)\n\}" % name, src, flags=re.S).group(1)
        out = []
        for line in body.strip().splitlines():
            parts = line.split()
            names = [p.strip(",") for p in parts[:-1]]
            out += [size[parts[-1]]] * len(names)
        return out

    trav_c = [size[t] for t in structs["groot_trav"].values()]
    assert go_fields("Trav") == trav_c == [4, 4, 4, 4, 2, 1, 1]       # Collect() casts groot_trav* to *Trav
    counts_c = [size[t] for t in structs["groot_counts"].values()]
    assert go_fields("Counts") == counts_c                            # (copied field by field, same order)
    # the flag constants
    for go_name, c_name in (("TravRC", "GROOT_TRAV_RC"), ("TravStartClip", "GROOT_TRAV_START_CLIP"), ("TravEndClip", "GROOT_TRAV_END_CLIP"),
                            ("TravFirst", "GROOT_TRAV_FIRST")):
        gv = int(re.search(r"%s\s*=\s*(\d+)" % go_name, src).group(1))
        cv = int(re.search(r"#define\s+%s\s+(\d+)u?" % c_name, HEADERS).group(1))
        assert gv == cv, go_name


def test_the_patch_uses_only_what_the_package_has():
    pkg = go_source(os.path.join(REPO, "cgo", "groothip", "groothip.go"))
    funcs = set(re.findall(r"^func (\w+)\(", pkg, flags=re.M))
    methods = set(re.findall(r"^func \(\w+ \*?\w+\) (\w+)\(", pkg, flags=re.M))
    types = set(re.findall(r"^type (\w+) ", pkg, flags=re.M))
    consts = set(re.findall(r"^\s+(Trav\w+)\s*=", pkg, flags=re.M)) | set(re.findall(r"^var (\w+) ", pkg, flags=re.M))
    fields = set(re.findall(r"^\s+(\w+)\s+[\w\.\[\]\*]+\s*(?://.*)?$", pkg, flags=re.M))
    patch = go_source(os.path.join(REPO, "cgo", "patch", "boss_hip.go"))
    for m in re.finditer(r"\bgroothip\.(\w+)", patch):
        assert m.group(1) in funcs | types | consts, f"boss_hip.go uses groothip.{m.group(1)}, which the package lacks"
    for m in re.finditer(r"\bctxs\[\w+\]\.(\w+)\(", patch):
        assert m.group(1) in methods, f"boss_hip.go calls Ctx.{m.group(1)}, which the package lacks"
    for m in re.finditer(r"\b(?:b|cur)\.wire\.(\w+)\(", patch):
        assert m.group(1) in methods, f"boss_hip.go calls Batch.{m.group(1)}, which the package lacks"
    for m in re.finditer(r"groothip\.Params\{(.*?)\}", patch, flags=re.S):
        for f in re.findall(r"(\w+):", m.group(1)):
            assert f in fields, f"groothip.Params has no field {f}"
    # the read length limit is not hard-wired: a longer read makes every ctx grow
    assert "Reopen(" in patch and "MaxLen()" in patch and "MaxReadLen: 512" not in patch
