"""Build-container script (needs /root/reference; NOT run by the tests): compiles the reference's two cffi
wrapper headers where they lie and records what the compiler makes of them -- sizeof and, per field, offset
and size of every struct -- into tests/golden/abi_layout.json, plus the names and argument counts of the
prototypes in _functionprototypes_wrapper.h.  The JSON is data (numbers and identifiers the drop-in boundary
must reproduce), not source.  tests/test_abi_layout.py holds include/c21cm_abi.h to it on any box.

Reference: src/py21cmfast/src/_inputparams_wrapper.h:6-202, _outputstructs_wrapper.h:6-105,
_functionprototypes_wrapper.h.

    python tests/golden/make_abi_layout.py [/root/reference]
"""

import json
import re
import subprocess
import sys
import tempfile
from pathlib import Path

HERE = Path(__file__).resolve().parent


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def parse_structs(text):
    """[(struct name, [field names])] of every `typedef struct X {...} X;` in declaration order."""
    out = []
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", strip_comments(text), flags=re.S):
        assert m.group(1) == m.group(3)
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                name = re.search(r"(\w+)\s*(\[[^\]]*\])?\s*$", part.strip())
                fields.append(name.group(1))
        out.append((m.group(1), fields))
    return out


def parse_prototypes(text):
    protos = {}
    for m in re.finditer(r"\b(\w+)\s*\(([^()]*)\)\s*;", strip_comments(text)):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return protos


def layout_of(headers, structs, include_dirs=()):
    """Compile + run a program printing sizeof / offsetof / field size for `structs` given `headers`."""
    lines = ["#include <stdio.h>", "#include <stddef.h>", "#include <stdbool.h>"]
    lines += [f'#include "{h}"' for h in headers]
    lines.append("int main(void){")
    for name, fields in structs:
        lines.append(f'printf("{name} . %zu 0\\n", sizeof({name}));')
        for f in fields:
            lines.append(f'printf("{name} {f} %zu %zu\\n", offsetof({name}, {f}), sizeof((({name} *)0)->{f}));')
    lines.append("return 0;}")
    with tempfile.TemporaryDirectory() as tmp:
        src, exe = Path(tmp) / "layout.c", Path(tmp) / "layout"
        src.write_text("\n".join(lines))
        cmd = ["gcc", "-std=gnu11"] + [f"-I{d}" for d in include_dirs] + [str(src), "-o", str(exe)]
        subprocess.run(cmd, check=True)
        text = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    out = {}
    for line in text.strip().splitlines():
        name, field, a, b = line.split()
        rec = out.setdefault(name, {"size": None, "fields": []})
        if field == ".":
            rec["size"] = int(a)
        else:
            rec["fields"].append([field, int(a), int(b)])  # declaration order: [name, offset, size]
    return out


def main():
    ref = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference") / "src" / "py21cmfast" / "src"
    heads = [ref / "_inputparams_wrapper.h", ref / "_outputstructs_wrapper.h"]
    structs = []
    for h in heads:
        structs += parse_structs(h.read_text())
    layout = layout_of([str(h) for h in heads], structs)
    protos = parse_prototypes((ref / "_functionprototypes_wrapper.h").read_text())
    doc = {
        "made_by": "tests/golden/make_abi_layout.py",
        "from": ["src/py21cmfast/src/_inputparams_wrapper.h", "src/py21cmfast/src/_outputstructs_wrapper.h",
                 "src/py21cmfast/src/_functionprototypes_wrapper.h"],
        "compiler": subprocess.run(["gcc", "--version"], capture_output=True, text=True).stdout.splitlines()[0],
        "structs": layout,
        "prototype_arg_counts": protos,
    }
    (HERE / "abi_layout.json").write_text(json.dumps(doc, indent=1, sort_keys=True) + "\n")
    print(f"{len(layout)} structs, {sum(len(v['fields']) for v in layout.values())} fields, {len(protos)} prototypes")


if __name__ == "__main__":
    main()
