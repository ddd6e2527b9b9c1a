"""Generate the ctypes struct declarations of INTEGRATION.md from include/freerl_hip.h.

The document's binding stub is what a reference maintainer copies; written by hand it went stale (round 2: six
fields short of the header).  This tool parses the header's `typedef struct` blocks and rewrites the region between
`<!-- ctypes-structs:begin -->` and `<!-- ctypes-structs:end -->`; tests/test_abi_and_host.py executes that region and
compares every sizeof / field offset with a gcc probe of the header.

    python tools/gen_ctypes_stub.py            # rewrite INTEGRATION.md in place
    python tools/gen_ctypes_stub.py --check    # exit 1 if the document is stale
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "freerl_hip.h")
DOC = os.path.join(ROOT, "INTEGRATION.md")
BEGIN, END = "<!-- ctypes-structs:begin -->", "<!-- ctypes-structs:end -->"

CTYPE = {"int": "C.c_int", "float": "C.c_float", "double": "C.c_double", "uint64_t": "C.c_uint64", "int64_t": "C.c_int64",
         "long long": "C.c_longlong", "uint8_t": "C.c_uint8"}


def parse_structs(src=None):
    """[(name, [(field, ctypes expression)])] in header order; nested structs refer to earlier ones by name."""
    src = open(HEADER).read() if src is None else src
    defines = {k: int(v) for k, v in re.findall(r"#define\s+(FRL_[A-Z_]+)\s+(\d+)", src)}
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = []
    for body, name in re.findall(r"typedef\s+struct\s+\w+\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        known = {n for n, _ in out}
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.match(r"(const\s+)?(long long|\w+)\s*(\*?)\s*(.*)$", decl)
            base, ptr, names = m.group(2), m.group(3), m.group(4)
            for nm in [x.strip() for x in names.split(",")]:
                p = ptr
                if nm.startswith("*"):
                    p, nm = "*", nm[1:].strip()
                arr = re.match(r"(\w+)\[(\w+)\]$", nm)
                if base in known:
                    t = base
                else:
                    t = CTYPE[base]
                if p:
                    t = "C.POINTER(%s)" % t
                if arr:
                    nm, dim = arr.group(1), arr.group(2)
                    t = "%s * %d" % (t, defines[dim] if dim in defines else int(dim))
                fields.append((nm, t))
        out.append((name, fields))
    return out, defines


def render():
    structs, defines = parse_structs()
    lines = ["```python", "# generated from include/freerl_hip.h by tools/gen_ctypes_stub.py -- do not edit by hand", "import ctypes as C", ""]
    for k in ("FRL_MAX_AGENTS", "FRL_STAT_COUNT", "FRL_COMM_ID_BYTES", "FRL_COMM_MAX_VALUES"):
        if k in defines:
            lines.append("%s = %d" % (k, defines[k]))
    for name, fields in structs:
        lines += ["", "", "class %s(C.Structure):" % name, "    _fields_ = ["]
        row = "        "
        for i, (f, t) in enumerate(fields):
            item = '("%s", %s)%s' % (f, t, "," if i + 1 < len(fields) else "]")
            if len(row) + len(item) > 118:
                lines.append(row.rstrip())
                row = "        "
            row += item + " "
        lines.append(row.rstrip())
    lines.append("```")
    return "\n".join(lines)


def extract(doc=None):
    """The python source between the markers of INTEGRATION.md (without the code fence)."""
    doc = open(DOC).read() if doc is None else doc
    a, b = doc.index(BEGIN) + len(BEGIN), doc.index(END)
    body = doc[a:b].strip()
    assert body.startswith("```python") and body.endswith("```")
    return body[len("```python"):-3]


def main(argv):
    doc = open(DOC).read()
    a, b = doc.index(BEGIN) + len(BEGIN), doc.index(END)
    new = doc[:a] + "\n" + render() + "\n" + doc[b:]
    if "--check" in argv:
        if new != doc:
            print("INTEGRATION.md's ctypes block is stale: run python tools/gen_ctypes_stub.py")
            return 1
        return 0
    open(DOC, "w").write(new)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
