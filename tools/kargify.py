#!/usr/bin/env python3
"""Rewrite `P.field` accesses of a fused kernel into per-phase KARG(...) locals (kernarg loads at the point of use).
Regions are delimited by STAMP(i); lines.  usage: kargify.py file kernel_name param_type"""
import re, sys

def transform(s, kernel_name, ptype):
    m = re.search(r"template <int DH>\n__global__ void __launch_bounds__\(\d+\)[^\n]*\n" + kernel_name, s)
    a = m.start()
    e = s.index("    STAMP(8);\n}", a) + len("    STAMP(8);\n}")
    body = s[a:e]
    body = body.replace("(const " + ptype + " P) {", "(const " + ptype + " P_unused) {\n#define PTYPE " + ptype)
    parts = re.split(r"(    STAMP\(\d\);\n)", body)
    out, region = [], 0
    for part in parts:
        if re.match(r"    STAMP\(\d\);\n", part):
            out.append(part)
            continue
        fields = sorted(set(re.findall(r"\bP\.(\w+)", part)))
        if fields:
            tag = "R%d" % region
            decl = "".join("    const auto %s_%s = KARG(%s, %s);\n" % (tag, f, ptype, f) for f in fields)
            for f in fields:
                part = re.sub(r"\bP\.%s\b" % f, "%s_%s" % (tag, f), part)
            if region == 0:
                i = part.index("#define PTYPE " + ptype) + len("#define PTYPE " + ptype) + 1
                part = part[:i] + decl + part[i:]
            else:
                part = decl + part
        region += 1
        out.append(part)
    return s[:a] + "".join(out) + "\n#undef PTYPE" + s[e:]

if __name__ == "__main__":
    p, k, t = sys.argv[1:4]
    s = open(p).read()
    open(p, "w").write(transform(s, k, t))
