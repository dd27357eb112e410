#!/usr/bin/env python3
"""Compare the gfx950 ISA of kernels between a git revision and the working tree (no GPU needed).

A change that is meant to leave a hot kernel alone (a new template parameter, a new variant beside it) can still
perturb its code: round 4's first version of the K1b side test cost the PLAIN instantiation 51 extra moves and 11 SGPR
spills -- 11 % on the headline -- although none of its source lines had changed.  This tool compiles
ahocorasick_rs_amd/csrc/kernels.hip of both trees with --save-temps and prints, per kernel whose (demangled) name
matches, the instruction counts and every opcode whose count differs.

usage: tools/isa_compare.py <git-rev> [name-substring ...]      e.g.  tools/isa_compare.py HEAD~1 k1b_prefilter k_tile_main
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ahocorasick_rs_amd", "csrc")


def build(tree_csrc: str, include: str, out: str) -> str:
    os.makedirs(out, exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-I" + include,
                           os.path.join(tree_csrc, "kernels.hip"), "-o", os.path.join(out, "kernels.o"), "--save-temps=obj"],
                          stderr=subprocess.DEVNULL, cwd=tree_csrc)
    return os.path.join(out, "kernels-hip-amdgcn-amd-amdhsa-gfx950.s")


def kernels(asm: str):
    text = open(asm).read()
    out = {}
    for m in re.finditer(r"^(_Z\w+):\s*; @", text, re.M):
        end = text.index(".Lfunc_end", m.end())
        body = text[m.end():end]
        ins = [l.strip().split()[0] for l in body.split("\n")[1:]
               if l.strip() and not l.strip().startswith((";", ".", "/")) and not l.strip().endswith(":")]
        out[m.group(1)] = collections.Counter(ins)
    return out


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, p.stdout.split("\n")))


def key(d):  # template arguments kept, parameter list dropped; "void " dropped
    return re.sub(r"\(.*", "", d.replace("void ", "")).replace("acx::", "")


def main():
    rev, subs = sys.argv[1], sys.argv[2:]
    with tempfile.TemporaryDirectory() as tmp:
        old = os.path.join(tmp, "old")
        os.makedirs(os.path.join(old, "ahocorasick_rs_amd", "csrc"))
        os.makedirs(os.path.join(old, "include"))
        for f in ("kernels.hip", "kernels.hpp", "device_types.hpp", "automaton.hpp"):
            open(os.path.join(old, "ahocorasick_rs_amd", "csrc", f), "w").write(
                subprocess.check_output(["git", "show", f"{rev}:ahocorasick_rs_amd/csrc/{f}"], cwd=ROOT, text=True))
        open(os.path.join(old, "include", "acx.h"), "w").write(
            subprocess.check_output(["git", "show", f"{rev}:include/acx.h"], cwd=ROOT, text=True))
        a = kernels(build(os.path.join(old, "ahocorasick_rs_amd", "csrc"), os.path.join(old, "include"), os.path.join(tmp, "oa")))
        b = kernels(build(CSRC, os.path.join(ROOT, "include"), os.path.join(tmp, "ob")))
    da, db = demangle(list(a)), demangle(list(b))
    ka = {key(da[n]): a[n] for n in a}
    kb = {key(db[n]): b[n] for n in b}
    # a template that gained trailing parameters: match "name<args" as a prefix when the new arguments are all false / defaults
    for name in sorted(kb):
        if subs and not any(s_ in name for s_ in subs):
            continue
        cand = name if name in ka else next((o for o in ka if name.startswith(o.rstrip(">")) and
                                             set(name[len(o.rstrip(">")):]) <= set(", false<>1024u")), None)
        if cand is None:
            print(f"{name}: new ({sum(kb[name].values())} instructions)")
            continue
        ca, cb = ka[cand], kb[name]
        diff = {k: (ca[k], cb[k]) for k in sorted(set(ca) | set(cb)) if ca[k] != cb[k]}
        print(f"{name}: {sum(ca.values())} -> {sum(cb.values())} instructions" + (f"  {diff}" if diff else "  (identical histogram)"))


if __name__ == "__main__":
    main()
