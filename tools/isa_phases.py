#!/usr/bin/env python
"""Static instruction budget of the march kernel's sample loop BY PHASE.

Compiles csrc/ugrid_march.hip for gfx950 with line tables (-gline-tables-only: the code is the shipped -O3 code, the .loc
directives only annotate it), takes the sample loop of one k_march instantiation (default k_march<3,false,6>: the S1 headline) and
attributes every instruction to the source function its .loc points into -- and, inside ug_march_tile itself, to the statement:

    position | contraction | normalise (u) | level coordinates (sin / cos) | axis set-up | brick address | cell polynomial |
    level sum + mean | alpha | compositing (w, T, thresholds) | compaction | loop control

Prints a table (VALU / SALU / VMEM per phase, share of the loop's VALU) and, with --json, writes it.  Host tool: needs hipcc only.
"""
import argparse
import collections
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "unboundednerfpytorch_amd", "csrc")

# statement classes inside ug_march_tile (first match wins), on the text of the source line
TILE_RULES = [
    (r"__ballot\(surv\)|mbcnt|ent\[idx\]|slot\[idx\]|nsurv", "compaction"),
    (r"__ballot\(!done\)|for \(int j", "loop control"),
    (r"t_table\[j\]|ox \+ dx \* t", "position"),
    (r"nrm|ug_rcp_refined|\* sc|a\.B - rn", "contraction"),
    (r"a\.lox|a\.loy|a\.loz", "normalise (u)"),
    (r"ug_sincos|\(float\)\(1 << k\)", "level coordinates (sin / cos)"),
    (r"ug_div_r\(dens", "level sum + mean"),
    (r"ug_density_level|dens \+=", "level sum + mean"),
    (r"ug_alpha|a\.shift", "alpha"),
    (r"alpha > a\.thres|w = T \* alpha|T = \(float\)|w > a\.thres|dsum|done = true|\(double\)T", "compositing (w, T, thresholds)"),
    (r"if \(!done\)|if \(keep\)|bool surv|float w = 0|float px = 0", "loop control"),
]
FUNC_PHASE = {
    "ug_sincos": "level coordinates (sin / cos)", "ug_sincos_small": "level coordinates (sin / cos)",
    "ug_axis_inrange": "axis set-up", "ug_alpha": "alpha",
    "ug_rcp_refined": "division helpers (contraction, u, mean, alpha)", "ug_div_r": "division helpers (contraction, u, mean, alpha)",
    "ug_norm3_torch": "contraction", "ug_lane": "compaction",
}


def function_of_line(path):
    """line number -> name of the enclosing top-level function of a header (brace depth 0 -> 1 transitions)"""
    out, cur, depth, pending = {}, None, 0, None
    for i, l in enumerate(open(path), 1):
        m = re.search(r"\b(ug_\w+|k_\w+)\s*\(", l)
        if depth == 0 and m and ("__device__" in l or "__global__" in l or pending):
            pending = m.group(1)
        elif depth == 0 and ("__device__" in l or "__global__" in l or "template" in l):
            pending = pending or True
        if depth == 0 and m and pending is True:
            pending = m.group(1)
        if depth > 0 or "{" in l:
            if depth == 0 and isinstance(pending, str):
                cur = pending
            out[i] = cur
        depth += l.count("{") - l.count("}")
        if depth == 0:
            pending = None if "}" in l else pending
    return out


PASS_THROUGH = ("ug_rcp_refined", "ug_div_r", "ug_norm3_torch", "ug_lane", "ug_readlane_f")     # helpers: charged to their caller's phase


def classify(frames, src_lines, funcs):
    """frames: [(file name, line)] innermost first (the .loc comment's inlined-at chain)"""
    for name, line in frames:
        name = os.path.basename(name)
        if name == "__clang_hip_math.h":
            continue                          # expf / logf / floorf ...: charged to the caller
        if name not in ("ugrid_render.h", "ugrid_math.h"):
            continue
        fn = funcs[name].get(line)
        text = src_lines[name][line - 1] if 0 < line <= len(src_lines[name]) else ""
        if fn in PASS_THROUGH:
            continue
        if fn == "ug_march_tile":
            for pat, ph in TILE_RULES:
                if re.search(pat, text):
                    return ph
            return "loop control"
        if fn == "ug_density_level":
            if "ug_axis_inrange" in text:
                return "axis set-up"
            if re.search(r"row|off|\(lvl", text):
                return "brick address + load"
            return "cell polynomial"
        if fn in FUNC_PHASE:
            return FUNC_PHASE[fn]
        return "other (%s)" % fn
    return "kernel wrapper / other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="_Z7k_marchILi3ELb0ELi6EEv", help="mangled-name prefix of the instantiation")
    ap.add_argument("--json", default=None)
    ap.add_argument("--flags", default="", help="extra hipcc flags (e.g. -DUG_X=1 for an A/B arm)")
    a = ap.parse_args()
    out = "/tmp/ugrid_march_g.s"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize",
                           "-gline-tables-only", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out,
                           os.path.join(CSRC, "ugrid_march.hip")] + a.flags.split(), stderr=subprocess.DEVNULL, cwd=CSRC)
    txt = open(out).read()
    src_lines = {n: open(os.path.join(CSRC, n)).read().split("\n") for n in ("ugrid_render.h", "ugrid_math.h")}
    funcs = {n: function_of_line(os.path.join(CSRC, n)) for n in src_lines}
    start = txt.index("\n" + a.kernel)
    body = txt[start: txt.index("s_endpgm", start)].split("\n")
    # the sample loop: from the label that is the target of the backward branch to that branch
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loop = None
    for i, l in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)|\s+s_branch\s+(\.LBB\d+_\d+)", l)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in labels and labels[tgt] < i and (loop is None or i - labels[tgt] > loop[1] - loop[0]):
                loop = (labels[tgt], i)
    lo, hi = loop
    # the loop's blocks may be laid out before its header label too: take every line marked "in Loop" plus [lo, hi]
    first = min([i for i, l in enumerate(body) if "in Loop: Header" in l] + [lo])
    cur = []
    per = collections.defaultdict(collections.Counter)
    for i in range(first, hi + 1):
        l = body[i]
        if re.match(r"\s+\.loc\s", l):
            cur = [(m.group(1), int(m.group(2))) for m in re.finditer(r"([\w.]+):(\d+):\d+", l.split(";", 1)[1])] if ";" in l else []
            continue
        if not l.startswith("\t") or l.strip().startswith((".", ";")):
            continue
        op = l.split()[0]
        kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "other"
        per[classify(cur, src_lines, funcs)][kind] += 1
    tot = sum(c["valu"] for c in per.values())
    rows = sorted(per.items(), key=lambda kv: -kv[1]["valu"])
    print("%-52s %6s %6s %6s %7s" % ("phase (%s, sample loop, static)" % a.kernel, "VALU", "SALU", "VMEM", "share"))
    for ph, c in rows:
        print("%-52s %6d %6d %6d %6.1f%%" % (ph, c["valu"], c["salu"], c["vmem"], 100.0 * c["valu"] / max(1, tot)))
    print("%-52s %6d %6d %6d" % ("total", tot, sum(c["salu"] for c in per.values()), sum(c["vmem"] for c in per.values())))
    if a.json:
        json.dump({"kernel": a.kernel, "flags": a.flags, "loop_static": {ph: dict(c) for ph, c in rows}, "valu_total": tot}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
