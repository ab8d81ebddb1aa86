#!/usr/bin/env python3
"""Loop structure of a kernel in hipcc -S output: for every backward branch the loop's length, VALU / LDS / VMEM / scratch
instruction counts.   usage: asm_loops.py file.s <mangled-name-prefix>"""
import re, sys
s = open(sys.argv[1]).read().split("\n")
pref = sys.argv[2]
start = next(i for i, l in enumerate(s) if l.startswith(pref) and l.rstrip().split(":")[0].startswith(pref) and ":" in l)
end = next(i for i in range(start, len(s)) if "s_endpgm" in s[i])
body = s[start:end + 1]
print(body[0].split(":")[0], len(body), "lines; scratch ops", sum("scratch_" in l for l in body), "valu", sum(bool(re.match(r"\s+v_", l)) for l in body))
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)", l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i:
            seg = body[labels[t]:i]
            c = lambda pat: sum(bool(re.match(pat, x)) for x in seg)
            pats = {"valu": r"\s+v_", "mfma": r"\s+v_mfma", "ds": r"\s+ds_", "vmem": r"\s+(buffer|global)_", "scratch": r"\s+scratch_", "salu": r"\s+s_"}
            print(f"  loop @{labels[t]}..{i}: {len(seg)} lines, " + ", ".join(f"{k} {c(v)}" for k, v in pats.items()))
