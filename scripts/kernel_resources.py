#!/usr/bin/env python3
"""Static resources of every kernel in the built objects (build/csrc/*.o): VGPRs, AGPRs, SGPRs, scratch bytes, LDS bytes.
Reads the code-object metadata with llvm-readelf --notes (no GPU needed).   usage: kernel_resources.py [substring ...]"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
OBJCOPY = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
FILT = "c++filt"
subs = sys.argv[1:]
rows = []
for obj in sorted((ROOT / "build" / (__import__("os").environ.get("KR_OBJDIR", "csrc"))).glob("*.o")):
    co, fb = Path("/tmp") / (obj.stem + ".co"), Path("/tmp") / (obj.stem + ".fatbin")
    subprocess.run([OBJCOPY, f"--dump-section=.hip_fatbin={fb}", str(obj)], capture_output=True, text=True)
    if not fb.exists():
        continue
    r = subprocess.run([BUNDLER, "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fb}", f"--output={co}", "--unbundle"],
                       capture_output=True, text=True)
    if r.returncode != 0 or not co.exists() or co.stat().st_size == 0:
        continue
    notes = subprocess.run([READELF, "--notes", str(co)], capture_output=True, text=True).stdout
    for blk in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", notes, re.S):
        t = blk.group(0)
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", t).group(1))
        name = re.search(r"\.name:\s+(\S+)", t).group(1)
        rows.append((obj.stem, name, g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
names = subprocess.run([FILT], input="\n".join(r[1] for r in rows), capture_output=True, text=True).stdout.splitlines()
print(f"{'object':22s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'scratch':>8s} {'lds':>7s}  kernel")
for r, n in zip(rows, names):
    n = n.replace("void ", "").split("(")[0]
    if subs and not any(s in n for s in subs):
        continue
    print(f"{r[0]:22s} {r[2]:5d} {r[3]:5d} {r[4]:5d} {r[5]:8d} {r[6]:7d}  {n}")
