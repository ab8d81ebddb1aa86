#!/bin/bash
# kernel_regs.sh <tu> [extra flags]: VGPRs / scratch bytes of every kernel of one translation unit (device assembly metadata; no GPU needed)
tu=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I "$root/klara.jl_amd/csrc" -I "$root/build/csrc" "$@" -S --cuda-device-only -o "$out/k.s" "$root/klara.jl_amd/csrc/$tu.hip" 2>/dev/null
python3 - "$out/k.s" <<'PY'
import re, sys, subprocess
t = open(sys.argv[1]).read()
rows = []
for blk in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", t, re.S):
    b = blk.group(0)
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    v = int(re.search(r"\.vgpr_count:\s+(\d+)", b).group(1)); s = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", b).group(1))
    rows.append((name, v, s))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for (n, v, s), d in zip(rows, names):
    print(f"{v:4d} {s:5d}  {d[:110]}")
PY
rm -rf "$out"
