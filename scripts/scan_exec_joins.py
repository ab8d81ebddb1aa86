#!/usr/bin/env python3
"""Static scan of the device assembly for register copies placed at the head of a join block BEFORE the execution mask is restored.

Background (streamed dense kernels): under register pressure the allocator parks architectural registers in accumulator
registers (v_accvgpr_write / v_accvgpr_read) or scratch.  If such a copy lands at the top of the block where a divergent if / else rejoins —
ahead of the `s_or_b64 exec, exec, s[..]` that re-enables the lanes of the other side — it only runs for the lanes of one side, and a value
defined on the other side is lost (k_dense_big<MH, 48, mean>, ROCm 7.2).  This script lists, per kernel, the blocks whose first instructions
up to the mask restore contain such a SAVE (v_accvgpr_write / scratch_store).  A hit is not proof of a wrong value (a save of a value that only
the active side defines is harmless) — it marks the places to look at, and the kernels written to avoid divergent branches around their register
arrays must have none (tests/test_host_api.py::test_streamed_dense_kernels_have_no_register_saves_under_a_partial_mask).

usage: scan_exec_joins.py <translation unit without .hip> [extra hipcc flags]      prints  <hits> <kernel>  lines, exit status 0
"""
import re, subprocess, sys, tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
COPY = re.compile(r"^\s*(v_accvgpr_write_b32|scratch_store)")      # SAVES of a live register (a reload for use inside the block is normal)
RESTORE = re.compile(r"^\s*s_or_b64 exec, exec,")
LABEL = re.compile(r"^\.LBB\d+_\d+:")
KERNEL = re.compile(r"^(_Z\w+):")


def scan(asm: str, window: int = 16):
    hits, kernel, lines = {}, None, asm.split("\n")
    for i, ln in enumerate(lines):
        m = KERNEL.match(ln)
        if m:
            kernel = m.group(1); hits.setdefault(kernel, [])
        if kernel is None or not LABEL.match(ln):
            continue
        seen_copy = []
        n = 0
        for ln2 in lines[i + 1:i + 1 + 4 * window]:
            if LABEL.match(ln2) or ln2.lstrip().startswith(("s_cbranch", "s_branch", "s_endpgm")):
                break
            if not ln2.strip() or ln2.lstrip().startswith(";"):
                continue
            n += 1
            if RESTORE.match(ln2):
                if seen_copy:
                    hits[kernel].append((ln.strip(), seen_copy))
                break
            if COPY.match(ln2):
                seen_copy.append(ln2.strip())
            if n >= window:
                break
    return hits


def main():
    tu, flags = sys.argv[1], sys.argv[2:]
    with tempfile.TemporaryDirectory() as d:
        out = Path(d) / "k.s"
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", str(ROOT / "klara.jl_amd" / "csrc"),
                        "-I", str(ROOT / "build" / "csrc"), *flags, "-S", "--cuda-device-only", "-o", str(out), str(ROOT / "klara.jl_amd" / "csrc" / f"{tu}.hip")],
                       check=True, capture_output=True)
        hits = scan(out.read_text())
    names = subprocess.run(["c++filt"], input="\n".join(hits), capture_output=True, text=True).stdout.split("\n")
    for (k, h), name in zip(hits.items(), names):
        print(f"{len(h):3d} {name[:100]}")
        for lab, copies in h[:3]:
            print(f"      {lab} {copies[:2]}")


if __name__ == "__main__":
    main()
