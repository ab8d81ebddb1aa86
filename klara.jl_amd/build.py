"""In-tree builds: libklara_hip.so (hipcc, gfx950 only) and the CPU oracle (gcc, test infrastructure)."""
from __future__ import annotations

import os
import subprocess
from pathlib import Path

_HERE = Path(__file__).resolve().parent
_ROOT = _HERE.parent


def _make(directory: Path, *targets: str) -> None:
    jobs = str(min(8, os.cpu_count() or 1))
    cmd = ["make", "-C", str(directory), f"-j{jobs}", *targets]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{' '.join(cmd)} failed:\n{r.stdout[-4000:]}\n{r.stderr[-4000:]}")


def build_library() -> Path:
    """hipcc --offload-arch=gfx950 ... -> klara.jl_amd/lib/libklara_hip.so (cross-compiles without a GPU)."""
    _make(_HERE / "csrc")
    out = _HERE / "lib" / "libklara_hip.so"
    if not out.exists():
        raise RuntimeError("libklara_hip.so was not produced")
    return out


def build_oracle() -> Path:
    """gcc -> oracle/libklara_oracle.so.  The oracle is only ever loaded by tests / smoke / bench baseline."""
    _make(_ROOT / "oracle")
    out = _ROOT / "oracle" / "libklara_oracle.so"
    if not out.exists():
        raise RuntimeError("libklara_oracle.so was not produced")
    return out
