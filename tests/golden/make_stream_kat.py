"""Writes tests/golden/stream_kat.json: known-answer values of the FROZEN random stream (include/klara_hip.h, KLARA_ABI_VERSION 6).

    python tests/golden/make_stream_kat.py          # refuses to overwrite an existing file unless --force is given

For a few (seed, global chain id, transition) keys: the raw Philox block of slot 0, the D = 100 and D = 7 proposal normals of the transition
as the samplers draw them (hex floats: bit patterns), its accept uniform, and the slice sampler's draws of two coordinates (log-uniform's
uniform, runiform, six shrink attempts).  Generated with the host build of detmath.h (the oracle); tests/test_stream_joint.py requires the host
build, the device build and the independent NumPy restatement (to libm accuracy) to reproduce them.  The file only changes together with
KLARA_ABI_VERSION."""
import ctypes as C
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import oracle_ffi as O  # noqa: E402
from klara_jl_amd import _lib as L  # noqa: E402

KEYS = [(20260927, 0, 0), (20260927, 65535, 999), (1, (1 << 33) + 7, 12345), ((1 << 63) + 11, 3, (1 << 40) - 1)]
OUT = ROOT / "tests" / "golden" / "stream_kat.json"


def entry(seed, chain, t):
    lib = O.load()
    e = {"seed": seed, "chain": chain, "transition": t, "block_slot0": [int(v) for v in O.stream_blocks(seed, chain, t, [0])[0]]}
    for d in (100, 7):
        z = np.empty(d); u = C.c_double(0.0)
        lib.ko_transition_normals(seed, chain, t, d, z.ctypes.data, C.byref(u))
        e[f"normals_d{d}"] = [float(v).hex() for v in z]
        e[f"accept_uniform_d{d}"] = float(u.value).hex()
    for i in (0, 99):
        out = np.empty(8)
        lib.ko_slice_draws(seed, chain, t, i, 6, out.ctypes.data)
        e[f"slice_draws_coord{i}"] = [float(v).hex() for v in out]
    return e


if __name__ == "__main__":
    if OUT.exists() and "--force" not in sys.argv:
        sys.exit(f"{OUT} exists: the stream is frozen (include/klara_hip.h).  --force only together with a new KLARA_ABI_VERSION.")
    doc = {"abi_version": int(L.KLARA_ABI_VERSION), "entries": [entry(*k) for k in KEYS]}
    OUT.write_text(json.dumps(doc, indent=1) + "\n")
    print("wrote", OUT)
