#!/usr/bin/env python3
"""Randomised parity sweep beyond the seeds the test suite runs: tests/test_gpu_parity.py's job generator (every sampler / target family / tuner / monitor,
random launch splitting, chain partitions on streams, chain offsets) for seeds FIRST .. FIRST + COUNT - 1, standard and "wide" (streamed dense, logistic
9..16 parameters) families, each job against the oracle bit for bit.  Prints the failing seeds with the assertion; exit status = number of failures.
usage: random_sweep.py FIRST COUNT [wide|split|logit]        (split: dense targets of 257 .. 1,024 dimensions on the workgroup-split layout; logit: 17 .. 256 parameters on the matrix cores)"""
import sys, traceback
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import test_gpu_parity as T

first, count = int(sys.argv[1]), int(sys.argv[2])
fam = sys.argv[3] if len(sys.argv) > 3 else "standard"
wide = fam == "wide"
bad = []
for seed in range(first, first + count):
    try:
        T._run_random(*T._random_case(seed, wide=wide, split=fam == "split", logit_mfma=fam == "logit"))
    except Exception as e:                     # noqa: BLE001
        bad.append(seed)
        print(f"seed {seed} ({fam}): {type(e).__name__}: {str(e)[:300]}", flush=True)
print(f"{fam} seeds {first}..{first + count - 1}: {count - len(bad)} passed, {len(bad)} failed {bad}", flush=True)
sys.exit(min(len(bad), 100))
