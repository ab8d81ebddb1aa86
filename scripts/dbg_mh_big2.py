#!/usr/bin/env python3
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import klara_jl_amd as K
from klara_jl_amd import _lib as L
import oracle_ffi as O
import cases
import test_gpu_parity as T
np.set_printoptions(linewidth=250)
c, rng = T._random_case(11, wide=True)
for nst in (1, 2, 3):
    eng = K.Engine(**cases.engine_kwargs(c, monitor=L.MON_ACCEPT, steps_per_launch=0, nstreams=1))
    job = O.OracleJob(**cases.oracle_kwargs(c, layout=eng.layout()))
    eng.set_state(c["x0"]) if c["x0"] is not None else eng.init_state_normal()
    job.set_state(c["x0"]) if c["x0"] is not None else job.init_state_normal()
    eng.run(nst); job.run(nst)
    x, lt, g = eng.state()
    m = eng.accept_mask()
    print("nst", nst, "gpu mask\n", m, "\noracle\n", job.accept)
    for ch in range(x.shape[0]):
        d = np.flatnonzero(x[ch] != job.X[ch])
        print(" chain", ch, "ndiff", d.size, "first idx", d[:12], "lt", lt[ch], job.LT[ch])
    eng.close()
print("---- values")
eng = K.Engine(**cases.engine_kwargs(c, monitor=L.MON_ACCEPT, steps_per_launch=0, nstreams=1))
job = O.OracleJob(**cases.oracle_kwargs(c, layout=eng.layout()))
eng.set_state(c["x0"]) if c["x0"] is not None else eng.init_state_normal()
job.set_state(c["x0"]) if c["x0"] is not None else job.init_state_normal()
x0 = eng.state()[0].copy()
eng.run(2); job.run(2)
x = eng.state()[0]
sg = np.asarray(c["mh_sigma"])
for ch in (5,):
    print("x0   ", x0[ch, 56:68])
    print("gpu  ", x[ch, 56:68])
    print("orac ", job.X[ch, 56:68])
    print("zgpu ", (x[ch, 56:68] - x0[ch, 56:68]) / sg[56:68])
    print("zora ", (job.X[ch, 56:68] - x0[ch, 56:68]) / sg[56:68])
# the normals of transition 0 and 1 for that chain from the oracle's stream
import ctypes as C
lib = O.load()
for t in (0, 1):
    z = np.zeros(c["target"].precision.shape[0]); u = np.zeros(1)
    lib.ko_transition_normals(C.c_uint64(c["seed"]), C.c_uint64(5), C.c_uint64(t), 192, z.ctypes.data, u.ctypes.data)
    print("oracle z t=%d" % t, z[56:68])
