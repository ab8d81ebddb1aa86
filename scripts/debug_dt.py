import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import numpy as np
import cases, oracle_ffi as O
import klara_jl_amd as K
from klara_jl_amd import _lib as L
for name in sys.argv[1:]:
    case = cases.make_case(name)
    eng = K.Engine(**cases.engine_kwargs(case, monitor=L.MON_ACCEPT, steps_per_launch=1))
    layout = eng.layout()
    job = O.OracleJob(**cases.oracle_kwargs(case, layout=layout))
    eng.init_state_normal(); job.init_state_normal()
    eng.run(1); job.run(1)
    m = eng.accept_mask()[0]; x, lt, g = eng.state()
    print(name, layout, "mask gpu", m[:16], "oracle", job.accept[0][:16])
    bad = np.where(m != job.accept[0])[0]
    print(" differing chains", bad[:20], "of", m.size)
    same = np.where(m == job.accept[0])[0]
    print(" on agreeing chains: x equal", np.array_equal(x[same], job.X[same]), "lt equal", np.array_equal(lt[same], job.LT[same]),
          "max |dlt|", np.max(np.abs(lt[same] - job.LT[same])) if same.size else None)
    acc_same = [c for c in same if m[c]]
    if acc_same:
        c = acc_same[0]
        print(" chain", c, "lt gpu/oracle", lt[c], job.LT[c], "x diff idx", np.where(x[c] != job.X[c])[0][:10])
