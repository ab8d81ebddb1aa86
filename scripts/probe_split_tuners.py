import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import klara_jl_amd as K
from klara_jl_amd import _lib as L
n=65536
for d in (512, 1024, 320):
    for tag, kw in (("plain", {}), ("rate", dict(tuner=L.TUNER_ACCEPT_RATE, targetrate=0.7, period=10)), ("da", dict(tuner=L.TUNER_DUAL_AVERAGING, targetrate=0.7, da_nadapt=10**6))):
        e = K.Engine(target=K.GaussDenseTarget.compound_symmetric(d, 0.5), nchains=n, nsteps=10**6, steps_per_launch=4, sampler=L.SAMPLER_HMC, leapstep=0.1*(256/d)**0.25, nleaps=10, **kw)
        e.init_state_normal(); e.run(8)
        r=[]
        for _ in range(3):
            e.run(16); ms, nl = e.last_run_ms(); r.append(ms)
        # leapfrogs per transition: plain 10; da: lambda/eps per chain (varies) -> report ms per 16 transitions
        print(f"HMC {tag} D={d}: ms per 16 transitions {' '.join('%.2f' % v for v in r)}  attrs {e.kernel_attributes(0, 4)[:2]}", flush=True)
        e.close()
