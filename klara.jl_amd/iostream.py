"""`:destination => :iostream` — CSV sink in the reference's on-disk format.

Reference: one file per monitored field in `filepath` — `value.csv`, `logtarget.csv`, `gradlogtarget.csv`,
`diagnosticvalues.csv` — one line per saved step, fields comma-joined with Julia's shortest round-trip float
printing and `true`/`false` for the accept diagnostics
(src/iostreams/ParameterIOStreams/BasicContParamIOStream.jl:64-82 file names, :152-159 `write(iostream, state)`;
src/jobs/jobs.jl:193-202).  One job with N chains writes one such directory per chain: `filepath` itself when
N == 1 (identical to the reference), `filepath/chain_<c>` (1-based, zero-padded) otherwise.
"""
from __future__ import annotations

import math
import os
from decimal import Decimal
from typing import Iterable, Optional

import numpy as np


def julia_float_repr(x: float) -> str:
    """Julia's `string(::Float64)`: shortest round-trip digits; fixed notation for 1e-4 <= |x| < 1e6, else
    `d.ddde±x`; always at least one fractional digit."""
    x = float(x)                                   # numpy scalars repr() as 'np.float64(..)'
    if math.isnan(x):
        return "NaN"
    if math.isinf(x):
        return "Inf" if x > 0 else "-Inf"
    if x == 0.0:
        return "-0.0" if math.copysign(1.0, x) < 0 else "0.0"
    sign = "-" if x < 0 else ""
    t = Decimal(repr(abs(x))).as_tuple()           # Python's repr is also the shortest round-trip representation
    ds = "".join(map(str, t.digits)).rstrip("0") or "0"
    pt = len(t.digits) + t.exponent                # digits before the decimal point
    if -4 < pt <= 6:                               # 0.0001 -> "0.0001", 0.00001 -> "1.0e-5", 999999.0 fixed, 1.0e6
        if pt <= 0:
            body = "0." + "0" * (-pt) + ds
        elif pt >= len(ds):
            body = ds + "0" * (pt - len(ds)) + ".0"
        else:
            body = ds[:pt] + "." + ds[pt:]
    else:
        body = ds[0] + "." + (ds[1:] or "0") + "e" + str(pt - 1)
    return sign + body


def _line(values: Iterable[float]) -> str:
    return ",".join(julia_float_repr(float(v)) for v in values) + "\n"


def write_chain(directory: str, suffix: str, value: Optional[np.ndarray], logtarget: Optional[np.ndarray],
                gradlogtarget: Optional[np.ndarray], accept: Optional[np.ndarray]) -> None:
    """value / gradlogtarget: (D, n) as in the NState; logtarget: (n,); accept: (n,) bools."""
    os.makedirs(directory, exist_ok=True)
    if value is not None:
        with open(os.path.join(directory, f"value.{suffix}"), "w") as f:
            f.writelines(_line(value[:, i]) for i in range(value.shape[1]))
    if logtarget is not None:
        with open(os.path.join(directory, f"logtarget.{suffix}"), "w") as f:
            f.writelines(julia_float_repr(float(v)) + "\n" for v in logtarget)
    if gradlogtarget is not None:
        with open(os.path.join(directory, f"gradlogtarget.{suffix}"), "w") as f:
            f.writelines(_line(gradlogtarget[:, i]) for i in range(gradlogtarget.shape[1]))
    if accept is not None:
        with open(os.path.join(directory, f"diagnosticvalues.{suffix}"), "w") as f:
            f.writelines(("true" if a else "false") + "\n" for a in accept)


class ChainWriter:
    """The open files of one chain's BasicContParamIOStream (BasicContParamIOStream.jl:64-82: one per monitored field, mode "w"):
    `append` writes the lines of newly saved steps, `flush` is `flush(iostream)` (:141-149), so a long job streams to disk with
    bounded memory instead of holding every saved step (jobs.jl:17-29 `:flush`)."""

    def __init__(self, directory: str, suffix: str, value: bool, logtarget: bool, gradlogtarget: bool, accept: bool,
                 likelihood_prior: bool = False):
        os.makedirs(directory, exist_ok=True)
        mk = lambda name, on: open(os.path.join(directory, f"{name}.{suffix}"), "w") if on else None
        # one file per monitored field, named after the field (BasicContParamIOStream.jl:64-82)
        self.files = {"value": mk("value", value), "loglikelihood": mk("loglikelihood", likelihood_prior), "logprior": mk("logprior", likelihood_prior),
                      "logtarget": mk("logtarget", logtarget),
                      "gradlogtarget": mk("gradlogtarget", gradlogtarget), "diagnosticvalues": mk("diagnosticvalues", accept)}

    def append(self, value=None, logtarget=None, gradlogtarget=None, accept=None, loglikelihood=None, logprior=None) -> None:
        f = self.files
        for name, series in (("loglikelihood", loglikelihood), ("logprior", logprior)):
            if f[name] is not None and series is not None:
                f[name].writelines(julia_float_repr(float(v)) + "\n" for v in series)
        if f["value"] is not None and value is not None:
            f["value"].writelines(_line(value[:, i]) for i in range(value.shape[1]))
        if f["logtarget"] is not None and logtarget is not None:
            f["logtarget"].writelines(julia_float_repr(float(v)) + "\n" for v in logtarget)
        if f["gradlogtarget"] is not None and gradlogtarget is not None:
            f["gradlogtarget"].writelines(_line(gradlogtarget[:, i]) for i in range(gradlogtarget.shape[1]))
        if f["diagnosticvalues"] is not None and accept is not None:
            f["diagnosticvalues"].writelines(("true" if a else "false") + "\n" for a in accept)

    def flush(self) -> None:
        for fh in self.files.values():
            if fh is not None:
                fh.flush()

    def close(self) -> None:
        for fh in self.files.values():
            if fh is not None:
                fh.close()


class ContMuvMarkovChain:
    """What `read(iostream, T)` hands back (BasicContParamIOStream.jl:254-281 -> read!, :215-252): the monitored fields of one chain,
    `value` / `gradlogtarget` as (size x n) matrices — one CSV line per saved step, transposed on reading (:224) —, `loglikelihood` /
    `logprior` / `logtarget` as vectors of length n (:219), `diagnosticvalues` as a (keys x n) matrix of the parsed `true` / `false`
    fields (:249-251); fields that were not monitored (no file) are empty, as in the reference."""

    def __init__(self, size: int, n: int):
        self.size, self.n = int(size), int(n)
        self.value = np.zeros((0, 0)); self.gradlogtarget = np.zeros((0, 0))
        self.loglikelihood = np.zeros(0); self.logprior = np.zeros(0); self.logtarget = np.zeros(0)
        self.diagnosticvalues = np.zeros((0, 0), dtype=bool)
        self.diagnostickeys = []


def read_chain(directory: str, suffix: str = "csv", diagnostickeys=("accept",)) -> ContMuvMarkovChain:
    """`read(BasicContParamIOStream(size, n, ...; filepath=directory, mode="r"), Float64)` for the fields this build writes: parses
    whichever of value / loglikelihood / logprior / logtarget / gradlogtarget / diagnosticvalues files exist in `directory`.  Julia's
    float printing round-trips (`julia_float_repr`), so a job written by the :iostream sink reads back bit for bit."""
    def rows(name):
        path = os.path.join(directory, f"{name}.{suffix}")
        if not os.path.exists(path):
            return None
        with open(path) as f:
            return [ln.rstrip("\n").split(",") for ln in f if ln.strip() != ""]

    def num(tok: str) -> float:
        return {"Inf": math.inf, "-Inf": -math.inf, "NaN": math.nan}.get(tok, None) if tok in ("Inf", "-Inf", "NaN") else float(tok)

    mats, vecs = {}, {}
    for name in ("value", "gradlogtarget"):
        r = rows(name)
        if r is not None:
            mats[name] = np.array([[num(t) for t in ln] for ln in r], dtype=np.float64).T.copy() if r else np.zeros((0, 0))
    for name in ("loglikelihood", "logprior", "logtarget"):
        r = rows(name)
        if r is not None:
            vecs[name] = np.array([num(ln[0]) for ln in r], dtype=np.float64)
    n = next((m.shape[1] for m in mats.values()), next((v.size for v in vecs.values()), 0))
    size = next((m.shape[0] for m in mats.values()), 0)
    chain = ContMuvMarkovChain(size, n)
    for name, m in mats.items():
        setattr(chain, name, m)
    for name, v in vecs.items():
        setattr(chain, name, v)
    d = rows("diagnosticvalues")
    if d is not None:
        chain.diagnosticvalues = np.array([[t.strip() == "true" for t in ln] for ln in d], dtype=bool).T.copy() if d else np.zeros((0, 0), dtype=bool)
        chain.diagnostickeys = list(diagnostickeys)[:chain.diagnosticvalues.shape[0]]
        if n == 0:
            chain.n = chain.diagnosticvalues.shape[1]
    return chain
