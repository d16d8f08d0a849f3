import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


sys.path.insert(0, str(GOLDEN))


def golden_names():
    """Every case of tests/golden/cases.py (the table make_golden.py writes the fixtures from)."""
    from cases import CASES
    return list(CASES)


def golden_model(name):
    """-> (QincoConfig, state dict) of a golden case: seeded synthetic weights (optionally with a real dataset's normalisation
    constants) or a checkpoint trained and saved by the reference (tests/golden/*.pt)."""
    from cases import case_model
    return case_model(name)


def load_golden(name):
    return dict(np.load(GOLDEN / f"{name}.npz"))


def make_oracle(cfg, sd):
    from oracle.qinco_oracle import OracleQINCo
    return OracleQINCo.from_config(cfg, sd)


def ref_codes(g):
    return g["codes_wrapper"] if "codes_wrapper" in g else g["codes_base"]


def selection_margins(oracle, x):
    """Per row and step m >= 1: the oracle's relative gap at the selection boundaries of that step -- between the last
    kept and the first dropped candidate distance, and (A > 0) between the A-th and (A+1)-th codeword of each beam's
    pre-selection table.  A code row may differ from the oracle only where this gap is at rounding level."""
    trace = {}
    xn = (np.asarray(x, np.float32) - oracle.data_mean) / oracle.data_std
    oracle.encode(xn, trace)
    out = []
    for m in range(1, oracle.M):
        d = np.sort(trace[f"dists{m}"], axis=-1)
        fo = min(oracle.B if m < oracle.M - 1 else 1, d.shape[1] - 1)
        mg = (d[:, fo] - d[:, fo - 1]) / np.maximum(np.abs(d[:, fo]), 1e-12)
        if f"dsub{m}" in trace:
            ds = np.sort(trace[f"dsub{m}"], axis=-1)
            a = trace[f"top{m}"].shape[-1]
            if a < ds.shape[-1]:
                mg = np.minimum(mg, ((ds[..., a] - ds[..., a - 1]) / np.maximum(np.abs(ds[..., a]), 1e-12)).min(axis=1))
        out.append(mg)
    return np.stack(out, axis=1) if out else np.zeros((len(x), 0), np.float32)


def assert_only_near_ties(oracle, x, got, want, near_tie, label=""):
    """got / want: (n, M) code rows.  Every mismatching row must (1) have an oracle selection margin below `near_tie` at or
    after its first differing column (= step) and (2) be reproduced EXACTLY by the oracle when its selections favour that
    row's path by a relative `near_tie` (OracleQINCo.encode(prefer=...)): the row is then an outcome of the reference
    algorithm itself under a rounding-level perturbation, not merely "close".  Returns the number of such rows."""
    bad = np.nonzero((got != want).any(axis=1))[0]
    if len(bad) == 0:
        return 0
    xb = np.asarray(x)[bad].astype(np.float32)
    mg = selection_margins(oracle, xb)
    for r, i in enumerate(bad):
        # (column 0 may differ too: with beams the surviving row can descend from another step-0 candidate -- the flip itself
        # still happened at some later selection; for an IVF model it can be the coarse arg-min, which the replay covers)
        first = int(np.nonzero(got[i] != want[i])[0][0])
        m = float(mg[r, max(first - 1, 0):].min())
        print(f"{label}: row {i} differs from step {first}; oracle margin there {m:.3e}")
        assert (first == 0 and oracle.ivf) or m < near_tie, f"{label}: row {i} differs although the oracle margin is {m:.3e}"
    replay, _ = oracle.encode((xb - oracle.data_mean) / oracle.data_std, prefer=got[bad], tie=near_tie)
    same = (replay.T == got[bad]).all(axis=1)
    assert same.all(), f"{label}: rows {bad[~same].tolist()} are not reachable by the oracle under a {near_tie:g} perturbation"
    return len(bad)


def lut_case_inputs(seed, M, K, D, ivf_K, Mt, n, IVF_M=5):
    """The seeded inputs of a look-up-decoder fixture case: the same draws, in the same order, as tests/golden/make_golden.py
    lut_case_inputs (numpy RandomState streams are stable across numpy versions)."""
    rs = np.random.RandomState(int(seed))
    M, K, D, ivf_K, Mt, n = (int(v) for v in (M, K, D, ivf_K, Mt, n))
    cb = rs.randn(Mt, K * K, D).astype(np.float32)
    comb = rs.randint(0, M + IVF_M, (2, Mt)).astype(np.int64)
    comb[:, 0] = (0, M)
    comb[:, 1] = (1, 2)
    imap = rs.randint(0, K, (ivf_K, IVF_M)).astype(np.int64)
    codes_MB = rs.randint(0, K, (M, n)).astype(np.int64)
    ivf = rs.randint(0, ivf_K, n).astype(np.int64)
    return cb, comb, imap, codes_MB, ivf


def lut_fixture_cases():
    """-> list of (label, codebook_MKD, combine_mvals_m, K_base, ivf_code_map, codes_MB, ivf_codes, mapped, xhat) of
    tests/golden/lut_decoders.npz: what the REFERENCE's PairwiseDecoderIVF.map_codes / forward returned for these inputs."""
    g = load_golden("lut_decoders")
    out = []
    for name in ("small", "wide"):
        kw = {k: int(g[f"pw_{name}_{k}"]) for k in ("seed", "M", "K", "D", "ivf_K", "Mt", "n")}
        cb, comb, imap, codes_MB, ivf = lut_case_inputs(**kw)
        if name == "small":     # stored whole: also pins the regeneration recipe itself
            for got, key in ((cb, "codebook_MKD"), (comb, "combine"), (imap, "ivf_code_map"), (codes_MB, "codes_MB"), (ivf, "ivf")):
                assert np.array_equal(got, g[f"pw_small_{key}"]), key
        out.append((name, cb, comb, kw["K"], imap, codes_MB, ivf, g[f"pw_{name}_mapped"], g[f"pw_{name}_xhat"]))
    return out
