"""Deterministic synthetic checkpoints and inputs in the reference's state-dict layout.

No trained weights or datasets are available offline, so parity and throughput are
measured on seeded synthetic models.  Everything is drawn from numpy RandomState so the GPU box regenerates
bit-identical weights without torch RNG.  Parameter names: qinco/model/qinco_base.py:229-260, 432-445.
"""
from __future__ import annotations

import numpy as np

from .config import QincoConfig

F32 = np.float32


def synth_state_dict(cfg: QincoConfig, seed: int = 1234, gain: float = 0.6,
                     data_std: float = 2.0, xhat_gain: float = 0.15) -> dict:
    rs = np.random.RandomState(seed)
    D, De, Dh = cfg.D, cfg.De, cfg.dh
    sd: dict = {}
    sd["data_mean"] = (0.5 * rs.randn(D)).astype(F32)
    sd["data_std"] = np.asarray(data_std, dtype=F32)

    def lin(o, i):
        return (rs.randn(o, i) * (gain / np.sqrt(i))).astype(F32)

    # codebook scale per step: 0.6^m up to 8 steps; deeper models decay more slowly so that the last step
    # is as large as an 8-step model's (0.6^8), else fp32 distances of the deep steps tie exactly by the hundreds
    decay = 0.6 if cfg.M <= 8 else 0.6 ** (8.0 / cfg.M)
    for m in range(cfg.M_total):
        p = f"steps.{m}."
        if m == 0 and cfg.ivf:
            # IVFBook (qinco_base.py:128-196): a frozen coarse codebook, already in normalised space
            sd[p + "ivf_centroids.weight"] = rs.randn(cfg.ivf_K, D).astype(F32)
            continue
        cb = (rs.randn(cfg.K, D) * (decay ** m)).astype(F32)
        sd[p + "codebook.weight"] = cb
        sd[p + "xtarget_mean"] = np.zeros(D, F32)   # training buffers, unused at inference
        sd[p + "xtarget_var"] = np.ones(D, F32)
        if m == 0:
            continue
        if cfg.A > 0:
            noise = rs.randn(cfg.K, D).astype(F32) * F32(0.1 * float(cb.std()))
            sd[p + "substep.codebook.weight"] = (cb + noise).astype(F32)
        wc = lin(De, De + D)
        wc[:, De:] *= F32(xhat_gain)   # keep f(c, xhat) dominated by c, as in a trained residual quantiser
        sd[p + "concat.mlp.weight"] = wc
        sd[p + "concat.mlp.bias"] = (0.05 * rs.randn(De)).astype(F32)
        for l in range(cfg.L):
            sd[p + f"residual_blocks.{l}.up_proj.weight"] = lin(Dh, De)
            sd[p + f"residual_blocks.{l}.down_proj.weight"] = lin(De, Dh)
        if De != D:
            sd[p + "in_proj.weight"] = lin(De, D)
            sd[p + "out_proj.weight"] = lin(D, De)
    return sd


def synth_vectors(cfg: QincoConfig, sd: dict, n: int, seed: int = 42) -> np.ndarray:
    """S0 inputs: x = mean + std * N(0, I) in fp32 (normalises to ~N(0, I))."""
    rs = np.random.RandomState(seed)
    z = rs.randn(n, cfg.D).astype(F32)
    return (z * F32(sd["data_std"]) + sd["data_mean"]).astype(F32)


def synth_codes(cfg: QincoConfig, n: int, seed: int = 7) -> np.ndarray:
    """Uniform random codes (M, n) int64 for decode tests / the structured S1 inputs."""
    rs = np.random.RandomState(seed)
    return np.stack([rs.randint(0, k, size=n) for k in cfg.K_vals]).astype(np.int64)


# Normalisation regimes of the reference's datasets (magnitudes of qinco/qinco_tasks.py:516-527: BigANN bytes with per-dimension
# means of 11-93 and a global std of 36.59; FB-ssnpp bytes around 128 with std 22.1; Contriever floats with std 0.0583; Deep1B
# floats with std 0.102).  Synthetic values of those magnitudes -- the tables themselves are not reproduced.
REGIMES = {
    "bigann": dict(dtype="u8", std=36.5888),
    "ssnpp": dict(dtype="u8", std=22.1006),
    "contriever": dict(dtype="f32", std=0.0583),
    "deep": dict(dtype="f32", std=0.1020),
}


def apply_regime(cfg: QincoConfig, sd: dict, regime: str, seed: int = 0) -> dict:
    """Replace data_mean / data_std of a synthetic checkpoint by constants of a real dataset's magnitude."""
    rs = np.random.RandomState(seed + 7001)
    D = cfg.D
    mean = {"bigann": lambda: 12.0 + 55.0 * rs.rand(D) ** 2,
            "ssnpp": lambda: 128.0 + 0.04 * rs.randn(D),
            "contriever": lambda: -0.015 + 0.01 * rs.randn(D),
            "deep": lambda: 0.03 * rs.randn(D)}[regime]()
    out = dict(sd)
    out["data_mean"] = mean.astype(F32)
    out["data_std"] = np.asarray(REGIMES[regime]["std"], dtype=F32)
    return out


def regime_vectors(cfg: QincoConfig, sd: dict, n: int, regime: str, seed: int = 42) -> np.ndarray:
    """Inputs in the dataset's own storage type: uint8 rows (clipped to [0, 255], like SIFT bytes) or float32 rows."""
    x = synth_vectors(cfg, sd, n, seed)
    if REGIMES[regime]["dtype"] == "u8":
        return np.clip(np.rint(x), 0, 255).astype(np.uint8)
    return x
