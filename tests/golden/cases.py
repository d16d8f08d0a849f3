"""The golden cases: ONE table for tests/golden/make_golden.py (which writes the fixtures, build container only) and for
tests/conftest.py (which reads them, everywhere).  A case names its model -- a seeded synthetic checkpoint, optionally with the
normalisation constants of a real dataset's magnitude, or a checkpoint file trained and saved by the imported reference
(make_trained.py) -- the number of input rows, and how the inputs are drawn."""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Optional

import numpy as np

from qinco_amd.config import QincoConfig, preset

HERE = Path(__file__).resolve().parent


@dataclass(frozen=True)
class Case:
    cfg: Optional[QincoConfig]        # None: read from the checkpoint
    seed: int
    n: int                            # encode rows stored in the fixture
    regime: Optional[str] = None      # qinco_amd.synth.REGIMES key: data_mean / data_std (and input dtype) of that magnitude
    ckpt: Optional[str] = None        # checkpoint file under tests/golden/ written by the reference's save_model
    data: Optional[str] = None        # trained cases: make_trained.clustered_rows kind ("u8" / "small") of the held-out inputs


CASES = {
    "tiny_proj_beam": Case(QincoConfig(D=32, M=4, K=256, L=2, de=64, dh=96, A=8, B=4), 11, 256),
    "tiny_proj_greedyA": Case(QincoConfig(D=32, M=4, K=256, L=2, de=64, dh=96, A=8, B=1), 12, 256),
    "tiny_id_qinco1": Case(QincoConfig(D=32, M=4, K=256, L=2, de=None, dh=64, A=0, B=1, qinco1_mode=True), 13, 256),
    "tiny_id_A0_beam": Case(QincoConfig(D=32, M=3, K=256, L=1, de=None, dh=64, A=0, B=3, qinco1_mode=False), 14, 64),
    "tiny_proj_dh128": Case(QincoConfig(D=32, M=4, K=256, L=3, de=64, dh=128, A=8, B=4), 18, 256),   # even block counts: the split-fp16 form
    "C1_qinco1_8x8": Case(preset("qinco1", D=128, M=8), 1235, 128),
    "C2_qinco2L_8x8_b8": Case(preset("qinco2-L", D=128, M=8, B=8), 1236, 64),
    "C2_qinco2L_8x8_b1": Case(preset("qinco2-L", D=128, M=8, B=1), 1236, 64),
    "C4_qinco2L_d768_b8": Case(preset("qinco2-L", D=768, M=8, B=8), 1238, 32),     # BASELINE configs[3] at its real depth
    "C3_qinco2L_16x8_b8": Case(preset("qinco2-L", D=128, M=16, B=8), 1237, 64),    # BASELINE configs[2]: M = 16 steps
    "C2_qinco2L_8x8_b32": Case(preset("qinco2-L", D=128, M=8, B=32), 1236, 32),    # the presets' own search width (qinco2-L.yaml:13)
    "tiny_smallK_wideB": Case(QincoConfig(D=32, M=4, K=64, L=2, de=64, dh=96, A=8, B=128), 17, 64),   # B > K: beam grows past beam_0
    "qinco1_d768": Case(preset("qinco1", D=768, M=3), 1241, 32),   # De = D = 768: the 16-row tile kernel's shape
    # IVF-QINCo: coarse step of ivf_K centroids, beam_0 = 1, first QINCo step takes max(A, B)
    "tiny_ivf_beam": Case(QincoConfig(D=32, M=3, K=256, L=2, de=64, dh=96, A=4, B=8, ivf_K=2048), 15, 256),
    "tiny_ivf_greedy_id": Case(QincoConfig(D=32, M=3, K=256, L=2, de=None, dh=64, A=8, B=1, ivf_K=1024), 16, 256),
    "ivf_qinco2S_d128": Case(preset("qinco2-S", D=128, M=4, B=8, ivf_K=65536), 1240, 64),
    # the reference's normalisation regimes with inputs in the datasets' own storage types (round 3): uint8 rows go through
    # torch.from_numpy(uint8).to(float32) on the reference side (search_tasks.py:109-110), through the GPU's byte path here
    "norm_bigann_u8": Case(preset("qinco2-S", D=128, M=4, B=8), 1301, 128, regime="bigann"),
    "norm_ssnpp_u8": Case(preset("qinco2-S", D=256, M=3, B=8), 1302, 96, regime="ssnpp"),
    "norm_contriever": Case(preset("qinco2-S", D=768, M=3, B=8), 1303, 48, regime="contriever"),
    "norm_deep_qinco1": Case(preset("qinco1", D=96, M=3, L=4), 1304, 96, regime="deep"),
    # checkpoints TRAINED by the imported reference on clustered data (make_trained.py) and written by its save_model
    "trained_qinco2S": Case(None, 2101, 192, ckpt="trained_qinco2S.pt", data="u8"),
    "trained_qinco2S_b1": Case(None, 2101, 128, ckpt="trained_qinco2S.pt", data="u8"),     # greedy on the same weights
    "trained_qinco1": Case(None, 2102, 128, ckpt="trained_qinco1.pt", data="small"),
    "trained_tiny_proj": Case(None, 2104, 192, ckpt="trained_tiny_proj.pt", data="small"),    # De != D: trained projections
    "trained_ivf_qinco2S": Case(None, 2103, 192, ckpt="trained_ivf_qinco2S.pt", data="u8"),  # frozen coarse codebook of 2048 in front
    # the headline kernel's own shape (de = dh = 384, L = 16), trained by the reference: beam search and the greedy override
    "trained_qinco2L": Case(None, 2105, 128, ckpt="trained_qinco2L.pt", data="u8"),
    "trained_qinco2L_b1": Case(None, 2105, 128, ckpt="trained_qinco2L.pt", data="u8"),
    "trained_qinco2L_d768": Case(None, 2106, 64, ckpt="trained_qinco2L_d768.pt", data="small"),   # ... on 768-d data (C4's kernel instance)
}
# Rows of a fixture on which the product's codes sit on the other side of a rounding-level tie of the reference, per arithmetic
# form, as MEASURED on the MI355X (profiles/r04_golden_tie_counts.txt: 0 on every fixture in both forms).  The GPU parity tests
# assert the count, so a regression from k to k + 1 tie-side rows fails instead of passing as "only near ties"; a case that is
# not listed expects 0.  Greedy fixtures (B == 1: qinco_inference.py:126, north_star "bit-identical") are compared with
# np.array_equal before anything else, whatever this table says.
EXPECTED_TIE_ROWS = {"fp32": {}, "split_f16": {}}


def expected_tie_rows(name: str, form: str = "fp32") -> int:
    return EXPECTED_TIE_ROWS[form].get(name, 0)


SEARCH_OVERRIDE = {"trained_qinco2S_b1": dict(B=1), "trained_qinco2L_b1": dict(B=1)}      # CLI-style override of the stored search width (utils.py:166-172)


def case_model(name: str):
    """-> (QincoConfig, state dict of fp32 numpy arrays) of a case."""
    from qinco_amd.synth import apply_regime, synth_state_dict
    c = CASES[name]
    if c.ckpt:
        from qinco_amd.checkpoint import load_checkpoint
        return load_checkpoint(str(HERE / c.ckpt), **SEARCH_OVERRIDE.get(name, {}))
    sd = synth_state_dict(c.cfg, c.seed)
    if c.regime:
        sd = apply_regime(c.cfg, sd, c.regime, c.seed)
    return c.cfg, sd


def clustered_rows(kind: str, n: int, D: int, seed: int, part: int = 0) -> np.ndarray:
    """Synthetic stand-in for a descriptor dataset: 48 clusters with their own low-rank anisotropic covariance, Laplace
    (heavy-tailed) coefficients, a per-dimension offset profile.  kind "u8": SIFT-like non-negative bytes; kind "small":
    float rows of magnitude ~0.1 (deep1M / contriever-like normalisation constants).  The mixture is a function of `seed`;
    `part` selects an independent draw of rows from it (0 = the training rows, others = held-out rows for the fixtures)."""
    rs = np.random.RandomState(seed)
    nc, r = 48, 12
    centres = rs.randn(nc, D) * 0.9
    bases = rs.randn(nc, r, D) / np.sqrt(r) * (0.3 + 1.4 * rs.rand(nc, r, 1))
    profile_u8 = 12.0 + 55.0 * rs.rand(D) ** 2           # per-dimension offset: a few tens, some dimensions much larger
    profile_small = 0.02 * rs.randn(D)
    rs = np.random.RandomState(seed + 10007 * (part + 1))
    which = rs.randint(0, nc, n)
    coef = rs.laplace(size=(n, r)) * 0.8
    z = centres[which] + np.einsum("nr,nrd->nd", coef, bases[which]) + 0.25 * rs.randn(n, D)
    if kind == "u8":
        return np.clip(np.rint(profile_u8 + 33.0 * z), 0, 255).astype(np.uint8)
    return (profile_small + 0.085 * z).astype(np.float32)


def rerank_lut_tables(cfg, sd):
    """Look-up tables of the rerank_ivf fixture's mid re-ranker, a function of the model's codebooks (rebuilt by the test instead of
    stored): pairs of QINCo steps (0, 1), (1, 2) and one pair of ivf_code_map columns (M, M + 1) -- map_codes appends those
    behind the M step columns (pairwise_decoder.py:126-130).  -> (tables (3, K^2, D) float32, combine_mvals_m (2, 3) int64)."""
    K, d, M = cfg.K, cfg.D, cfg.M
    cb = [np.asarray(sd[f"steps.{m + 1}.codebook.weight"], np.float32) for m in range(M)]
    pairs = [(0, 1), (1, 2), (M, M + 1)]
    tabs = [(cb[0][:, None, :] + 0.5 * cb[1][None, :, :]), (cb[1][:, None, :] + 0.5 * cb[2][None, :, :]),
            (0.25 * cb[0][:, None, :] - 0.125 * cb[2][None, :, :])]
    tables = np.stack([t.reshape(K * K, d) for t in tabs]).astype(np.float32)
    return tables, np.array([[a for a, _ in pairs], [b for _, b in pairs]], np.int64)
