"""Hyper-parameters of a QINCo / QINCo2 model on the encode/decode path.

Mirrors the reference's checkpoint["parameters"] block (qinco/utils.py:100-137) and the model_args presets
(config/model_args/{qinco1,qinco2-S,qinco2-M,qinco2-L}.yaml).
"""
from __future__ import annotations

from dataclasses import dataclass, replace
from typing import Optional


@dataclass(frozen=True)
class QincoConfig:
    D: int                      # cfg._D, checkpoint["data_dim"]
    M: int = 8                  # steps
    K: int = 256                # codebook size
    L: int = 16                 # residual blocks per step
    de: Optional[int] = None    # embedding dim; None = D (QINCo1)
    dh: int = 256               # hidden dim
    A: int = 0                  # pre-selected candidates (0 = all K)
    B: int = 1                  # beam size
    qinco1_mode: bool = False   # res_codeword_coeff = 0 (qinco_inference.py:29)
    ivf_K: Optional[int] = None  # IVF-QINCo: step 0 is a frozen coarse codebook of ivf_K centroids (ivf_in_use)

    @property
    def De(self) -> int:
        return self.de or self.D

    @property
    def ivf(self) -> bool:
        return bool(self.ivf_K)

    @property
    def M_total(self) -> int:
        """cfg._M_ivf: columns of the code matrix (IVF id first) (qinco_tasks.py:379-383)."""
        return self.M + (1 if self.ivf else 0)

    @property
    def K_vals(self) -> list:
        """cfg._K_vals."""
        return ([self.ivf_K] if self.ivf else []) + [self.K] * self.M

    def n_codes(self, m: int) -> int:
        """Candidates pre-selected at step m: A, or max(A, B) for the first QINCo step of an IVF model
        (QincoSubstep._n_codes, qinco_base.py:108-112)."""
        if self.A and self.ivf and m == 1:
            return max(self.A, self.B)
        return self.A

    def with_search(self, A: Optional[int] = None, B: Optional[int] = None) -> "QincoConfig":
        """CLI-style override of A / B (utils.py:166-172: A > 0 is illegal on an A = 0 model)."""
        A = self.A if A is None else A
        B = self.B if B is None else B
        if A > 0 and self.A == 0:
            raise ValueError("Can't evaluate a model trained with A=0 (no candidates pre-selection) "
                             "using a non-zero A value.")
        return replace(self, A=A, B=B)

    def parameters_dict(self) -> dict:
        """The "parameters" entry save_model would write (None-valued keys are dropped there)."""
        d = {"K": self.K, "M": self.M, "dh": self.dh, "L": self.L, "A": self.A, "B": self.B,
             "qinco1_mode": self.qinco1_mode}
        if self.de is not None:
            d["de"] = self.de
        if self.ivf:
            d["ivf_in_use"] = True
            d["ivf_K"] = self.ivf_K
        return d

    # algorithmic FLOPs
    def mlp_flops_per_row(self) -> float:
        f = 2.0 * (self.De + self.D) * self.De + 4.0 * self.L * self.De * self.dh
        if self.De != self.D:
            f += 4.0 * self.D * self.De
        return f

    def encode_flops_per_vector(self) -> float:
        Mt = self.M_total
        total = 2.0 * self.D * self.K_vals[0]
        F = 1 if (Mt == 1 or self.ivf) else min(self.B, self.K)
        for m in range(1, Mt):
            Ae = self.n_codes(m) or self.K
            total += F * Ae * self.mlp_flops_per_row()
            if self.A:
                total += F * self.K * 2.0 * self.D
            total += F * Ae * 2.0 * self.D
            F = min(self.B if m < Mt - 1 else 1, F * Ae)
        return total

    def decode_flops_per_vector(self) -> float:
        return (self.M_total - 1) * self.mlp_flops_per_row()


def preset(name: str, D: int, M: int = 8, **over) -> QincoConfig:
    """config/model_args presets.  B defaults to the presets' value (32 for QINCo2); BASELINE.json uses B=8."""
    base = {
        "qinco1": dict(L=16, de=None, dh=256, A=0, B=1, qinco1_mode=True),
        "qinco2-S": dict(L=2, de=128, dh=256, A=16, B=32),
        "qinco2-M": dict(L=4, de=384, dh=384, A=16, B=32),
        "qinco2-L": dict(L=16, de=384, dh=384, A=16, B=32),
    }[name]
    base.update(over)
    return QincoConfig(D=D, M=M, K=256, **base)   # `over` may carry ivf_K (IVF-qinco2_* models)


# The BASELINE.json configs (SURVEY.md section 8 constants).
BASELINE_CONFIGS = {
    "C1": preset("qinco1", D=128, M=8),
    "C2": preset("qinco2-L", D=128, M=8, B=8),
    "C3": preset("qinco2-L", D=128, M=16, B=8),
    "C4": preset("qinco2-L", D=768, M=8, B=8),
    # IVF-qinco2 models of the reference's large-scale search (README "IVF-qinco2_*"), ivf_K = 2^20
    "IVF_L": preset("qinco2-L", D=128, M=8, B=8, ivf_K=1 << 20),
    "IVF_S": preset("qinco2-S", D=128, M=8, B=8, ivf_K=1 << 20),
    # the smaller presets of the reference (config/model_args/qinco2-S.yaml, qinco2-M.yaml) on BigANN-shaped data
    "S": preset("qinco2-S", D=128, M=8, B=8),
    "M": preset("qinco2-M", D=128, M=8, B=8),
    # qinco2-S on the other datasets' dimensions (Deep1B 96, FB-ssnpp 256, Contriever 768: in/out projections)
    "S_d96": preset("qinco2-S", D=96, M=8, B=8),
    "S_d256": preset("qinco2-S", D=256, M=8, B=8),
    "S_d768": preset("qinco2-S", D=768, M=8, B=8),
    # QINCo1 on 768-d data (De = D = 768): the 16-row tile kernel (csrc/mlp16_kernel.hpp)
    "Q1_768": preset("qinco1", D=768, M=8),
}
