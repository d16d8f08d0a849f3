"""Checkpoint reader for the reference's .pt layout (qinco/utils.py:100-200).

File = torch.save({"epoch", "model": state_dict, "optimizer", "scheduler", "logger",
                   "parameters": {K, M, de, dh, L, A, B, ivf_in_use, ivf_K, qinco1_mode} (None-valued keys are
                   absent, e.g. `de` for QINCo1), "data_dim": D}).
PyTorch is used here only as the weight loader (torch.load(map_location="cpu", weights_only=True), :161-163).
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .config import QincoConfig


def config_from_checkpoint(ckpt: dict, A: Optional[int] = None, B: Optional[int] = None) -> QincoConfig:
    """Hyper-parameters come from ckpt["parameters"]; explicit A / B override the stored ones like the CLI does
    (utils.py:166-172), and asking for A > 0 on an A = 0 model raises the reference's ValueError."""
    if "parameters" not in ckpt:
        raise ValueError("Missing model parameters is acceptable only for converting a model!")  # utils.py:175-177
    p = ckpt["parameters"]
    stored_A = int(p.get("A") or 0)
    if A is not None and A > 0 and not stored_A:
        raise ValueError("Can't evaluate a model trained with A=0 (no candidates pre-selection) "
                         "using a non-zero A value.")
    return QincoConfig(D=int(ckpt["data_dim"]), M=int(p["M"]), K=int(p["K"]), L=int(p["L"]),
                       de=(int(p["de"]) if p.get("de") else None), dh=int(p["dh"]),
                       A=stored_A if A is None else int(A), B=int(p.get("B") or 1) if B is None else int(B),
                       qinco1_mode=bool(p.get("qinco1_mode", False)),
                       ivf_K=(int(p["ivf_K"]) if p.get("ivf_in_use") and p.get("ivf_K") else None))


def state_dict_to_numpy(sd: dict) -> dict:
    out = {}
    for k, v in sd.items():
        k = k.replace("module.", "")  # load_model strips DDP prefixes (utils.py:195-199)
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        v = np.asarray(v, dtype=np.float32)
        out[k] = np.ascontiguousarray(v) if v.ndim else v   # keep data_std 0-d
    return out


def load_checkpoint(path: str, A: Optional[int] = None, B: Optional[int] = None):
    """-> (QincoConfig, {name: fp32 ndarray})."""
    import torch
    ckpt = torch.load(str(path), map_location=torch.device("cpu"), weights_only=True)
    cfg = config_from_checkpoint(ckpt, A, B)
    return cfg, state_dict_to_numpy(ckpt["model"])


def save_checkpoint(path: str, cfg: QincoConfig, sd: dict) -> None:
    """Write a checkpoint the reference's load_saved_model_data / load_model accept (same keys as save_model)."""
    import torch
    torch.save({"epoch": None, "model": {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()},
                "optimizer": None, "scheduler": None, "logger": None,
                "parameters": cfg.parameters_dict(), "data_dim": cfg.D}, str(path))
