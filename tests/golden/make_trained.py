"""Train the IMPORTED reference model for a few hundred optimiser steps and save it with the reference's own save_model
(build container only: needs /root/reference; the resulting .pt files travel as fixtures, the reference does not).

    python tests/golden/make_trained.py            # -> tests/golden/trained_qinco2S.pt, trained_qinco1.pt

Why: every other fixture uses qinco_amd.synth (Gaussian weights at one gain).  A trained network has statistics that
generator cannot give -- dead ReLU units, down-projections that grew from zero, correlated codebooks refined from a residual
quantiser, pre-selection codebooks that track the main ones, activation scales set by real normalisation constants -- and
the split-fp16 form chooses its operand scalings from the weights.  The reference trains in this container without faiss:
QINCo.forward(step="train") (qinco/model/qinco_base.py:487-503, 524-539) needs torch + einops only, and
initialize_qinco_codebooks (:27-44) takes any list of centroids.  What is done here, with the reference's own code for every
model-side piece:
  * data: a clustered, anisotropic, heavy-tailed mixture (numpy), in two regimes -- "bigann-like" uint8 rows (non-negative,
    clipped at 0 / 255, per-dimension means of a few tens, global std of a few tens: the magnitudes of
    qinco_tasks.py:516-527) and "deep-like" small-magnitude float rows (std ~0.1);
  * data_mean / data_std: per-dimension mean and global std of the training rows, as QincoTrainTask does
    (qinco_tasks.py:430-431);
  * codebooks: a numpy residual k-means (the RQ the reference gets from faiss), handed to initialize_qinco_codebooks;
  * optimisation: losses of model(batch, step="train") summed like aggregate_losses (qinco_tasks.py:171-176), AdamW at the
    reference's learning rate with its gradient clipping (qinco_cfg.yaml:34-37) for QINCo2, Adam 1e-4 for QINCo1
    (qinco1.yaml:18-21) -- scaled up a little so that a few hundred steps move the weights;
  * saved through qinco.utils.save_model (utils.py:100-137).
Only the checkpoint (data) is written.
"""
from __future__ import annotations

import sys
import time
from pathlib import Path
from types import SimpleNamespace as NS

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(HERE))
sys.path.insert(0, "/root/reference")

from cases import clustered_rows  # noqa: E402  (pure numpy: also used by the tests, which never import the reference)
from qinco.model import QINCo  # noqa: E402  (the reference)
from qinco.model.qinco_base import IVFBook, initialize_qinco_codebooks  # noqa: E402
from qinco.utils import SharedCfgState, save_model  # noqa: E402

torch.set_num_threads(8)


class Acc:
    device = torch.device("cpu")
    is_main_process = True
    num_processes = 1
    process_index = 0

    def print(self, *a, **k):
        pass


def numpy_rq(xn: np.ndarray, M: int, K: int, seed: int, iters: int = 8) -> list:
    """Residual k-means in the normalised space (stands in for the faiss RQ of the reference's training recipe)."""
    rs = np.random.RandomState(seed)
    res = xn.astype(np.float32).copy()
    books = []
    for _ in range(M):
        cent = res[rs.choice(len(res), K, replace=False)].copy()
        for _ in range(iters):
            d = (res * res).sum(1)[:, None] + (cent * cent).sum(1)[None] - 2.0 * res @ cent.T
            a = d.argmin(1)
            for k in range(K):
                sel = a == k
                cent[k] = res[sel].mean(0) if sel.any() else res[rs.randint(len(res))]
        d = (res * res).sum(1)[:, None] + (cent * cent).sum(1)[None] - 2.0 * res @ cent.T
        res = res - cent[d.argmin(1)]
        books.append(cent.astype(np.float32))
    return books


SPECS = {
    # qinco2-S-shaped (config/model_args/qinco2-S.yaml: de 128, dh 256, L 2, A 16) on bigann-like bytes
    "trained_qinco2S": dict(kind="u8", D=128, M=4, K=256, L=2, de=128, dh=256, A=16, B=8, qinco1_mode=False,
                            steps=1500, batch=256, lr=8e-4, opt="adamw", clip=0.1, seed=2101),
    # QINCo1-shaped (qinco1.yaml: de null, dh 256, A 0, B 1, qinco1_mode) with L = 4 on small-magnitude floats
    "trained_qinco1": dict(kind="small", D=128, M=4, K=256, L=4, de=None, dh=256, A=0, B=1, qinco1_mode=True,
                           steps=700, batch=192, lr=4e-4, opt="adam", clip=0.0, seed=2102),
    # De != D: trained in_proj / out_proj (the S shape has Identity there), on a small geometry (32 -> 64, hidden 128: the one small geometry with a split-fp16 instance too)
    "trained_tiny_proj": dict(kind="small", D=32, M=4, K=256, L=2, de=64, dh=128, A=8, B=8, qinco1_mode=False,
                              steps=1200, batch=256, lr=8e-4, opt="adamw", clip=0.1, seed=2104),
    # IVF-QINCo2-S-shaped: a frozen coarse codebook of 2048 k-means centroids in front (qinco_tasks.py:277-300), M = 4 steps after it
    "trained_ivf_qinco2S": dict(kind="u8", D=128, M=4, K=256, L=2, de=128, dh=256, A=16, B=8, qinco1_mode=False, ivf_K=2048,
                                steps=1000, batch=256, lr=8e-4, opt="adamw", clip=0.1, seed=2103),
    # the HEADLINE kernel's own shape (qinco2-L: de = dh = 384, sixteen residual blocks) with one QINCo step behind step 0: sixteen
    # blocks deep is where trained activation growth meets the folded head's fp32 association and the split form's scalings
    "trained_qinco2L": dict(kind="u8", D=128, M=2, K=256, L=16, de=384, dh=384, A=16, B=8, qinco1_mode=False,
                            steps=1200, batch=96, lr=8e-4, opt="adamw", clip=0.1, seed=2105),
    # ... and on 768-d data (contriever-like small-magnitude floats): the same blocks between trained in_proj / out_proj of 768 x 384
    "trained_qinco2L_d768": dict(kind="small", D=768, M=2, K=256, L=16, de=384, dh=384, A=16, B=8, qinco1_mode=False,
                                 steps=600, batch=64, lr=8e-4, opt="adamw", clip=0.1, seed=2106),
}


def numpy_kmeans(x: np.ndarray, K: int, seed: int, iters: int = 10) -> np.ndarray:
    """Plain Lloyd iterations on raw rows (stands in for the faiss k-means that makes the reference's IVF centroid files)."""
    rs = np.random.RandomState(seed)
    cent = x[rs.choice(len(x), K, replace=False)].astype(np.float32).copy()
    for _ in range(iters):
        a = ((x * x).sum(1)[:, None] + (cent * cent).sum(1)[None] - 2.0 * x @ cent.T).argmin(1)
        for k in range(K):
            sel = a == k
            cent[k] = x[sel].mean(0) if sel.any() else x[rs.randint(len(x))]
    return cent


def train(name: str) -> Path:
    s = SPECS[name]
    torch.manual_seed(s["seed"])
    D, M, K = s["D"], s["M"], s["K"]
    xtr = clustered_rows(s["kind"], 24576, D, s["seed"]).astype(np.float32)
    mean, std = xtr.mean(0).astype(np.float32), float(xtr.std())                   # qinco_tasks.py:430-431
    ivf_K = s.get("ivf_K")
    cfg = SharedCfgState(dict(output=str(HERE / f"{name}.pt"), K=K, M=M, de=s["de"], dh=s["dh"], L=s["L"], A=s["A"], B=s["B"],
                              ivf_in_use=(True if ivf_K else None), ivf_K=ivf_K, qinco1_mode=s["qinco1_mode"], task="train",
                              enc_max_bs=1 << 20, codebook_noise_init=0.1, inference=False, batch=s["batch"]))
    cfg._D, cfg._M_ivf, cfg._K_vals, cfg._ivf_book = D, M + (1 if ivf_K else 0), ([ivf_K] if ivf_K else []) + [K] * M, None
    cfg._qinco_jit = False
    cfg._accelerator = Acc()
    cfg._data_mean, cfg._data_std = mean, std
    cfg._cur_epoch = cfg._optimizer = cfg._scheduler = cfg._melog = None
    xn = (xtr[:16384] - mean) / std
    if ivf_K:   # initialize_model (qinco_tasks.py:277-300): raw-space centroids, stored normalised with the data's mean / std
        cent = numpy_kmeans(xtr[:16384], ivf_K, s["seed"] + 3)
        cent_n = ((cent - mean) / std).astype(np.float32)
        cfg._ivf_book = IVFBook(cfg, cent_n)
        a = ((xn * xn).sum(1)[:, None] + (cent_n * cent_n).sum(1)[None] - 2.0 * xn @ cent_n.T).argmin(1)
        xn = xn - cent_n[a]                        # the QINCo steps quantise what the coarse centroid leaves
    model = QINCo(cfg)
    books = numpy_rq(xn, M, K, s["seed"] + 1)
    # initialize_qinco_codebooks expects centroids in DATA space: without IVF it normalises step 0 with the mean, everything
    # else by std only; with IVF the list is indexed by the non-IVF steps (get_codebooks_refs skips the IVFBook)
    rq = [torch.from_numpy(b * std + (mean if (m == 0 and not ivf_K) else 0.0)).float() for m, b in enumerate(books)]
    initialize_qinco_codebooks(cfg, model, rq)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = (torch.optim.AdamW if s["opt"] == "adamw" else torch.optim.Adam)(params, lr=s["lr"])
    model.train()
    xt = torch.from_numpy(xtr)
    rs = np.random.RandomState(s["seed"] + 2)
    t0 = time.time()
    for it in range(s["steps"]):
        batch = xt[torch.from_numpy(rs.choice(len(xt), s["batch"], replace=False))]
        _, _, losses = model(batch, step="train")
        loss = torch.sum(torch.stack(list(losses.values())))                       # aggregate_losses (qinco_tasks.py:171-176)
        opt.zero_grad()
        loss.backward()
        if s["clip"]:
            torch.nn.utils.clip_grad_value_(params, s["clip"])                     # qinco_tasks.py:197
        opt.step()
        if it % 50 == 0 or it == s["steps"] - 1:
            print(f"{name}: step {it:4d}  " + "  ".join(f"{k} {float(v):.4f}" for k, v in losses.items())
                  + f"   ({time.time() - t0:.0f} s)", flush=True)
    model.eval()
    save_model(cfg, Acc(), model)
    sd = model.state_dict()
    dead = []
    with torch.no_grad():   # how un-synthetic did it get: dead hidden units on a probe batch, weight scales
        xb = (xt[:512] - model.data_mean) / model.data_std
        codes, _ = model.encode(xb)
        xhat = torch.zeros_like(xb)
        for m, st in enumerate(model.steps):
            if m:
                z = st.concat(st.in_proj(st.codebook(codes[m])), xhat)
                for blk in st.residual_blocks:
                    h = torch.relu(blk.up_proj(z))
                    dead.append(float((h.max(0).values <= 0).float().mean()))
                    z = z + blk.down_proj(h)
            xhat = xhat + st.decode(codes[m], xhat)
    print(f"{name}: saved {cfg.output} ({Path(cfg.output).stat().st_size / 1e6:.1f} MB); data_std {std:.4f}, "
          f"|down_proj| max {max(float(v.abs().max()) for k, v in sd.items() if 'down_proj' in k):.3f}, "
          f"dead hidden units per block {np.round(dead, 3).tolist()}")
    return Path(cfg.output)


if __name__ == "__main__":
    for n in (sys.argv[1:] or list(SPECS)):
        train(n)
