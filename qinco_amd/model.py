"""Drop-in model object for the reference's encode / decode call sites.

`QINCoHIP` exposes the surface that qinco_tasks / search_tasks use on a QINCo / QINCoInferenceWrapper
(SURVEY.md 8b): model(x, step="encode") -> (M, N) int64, model(codes, step="decode") -> (N, D) float32,
.encode(x_norm) -> (codes_MB, xhat_BD), .decode(codes_MB), .built / .build(), .load_state_dict(sd), .state_dict(), .eval(), .train(),
.to(device), .data_mean / .data_std, get_codebooks_refs(), .qinco_model.steps[0].ivf_centroids.weight.  All arithmetic runs in
libqinco_hip (HIP, gfx950); there is no CPU implementation behind this class.

Container in = container out, like the reference's module (qinco_inference.py:272-283):
  * torch CUDA tensor -> torch CUDA tensor on the same device (device pointers straight into the C ABI, asynchronous on the current
    stream);
  * torch CPU tensor -> torch CPU tensor (the reference's `cfg.cpu=true` callers -- task=eval_time asserts cpu,
    qinco_tasks.py:487-492 -- index the result and call `.cpu()` / `.item()` on it, qinco_tasks.py:101-125; the vectors travel to
    the GPU through the pinned host pipeline of qinco_encode_host / qinco_decode_host);
  * numpy array -> numpy array.
When torch is importable the class IS a torch.nn.Module (without parameters: the weights live in the handle's device memory), so
`accelerator.prepare(model)` (qinco_tasks.py:499-505), `unwrap(model)` (qinco/utils.py:230-237), `.eval()`, `.train()`, `.to(...)`,
`torch.no_grad` / `inference_mode` blocks and forward hooks all behave as they do for the reference's model; without torch it is a
plain object with the same methods.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .checkpoint import load_checkpoint, state_dict_to_numpy
from .config import QincoConfig
from .engine import QincoEngine, _is_torch

try:                                    # the Module face needs torch; the arithmetic does not
    import torch as _torch
    _Base = _torch.nn.Module
except ImportError:                     # pragma: no cover -- this image always has torch
    _torch = None

    class _Base:                        # the handful of Module methods the reference's callers use
        training = False

        def __call__(self, *a, **kw):
            return self.forward(*a, **kw)

        def eval(self):
            self.training = False
            return self

        def train(self, mode: bool = True):
            self.training = bool(mode)
            return self

        def to(self, *a, **kw):
            return self


def _host_like(arg, out):
    """A host result in the container the caller passed: torch CPU tensor for a torch CPU tensor, numpy otherwise."""
    if _torch is not None and _is_torch(arg) and not arg.is_cuda and isinstance(out, np.ndarray):
        return _torch.from_numpy(out)
    return out


class QINCoHIP(_Base):
    def __init__(self, cfg: QincoConfig, state_dict: Optional[dict] = None, max_batch: int = 8192,
                 device: Optional[int] = None, split_f16: bool = False):
        super().__init__()
        self.cfg = cfg
        self.max_batch = max_batch
        self.device = device
        self.split_f16 = split_f16   # opt-in: QincoEngine(split_f16=...)
        self.built = False
        self.engine: Optional[QincoEngine] = None
        self._sd: Optional[dict] = None
        if state_dict is not None:
            self.load_state_dict(state_dict)

    # ---- construction ---------------------------------------------------------------------------
    @classmethod
    def from_checkpoint(cls, path: str, A: Optional[int] = None, B: Optional[int] = None, **kw) -> "QINCoHIP":
        cfg, sd = load_checkpoint(path, A, B)
        return cls(cfg, sd, **kw)

    def load_state_dict(self, state_dict: dict, *_, **__):
        """QINCoInferenceWrapper.load_state_dict: load, then rebuild (qinco_inference.py:285-288)."""
        self._sd = state_dict_to_numpy(state_dict)
        self.build()

    def state_dict(self, *_, **__):
        """The weights as loaded (tensors when torch is here): what `save_model` (qinco/utils.py) would write back."""
        if self._sd is None:
            return {}
        if _torch is None:
            return dict(self._sd)
        return {k: _torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else _torch.as_tensor(v) for k, v in self._sd.items()}

    def build(self):
        if self._sd is None:
            raise RuntimeError("build() needs weights: call load_state_dict first")
        if not float(np.asarray(self._sd.get("data_std", 0.0)).reshape(-1)[0]) > 0:
            raise AssertionError("data_std must be > 0")  # qinco_base.py:526
        if self.engine is not None:
            self.engine.close()
        self.engine = QincoEngine(self.cfg, self._sd, max_batch=self.max_batch, device=self.device,
                                  split_f16=self.split_f16)
        self.data_mean = self._sd["data_mean"]
        self.data_std = self._sd["data_std"]
        self.built = True

    @property
    def qinco_model(self):
        """The inner-model attribute path the IVF search reads (search_tasks.py:449):
        `model.qinco_model.steps[0].ivf_centroids.weight` (and `.steps[m].codebook.weight`), as read-only arrays
        (torch tensors when the process has imported torch, like the reference's parameters)."""
        from types import SimpleNamespace as NS

        def wrap(a):
            import sys
            torch = sys.modules.get("torch")   # a tensor for torch users, without importing torch for the others
            return torch.from_numpy(np.ascontiguousarray(a)) if torch is not None else a
        steps = []
        for m in range(self.cfg.M_total):
            st = NS(codebook=NS(weight=wrap(self._sd[f"steps.{m}.codebook.weight"]))) if (
                f"steps.{m}.codebook.weight" in self._sd) else NS()
            if m == 0 and self.cfg.ivf:
                st.ivf_centroids = NS(weight=wrap(self._sd["steps.0.ivf_centroids.weight"]))
            steps.append(st)
        return NS(steps=steps)

    def to(self, *args, **kwargs):
        """A no-op that returns the model (the handle is bound to the GPU it was created on; qinco_inference.py:303's `.to(device)`
        and accelerate's device placement both land here).  Asking for another GPU than the handle's is refused."""
        dev = args[0] if args else kwargs.get("device")
        if _torch is not None and isinstance(dev, (str, _torch.device)):
            dev = _torch.device(dev)
            if dev.type == "cuda" and dev.index is not None and self.engine is not None and self.engine.device is not None \
                    and dev.index != self.engine.device:
                raise ValueError(f"QINCoHIP lives on cuda:{self.engine.device}; build another one on {dev} (QINCoHIP(..., device={dev.index}))")
        return self

    def set_search(self, A: Optional[int] = None, B: Optional[int] = None):
        self.cfg = self.cfg.with_search(A, B)
        self.engine.set_beam(self.cfg.A, self.cfg.B)

    def get_codebooks_refs(self):
        """qinco_base.py:541-549: per step, [codebook] (+ [substep codebook])."""
        refs = []
        for m in range(self.cfg.M_total):
            if m == 0 and self.cfg.ivf:
                continue    # IVFBook steps are skipped by the reference (qinco_base.py:544)
            r = [self._sd[f"steps.{m}.codebook.weight"]]
            k = f"steps.{m}.substep.codebook.weight"
            if k in self._sd:
                r.append(self._sd[k])
            refs.append(r)
        return refs

    # ---- forward --------------------------------------------------------------------------------
    def forward(self, x_in, *args, step: str = "train", **kwargs):
        """QINCoInferenceWrapper.forward (qinco_inference.py:272-283)."""
        assert step in ["encode", "decode"]
        if not self.built:
            raise RuntimeError("model not built")
        if step == "encode":
            return self._t(_host_like(x_in, self.engine.encode(x_in)))
        return _host_like(x_in, self.engine.decode(self._t(x_in)))

    @staticmethod
    def _t(a):
        """(n, M) <-> (M, n): the model speaks (M, N), the ABI and the files (N, M) (search_tasks.py:115)."""
        return a.T if not _is_torch(a) else a.transpose(0, 1)

    def encode(self, x_norm):
        """QINCoInferenceWrapper.encode(x_target_BD) -> (codes_MB, xhat_BD), both in normalised space (:340-350)."""
        codes, xhat = self.engine.encode(x_norm, return_xhat=True, normalised=True)
        return self._t(_host_like(x_norm, codes)), _host_like(x_norm, xhat)

    def decode(self, codes_MB):
        """QINCoInferenceWrapper.decode(codes_MB) -> normalised reconstruction (:330-337)."""
        return _host_like(codes_MB, self.engine.decode(self._t(codes_MB), normalised=True))
