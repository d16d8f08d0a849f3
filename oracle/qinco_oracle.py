"""CPU oracle for the QINCo / QINCo2 encode-decode path -- TEST INFRASTRUCTURE ONLY.

This is a plain-numpy (fp32) restatement of the reference algorithm, written from the reference's behaviour
with every function citing the reference file:line it follows (paths relative to the upstream repo root).
It is the checker for the HIP engine: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import it.  The product (qinco_amd) never routes through this file and fails loudly when the HIP
library is missing.

Parity pin: tests/test_oracle_vs_golden.py checks this oracle against tests/golden/*.npz, which were
produced by importing the reference itself (tests/golden/make_golden.py, run in the build container where
/root/reference exists): codes bit-equal, reconstructions to <=1e-5 relative.

State dict = {reference parameter name: np.ndarray fp32} (qinco/model/qinco_base.py:229-260, 432-445).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------------------------------
# distances (qinco/utils.py)
# --------------------------------------------------------------------------------------------------
def approx_pairwise_distance(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """utils.py:336-346: |a|^2[:,None] + |b|^2 - 2 a @ b.T  (no clamp; may be slightly negative)."""
    anorms = (a * a).sum(-1, dtype=F32)
    bnorms = (b * b).sum(-1, dtype=F32)
    return (anorms[:, None] + bnorms) - F32(2) * (a @ b.T)


def approx_compute_batch_distances(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """utils.py:377-383 with a = x[:,None,:] (n,1,D), b = candidates (n,C,D) -> (n,C)."""
    anorms = (a * a).sum(-1, dtype=F32)  # (n,1)
    bnorms = (b * b).sum(-1, dtype=F32)  # (n,C)
    ab = np.matmul(a, b.transpose(0, 2, 1))[:, 0, :]  # (n,C)
    return (anorms + bnorms) - F32(2) * ab


def topk_smallest(d: np.ndarray, k: int) -> np.ndarray:
    """dists.topk(k, largest=False).indices (ascending).  Ties broken by lower index (stable sort);
    torch leaves the order of exact ties unspecified, argmin returns the first minimum."""
    return np.argsort(d, axis=-1, kind="stable")[..., :k]


# --------------------------------------------------------------------------------------------------
# the codeword MLP  f(c, xhat)
# --------------------------------------------------------------------------------------------------
class StepWeights:
    """Parameters of one QINCoStep (qinco_base.py:207-260)."""

    def __init__(self, sd: dict, m: int, L: int):
        p = f"steps.{m}."
        if m == 0 and p + "ivf_centroids.weight" in sd:  # IVFBook (qinco_base.py:128-196)
            self.codebook = sd[p + "ivf_centroids.weight"]
            self.sub_codebook = None
            return
        self.codebook = sd[p + "codebook.weight"]
        self.sub_codebook = sd.get(p + "substep.codebook.weight")
        if m == 0:
            return
        self.cat_w = sd[p + "concat.mlp.weight"]
        self.cat_b = sd[p + "concat.mlp.bias"]
        self.in_proj = sd.get(p + "in_proj.weight")
        self.out_proj = sd.get(p + "out_proj.weight")
        self.up = [sd[p + f"residual_blocks.{l}.up_proj.weight"] for l in range(L)]
        self.down = [sd[p + f"residual_blocks.{l}.down_proj.weight"] for l in range(L)]


def step_forward(w: StepWeights, c: np.ndarray, xhat: np.ndarray, qinco1_mode: bool) -> np.ndarray:
    """QINCoInferenceStep.forward (qinco_inference.py:31-40) = QINCoStep.forward (qinco_base.py:262-280).

    c, xhat: (..., D).  z = in_proj(c); z = z + Linear([z ; xhat]) (QConcat.forward qinco_base.py:60-64,
    cat order (z, xhat)); L x  z = z + down(relu(up(z))) (QBlockFFN.forward :93-97);
    out = out_proj(z) + coeff * c with coeff = 0 in qinco1_mode else 1 (qinco_inference.py:29,40).
    """
    z = c if w.in_proj is None else c @ w.in_proj.T
    cc = np.concatenate([z, np.broadcast_to(xhat, z.shape[:-1] + xhat.shape[-1:])], axis=-1)
    z = z + (cc @ w.cat_w.T + w.cat_b)
    for up, down in zip(w.up, w.down):
        h = np.maximum(z @ up.T, F32(0))
        z = z + h @ down.T
    out = z if w.out_proj is None else z @ w.out_proj.T
    if not qinco1_mode:
        out = out + c
    return out


def step_forward_torch(w: StepWeights, c: np.ndarray, xhat: np.ndarray, qinco1_mode: bool) -> np.ndarray:
    """step_forward with the same op sequence on torch CPU tensors (ATen matmul / relu / add: multi-threaded,
    like the reference's own CPU path).  Same restatement, different array library: used by bench.py's
    cpu_baseline so that the reported CPU figure is not limited by numpy's single-threaded element-wise ops."""
    import torch
    t = torch.from_numpy
    with torch.no_grad():
        ct = t(np.ascontiguousarray(c))
        z = ct if w.in_proj is None else ct @ t(w.in_proj).T
        xh = t(np.ascontiguousarray(np.broadcast_to(xhat, z.shape[:-1] + xhat.shape[-1:])))
        z = z + (torch.cat([z, xh], dim=-1) @ t(w.cat_w).T + t(w.cat_b))
        for up, down in zip(w.up, w.down):
            z = z + torch.relu(z @ t(up).T) @ t(down).T
        out = z if w.out_proj is None else z @ t(w.out_proj).T
        if not qinco1_mode:
            out = out + ct
    return out.numpy()


# --------------------------------------------------------------------------------------------------
# model
# --------------------------------------------------------------------------------------------------
class OracleQINCo:
    """Oracle twin of QINCoInferenceWrapper (qinco_inference.py:257-353) on the CPU fp32 path."""

    def __init__(self, sd: dict, *, M: int, K: int, L: int, A: int, B: int, qinco1_mode: bool, ivf: bool = False,
                 backend: str = "numpy"):
        """M = number of steps including the IVF step (cfg._M_ivf).  backend: "numpy" (the pinned checker) or "torch"
        (the codeword MLP on torch CPU ops, see step_forward_torch)."""
        assert backend in ("numpy", "torch")
        self._mlp = step_forward if backend == "numpy" else step_forward_torch
        self.sd = {k: np.ascontiguousarray(np.asarray(v, dtype=F32)) for k, v in sd.items()
                   if not k.endswith(("xtarget_mean", "xtarget_var"))}
        self.M, self.K, self.L, self.A, self.B = M, K, L, A, B
        self.qinco1_mode = bool(qinco1_mode)
        self.ivf = bool(ivf)
        self.data_mean = self.sd["data_mean"]
        self.data_std = F32(self.sd["data_std"])
        assert self.data_std > 0, "data_std must be > 0 (qinco_base.py:526)"
        self.steps = [StepWeights(self.sd, m, L) for m in range(M)]
        self.D = self.steps[0].codebook.shape[1]
        self._prefer, self._tie = None, F32(0)
        if A > 0 and M > 1 and self.steps[1].sub_codebook is None:
            raise ValueError("Can't evaluate a model trained with A=0 (no candidates pre-selection) "
                             "using a non-zero A value.")  # utils.py:169-172

    @classmethod
    def from_config(cls, cfg, sd: dict, backend: str = "numpy") -> "OracleQINCo":
        """cfg: any object with M_total, K, L, A, B, qinco1_mode, ivf (qinco_amd.config.QincoConfig)."""
        return cls(sd, M=cfg.M_total, K=cfg.K, L=cfg.L, A=cfg.A, B=cfg.B, qinco1_mode=cfg.qinco1_mode, ivf=cfg.ivf,
                   backend=backend)

    # ---- forward (qinco_inference.py:272-283) ---------------------------------------------------
    def __call__(self, x_in, step: str):
        assert step in ("encode", "decode")
        if step == "encode":
            x = (np.asarray(x_in, dtype=F32) - self.data_mean) / self.data_std
            codes, _ = self.encode(x)
            return codes
        xhat = self.decode(np.asarray(x_in))
        return xhat * self.data_std + self.data_mean

    # ---- encode (QINCoInferenceEncoder.forward qinco_inference.py:239-254) -----------------------
    def encode(self, x: np.ndarray, trace: dict | None = None, prefer: np.ndarray | None = None, tie: float = 2e-5):
        """prefer (test diagnostics only; None = the reference algorithm): (n, M) code rows.  Every selection then favours the
        candidates that lie on the preferred row's path by a relative `tie` of their distance -- a rounding-level
        perturbation.  If the result equals `prefer`, that row is an outcome the reference algorithm itself reaches when
        its near-ties fall the other way; this is how the tests decide whether a differing GPU row is legitimate."""
        n, D = x.shape
        M, K, A, B = self.M, self.K, self.A, self.B
        self._prefer, self._tie = (None if prefer is None else np.asarray(prefer).T), F32(tie)   # (M, n)
        K0 = self.steps[0].codebook.shape[0]
        beam_0 = 1 if (M == 1 or self.ivf) else min(B, K0)  # :237; a one-step model must end with one beam
        d0 = approx_pairwise_distance(x, self.steps[0].codebook)
        if self._prefer is not None:
            d0 = self._favour(d0, np.arange(K0)[None, :] == self._prefer[0][:, None])
        codes0 = topk_smallest(d0, beam_0)  # argmin when beam_0 == 1 (:243-245)
        xhat = self.steps[0].codebook[codes0]  # (n, F, D)
        codes = codes0[None]  # (1, n, F)
        if trace is not None:
            trace["d0"] = d0
        for m in range(1, M):
            F_out = B if m < M - 1 else 1  # :152
            if A > 0:
                # first QINCo step of an IVF model needs max(A, B) candidates (qinco_base.py:108-112)
                n_codes = max(A, B) if (self.ivf and m == 1) else A
                xhat, codes = self._step_preselect(self.steps[m], x, xhat, codes, n_codes, F_out, trace, m)
            else:
                xhat, codes = self._step_all(self.steps[m], x, xhat, codes, F_out, trace, m)
        return codes[:, :, 0].astype(np.int64), xhat[:, 0, :]

    def _favour(self, d, mask):
        return np.where(mask, d - self._tie * np.abs(d), d).astype(F32)

    def _on_path(self, codes):
        """(n, F): beams whose code history equals the preferred row's prefix."""
        return (codes == self._prefer[: codes.shape[0], :, None]).all(axis=0)

    def _step_preselect(self, w, x, xhat, codes, A, F_out, trace, m):
        """QINCoInferenceStepEncoder.forward (qinco_inference.py:156-224)."""
        n, F, D = xhat.shape
        Mc = codes.shape[0]
        xtarget = x[:, None, :] - xhat  # :171
        d_sub = approx_pairwise_distance(xtarget.reshape(n * F, D), w.sub_codebook)  # :172
        if self._prefer is not None:
            want = np.repeat(self._prefer[m], F)
            d_sub = self._favour(d_sub, self._on_path(codes).reshape(n * F, 1) & (np.arange(d_sub.shape[1])[None, :] == want[:, None]))
        top = topk_smallest(d_sub, A)  # (n*F, A) :173
        cw = w.codebook[top].reshape(n, F, A, D)  # :175
        out = self._mlp(w, cw, xhat[:, :, None, :], self.qinco1_mode)  # :178-188
        cand = out + xhat[:, :, None, :]  # :190-191
        cand_flat = cand.reshape(n, F * A, D)
        dists = approx_compute_batch_distances(x[:, None, :], cand_flat)  # :194-199
        dists_ref = dists
        if self._prefer is not None:
            sel = self._on_path(codes)[:, :, None] & (top.reshape(n, F, A) == self._prefer[m][:, None, None])
            dists = self._favour(dists, sel.reshape(n, F * A))
        F_out = min(F_out, F * A)
        idx = topk_smallest(dists, F_out)  # (n, F_out) :200
        real = np.take_along_axis(top.reshape(n, F * A), idx, axis=-1)  # :203-204
        hist = np.repeat(codes, A, axis=-1)  # :207  (Mc, n, F*A): parent beam = idx // A
        hist = np.take_along_axis(hist, np.broadcast_to(idx[None], (Mc, n, F_out)), axis=-1)  # :208-210
        xnext = np.take_along_axis(cand_flat, idx[:, :, None], axis=1)  # :213-219
        if trace is not None:
            trace[f"top{m}"] = top.reshape(n, F, A)
            trace[f"dsub{m}"] = d_sub.reshape(n, F, -1)
            trace[f"dists{m}"] = dists_ref
        return xnext, np.concatenate([hist, real[None]], axis=0)  # :222

    def _step_all(self, w, x, xhat, codes, F_out, trace, m):
        """A = 0: every codeword is a candidate.  QINCoInferenceStepEncoderNoSubstep.forward
        (qinco_inference.py:78-140) is greedy only; for F_out > 1 this follows the base model
        QINCoStep.encode (qinco_base.py:325-372: realidx = idx % K, parents = idx // K)."""
        n, F, D = xhat.shape
        K = w.codebook.shape[0]
        Mc = codes.shape[0]
        cw = np.broadcast_to(w.codebook[None, None], (n, F, K, D))
        out = self._mlp(w, cw, xhat[:, :, None, :], self.qinco1_mode)
        cand = out + xhat[:, :, None, :]
        cand_flat = cand.reshape(n, F * K, D)
        dists = approx_compute_batch_distances(x[:, None, :], cand_flat)
        dists_ref = dists
        if self._prefer is not None:
            sel = self._on_path(codes)[:, :, None] & (np.arange(K)[None, None, :] == self._prefer[m][:, None, None])
            dists = self._favour(dists, sel.reshape(n, F * K))
        F_out = min(F_out, F * K)
        idx = topk_smallest(dists, F_out)
        real = idx % K
        hist = np.repeat(codes, K, axis=-1)
        hist = np.take_along_axis(hist, np.broadcast_to(idx[None], (Mc, n, F_out)), axis=-1)
        xnext = np.take_along_axis(cand_flat, idx[:, :, None], axis=1)
        if trace is not None:
            trace[f"dists{m}"] = dists_ref
        return xnext, np.concatenate([hist, real[None]], axis=0)

    # ---- decode (QINCoInferenceDecoder.forward qinco_inference.py:66-75) --------------------------
    def decode(self, codes_MB: np.ndarray) -> np.ndarray:
        codes_MB = np.asarray(codes_MB)
        assert codes_MB.shape[0] == self.M
        for m in range(self.M):
            if codes_MB[m].size and (codes_MB[m].min() < 0 or codes_MB[m].max() >= self.steps[m].codebook.shape[0]):
                raise IndexError("code out of range")
        xhat = self.steps[0].codebook[codes_MB[0]].copy()
        for m in range(1, self.M):
            w = self.steps[m]
            xhat = xhat + self._mlp(w, w.codebook[codes_MB[m]], xhat, self.qinco1_mode)  # :72-74
        return xhat


def mse(x: np.ndarray, xhat: np.ndarray, mse_scale: float = 1.0) -> float:
    """AnyVectMSE (qinco/metrics.py:41-58): sum_i |x_i - xhat_i|^2 * mse_scale / N."""
    diff = np.asarray(x, dtype=np.float64) - np.asarray(xhat, dtype=np.float64)
    return float((diff * diff).sum() * mse_scale / len(x))


# --------------------------------------------------------------------------------------------------
# look-up decoders downstream of the path.  Pinned since round 5: tests/golden/lut_decoders.npz and the mid stage
# of tests/golden/rerank_ivf.npz hold what the reference's own PairwiseDecoderIVF.forward / map_codes and
# reconstruct_from_fixed_codebooks returned (tests/golden/make_golden.py imports qinco/search/{pairwise_decoder,search_utils}.py
# with empty placeholder modules for the faiss / torcheval imports none of those lines uses);
# tests/test_oracle_vs_golden.py::test_oracle_lookup_decoders_equal_the_reference checks these restatements bit for bit.
# --------------------------------------------------------------------------------------------------
def reconstruct_from_fixed_codebooks(codes: np.ndarray, codebooks: np.ndarray) -> np.ndarray:
    """search_utils.py:105-115."""
    M = codes.shape[1]
    assert codebooks.shape[0] == M
    recons = codebooks[0, codes[:, 0]].astype(F32).copy()
    for m in range(1, M):
        recons += codebooks[m, codes[:, m]]
    return recons


def pairwise_decoder_forward(codes_MB, ivf_codes, codebook_MKD, combine_mvals_m, K_base, ivf_code_map=None) -> np.ndarray:
    """PairwiseDecoderIVF.map_codes (:126-130) + forward (:88-93)."""
    codes_MB = np.asarray(codes_MB)
    if ivf_code_map is not None:
        assert ivf_codes.ndim == 1
        codes_MB = np.concatenate([codes_MB, ivf_code_map[ivf_codes].T])
    comb = codes_MB[combine_mvals_m[0]] * K_base + codes_MB[combine_mvals_m[1]]
    xhat = codebook_MKD[0][comb[0]].astype(F32).copy()
    for cb, c in zip(codebook_MKD[1:], comb[1:]):
        xhat += cb[c]
    return xhat


# --------------------------------------------------------------------------------------------------
# f3: small-db search.  run_search_full_direct_small_db (qinco/search/search_tasks.py:551-603) and
# compute_recalls (:275-282).  search_tasks.py needs faiss to import; the golden tests/golden/search_small_db.npz was
# produced by driving these lines with the reference's own model and approx_pairwise_distance (make_golden.py).
# --------------------------------------------------------------------------------------------------
def small_db_shortlists(xhat_db: np.ndarray, queries: np.ndarray, nshort: int = 100, bs: int = 100):
    """search_tasks.py:585-597: per query batch, approx_pairwise_distance to all reconstructions, argsort, first
    `nshort`.  Returns (ids (Q, nshort) int64, the matching distances)."""
    ids, dd = [], []
    for i0 in range(0, len(queries), bs):
        d = approx_pairwise_distance(queries[i0:i0 + bs].astype(F32), xhat_db.astype(F32))
        order = np.argsort(d, axis=-1, kind="stable")[:, :nshort]
        ids.append(order)
        dd.append(np.take_along_axis(d, order, axis=-1))
    return np.concatenate(ids).astype(np.int64), np.concatenate(dd).astype(F32)


def compute_recalls(I: np.ndarray, gt: np.ndarray) -> dict:
    """search_tasks.py:275-282."""
    assert I.ndim == 2 and gt.ndim == 2
    return {rank: float((I[:, :rank] == gt[:, :1]).sum() / gt.shape[0]) for rank in (1, 10, 100)}
