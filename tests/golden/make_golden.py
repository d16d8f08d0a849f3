"""Generate the golden vectors under tests/golden/ by IMPORTING the reference (build container only).

    python tests/golden/make_golden.py            # needs /root/reference (never present on the GPU box)

For each case: a seeded synthetic checkpoint (qinco_amd.synth, numpy RandomState -> regenerated bit-identically
by the tests) is loaded into the reference's QINCo model; inputs are run through BOTH reference
implementations (QINCoInferenceWrapper = TorchScript inference path, and the base QINCo model) on the CPU fp32
path, and the outputs are stored: codes (N, M), decoded vectors, the normalised reconstruction returned by
encode, per-step pre-selection ids and the candidate-distance margin at each selection (to flag near-ties).
Also writes tiny_ckpt.pt through the reference's own save_model to pin the checkpoint layout reader.

Only data (inputs / expected outputs) is written; no reference source is copied.
"""
from __future__ import annotations

import os
import sys
from pathlib import Path
from types import SimpleNamespace as NS

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, "/root/reference")

from qinco.model import QINCo, QINCoInferenceWrapper  # noqa: E402  (the reference)
from qinco.model.qinco_base import IVFBook  # noqa: E402

sys.path.insert(0, str(HERE))
from cases import CASES, case_model, clustered_rows, rerank_lut_tables  # noqa: E402  (the case table, shared with tests/conftest.py)
from qinco_amd.config import QincoConfig  # noqa: E402
from qinco_amd.synth import regime_vectors, synth_codes, synth_vectors  # noqa: E402
from oracle.qinco_oracle import OracleQINCo  # noqa: E402

torch.set_num_threads(8)


class Acc:
    device = torch.device("cpu")
    is_main_process = True
    num_processes = 1
    process_index = 0

    def print(self, *a, **k):
        pass


def ref_cfg(cfg: QincoConfig, batch: int = 64):
    return NS(A=cfg.A, B=cfg.B, K=cfg.K, L=cfg.L, de=cfg.de, dh=cfg.dh, M=cfg.M, _D=cfg.D, _M_ivf=cfg.M_total,
              _K_vals=list(cfg.K_vals), _ivf_book=None, qinco1_mode=cfg.qinco1_mode, _qinco_jit=False,
              _accelerator=Acc(), task="eval", enc_max_bs=65536, ivf_in_use=(True if cfg.ivf else None), batch=batch,
              codebook_noise_init=0.1, ivf_K=cfg.ivf_K, inference=True)


def build_reference(cfg: QincoConfig, sd: dict):
    rc = ref_cfg(cfg)
    with torch.no_grad():
        if cfg.ivf:  # initialize_model: cfg._ivf_book = IVFBook(cfg, centroids) (qinco_tasks.py:277-285)
            rc._ivf_book = IVFBook(rc, np.asarray(sd["steps.0.ivf_centroids.weight"]))
        model = QINCo(rc)
        missing = model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
    model.eval()
    wrapper = None
    if not (cfg.A == 0 and cfg.B > 1):  # the inference wrapper is greedy-only for A = 0
        wrapper = QINCoInferenceWrapper(rc, model)
        wrapper.build()
    return model, wrapper


def run_case(name: str) -> dict:
    c = CASES[name]
    cfg, sd = case_model(name)
    seed, n = c.seed, c.n
    model, wrapper = build_reference(cfg, sd)
    oracle = OracleQINCo.from_config(cfg, sd)

    if c.ckpt:          # held-out rows of the mixture the checkpoint was trained on
        x0 = clustered_rows(c.data, n, cfg.D, seed, part=1)
    elif c.regime:      # the dataset's own storage type (uint8 / float32) at its normalisation magnitude
        x0 = regime_vectors(cfg, sd, n, c.regime, seed=seed + 1)
    else:
        x0 = synth_vectors(cfg, sd, n, seed=seed + 1)
    # S1 "structured" half: x = decode(random codes) + small noise, so that beams really compete
    rc = synth_codes(cfg, n // 2, seed=seed + 2)
    with torch.no_grad():
        xs = model(torch.from_numpy(rc), step="decode").numpy()
    xs = xs + 0.05 * float(sd["data_std"]) * np.random.RandomState(seed + 3).randn(*xs.shape)
    if x0.dtype == np.uint8:
        xs = np.clip(np.rint(xs), 0, 255)
    x = np.concatenate([x0[: n - n // 2], xs.astype(x0.dtype)])

    out = {"x": x}
    wrapper_ok = wrapper is not None
    with torch.no_grad():
        xt = torch.from_numpy(x).to(torch.float32)      # uint8 rows: search_tasks.py:109-110
        try:
            codes_base = model(xt, step="encode").numpy()  # (M, N)
        except RuntimeError as e:   # the base model's step 0 asks topk for B > K entries; the inference wrapper clamps beam_0
            assert cfg.B > cfg.K and wrapper_ok, e
            codes_base = None
        if wrapper_ok:
            codes_w = wrapper(xt, step="encode").numpy()
            _, xhat_norm = wrapper.encode((xt - wrapper.data_mean) / wrapper.data_std)
            out["codes_wrapper"] = codes_w.T.copy()
            out["xhat_norm_wrapper"] = xhat_norm.numpy()
            dec = wrapper(torch.from_numpy(codes_w), step="decode").numpy()
        else:
            dec = model(torch.from_numpy(codes_base), step="decode").numpy()
        if codes_base is not None:
            out["codes_base"] = codes_base.T.copy()
        out["decoded"] = dec
        # decode of arbitrary codes (not produced by encode)
        rc2 = synth_codes(cfg, 96, seed=seed + 5)
        out["rand_codes"] = rc2.T.copy()
        if wrapper_ok:
            out["rand_decoded_wrapper"] = wrapper(torch.from_numpy(rc2), step="decode").numpy()
        out["rand_decoded_base"] = model(torch.from_numpy(rc2), step="decode").numpy()

    # oracle trace: pre-selection ids and selection margins (near-tie diagnostics)
    trace: dict = {}
    xn = (x.astype(np.float32) - oracle.data_mean) / oracle.data_std
    codes_o, xhat_o = oracle.encode(xn, trace)
    codes_ref = out["codes_wrapper"] if "codes_wrapper" in out else out["codes_base"]
    agree = float((codes_o.T == codes_ref).all(axis=1).mean())
    margins = []
    for m in range(1, cfg.M_total):
        d = np.sort(trace[f"dists{m}"], axis=-1)
        fo = min(cfg.B if m < cfg.M_total - 1 else 1, d.shape[1] - 1)
        mg = (d[:, fo] - d[:, fo - 1]) / np.maximum(np.abs(d[:, fo]), 1e-12)
        if f"dsub{m}" in trace:   # pre-selection boundary: A-th vs (A+1)-th codeword of each beam's table
            ds = np.sort(trace[f"dsub{m}"], axis=-1)
            a = trace[f"top{m}"].shape[-1]
            if a < ds.shape[-1]:
                pm = ((ds[..., a] - ds[..., a - 1]) / np.maximum(np.abs(ds[..., a]), 1e-12)).min(axis=1)
                mg = np.minimum(mg, pm)
        margins.append(mg)
        if f"top{m}" in trace:
            out[f"oracle_top{m}"] = trace[f"top{m}"].astype(np.int16)
    if cfg.ivf:  # margin of the coarse assignment (step 0)
        d0 = np.sort(trace["d0"], axis=-1)
        out["ivf_rel_margin"] = ((d0[:, 1] - d0[:, 0]) / np.maximum(np.abs(d0[:, 1]), 1e-12)).astype(np.float32)
    out["select_rel_margin"] = np.stack(margins, axis=1).astype(np.float32)  # (N, M_total-1)
    mse_ref = float(((x.astype(np.float32) - dec) ** 2).sum(-1).mean())
    out["mse"] = np.float64(mse_ref)
    print(f"{name:24s} N={len(x):4d} wrapper==base: "
          f"{bool((out.get('codes_base', codes_ref) == codes_ref).all())}  oracle==ref rows: {agree:.4f}  "
          f"min margin {out['select_rel_margin'].min():.2e}  mse {mse_ref:.4f}")
    return out


def write_tiny_checkpoint():
    """A checkpoint written by the reference's own save_model (qinco/utils.py:100-137)."""
    from qinco.utils import SharedCfgState, save_model
    cfg, sd = case_model("tiny_proj_beam")
    model, _ = build_reference(cfg, sd)
    path = HERE / "tiny_ckpt.pt"
    c = SharedCfgState(dict(output=str(path), K=cfg.K, M=cfg.M, de=cfg.de, dh=cfg.dh, L=cfg.L, A=cfg.A, B=cfg.B,
                            ivf_in_use=None, ivf_K=None, qinco1_mode=cfg.qinco1_mode))
    c._cur_epoch = c._optimizer = c._scheduler = c._melog = None
    c._D = cfg.D
    save_model(c, Acc(), model)
    print("wrote", path, os.path.getsize(path), "bytes")


def run_search_case() -> dict:
    """Small-db search.  qinco/search/search_tasks.py needs faiss to import, so the harness lines of
    run_search_full_direct_small_db (:551-603) are driven here step by step with the reference's own model and its own
    approx_pairwise_distance (qinco/utils.py:336-346): encode + decode the database with the reference wrapper,
    distances of query batches of 100 to the reconstructions, argsort, first 100 columns, recall of gt[:, 0]."""
    from qinco.utils import approx_pairwise_distance
    cfg, sd = case_model("tiny_proj_beam")
    model, wrapper = build_reference(cfg, sd)
    N, Q, nshort = 3000, 150, 100
    # clustered database so that near neighbours exist: vectors = decode(random codes) + noise; queries = db rows + noise
    rs = np.random.RandomState(77)
    with torch.no_grad():
        base = model(torch.from_numpy(synth_codes(cfg, N, seed=78)), step="decode").numpy()
    db = (base + 0.5 * float(sd["data_std"]) * rs.randn(N, cfg.D)).astype(np.float32)
    qsrc = rs.choice(N, Q, replace=False)
    queries = (db[qsrc] + 2.0 * float(sd["data_std"]) * rs.randn(Q, cfg.D)).astype(np.float32)
    d_exact = ((queries[:, None, :].astype(np.float64) - db[None].astype(np.float64)) ** 2).sum(-1)
    gt = np.argsort(d_exact, axis=1, kind="stable")[:, :100].astype(np.int64)
    with torch.no_grad():
        parts = []
        for i0 in range(0, N, 1024):
            b = torch.from_numpy(db[i0:i0 + 1024])
            parts.append(wrapper(wrapper(b, step="encode"), step="decode"))
        xhat = torch.concat(parts, dim=0)
        qt = torch.from_numpy(queries)
        sl, ds = [], []
        for i0 in range(0, Q, 100):
            d = approx_pairwise_distance(qt[i0:i0 + 100].unsqueeze(1), xhat).squeeze(1)
            order = d.argsort(dim=-1)[:, :nshort]
            sl.append(order)
            ds.append(torch.sort(d, dim=-1).values[:, :nshort + 1])
        shortlists = torch.concat(sl).numpy()
        dsorted = torch.concat(ds).numpy()
    recalls = [float((shortlists[:, :r] == gt[:, :1]).sum() / gt.shape[0]) for r in (1, 10, 100)]
    gap = (dsorted[:, 1:] - dsorted[:, :-1]) / np.maximum(np.abs(dsorted[:, 1:]), 1e-12)
    print(f"search_small_db          N={N} Q={Q} recalls {recalls}  min rel gap {gap.min():.2e}")
    return {"db": db, "queries": queries, "gt": gt, "xhat": xhat.numpy(), "shortlists": shortlists.astype(np.int64),
            "dist_sorted": dsorted.astype(np.float32), "recalls": np.asarray(recalls, np.float64)}


def import_reference_search_modules():
    """-> (qinco.search.pairwise_decoder, qinco.search.search_utils) of the reference, imported in this container.

    qinco/search/search_utils.py does `import faiss` at module level and qinco/search/pairwise_decoder.py imports qinco/metrics.py,
    whose two metric classes subclass torcheval.metrics.Metric; neither package is installed here (no network).  NONE of the
    reference lines driven by run_lut_case / run_rerank_case touches either of them (PairwiseDecoderIVF.forward / map_codes are
    three indexing statements on nn.Parameters, reconstruct_from_fixed_codebooks is numpy indexing): for the duration of the two
    imports EMPTY placeholder modules stand in sys.modules -- `faiss` with no attribute at all, `torcheval.metrics` with a bare
    `Metric` class for the two class statements to name as a base -- and are removed again before anything runs.  They implement
    nothing; a reference line that needed the real packages would raise AttributeError here."""
    import types
    added = [n for n in ("faiss", "torcheval", "torcheval.metrics") if n not in sys.modules]
    for n in added:
        sys.modules[n] = types.ModuleType(n)
    if "torcheval.metrics" in added:
        sys.modules["torcheval.metrics"].Metric = type("Metric", (), {})
        sys.modules["torcheval"].metrics = sys.modules["torcheval.metrics"]
    try:
        from qinco.search import pairwise_decoder, search_utils
    finally:
        for n in added:
            del sys.modules[n]
    return pairwise_decoder, search_utils


def reference_pairwise_decoder(pd_mod, codebook_MKD, combine_mvals_m, K_base, ivf_code_map):
    """A PairwiseDecoderIVF of the reference with its inference state filled in by hand: the constructor trains or loads a file
    (pairwise_decoder.py:19-58); forward / map_codes (:88-93, :126-130) read exactly these four attributes."""
    dec = pd_mod.PairwiseDecoderIVF.__new__(pd_mod.PairwiseDecoderIVF)
    torch.nn.Module.__init__(dec)
    dec.K_base = int(K_base)
    dec.codebook_MKD = torch.nn.Parameter(torch.from_numpy(np.ascontiguousarray(codebook_MKD, dtype=np.float32)), requires_grad=False)
    dec.combine_mvals_m = torch.nn.Parameter(torch.from_numpy(np.ascontiguousarray(combine_mvals_m, dtype=np.int64)), requires_grad=False)
    dec.ivf_code_map = torch.nn.Parameter(torch.from_numpy(np.ascontiguousarray(ivf_code_map, dtype=np.int64)), requires_grad=False)
    return dec


def lut_case_inputs(seed: int, M: int, K: int, D: int, ivf_K: int, Mt: int, n: int, IVF_M: int = 5):
    """Seeded inputs of a look-up-decoder case (the tests regenerate the large ones from the seed instead of storing them)."""
    rs = np.random.RandomState(seed)
    cb = rs.randn(Mt, K * K, D).astype(np.float32)
    comb = rs.randint(0, M + IVF_M, (2, Mt)).astype(np.int64)
    comb[:, 0] = (0, M)                 # at least one pair reads an IVF-map column, one a plain pair of steps
    comb[:, 1] = (1, 2)
    imap = rs.randint(0, K, (ivf_K, IVF_M)).astype(np.int64)
    codes_MB = rs.randint(0, K, (M, n)).astype(np.int64)
    ivf = rs.randint(0, ivf_K, n).astype(np.int64)
    return cb, comb, imap, codes_MB, ivf


LUT_CASES = {"small": dict(seed=77, M=8, K=16, D=32, ivf_K=200, Mt=11, n=500),          # stored whole
             "wide": dict(seed=78, M=8, K=64, D=128, ivf_K=4096, Mt=12, n=1000)}         # 25 MB of tables: regenerated from the seed


def run_lut_case() -> dict:
    """SURVEY 8f4's look-up decoders, recorded from the REFERENCE's own code (import_reference_search_modules):
    PairwiseDecoderIVF.forward (qinco/search/pairwise_decoder.py:88-93) with map_codes (:126-130) -- which always takes the IVF ids
    (it asserts their shape) -- and reconstruct_from_fixed_codebooks (qinco/search/search_utils.py:105-115), whose input has no
    IVF column."""
    pd_mod, su_mod = import_reference_search_modules()
    assert pd_mod.PairwiseDecoderIVF.IVF_M == 5
    out = {}
    for name, kw in LUT_CASES.items():
        cb, comb, imap, codes_MB, ivf = lut_case_inputs(**kw)
        dec = reference_pairwise_decoder(pd_mod, cb, comb, kw["K"], imap)
        with torch.no_grad():
            mapped = dec.map_codes(torch.from_numpy(codes_MB), torch.from_numpy(ivf))
            xhat = dec(torch.from_numpy(codes_MB), torch.from_numpy(ivf))
            xhat32 = dec(torch.from_numpy(codes_MB.astype(np.int32)), torch.from_numpy(ivf.astype(np.int32)))    # (the search hands over int32 columns)
        assert torch.equal(xhat, xhat32)
        out[f"pw_{name}_mapped"] = mapped.numpy()
        out[f"pw_{name}_xhat"] = xhat.numpy()
        for k, v in kw.items():
            out[f"pw_{name}_{k}"] = np.int64(v)
        if name == "small":
            out.update(pw_small_codebook_MKD=cb, pw_small_combine=comb, pw_small_ivf_code_map=imap, pw_small_codes_MB=codes_MB, pw_small_ivf=ivf)
        print(f"lut_decoders pairwise {name:6s} n={kw['n']} Mt={kw['Mt']} K^2={kw['K'] ** 2} D={kw['D']}")
    rs = np.random.RandomState(79)
    for name, (n, M, K, D, dt) in {"u8": (700, 8, 256, 32, np.uint8), "i64": (257, 16, 64, 96, np.int64), "one": (1, 4, 64, 32, np.int32)}.items():
        cbs = rs.randn(M, K, D).astype(np.float32)
        codes = rs.randint(0, K, (n, M)).astype(dt)
        rec = su_mod.reconstruct_from_fixed_codebooks(codes, cbs)
        out.update({f"fixed_{name}_codebooks": cbs, f"fixed_{name}_codes": codes, f"fixed_{name}_recons": np.asarray(rec, np.float32)})
        print(f"lut_decoders fixed    {name:6s} n={n} M={M} K={K} D={D}")
    return out


def run_rerank_case() -> dict:
    """The re-rank stages of run_search_ivf (qinco/search/search_tasks.py:447-507; the module itself needs faiss to import), driven
    line by line with the reference's own pieces: compute_batch_distances(approx=True) (qinco/utils.py:349-383), torch.argsort /
    take_along_dim, and the reference's inference wrapper for the QINCo decode in batches of cfg.search.batch_size.  The shortlist
    that faiss's index.search_and_return_codes would hand over is made here (each query's true neighbours among its own
    reconstructions + random rows).  The mid re-ranker is the reference's own PairwiseDecoderIVF (import_reference_search_modules)
    called exactly as search_tasks.py:451 calls it -- mid_reranker(codes_int32_T[1:], codes_int32_T[0]) -- over look-up tables made
    from the model's codebooks and a seeded ivf_code_map: every stage of the fixture is the reference's arithmetic."""
    from qinco.utils import compute_batch_distances
    pd_mod, _ = import_reference_search_modules()
    cfg, sd = case_model("tiny_ivf_beam")
    model, wrapper = build_reference(cfg, sd)
    M, d = cfg.M, cfg.D
    N, nq, n_short_ivf, nshort, bs = 3000, 24, 160, 40, 512
    rs = np.random.RandomState(91)
    with torch.no_grad():
        base = model(torch.from_numpy(synth_codes(cfg, N, seed=92)), step="decode").numpy()
        db = (base + 0.3 * float(sd["data_std"]) * rs.randn(N, d)).astype(np.float32)
        codes_db = wrapper(torch.from_numpy(db), step="encode").numpy().T.astype(np.int32)      # (N, M + 1): IVF id first
        xhat_db = wrapper(torch.from_numpy(codes_db.T.astype(np.int64)), step="decode").numpy()
    xq = (db[rs.choice(N, nq, replace=False)] + 1.0 * float(sd["data_std"]) * rs.randn(nq, d)).astype(np.float32)
    dq = ((xq[:, None, :].astype(np.float64) - xhat_db[None].astype(np.float64)) ** 2).sum(-1)
    near = np.argsort(dq, axis=1, kind="stable")[:, :n_short_ivf // 2]
    I = np.stack([rs.permutation(np.concatenate([near[q], rs.choice(np.setdiff1d(np.arange(N), near[q]), n_short_ivf - near.shape[1],
                                                                   replace=False)])) for q in range(nq)]).astype(np.int64)
    codes_int32 = codes_db[I.reshape(-1)]                                                     # (nq * n_short_ivf, M + 1)
    # the mid re-ranker's stand-in: a pairwise look-up table T[j][c_a K + c_b] over (step 1, step 2), ..., rough reconstructions
    K = cfg.K
    tables, comb = rerank_lut_tables(cfg, sd)
    ivf_code_map = np.random.RandomState(93).randint(0, K, (cfg.ivf_K, pd_mod.PairwiseDecoderIVF.IVF_M)).astype(np.int64)
    mid_reranker = reference_pairwise_decoder(pd_mod, tables, comb, K, ivf_code_map)
    with torch.no_grad():
        ct = torch.from_numpy(codes_int32).T
        mid = mid_reranker(ct[1:], ct[0]).numpy()                                             # search_tasks.py:451
    ivf_book_t = model.steps[0].ivf_centroids.weight.detach()
    # (the tables are a function of the model's codebooks: the test rebuilds them with rerank_lut_tables instead of storing 25 MB)
    out = {"xq": xq, "I": I, "codes_int32": codes_int32, "pair_combine": comb, "ivf_code_map": ivf_code_map, "mid_shortlist": mid,
           "ivf_book": ivf_book_t.numpy(), "nshort": np.int64(nshort), "batch_size": np.int64(bs)}
    with torch.no_grad():
        xq_distances = torch.from_numpy(xq)
        I_t, codes_t = torch.from_numpy(I), torch.from_numpy(codes_int32)
        # ---- Part 3 (search_tasks.py:447-472)
        shortlist = torch.from_numpy(mid).clone()
        shortlist += ivf_book_t[codes_t[:, 0].long()]
        shortlist = shortlist.reshape(nq, n_short_ivf, d)
        D_refined = compute_batch_distances(xq_distances.reshape(nq, 1, d), shortlist, approx=True).reshape(nq, n_short_ivf)
        idx = torch.argsort(D_refined, axis=1, stable=True)
        codes_refined = torch.take_along_dim(codes_t.reshape(nq, n_short_ivf, M + 1), idx[:, :nshort, None], dim=1)
        I_mid = torch.take_along_dim(I_t, idx[:, :nshort], dim=1)
        out["mid_dist_sorted"] = torch.sort(D_refined, dim=1).values.numpy()
        out["I_mid"], out["codes_mid"] = I_mid.numpy(), codes_refined.numpy()
        codes2 = codes_refined.reshape(nq * nshort, M + 1)
        # ---- Part 4 (:475-486): QINCo decode in batches of cfg.search.batch_size
        parts = [wrapper(codes2[i:i + bs].T.long(), step="decode") for i in range(0, len(codes2), bs)]
        shortlist_t = torch.concatenate(parts).reshape(nq, nshort, d)
        # ---- Part 5 (:497-507)
        D2 = compute_batch_distances(xq_distances.reshape(nq, 1, d), shortlist_t, approx=True).reshape(nq, nshort)
        idx2 = torch.argsort(D2, axis=1, stable=True)
        out["final_dist_sorted"] = torch.sort(D2, dim=1).values.numpy()
        out["I_refined"] = torch.take_along_dim(I_mid, idx2[:, :100], dim=1).numpy()
        out["decoded_shortlist"] = shortlist_t.numpy()
    g1 = np.diff(out["mid_dist_sorted"], axis=1) / np.maximum(np.abs(out["mid_dist_sorted"][:, 1:]), 1e-12)
    g2 = np.diff(out["final_dist_sorted"], axis=1) / np.maximum(np.abs(out["final_dist_sorted"][:, 1:]), 1e-12)
    print(f"rerank_ivf               nq={nq} shortlist {n_short_ivf} -> {nshort} -> {out['I_refined'].shape[1]}; min rel gaps {g1.min():.2e} {g2.min():.2e}")
    return out


if __name__ == "__main__":
    only = sys.argv[1:]
    if not only or "lut_decoders" in only:
        np.savez_compressed(HERE / "lut_decoders.npz", **run_lut_case())
    if not only or "rerank_ivf" in only:
        np.savez_compressed(HERE / "rerank_ivf.npz", **run_rerank_case())
    if not only or "search_small_db" in only:
        np.savez_compressed(HERE / "search_small_db.npz", **run_search_case())
    for name in CASES:
        if only and name not in only:
            continue
        out = run_case(name)
        np.savez_compressed(HERE / f"{name}.npz", **out)
    if not only:
        write_tiny_checkpoint()
