"""SURVEY 8f3 (small-db search) and 8a13 (compute_MSE): oracle vs the golden produced with the reference's own model
and approx_pairwise_distance (CPU tier), and the GPU top-k / squared-error kernels vs the oracle and the golden.

Tolerance: distances are fp32 with a summation order that differs from MKL's, so a shortlist may differ from the
reference's only where two sorted distances are closer than 2e-5 relative; everywhere else ids are identical.
Recalls must be equal."""
import numpy as np
import pytest

from conftest import golden_model, load_golden

REL = 2e-5


def _check_shortlists(ids, dist, want_ids, want_sorted):
    """ids/dist: (Q, k) got; want_sorted: (Q, k+1) reference distances ascending."""
    k = want_ids.shape[1]
    assert ids.shape == want_ids.shape and ids.dtype == np.int64
    assert np.allclose(dist, want_sorted[:, :k], rtol=1e-5, atol=1e-5 * np.abs(want_sorted).max())
    assert (np.diff(dist, axis=1) >= 0).all()
    gap = (want_sorted[:, 1:] - want_sorted[:, :-1]) / np.maximum(np.abs(want_sorted[:, 1:]), 1e-12)
    # position j may differ only if it is in a near-tie with a neighbour (gap before or after it below REL)
    tie = np.zeros_like(want_ids, dtype=bool)
    tie[:, :] = gap[:, :k] < REL
    tie[:, 1:] |= gap[:, :k - 1] < REL
    diff = ids != want_ids
    assert not (diff & ~tie).any(), f"{(diff & ~tie).sum()} shortlist entries differ outside near-ties"
    return int(diff.sum())


def test_oracle_search_matches_golden():
    from oracle.qinco_oracle import compute_recalls, small_db_shortlists
    g = load_golden("search_small_db")
    ids, dist = small_db_shortlists(g["xhat"], g["queries"], nshort=100)
    _check_shortlists(ids, dist, g["shortlists"], g["dist_sorted"])
    rec = compute_recalls(ids, g["gt"])
    assert [rec[r] for r in (1, 10, 100)] == list(g["recalls"])


def test_oracle_search_pipeline_reproduces_reference_reconstructions():
    """encode -> decode of the database through the oracle gives the golden's xhat (reference wrapper), so the whole
    run_search_full_direct_small_db pipeline is pinned, not only its last stage."""
    from conftest import make_oracle
    from qinco_amd import synth_state_dict
    g = load_golden("search_small_db")
    o = make_oracle(*golden_model("tiny_proj_beam"))
    n = 512
    xhat = o(o(g["db"][:n], step="encode"), step="decode")
    rel = np.abs(xhat - g["xhat"][:n]).max(axis=1) / np.abs(g["xhat"][:n]).max()
    assert (rel < 1e-5).mean() > 0.99  # rows may differ only at beam near-ties


def test_compute_recalls_and_timer_host_logic():
    from oracle.qinco_oracle import compute_recalls as ref
    from qinco_amd.evaluate import Timer
    from qinco_amd.search import compute_recalls
    rs = np.random.RandomState(0)
    I = rs.randint(0, 50, (200, 100))
    gt = rs.randint(0, 50, (200, 3))
    assert compute_recalls(I, gt) == ref(I, gt)
    with pytest.raises(AssertionError):
        compute_recalls(I[0], gt)
    t = Timer()
    with t:
        pass
    with t:
        pass
    assert t.get() >= 0.0


def test_compute_mse_protocol_cpu():
    """compute_MSE bookkeeping with a stand-in model (numpy path): warm-up does not count, MSE = sum / n * scale."""
    from oracle.qinco_oracle import mse as ref_mse
    from qinco_amd.evaluate import compute_MSE
    rs = np.random.RandomState(1)
    batches = [rs.randn(n, 8).astype(np.float32) for n in (64, 64, 17)]
    calls = {"encode": 0, "decode": 0}

    class Model:
        def __call__(self, x, step):
            calls[step] += 1
            if step == "encode":
                return np.round(x).astype(np.int64).T      # (M, n) like the reference
            return x.T.astype(np.float32)

    res = compute_MSE(Model(), batches, mse_scale=3.0, warm_start=True)
    x = np.concatenate(batches)
    assert res["n_vecs"] == len(x)
    assert np.isclose(res["MSE"], ref_mse(x, np.round(x), 3.0), rtol=1e-12)
    assert calls == {"encode": 6, "decode": 6}          # 3 warm-up + 3 timed
    assert res["encode_us_per_vec"] >= 0 and res["decode_us_per_vec"] >= 0
    res2 = compute_MSE(Model(), batches, warm_start=False)
    assert np.isclose(res2["MSE"] * 3.0, res["MSE"], rtol=1e-12)


# ------------------------------------------------------------------------------------------------------
# GPU tier
# ------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_knn_matches_golden_host_and_device():
    import torch
    from qinco_amd.search import KnnSearcher, compute_recalls
    g = load_golden("search_small_db")
    knn = KnnSearcher(32)
    ids, dist = knn.search(g["xhat"], g["queries"], k=100, return_dist=True)
    _check_shortlists(ids, dist, g["shortlists"], g["dist_sorted"])
    rec = compute_recalls(ids, g["gt"])
    assert [rec[r] for r in (1, 10, 100)] == list(g["recalls"])
    ids_d, dist_d = knn.search(torch.from_numpy(g["xhat"]).cuda(), torch.from_numpy(g["queries"]).cuda(), k=100,
                               return_dist=True)
    assert np.array_equal(ids_d.cpu().numpy(), ids) and np.array_equal(dist_d.cpu().numpy(), dist)
    knn.close()


@pytest.mark.gpu
@pytest.mark.parametrize("D,N,Q,k", [(128, 20000, 70, 100), (96, 5001, 33, 10), (768, 3000, 40, 100),
                                     (32, 100, 5, 100), (64, 257, 1, 1), (256, 4096, 64, 2048)])
def test_knn_vs_oracle_shapes(D, N, Q, k):
    from oracle.qinco_oracle import approx_pairwise_distance
    from qinco_amd.search import KnnSearcher
    rs = np.random.RandomState(D + N)
    db = rs.randn(N, D).astype(np.float32)
    q = (db[rs.choice(N, Q)] + 0.5 * rs.randn(Q, D)).astype(np.float32)
    knn = KnnSearcher(D)
    ids, dist = knn.search(db, q, k=k, return_dist=True)
    d = approx_pairwise_distance(q, db)
    order = np.argsort(d, axis=1, kind="stable")
    kk = min(k + 1, N)
    want_sorted = np.take_along_axis(d, order[:, :kk], axis=1)
    if kk == k:  # k == N: no (k+1)-th element
        want_sorted = np.concatenate([want_sorted, np.full((Q, 1), np.inf, np.float32)], axis=1)
    _check_shortlists(ids, dist, order[:, :k].astype(np.int64), want_sorted)
    # the returned distances are the distances of the returned ids
    assert np.allclose(np.take_along_axis(d, ids, axis=1), dist, rtol=1e-5, atol=1e-5 * np.abs(d).max())  # cancellation in |q|^2+|x|^2-2q.x
    for row in ids:
        assert len(set(row.tolist())) == k
    knn.close()


@pytest.mark.gpu
def test_knn_exact_ties_resolve_to_lower_index_and_radix_refinement():
    """Duplicated database rows give exactly equal distances: the stable order (lower id first) must come out, also
    when more than the LDS buffer's worth of keys share the k-th key's leading bits (forces extra radix passes)."""
    from qinco_amd.search import KnnSearcher
    rs = np.random.RandomState(5)
    D, N = 32, 40000
    base = rs.randn(4, D).astype(np.float32)
    db = np.repeat(base, N // 4, axis=0)           # 4 distinct rows, 10000 copies each
    q = base[[2, 0]] + 0.01
    knn = KnnSearcher(D)
    ids, dist = knn.search(db, q, k=50, return_dist=True)
    assert np.array_equal(ids[0], np.arange(20000, 20050)) and np.array_equal(ids[1], np.arange(50))
    assert (dist[0] == dist[0, 0]).all()
    knn.close()


def _knn_both_forms(D, db, q, k, min_n=1024):
    """The same search through the filtered form (forced from min_n rows on) and the unfiltered form: ids and distances.  The
    filtered form runs twice where it has two kernels (D <= 128): every wave in both roles (the default) and one MFMA wave + one
    filter wave per SIMD (opt-in, csrc/knn_roles_kernel.hpp) -- the two must agree bit for bit before the first is handed back."""
    import torch
    from qinco_amd.search import KnnSearcher
    dbt, qt = torch.from_numpy(db).cuda(), torch.from_numpy(q).cuda()
    out = []
    for filtered, roles in ((True, None), (True, True), (False, None)):
        knn = KnnSearcher(D, filtered=filtered, filter_min_n=min_n, roles=roles)
        ids, dist = knn.search(dbt, qt, k=k, return_dist=True)
        st = knn.last_stats()
        st["role_workgroups"] = knn.roles_stats()
        out.append((ids.cpu().numpy(), dist.cpu().numpy(), st))
        knn.close()
    (ids_s, dist_s, st_s), (ids_r, dist_r, st_r) = out[0], out[1]
    assert np.array_equal(ids_r, ids_s) and np.array_equal(dist_r.view(np.uint32), dist_s.view(np.uint32)), "two-role kernel differs"
    assert {k_: v for k_, v in st_r.items() if k_ != "role_workgroups"} == {k_: v for k_, v in st_s.items() if k_ != "role_workgroups"}
    ran_roles = sum(st_r["role_workgroups"].values())
    assert (ran_roles > 0) == (D <= 128 and st_r["filtered"] > 0) and sum(st_s["role_workgroups"].values()) == 0, (st_r, st_s)
    for o in out:
        o[2].pop("role_workgroups")
    return [out[0], out[2]]


@pytest.mark.gpu
@pytest.mark.parametrize("D,N,Q,k", [(128, 150_001, 70, 100), (32, 70_000, 4200, 10), (768, 66_000, 33, 100), (96, 20_000, 40, 600),
                                     (64, 9_000, 5, 1), (256, 131_072, 64, 200)])
def test_knn_filtered_form_is_bit_identical_to_the_table_form(D, N, Q, k):
    """Round 5: large databases never write the (queries x n) table -- thresholds from a strided sample, survivors of the whole
    table appended to per-query lists, those sorted (csrc/knn_kernel.hpp).  Same ids, same distance bits as the unfiltered
    kernels, and against the oracle's stable argsort on a few rows."""
    from oracle.qinco_oracle import approx_pairwise_distance
    rs = np.random.RandomState(D + N + k)
    db = rs.randn(N, D).astype(np.float32)
    q = (db[rs.choice(N, Q)] + 0.5 * rs.randn(Q, D)).astype(np.float32)
    (ids_f, dist_f, st_f), (ids_t, dist_t, st_t) = _knn_both_forms(D, db, q, k)
    assert st_t["filtered"] == 0
    stride = min(32, 8192 // (4 * k))
    if stride >= 4:
        assert st_f["filtered"] == st_f["chunks"] >= 1 and st_f["redone_unfiltered"] == 0, st_f
    else:
        assert st_f["filtered"] == 0, st_f       # k = 600: the sample would cost more than the table it saves
    assert np.array_equal(ids_f, ids_t) and np.array_equal(dist_f.view(np.uint32), dist_t.view(np.uint32))
    rows = rs.choice(Q, min(Q, 8), replace=False)
    d = approx_pairwise_distance(q[rows], db)
    order = np.argsort(d, axis=1, kind="stable")
    want_sorted = np.take_along_axis(d, order[:, :k + 1], axis=1)
    _check_shortlists(ids_f[rows], dist_f[rows], order[:, :k].astype(np.int64), want_sorted)


@pytest.mark.gpu
def test_knn_filtered_form_at_the_reference_scale():
    """bigann1M's shape (search_tasks.py:551-603: 10^6 reconstructions, D = 128, top-100) with 3000 queries -- two chunks of queries,
    the second ragged: the filtered form (the default there) against the table form, every id and distance bit; no chunk redone."""
    import torch
    from qinco_amd.search import KnnSearcher
    g = torch.Generator(device="cuda").manual_seed(3)
    db = torch.randn(1_000_000, 128, device="cuda", generator=g)
    q = db[torch.randint(0, 1_000_000, (3000,), device="cuda", generator=g)] + 0.7 * torch.randn(3000, 128, device="cuda", generator=g)
    out = {}
    for filtered in (True, False):
        knn = KnnSearcher(128, filtered=filtered)
        ids, dist = knn.search(db, q, k=100, return_dist=True)
        out[filtered] = (ids, dist, knn.last_stats())
        knn.close()
    assert out[True][2] == {"chunks": 2, "filtered": 2, "redone_unfiltered": 0} and out[False][2]["filtered"] == 0
    assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1].view(torch.int32), out[False][1].view(torch.int32))
    assert bool((out[True][1][:, 1:] >= out[True][1][:, :-1]).all())
    # the nearest row of a query that is a perturbed database row is (almost always) that row: a sanity check on the ids themselves
    d0 = ((db[out[True][0][:, 0]] - q) ** 2).sum(1)
    assert torch.allclose(d0, out[True][1][:, 0], rtol=1e-3, atol=1e-2)


@pytest.mark.gpu
def test_knn_filtered_form_falls_back_on_the_device_when_a_list_overflows():
    """Adversarial data for the filter: 4 distinct rows, 20 000 copies each -- every copy of the nearest row sits exactly AT the
    threshold, 20 000 candidates for 8192 slots.  The chunk's flag is raised and the unfiltered kernels queued behind it redo
    the chunk: the stable order (lower id first) must come out; a clustered database in sorted order (a sample that is
    unrepresentative row-prefix-wise but fine strided) stays on the filtered form."""
    rs = np.random.RandomState(5)
    D, N = 32, 80_000
    base = rs.randn(4, D).astype(np.float32)
    db = np.repeat(base, N // 4, axis=0)
    q = base[[2, 0]] + 0.01
    (ids_f, dist_f, st_f), (ids_t, dist_t, st_t) = _knn_both_forms(D, db, q, 50)
    assert st_f == {"chunks": 1, "filtered": 1, "redone_unfiltered": 1}, st_f
    assert np.array_equal(ids_f[0], np.arange(40000, 40050)) and np.array_equal(ids_f[1], np.arange(50))
    assert np.array_equal(ids_f, ids_t) and np.array_equal(dist_f.view(np.uint32), dist_t.view(np.uint32))
    # clustered and sorted by cluster: 64 clusters of 2000 rows; queries near cluster centres
    centres = 4.0 * rs.randn(64, D).astype(np.float32)
    db = (np.repeat(centres, 2000, axis=0) + 0.3 * rs.randn(128_000, D)).astype(np.float32)
    q = (centres[rs.choice(64, 40)] + 0.3 * rs.randn(40, D)).astype(np.float32)
    (ids_f, dist_f, st_f), (ids_t, dist_t, _) = _knn_both_forms(D, db, q, 100)
    assert st_f["filtered"] == 1 and st_f["redone_unfiltered"] == 0, st_f
    assert np.array_equal(ids_f, ids_t) and np.array_equal(dist_f.view(np.uint32), dist_t.view(np.uint32))
    # NaN in the database (a corrupt row): both forms order keys the same way
    db[777] = np.nan
    db[780] = np.nan                                  # 780 is a row of the strided sample (stride 20), 777 is not
    db[rs.choice(128_000, 300, replace=False), 5] = np.nan
    (ids_f, dist_f, st_f), (ids_t, dist_t, _) = _knn_both_forms(D, db, q, 100)
    assert st_f["redone_unfiltered"] == 0
    assert np.array_equal(ids_f, ids_t) and np.array_equal(dist_f.view(np.uint32), dist_t.view(np.uint32))
    assert not np.isnan(dist_f).any()                   # a NaN distance is THE positive quiet NaN in both forms and sorts last
    # a NaN QUERY: every distance NaN, every key equal up to the index -> rows 0 .. k-1; the filtered form's lists overflow -> redone
    q[3, 7] = np.nan
    (ids_f, dist_f, st_f), (ids_t, dist_t, _) = _knn_both_forms(D, db, q, 100)
    assert st_f["redone_unfiltered"] == 1
    assert np.array_equal(ids_f[3], np.arange(100)) and np.isnan(dist_f[3]).all()
    assert np.array_equal(ids_f, ids_t) and np.array_equal(dist_f.view(np.uint32), dist_t.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("D", [128, 768, 32])
def test_knn_smallest_query_stream_option(D):
    """QINCO_KNN_OPT_QUERY_BYTES at its documented minimum (4096 B): less than one 32-row block of query fragments for D >= 64 --
    the chunk must come out as 32 rows, not 0 (round-5 review: a division by zero), and the ids must not depend on the chunking."""
    import torch
    from qinco_amd import _lib
    from qinco_amd.search import KnnSearcher
    rs = np.random.RandomState(D)
    db = torch.from_numpy(rs.randn(70_000, D).astype(np.float32)).cuda()
    q = torch.from_numpy(rs.randn(100, D).astype(np.float32)).cuda()
    outs = []
    for qb in (None, 4096):
        knn = KnnSearcher(D)
        if qb is not None:
            _lib.check(knn.lib.qinco_knn_set_option(knn._h, 2, qb))
        ids, dist = knn.search(db, q, k=10, return_dist=True)
        outs.append((ids.cpu().numpy(), dist.cpu().numpy(), knn.last_stats()))
        knn.close()
    assert outs[1][2]["chunks"] == 4 and outs[0][2]["chunks"] == 1              # 100 queries in chunks of 32
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1].view(np.uint32), outs[1][1].view(np.uint32))
    knn = KnnSearcher(D)
    with pytest.raises(ValueError):
        _lib.check(knn.lib.qinco_knn_set_option(knn._h, 2, 4095))
    knn.close()


@pytest.mark.gpu
def test_knn_argument_errors_and_empty():
    from qinco_amd.search import KnnSearcher
    with pytest.raises(NotImplementedError):
        KnnSearcher(48)
    knn = KnnSearcher(32)
    db = np.zeros((10, 32), np.float32)
    assert knn.search(db, np.zeros((0, 32), np.float32), k=5).shape == (0, 5)
    with pytest.raises(ValueError):
        knn.search(db, db, k=11)
    with pytest.raises(ValueError):
        knn.search(db, np.zeros((3, 16), np.float32), k=1)
    knn.close()


@pytest.mark.gpu
def test_search_small_db_pipeline_and_compute_mse_on_gpu():
    """The whole f3 pipeline on the HIP path against the golden produced with the reference model: recalls equal,
    shortlists equal outside near-ties; compute_MSE over the same database equals the oracle's MSE of the golden
    reconstructions to 1e-5."""
    import torch
    from oracle.qinco_oracle import mse as ref_mse
    from qinco_amd import synth_state_dict
    from qinco_amd.evaluate import compute_MSE, sqerr_sum
    from qinco_amd.model import QINCoHIP
    from qinco_amd.search import search_small_db
    g = load_golden("search_small_db")
    cfg, sd_ = golden_model("tiny_proj_beam")
    model = QINCoHIP(cfg, sd_, max_batch=1024)
    res = search_small_db(model, g["db"], g["queries"], g["gt"], batch=1024)
    xhat = res["xhat"].cpu().numpy()
    same = np.abs(xhat - g["xhat"]).max(axis=1) <= 1e-5 * np.abs(g["xhat"]).max()
    assert same.mean() > 0.99
    if same.all():
        assert [res["recalls"][r] for r in (1, 10, 100)] == list(g["recalls"])
    else:  # a beam near-tie changed a reconstruction: recalls may move by that many queries at most
        assert np.allclose([res["recalls"][r] for r in (1, 10, 100)], g["recalls"], atol=(~same).sum() / len(g["gt"]))
    batches = [torch.from_numpy(g["db"][i:i + 1024]).cuda() for i in range(0, len(g["db"]), 1024)]
    ev = compute_MSE(model, batches, mse_scale=1.0)
    assert ev["n_vecs"] == len(g["db"])
    assert np.isclose(ev["MSE"], ref_mse(g["db"], g["xhat"]), rtol=1e-5)
    a = torch.randn(100003, device="cuda")
    b = torch.randn(100003, device="cuda")
    assert np.isclose(sqerr_sum(a, b), float(((a.double() - b.double()) ** 2).sum()), rtol=1e-9)


@pytest.mark.gpu
def test_rerank_ivf_matches_the_reference_stages():
    """run_search_ivf's re-rank stages (search_tasks.py:447-507) as qinco_amd.search.rerank_ivf over the look-up decoder, the
    re-rank kernel (csrc/rerank_kernel.hpp) and QINCoHIP's decode in batches, against a fixture made by driving those lines with
    the reference's own compute_batch_distances, argsort / take_along_dim and inference wrapper (tests/golden/make_golden.py
    run_rerank_case).  The mid re-ranker of the fixture is the reference's own PairwiseDecoderIVF, called as search_tasks.py:451
    calls it: the look-up decoder's output (mid_shortlist) must be the same bits, and everything downstream is the reference's too.
    A kept id may differ from the reference's only where the reference's own sorted distances are within rounding of each other."""
    import torch
    from conftest import golden_model, load_golden
    from qinco_amd.lut import PairwiseDecoder
    from qinco_amd.model import QINCoHIP
    from qinco_amd.search import rerank_ivf
    g = load_golden("rerank_ivf")
    cfg, sd = golden_model("tiny_ivf_beam")
    K, d, M = cfg.K, cfg.D, cfg.M
    from cases import rerank_lut_tables
    tables, comb = rerank_lut_tables(cfg, sd)
    assert np.array_equal(comb, g["pair_combine"])
    dec = PairwiseDecoder(tables, comb, K_base=K, ivf_code_map=g["ivf_code_map"])
    mid = dec(g["codes_int32"][:, 1:].T, g["codes_int32"][:, 0])
    assert np.array_equal(np.asarray(mid), g["mid_shortlist"])       # gathers + fp32 adds in the reference's order: the same bits
    model = QINCoHIP(cfg, sd, max_batch=1024)
    nshort, bs = int(g["nshort"]), int(g["batch_size"])
    out = rerank_ivf(model, g["xq"], g["I"], g["codes_int32"], nshort=nshort, mid_reranker=lambda c, i: dec(c.cpu().numpy(), i.cpu().numpy()),
                     ivf_book=g["ivf_book"], batch_size=bs)
    torch.cuda.synchronize()

    def same_up_to_ties(got, want, dist_sorted, label):
        got, want = np.asarray(got.cpu()), np.asarray(want)
        assert got.shape == want.shape, (label, got.shape, want.shape)
        gaps = np.diff(dist_sorted, axis=1) / np.maximum(np.abs(dist_sorted[:, 1:]), 1e-12)
        for q, t in zip(*np.nonzero(got != want)):
            near = gaps[q, max(t - 1, 0): t + 1]
            assert near.size and near.min() < 1e-5, f"{label}: query {q} rank {t}: {got[q, t]} != {want[q, t]} without a tie"
        return int((got != want).any(axis=1).sum())
    n_mid = same_up_to_ties(out["I_mid"], g["I_mid"], g["mid_dist_sorted"], "stage 3")
    if n_mid == 0:   # (the later stages work on the same survivors as the reference's only then)
        assert np.array_equal(np.asarray(out["codes_mid"].cpu()), g["codes_mid"])
        ref_dec = g["decoded_shortlist"]
        assert np.abs(np.asarray(out["decoded"].cpu()) - ref_dec).max() <= 1e-5 * np.abs(ref_dec).max()
        same_up_to_ties(out["I"], g["I_refined"], g["final_dist_sorted"], "stage 5")
    print(f"rerank_ivf: {n_mid} queries with a stage-3 tie swap")
    dec.close()
    model.engine.close()
