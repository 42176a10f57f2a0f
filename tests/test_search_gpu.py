"""Parity of the HIP search path (through the C ABI) against the oracle.  GPU only."""
import json
import os

import numpy as np
import pytest

from oracle import retrieval as R
from tests import golden_cases as G
from tests.gpu_util import assert_topk_matches, exact_topk_fast

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from semantic_query_engine_amd import Context
    c = Context(0)
    info = c.device_info()
    assert info["cu_count"] >= 64
    return c


def _index(ctx, x, dim=None):
    from semantic_query_engine_amd import VectorIndex
    idx = VectorIndex(ctx, dim or x.shape[1])
    idx.add(x)
    return idx


def test_normalize_matches_reference_golden(ctx, golden_dir):
    exp = np.load(os.path.join(golden_dir, "normalize_rows.npz"))["expected"]
    e = G.normalize_case()
    idx = _index(ctx, e)
    got = idx.get_rows(np.arange(32))
    assert not np.isnan(got).any() and np.all(got[3] == 0.0)
    big = np.abs(exp) > 1e-30
    assert np.allclose(got[big], exp[big], rtol=5e-7, atol=0)     # fp32 sum order: a few ulp
    assert np.allclose(got, exp, rtol=5e-7, atol=1e-37)


def test_knn_golden_small(ctx, golden_dir):
    g = np.load(os.path.join(golden_dir, "knn_small.npz"))
    x, q = G.knn_case()
    idx = _index(ctx, x)
    assert len(idx) == 4096
    cos, ids = idx.search(q, 10)
    xn, qn = R.normalize_rows(x), R.normalize_rows(q)
    assert_topk_matches(cos, ids, g["cos"], g["ids"], xn, qn)
    assert ids[0].tolist() == [5] + list(range(3000, 3009))       # 41-way exact tie -> lowest ids
    assert ids[15].tolist() == list(range(10))                    # zero query: all cosines 0
    # single-query calls (the reference's B=1 pattern, main.py:355) give the same answers
    for b in (0, 3, 15):
        c1, i1 = idx.search(q[b:b + 1], 10)
        assert np.array_equal(i1[0], ids[b]) and np.allclose(c1[0], cos[b], atol=1e-6)


@pytest.mark.parametrize("n,b,k", [(1, 1, 3), (7, 3, 10), (255, 2, 5), (257, 65, 10), (1000, 64, 1),
                                   (5000, 130, 10), (20011, 37, 32), (3000, 300, 100)])
def test_shapes_and_edges(ctx, n, b, k):
    rng = np.random.default_rng(n * 31 + b)
    x = rng.standard_normal((n, 256)).astype(np.float32)
    q = rng.standard_normal((b, 256)).astype(np.float32)
    q[0] = x[n // 2] * 3.0 + 0.01 * q[0]
    idx = _index(ctx, x)
    cos, ids = idx.search(q, k)
    ref_cos, ref_ids = R.knn_search(x, q, k)
    assert_topk_matches(cos, ids, ref_cos, ref_ids, R.normalize_rows(x), R.normalize_rows(q))
    assert ids[0, 0] == n // 2
    if k > n:
        assert np.all(ids[:, n:] == -1) and np.all(np.isneginf(cos[:, n:]))


def test_empty_index_and_incremental_add(ctx):
    from semantic_query_engine_amd import VectorIndex
    rng = np.random.default_rng(2)
    idx = VectorIndex(ctx, 128)
    q = rng.standard_normal((3, 128)).astype(np.float32)
    cos, ids = idx.search(q, 4)
    assert np.all(ids == -1) and np.all(np.isneginf(cos))
    x = rng.standard_normal((3000, 128)).astype(np.float32)
    for lo, hi in [(0, 1), (1, 300), (300, 1500), (1500, 3000)]:      # grows across reallocation
        idx.add(x[lo:hi])
    assert len(idx) == 3000
    cos, ids = idx.search(q, 4)
    ref_cos, ref_ids = R.knn_search(x, q, 4)
    assert_topk_matches(cos, ids, ref_cos, ref_ids, R.normalize_rows(x), R.normalize_rows(q))
    # overwrite rows in place (re-indexing an existing _id)
    x[10] = q[0] * 2.0
    x[2000] = -q[1]
    idx.update(np.array([10, 2000]), x[[10, 2000]])
    cos, ids = idx.search(q, 4)
    ref_cos, ref_ids = R.knn_search(x, q, 4)
    assert_topk_matches(cos, ids, ref_cos, ref_ids, R.normalize_rows(x), R.normalize_rows(q))
    assert ids[0, 0] == 10 and abs(cos[0, 0] - 1.0) < 1e-6


def test_adversarial_orderings(ctx):
    """Scores that increase with the row id defeat the running threshold (every row is a
    candidate, maximum compaction load); all-equal rows are one giant tie."""
    n, d = 30000, 64
    theta = np.linspace(1.5, 0.0, n)
    x = np.zeros((n, d), np.float32)
    x[:, 0], x[:, 1] = np.cos(theta), np.sin(theta)
    q = np.zeros((2, d), np.float32)
    q[0, 0] = 1.0                 # ascending scores
    q[1, 1] = 1.0                 # descending scores
    idx = _index(ctx, x)
    cos, ids = idx.search(q, 10)
    ref_cos, ref_ids = R.knn_search(x, q, 10)
    assert_topk_matches(cos, ids, ref_cos, ref_ids, R.normalize_rows(x), R.normalize_rows(q), tol=1e-7)
    # ~1200 rows sit within bf16 resolution of the best score: the certificate must have failed and
    # the exact fp32 rescan produced the answer
    assert ctx.stats()["uncertified"] >= 1
    idx.set_option("certify", 0)
    cos_nc, ids_nc = idx.search(q, 10)
    assert not np.array_equal(ids_nc[0], ref_ids[0])          # what the bf16 scan alone would return
    idx.set_option("certify", 1)
    same = np.tile(np.linspace(-1, 1, d, dtype=np.float32), (7000, 1))
    idx2 = _index(ctx, same)
    cos, ids = idx2.search(same[:3] * 5.0, 10)
    assert np.array_equal(ids, np.tile(np.arange(10), (3, 1))) and np.allclose(cos, 1.0, atol=1e-6)


def test_config2_shape_recall(ctx):
    """BASELINE config 2 at reduced N (the oracle must finish in seconds): N=200k x 1024,
    B=1024, k=10, recall@10 = 1.0 vs exact, cosines within 1e-3."""
    rng = np.random.default_rng(0)
    n, b = 200_000, 1024
    x = rng.standard_normal((n, 1024), dtype=np.float32)
    q = rng.standard_normal((b, 1024), dtype=np.float32)
    plant = rng.integers(0, n, b // 2)
    q[: b // 2] = x[plant] + 0.1 * q[: b // 2]
    idx = _index(ctx, x)
    cos, ids = idx.search(q, 10)
    ref_cos, ref_ids = exact_topk_fast(x, q, 10)
    assert R.recall_at_k(ids, ref_ids) == 1.0
    assert_topk_matches(cos, ids, ref_cos, ref_ids, R.normalize_rows(x), R.normalize_rows(q))
    assert np.array_equal(ids[: b // 2, 0], plant)
    assert np.abs(cos - ref_cos).max() < 1e-5
    assert ctx.stats()["uncertified"] <= b // 50              # the certificate holds for almost every query


def test_config2_own_size(ctx):
    """BASELINE config 2 at ITS OWN size (r02 verdict: the 200 k-row form above was the only one): 1 x MI355X,
    brute-force cosine top-10 over 1 M synthetic 1024-d fp32 vectors, batch = 1024 queries; recall@10 = 1.0 against the
    NumPy exact answer (fp32 BLAS shortlist, float64 scoring of the shortlist), cosines within 1e-3 (measured: 1e-5),
    ids exact, half of the queries with a planted neighbour."""
    rng = np.random.default_rng(0)
    n, b = 1_000_000, 1024
    x = rng.standard_normal((n, 1024), dtype=np.float32)
    q = rng.standard_normal((b, 1024), dtype=np.float32)
    plant = rng.integers(0, n, b // 2)
    q[: b // 2] = x[plant] + 0.1 * q[: b // 2]
    idx = _index(ctx, x)
    cos, ids = idx.search(q, 10)
    ref_cos, ref_ids = exact_topk_fast(x, q, 10)
    assert R.recall_at_k(ids, ref_ids) == 1.0
    assert_topk_matches(cos, ids, ref_cos, ref_ids, R.normalize_rows(x), R.normalize_rows(q))
    assert np.array_equal(ids[: b // 2, 0], plant)
    assert np.abs(cos - ref_cos).max() < 1e-5
    assert ctx.stats()["uncertified"] == 0
    idx.close() if hasattr(idx, "close") else None


def test_profiling_stats(ctx):
    rng = np.random.default_rng(4)
    x = rng.standard_normal((4096, 1024)).astype(np.float32)
    idx = _index(ctx, x)
    ctx.stats_reset()
    ctx.set_profiling(True)
    for _ in range(3):
        idx.search(x[:8], 10)
    st = ctx.stats()
    ctx.set_profiling(False)
    assert st["scan_calls"] == 3 and st["search_calls"] == 3 and st["scan_ms"] > 0
    assert st["scan_flops"] == 2 * 4096 * 1024 * 8 and st["scan_rows"] == 4096


@pytest.mark.parametrize("seed", range(24))
def test_random_shape_sweep(ctx, seed):
    """Seeded sweep over dimension, row count (tile edges, several chunks), batch (both scan kernels, padded
    query blocks) and k (up to the 256 limit), with planted exact matches and duplicated rows."""
    rng = np.random.default_rng(1000 + seed)
    d = int(rng.choice([64, 128, 192, 512, 1024, 2048]))
    n = int(rng.choice([255, 256, 257, 511, 513, 4097, 16384, 70001, 131072 + 255]))
    b = int(rng.choice([1, 2, 63, 64, 65, 100, 128, 129, 255, 256, 257, 300, 700, 1100]))
    k = int(rng.choice([1, 3, 10, 64, 100, 256]))
    x = rng.standard_normal((n, d)).astype(np.float32)
    if n > 600:
        x[n // 3] = x[5]                                   # exact duplicate: the lower id must come first
        x[17] = 0.0                                        # all-zero row: cosine 0 with everything
    q = rng.standard_normal((b, d)).astype(np.float32)
    q[0] = 2.5 * x[5]
    if b > 1:
        q[b - 1] = x[n - 1] + 0.05 * q[b - 1]
    idx = _index(ctx, x)
    cos, ids = idx.search(q, k)
    xn, qn = R.normalize_rows(x), R.normalize_rows(q)
    if n * b <= 40_000_000:
        ref_cos, ref_ids = R.knn_search(x, q, k)
    else:
        from gpu_util import exact_topk_fast
        ref_cos, ref_ids = exact_topk_fast(x, q, k, extra=64)
    assert_topk_matches(cos, ids, ref_cos, ref_ids, xn, qn)
    assert ids[0, 0] == 5 and abs(cos[0, 0] - 1.0) < 1e-5
    if n > 600 and k >= 2:
        assert ids[0, 1] == n // 3
    if b > 1:
        assert ids[b - 1, 0] == n - 1


@pytest.mark.parametrize("n_hard,b", [(7, 200), (150, 200), (40, 40)])
def test_partial_certificate_failures(ctx, n_hard, b):
    """A tight cluster of near-identical rows (differences below bf16 resolution) next to random rows:
    queries aimed at the cluster cannot be certified from bf16 scores and go through the compacted collect
    pass (the 64-query kernel for a handful of failures, 256-query blocks beyond that); the others are
    certified.  Every result must equal the float64 oracle."""
    rng = np.random.default_rng(77 + n_hard)
    d, n = 256, 30000
    x = rng.standard_normal((n, d)).astype(np.float32)
    centre = rng.standard_normal(d).astype(np.float32)
    cluster_rows = rng.choice(n, 3000, replace=False)
    x[cluster_rows] = centre + 3e-3 * rng.standard_normal((3000, d)).astype(np.float32)
    q = rng.standard_normal((b, d)).astype(np.float32)
    hard = rng.choice(b, n_hard, replace=False)
    q[hard] = centre + 3e-3 * rng.standard_normal((n_hard, d)).astype(np.float32)
    idx = _index(ctx, x)
    ctx.stats_reset()
    cos, ids = idx.search(q, 10)
    unc = ctx.stats()["uncertified"]
    assert unc >= n_hard                                  # every hard query failed its certificate ...
    assert unc <= n_hard + 3                              # ... and (almost) only those
    ref_cos, ref_ids = R.knn_search(x, q, 10)
    assert_topk_matches(cos, ids, ref_cos, ref_ids, R.normalize_rows(x), R.normalize_rows(q))
    # the hard queries' neighbours all come from the cluster, which the bf16 scan alone cannot order
    cl = set(cluster_rows.tolist())
    assert all(set(ids[h].tolist()) <= cl for h in hard)


@pytest.mark.parametrize("b", [8, 64, 100, 128, 200, 300])
def test_forced_collect_pass_is_exact(ctx, b):
    """rescore_k = k leaves the first pass no margin, so EVERY query fails its certificate and the answer comes
    from the compacted collect pass.  Random data: the top-k sets must equal the float64 oracle's exactly
    (a weaker, tolerance-based check once hid rows lost by the collect pass)."""
    rng = np.random.default_rng(50 + b)
    n, d, k = 60000, 1024, 10
    x = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((b, d)).astype(np.float32)
    idx = _index(ctx, x)
    idx.set_option("rescore_k", k)
    s = R.normalize_rows(q).astype(np.float64) @ R.normalize_rows(x).astype(np.float64).T
    want = [set(np.argsort(-s[i], kind="stable")[:k].tolist()) for i in range(b)]
    for _ in range(3):
        ctx.stats_reset()
        cos, ids = idx.search(q, k)
        assert ctx.stats()["uncertified"] == b
        bad = [i for i in range(b) if set(ids[i].tolist()) != want[i]]
        assert not bad, bad


def test_random_session_against_oracle(ctx):
    """One index through a seeded sequence of adds, in-place updates and searches with changing batch size,
    k and candidate depth (certified and forced-collect searches interleaved): internal buffers are reused
    across calls of different shapes, every answer must match the oracle on the rows present at that time."""
    from semantic_query_engine_amd import VectorIndex
    rng = np.random.default_rng(2024)
    d = 192
    idx = VectorIndex(ctx, d)
    x = np.zeros((0, d), np.float32)
    for step in range(14):
        add = rng.standard_normal((int(rng.choice([1, 300, 5000, 20000])), d)).astype(np.float32)
        idx.add(add)
        x = np.concatenate([x, add], 0)
        if x.shape[0] > 10 and step % 3 == 1:
            rows = rng.choice(x.shape[0], 5, replace=False)
            new = rng.standard_normal((5, d)).astype(np.float32)
            idx.update(rows, new)
            x[rows] = new
        b = int(rng.choice([1, 7, 64, 90, 130, 400]))
        k = int(rng.choice([1, 5, 10, 40]))
        q = rng.standard_normal((b, d)).astype(np.float32)
        if x.shape[0] > 3:
            q[0] = 1.7 * x[rng.integers(0, x.shape[0])]
        idx.set_option("rescore_k", float(k if step % 4 == 2 else 0))      # every fourth search: all queries collected
        cos, ids = idx.search(q, k)
        ref_cos, ref_ids = R.knn_search(x, q, k)
        assert_topk_matches(cos, ids, ref_cos, ref_ids, R.normalize_rows(x), R.normalize_rows(q))
    assert len(idx) == x.shape[0]


@pytest.mark.parametrize("b,k", [(200, 16), (600, 10), (64, 16), (300, 1), (300, 3), (300, 17), (300, 32), (300, 33)])
def test_k_row_bound_keeps_rows_inside_the_error_band(ctx, b, k):
    """The scan drops rows below (a score k distinct rows reach) - 2 eps without a certificate (scan_common.h:
    refresh_apply).  Per query, 40 planted rows whose true cosines differ by 1e-5 -- far below the bf16 noise of
    the scan scores (~1e-4) -- so the bf16 order of the true top-k is scrambled and several of them score BELOW
    the k-row bound itself; only the 2 eps slack keeps them.  Enough rows for the cross-chunk bound to be active
    (>= 64 chunks).  Ids and order must equal the float64 oracle.  The k values cover every form of the bound: the
    k-th largest of the 16 group maxima (k <= 16), the minimum over pairs of columns (17 .. 32), and none beside the
    kp-row bound (33 and more)."""
    rng = np.random.default_rng(900 + b + k)
    d, n, planted = 256, 40000, 40
    x = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((b, d)).astype(np.float32)
    qn = R.normalize_rows(q).astype(np.float64)
    rows = rng.permutation(n)[:b * planted].reshape(b, planted)
    for i in range(b):
        noise = rng.standard_normal((planted, d))
        noise -= (noise @ qn[i])[:, None] * qn[i][None, :]
        noise /= np.linalg.norm(noise, axis=1, keepdims=True)
        c = 0.6 - 1e-5 * rng.permutation(planted)
        x[rows[i]] = (c[:, None] * qn[i][None, :] + np.sqrt(1 - c * c)[:, None] * noise).astype(np.float32)
    idx = _index(ctx, x)
    cos, ids = idx.search(q, k)
    ref_cos, ref_ids = R.knn_search(x, q, k)
    assert_topk_matches(cos, ids, ref_cos, ref_ids, R.normalize_rows(x), R.normalize_rows(q))
    assert all(set(ids[i].tolist()) <= set(rows[i].tolist()) for i in range(b))


@pytest.mark.parametrize("d,n,b,k", [(4096, 20000, 150, 10), (8192, 17000, 40, 10), (8192, 17000, 136, 3)])
def test_certificate_margin_scales_with_dim(ctx, d, n, b, k):
    """r02 verdict / advisor: scan_eps carried a fixed 2e-4 for the rounding of the two fp32 accumulations, which the
    worst case K * 2^-23 exceeds above dim ~1700 while sqe_index_create admits dim 8192; the term now grows with K
    (kernels.h).  Near-tie rows at the k-th place at dims 4096 and 8192 -- planted cosines 4e-5 apart, below the bf16
    noise of the scan, enough chunks for the k-row bound (which drops rows with NO certificate behind them) to be on
    -- must come back exactly as the float64 oracle orders them (rows closer than the fp32 re-score can resolve at
    these dims, 2e-5, may swap)."""
    rng = np.random.default_rng(7000 + d + b)
    planted = 30
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((b, d), dtype=np.float32)
    qn = R.normalize_rows(q).astype(np.float64)
    rows = rng.permutation(n)[:b * planted].reshape(b, planted)
    for i in range(b):
        noise = rng.standard_normal((planted, d))
        noise -= (noise @ qn[i])[:, None] * qn[i][None, :]
        noise /= np.linalg.norm(noise, axis=1, keepdims=True)
        c = 0.5 - 4e-5 * rng.permutation(planted)
        x[rows[i]] = (c[:, None] * qn[i][None, :] + np.sqrt(1 - c * c)[:, None] * noise).astype(np.float32)
    idx = _index(ctx, x)
    cos, ids = idx.search(q, k)
    ref_cos, ref_ids = R.knn_search(x, q, k)
    assert_topk_matches(cos, ids, ref_cos, ref_ids, R.normalize_rows(x), R.normalize_rows(q), tol=2e-5)
    assert all(set(ids[i].tolist()) <= set(rows[i].tolist()) for i in range(b))
    idx.close() if hasattr(idx, "close") else None
