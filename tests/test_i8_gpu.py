"""The int8 first-pass scan (scan_mode = SQE_SCAN_INT8_RESCORE: quant.hip, scan_i8.hip, select_i8.hip) against the exact
oracle.  Whatever the int8 rounding does, the answer must be the exact fp32 top-k: ids bit-exact (rows closer than fp32
resolves may swap), cosines within 1e-3 (north_star) -- measured ~1e-7, they are fp32 re-scores.  The cases cover what is
new in this path: the sample-derived thresholds (good, too high, too low), the per-row scales (rows with one huge
element, zero rows), list overflow and certificate failure (-> the bf16 collect pass), a partial last tile, lazy
quantisation across appends and index growth, overwritten rows.  GPU only."""
import numpy as np
import pytest

from oracle import retrieval as R
from tests.gpu_util import assert_topk_matches, exact_topk_fast

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from semantic_query_engine_amd import Context
    return Context(0)


def _i8_index(ctx, dim, step=8, m=32, min_rows=0):
    from semantic_query_engine_amd import SCAN_INT8_RESCORE, VectorIndex
    idx = VectorIndex(ctx, dim)
    idx.set_option("scan_mode", SCAN_INT8_RESCORE)
    idx.set_option("i8_min_rows", min_rows)
    idx.set_option("i8_sample_step", step)
    idx.set_option("i8_sample_m", m)
    return idx


def _check(ctx, idx, x, q, k, want_i8=True, **kw):
    ctx.stats_reset()
    cos, ids = idx.search(q, k)
    st = ctx.stats()
    ref_cos, ref_ids = exact_topk_fast(x, q, k)
    assert_topk_matches(cos, ids, ref_cos, ref_ids, R.normalize_rows(x), R.normalize_rows(q), **kw)
    assert np.abs(cos - ref_cos)[ref_ids >= 0].max() < 1e-5
    if want_i8:
        assert st["i8_collected"] > 0 and st["sample_ms"] >= 0.0      # the int8 path answered (not the bf16 scan)
    return cos, ids, st


@pytest.mark.parametrize("n,d,b,k,unc_max", [(300_000, 1024, 300, 10, 0.02), (120_001, 256, 1024, 10, 0.02), (70_000, 512, 129, 1, 0.02),
                                             (100_000, 1024, 700, 32, 0.5),
                                             # the staged 64- / 128-query kernels (HBM-bound batches; the reference sends ONE query)
                                             (200_000, 1024, 1, 10, 1.0), (150_000, 512, 64, 10, 0.05), (150_003, 256, 100, 5, 0.05),
                                             (150_000, 1024, 128, 10, 0.05)])
def test_gaussian_rows_exact_and_certified(ctx, n, d, b, k, unc_max):
    """Sample every 8th tile, threshold = 64th best cosine of the sample (~512 rows collected per query): at these index
    sizes the 10th -> 512th gap is ~1 sigma against an int8 bound of ~0.75 sigma (the 10 M-row defaults, every 50th tile
    and the 32nd best, leave 1.15 sigma); with k = 32 the gap is thinner than the bound for many queries, which then
    take the bf16 pass -- the answers are exact either way."""
    rng = np.random.default_rng(n + b)
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((b, d), dtype=np.float32)
    plant = rng.integers(0, n, b // 2)
    q[: b // 2] = x[plant] + 0.1 * q[: b // 2]
    idx = _i8_index(ctx, d, step=8, m=64)
    idx.add(x)
    cos, ids, st = _check(ctx, idx, x, q, k)
    assert np.array_equal(ids[: b // 2, 0], plant[: b // 2])
    assert st["i8_overflows"] == 0
    assert st["uncertified"] <= max(1, int(b * unc_max)), st         # the int8 certificate holds for (almost) every query
    assert st["i8_rescored"] <= st["i8_collected"]
    # another batch size = another kernel configuration -- same answers
    c1, i1 = idx.search(q[:7], k)
    assert np.array_equal(i1, ids[:7]) and np.allclose(c1, cos[:7], atol=2e-6)
    idx.close()


def test_default_options_on_a_shard_sized_index(ctx):
    """1.1 M x 256 rows with the DEFAULT options (int8 is the default of a flat index from 1 M rows on): 4,297 tiles, where the
    threshold pass adapts its sample -- at least 128 tiles (every 33rd instead of every 100th) and a deeper place of it (api.hip) --
    the size of one shard of the 10 M-row index on eight devices.  Exact answers, the int8 path answers, (almost) all certified."""
    from semantic_query_engine_amd import VectorIndex
    n, d, b, k = 1_100_000, 256, 512, 10
    rng = np.random.default_rng(77)
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((b, d), dtype=np.float32)
    plant = rng.integers(0, n, b // 2)
    q[: b // 2] = x[plant] + 0.1 * q[: b // 2]
    idx = VectorIndex(ctx, d)
    idx.add(x)
    cos, ids, st = _check(ctx, idx, x, q, k)
    assert np.array_equal(ids[: b // 2, 0], plant[: b // 2])
    assert st["i8_overflows"] == 0
    assert st["uncertified"] <= max(1, int(b * 0.05)), st
    # ~2,000 keys per query expected (step x place ~ 2,000): the adapted sample keeps the collection where the options put it
    assert 500 * b <= st["i8_collected"] <= 6000 * b, st
    idx.close()


@pytest.mark.parametrize("b", [300, 40])
def test_bf16_threshold_pass_option(ctx, b):
    """`i8_sample_int8 = 0` keeps the r03a threshold pass (bf16 scan + fp32 re-score of the row sample) in front of the int8
    collect scan; the default is the int8 sample scan + order statistic.  Both place the collection, neither decides an answer:
    same ids, same cosines."""
    n, d, k = 150_000, 512, 10
    rng = np.random.default_rng(n + b)
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((b, d), dtype=np.float32)
    out = []
    for form in (1, 0):
        idx = _i8_index(ctx, d, step=8, m=64)
        idx.set_option("i8_sample_int8", form)
        idx.add(x)
        cos, ids, st = _check(ctx, idx, x, q, k)
        assert st["uncertified"] <= max(1, int(b * 0.05)), st
        out.append((cos, ids, st["i8_collected"]))
        idx.close()
    assert np.array_equal(out[0][1], out[1][1]) and np.allclose(out[0][0], out[1][0], atol=2e-6)
    assert 0.3 < out[0][2] / out[1][2] < 3.0          # the two estimates of the same order statistic collect alike


def test_identical_calls_collect_the_same_keys(ctx):
    """The number of keys the int8 scan collects is a function of EVERY estimated score against a fixed threshold: it must not
    move between identical calls.  (r03's one-barrier schedule let group 0 retire its DMA pieces behind the barrier its sibling
    waves' reads relied on, and this count moved by a few keys in two million; tests/test_i8_exact_gpu.py now checks one launch
    bit for bit against NumPy, this test keeps the repeated-call symptom; tools/repeat_scan.py is its long form.)"""
    n, d, b, k = 600_000, 1024, 700, 10
    rng = np.random.default_rng(17)
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((b, d), dtype=np.float32)
    idx = _i8_index(ctx, d, step=8, m=64)
    idx.add(x)
    seen, first = set(), None
    for it in range(25):
        ctx.stats_reset()
        cos, ids = idx.search(q, k)
        st = ctx.stats()
        seen.add((st["i8_collected"], st["i8_rescored"], st["uncertified"]))
        if first is None:
            first = (cos, ids)
        else:
            assert np.array_equal(ids, first[1]) and np.array_equal(cos, first[0])
    assert len(seen) == 1 and next(iter(seen))[0] > 0, seen
    idx.close()


def test_near_ties_at_the_kth_place(ctx):
    """40 planted rows per query whose true cosines differ by 1e-5 -- the int8 scores (noise ~1e-3) scramble them
    completely; the staged re-score must still return the exact order."""
    rng = np.random.default_rng(17)
    d, n, b, k, planted = 256, 60000, 200, 10, 40
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
    idx = _i8_index(ctx, d, step=4)
    idx.add(x)
    cos, ids, st = _check(ctx, idx, x, q, k)
    assert all(set(ids[i].tolist()) <= set(rows[i].tolist()) for i in range(b))
    idx.close()


@pytest.mark.parametrize("anchored", [True, False])
def test_clustered_rows_stay_exact(ctx, anchored):
    """Tightly clustered rows: ~2,000 rows sit inside the int8 error band of the k-th place.  With the threshold anchored on the
    sample's best true cosine (select_i8.hip: i8_sample_select_kernel) the scan collects the whole crowd and the certificate holds
    by construction; without the anchor (key budget 1: it is never used -- r03's behaviour) the m-th sample score lies inside the
    crowd, the proof fails and the bf16 collect pass answers.  The exact top-k either way."""
    rng = np.random.default_rng(23)
    d, n, b, k = 256, 80000, 300, 10
    cen = rng.standard_normal((40, d)).astype(np.float32)
    x = (cen[rng.integers(0, 40, n)] + 0.05 * rng.standard_normal((n, d))).astype(np.float32)
    q = (cen[rng.integers(0, 40, b)] + 0.05 * rng.standard_normal((b, d))).astype(np.float32)
    idx = _i8_index(ctx, d, step=4)
    if not anchored:
        idx.set_option("i8_key_budget", 1)
    idx.add(x)
    cos, ids, st = _check(ctx, idx, x, q, k, tol=5e-6)
    if anchored:
        assert st["uncertified"] <= b // 20 and st["i8_overflows"] <= b // 20, st
        assert st["i8_collected"] > 1000 * b, st             # the crowds were collected whole
    else:
        assert st["uncertified"] > 0                         # this data is what the fallback is for
    idx.close()


def test_text_like_clusters_certify_in_the_int8_pass(ctx):
    """SURVEY 8(d)'s clustered set in small: Gaussian centres, rows = centre + 0.3 x noise, ~2,400 rows per centre -- what text
    embeddings look like.  Every member of the query's cluster lies within the int8 bound of the 10th place (in-cluster cosines
    0.917 +- 0.003 against eps ~0.02); r03 paid the int8 pass and then the bf16 pass for every such query.  The anchored threshold
    falls into the gap below the cluster, the int8 pass certifies."""
    rng = np.random.default_rng(29)
    d, n, b, k, ncen = 1024, 240_000, 256, 10, 100
    cen = rng.standard_normal((ncen, d)).astype(np.float32)
    x = cen[rng.integers(0, ncen, n)] + 0.3 * rng.standard_normal((n, d), dtype=np.float32)
    q = cen[rng.integers(0, ncen, b)] + 0.3 * rng.standard_normal((b, d), dtype=np.float32)
    idx = _i8_index(ctx, d, step=8, m=20)
    idx.add(x)
    cos, ids, st = _check(ctx, idx, x, q, k, tol=5e-6)
    assert st["uncertified"] <= b // 50 and st["i8_overflows"] == 0, st
    assert 1500 * b < st["i8_collected"] < 4000 * b, st      # ~ one cluster per query
    idx.close()


def test_a_crowd_in_one_chunk_goes_through_the_overflow_pool(ctx):
    """3,000 near-identical rows appended in ONE call (the chunks of one document, a cluster added at once) sit in one or two row
    chunks: a query aimed at them collects thousands of keys from a (chunk, query) list of 512 slots.  What does not fit goes to
    the query's overflow pool (kernels.h: I8_OVF_CAP) and the int8 pass still certifies (r03: list overflow -> bf16 pass)."""
    rng = np.random.default_rng(37)
    d, n, b, k, crowd = 512, 200_000, 300, 10, 3000
    x = rng.standard_normal((n + crowd, d), dtype=np.float32)
    centre = rng.standard_normal((1, d), dtype=np.float32)
    x[n:] = centre + 3e-3 * rng.standard_normal((crowd, d), dtype=np.float32)
    q = rng.standard_normal((b, d), dtype=np.float32)
    q[:8] = centre + 3e-3 * rng.standard_normal((8, d), dtype=np.float32)
    idx = _i8_index(ctx, d, step=8, m=20)
    idx.add(x)
    cos, ids, st = _check(ctx, idx, x, q, k, tol=5e-6)
    assert np.all(ids[:8] >= n)
    assert st["uncertified"] <= 2 and st["i8_overflows"] == 0, st
    from semantic_query_engine_amd import engine as E
    L = idx.i8_last()
    pool = idx.i8_read(E.I8_POOL_COUNTS, np.int32, L["b_pad"])
    assert np.all(pool[:8] > 1000) and not pool[8:].any(), pool[:12]
    idx.close()


@pytest.mark.parametrize("step,m", [(64, 1), (1, 64)])
def test_thresholds_too_high_or_too_low(ctx, step, m):
    """(64, 1): the threshold is the BEST cosine of a 1.5 % sample -- far above the 10th best of the index, the proof
    fails for most queries.  (1, 64): every tile is in the sample and the threshold is the 64th best of the index itself
    -- few rows collected, proof margin thin.  Both must return the exact answer."""
    rng = np.random.default_rng(31 + step)
    d, n, b, k = 512, 150_000, 260, 10
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((b, d), dtype=np.float32)
    idx = _i8_index(ctx, d, step=step, m=max(m, k))
    idx.add(x)
    _check(ctx, idx, x, q, k)
    idx.close()


def test_rows_with_outlier_elements_zero_rows_and_partial_tile(ctx):
    """Per-row scales: rows dominated by ONE element (scale 8 x the typical one), all-zero rows (main.py:315-316 keeps
    them zero), rows scaled by 1e-20 / 1e+20 before normalisation; n is not a multiple of the 256-row tile."""
    rng = np.random.default_rng(41)
    d, n, b, k = 256, 50_003, 256, 10
    x = rng.standard_normal((n, d), dtype=np.float32)
    spikes = rng.permutation(n)[:500]
    x[spikes, rng.integers(0, d, 500)] += 40.0                # cosine ~0.93 with the axis
    x[rng.permutation(n)[:50]] = 0.0
    x[100:110] *= 1e-20
    x[200:210] *= 1e+18
    q = rng.standard_normal((b, d), dtype=np.float32)
    q[:64] = x[spikes[:64]] + 0.5 * q[:64]                    # queries whose neighbours are spike rows
    q[64:70] = x[200:206] * 1e-10                             # neighbours among the huge rows (the tiny ones normalise to ~0: 1e-9 rule)
    idx = _i8_index(ctx, d, step=4)
    idx.set_option("i8_max_resid", 1.0)                      # keep the int8 pass on although the spike rows quantise badly
    idx.add(x)
    cos, ids, st = _check(ctx, idx, x, q, k)
    assert np.array_equal(ids[:64, 0], spikes[:64])
    assert np.array_equal(ids[64:70, 0], np.arange(200, 206))
    idx.close()


def test_a_row_that_quantises_badly_switches_the_index_to_bf16(ctx):
    """The int8 bound uses the LARGEST rounding residual of the index: one row dominated by a single element (residual ~0.05
    at dim 1024 against 0.014 for Gaussian rows) would make every certificate fail and every query pay the int8 pass AND the bf16 pass.  The
    index notices (option i8_max_resid, default 0.02) and answers with the bf16 scan; overwriting the row brings int8 back."""
    rng = np.random.default_rng(61)
    d, n, b, k = 1024, 60_000, 300, 10
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((b, d), dtype=np.float32)
    idx = _i8_index(ctx, d, step=4, m=64)
    idx.add(x)
    _, _, st = _check(ctx, idx, x, q, k)
    assert st["i8_collected"] > 0
    onehot = rng.standard_normal((1, d), dtype=np.float32)   # one element 40 x the others: its scale is set by that element and the
    onehot[0, 5] = 40.0                                        #   other 1023 round at ~1/3 of their own size (an exact one-hot row would
    x2 = np.concatenate([x, onehot])                           #   quantise perfectly: 127 and zeros)
    idx.add(onehot)
    _, _, st = _check(ctx, idx, x2, q, k, want_i8=False)
    assert st["i8_collected"] == 0                                        # the bf16 scan answered
    idx.set_option("i8_max_resid", 1.0)                                   # forced: int8 pass, (almost) every proof fails, bf16 pass -- still exact
    _, _, st = _check(ctx, idx, x2, q, k)
    assert st["uncertified"] >= b * 3 // 4, st                            # (the anchored threshold still certifies a few queries)
    idx.close()


def test_appends_growth_and_overwrites(ctx):
    """The int8 copy is filled lazily by the first search after an add and follows index growth and row overwrites."""
    rng = np.random.default_rng(53)
    d, b, k = 256, 300, 5
    x = rng.standard_normal((90_000, d), dtype=np.float32)
    q = rng.standard_normal((b, d), dtype=np.float32)
    idx = _i8_index(ctx, d, step=4)
    idx.add(x[:20_000])
    _check(ctx, idx, x[:20_000], q, k)
    idx.add(x[20_000:20_001])                                 # one row: lands in a tile that is already quantised
    idx.add(x[20_001:61_234])                                 # grows the index (new buffers), partial tiles on both ends
    _check(ctx, idx, x[:61_234], q, k)
    idx.add(x[61_234:])
    upd = np.array([3, 19_999, 20_000, 61_233, 89_999])
    x[upd] = q[:5] * 2.0
    idx.update(upd, x[upd])
    cos, ids, st = _check(ctx, idx, x, q, k)
    assert ids[:5, 0].tolist() == upd.tolist() and np.all(cos[:5, 0] > 0.999999)
    idx.close()


def test_full_size_properties_int8():
    """BASELINE size (10 M x 1024, batch 1024, top-10) in int8 mode, through what the oracle cannot check directly:
    planted neighbours first, and the whole result EQUAL to the bf16-mode result of the same index (both are exact)."""
    import torch
    from semantic_query_engine_amd import SCAN_BF16_RESCORE, SCAN_INT8_RESCORE, Context, VectorIndex
    from tests.test_fullsize_gpu import BLOCK, D, K, N, _block
    dev = torch.device("cuda", 0)
    ctx = Context(0)
    idx = VectorIndex(ctx, D)
    idx.reserve(N)
    nblocks = (N + BLOCK - 1) // BLOCK
    for blk in range(nblocks):
        rows = min(BLOCK, N - blk * BLOCK)
        xb = _block(blk, rows, dev)
        torch.cuda.synchronize()
        idx.add_device(xb.data_ptr(), rows)
        ctx.synchronize()
        del xb
    B = 1024
    g = torch.Generator(device=dev).manual_seed(11)
    q = torch.randn((B, D), generator=g, device=dev)
    plant = torch.arange(B // 2, device=dev) * (N // B) + 5
    for blk in range(nblocks):
        lo, hi = blk * BLOCK, min(N, (blk + 1) * BLOCK)
        sel = torch.nonzero((plant >= lo) & (plant < hi)).flatten()
        if sel.numel():
            xb = _block(blk, hi - lo, dev)
            q[sel] = xb[plant[sel] - lo] * 1.3 + 0.2 * q[sel]
            del xb
    out = {}
    for mode in (SCAN_BF16_RESCORE, SCAN_INT8_RESCORE):
        idx.set_option("scan_mode", mode)
        cos = torch.empty((B, K), device=dev)
        ids = torch.empty((B, K), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        ctx.stats_reset()
        idx.search_device(q.data_ptr(), B, K, cos.data_ptr(), ids.data_ptr())
        ctx.synchronize()
        out[mode] = (cos.cpu(), ids.cpu(), ctx.stats())
    c16, i16, s16 = out[SCAN_BF16_RESCORE]
    c8, i8, s8 = out[SCAN_INT8_RESCORE]
    assert s8["i8_collected"] > 0 and s16["i8_collected"] == 0
    assert torch.equal(i8[: B // 2, 0], plant.cpu())
    assert torch.equal(i8, i16) and torch.allclose(c8, c16, atol=2e-6)
    assert s8["uncertified"] <= 4 and s8["i8_overflows"] == 0, s8
    # the re-score reads a few hundred rows per query, the collection a couple of thousand keys
    assert s8["i8_rescored"] < 1024 * 1200 and s8["i8_collected"] < 1024 * 4000, s8


def test_int8_first_pass_inside_a_device_group():
    """The shards of a multi-device context are ordinary FLAT indexes: each runs the int8 first pass on its rows (options
    set on the group index reach every shard), the [B,k] exchange and merge are unchanged, sqe_stats sums the members."""
    from semantic_query_engine_amd import EXCHANGE_COPY, SCAN_INT8_RESCORE, Context, VectorIndex
    ctx = Context(devices=[0, 0, 0], exchange=EXCHANGE_COPY)
    rng = np.random.default_rng(71)
    d, n, b, k = 256, 180_000, 300, 10
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((b, d), dtype=np.float32)
    plant = rng.integers(0, n, 100)
    q[:100] = x[plant] + 0.1 * q[:100]
    idx = VectorIndex(ctx, d)
    for key, val in (("scan_mode", SCAN_INT8_RESCORE), ("i8_min_rows", 0), ("i8_sample_step", 4), ("i8_sample_m", 64)):
        idx.set_option(key, val)
    idx.add(x)
    ctx.stats_reset()
    cos, ids = idx.search(q, k)
    st = ctx.stats()
    ref_cos, ref_ids = exact_topk_fast(x, q, k)
    assert_topk_matches(cos, ids, ref_cos, ref_ids, R.normalize_rows(x), R.normalize_rows(q))
    assert np.array_equal(ids[:100, 0], plant)
    assert st["i8_collected"] > 3 * b * 50, st               # all three shards collected
    idx.close()
