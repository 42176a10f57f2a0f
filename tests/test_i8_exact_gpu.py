"""Bit-exact parity of the int8 kernels' INTEGER arithmetic (scan_i8.hip: scan_i8_pp_kernel, scan_i8_small_kernel,
sample_i8_pp_kernel), through the C ABI (sqe_index_i8_last / sqe_index_i8_read).

The collect scan computes exact int32 dot products of the int8 copies and applies a fixed integer predicate, so what ONE launch
appends is a function of its operands alone:

    { (acc[r, q] * s[r], r) : acc[r, q] * s[r] >= thr[q], r < rows, q < B },        acc = X8 . Q8^T  (int32)

The tests read the launch's own operands back (the int8 copy, the row scales, the quantised queries, the thresholds), recompute
that set in NumPy and compare it with the keys the launch wrote -- every one of the rows x B scores takes part, so one stale
operand byte anywhere in the launch shows (the final top-k would not: the fp32 re-score, the certificate and the bf16 fallback
mask exactly that kind of corruption).  The threshold pass is checked the same way: the two best (score, row) of every lane
stream.  Integer work: the bar is equality.  This file runs before test_i8_gpu.py on purpose."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from semantic_query_engine_amd import Context
    return Context(0)


def _i8_index(ctx, dim, step, m):
    from semantic_query_engine_amd import SCAN_INT8_RESCORE, VectorIndex
    idx = VectorIndex(ctx, dim)
    idx.set_option("scan_mode", SCAN_INT8_RESCORE)
    idx.set_option("i8_min_rows", 0)
    idx.set_option("i8_sample_step", step)
    idx.set_option("i8_sample_m", m)
    return idx


def _chunk_range(n_tiles, n_chunks, c):
    """scan_common.h: chunk_tile_range -- tiles dealt out evenly, the first (n_tiles % n_chunks) chunks one longer"""
    base, rem = divmod(n_tiles, n_chunks)
    begin = c * base + min(c, rem)
    return begin, begin + base + (1 if c < rem else 0)


def _untile(raw, L, t0, t1):
    """tiles [t0, t1) of the tiled int8 copy -> int8 [(t1 - t0) * tile_rows, dim] row-major"""
    hs, tr = L["dim"] // 64, L["tile_rows"]
    t = raw.reshape(t1 - t0, L["tile_stride"])[:, : hs * tr * 64].reshape(t1 - t0, hs, tr, 64)
    return np.ascontiguousarray(t.transpose(0, 2, 1, 3)).reshape((t1 - t0) * tr, L["dim"])


def _exact_acc(x8, q8, device=None):
    """X8 . Q8^T exactly.  |acc| <= dim * 127^2 < 2^24 for dim <= 1040, and every partial sum is an integer below 2^24 too, so an
    fp32 GEMM is exact whatever its summation order; larger dims go through float64.  device: a torch device for the big cases
    (checked against the NumPy product on the first rows by the caller)."""
    big = x8.shape[1] * 127 * 127 >= (1 << 24)
    if device is None:
        dt = np.float64 if big else np.float32
        return (x8.astype(dt) @ q8.astype(dt).T).astype(np.int64)
    import torch
    dt = torch.float64 if big else torch.float32
    xa = torch.from_numpy(x8).to(device).to(dt)
    qa = torch.from_numpy(q8).to(device).to(dt)
    return (xa @ qa.T).to(torch.int64).cpu().numpy()


def _expected_keys(acc, scales, thr, row0, rows, B):
    """-> (query, key) of every (row, query) that passes the collect predicate, as the kernel packs a key"""
    score = acc[:, :B] * scales[:, None].astype(np.int64)
    assert np.abs(score).max() < (1 << 31)
    r, q = np.nonzero(score >= thr[None, :B].astype(np.int64))
    keep = row0 + r < rows
    r, q = r[keep], q[keep]
    sc = score[r, q].astype(np.int64)
    key = (((sc & 0xFFFFFFFF) ^ 0x80000000).astype(np.uint64) << np.uint64(32)) | (0xFFFFFFFF - (row0 + r)).astype(np.uint64)
    return q.astype(np.int64), key


def _launch_lists(idx, L):
    """-> (query, key, chunk) of every key the collect launch wrote -- its (chunk, query) lists, then the queries' overflow pools
    (chunk -1: the keys that found their list full) -- and the queries whose POOL overflowed (keys were dropped)"""
    from semantic_query_engine_amd import engine as E
    nc, bp, cap, pcap = L["n_chunks"], L["b_pad"], L["list_cap"], L["pool_cap"]
    cnt = idx.i8_read(E.I8_LIST_COUNTS, np.int32, nc * bp).reshape(nc, bp)
    lists = idx.i8_read(E.I8_LISTS, np.uint64, nc * bp * cap).reshape(nc, bp, cap)
    pcnt = idx.i8_read(E.I8_POOL_COUNTS, np.int32, bp)
    assert cnt.min() >= 0 and pcnt.min() >= 0
    assert not cnt[:, L["B"]:].any() and not pcnt[L["B"]:].any(), "a padding query collected keys"
    assert np.array_equal(np.maximum(cnt - cap, 0).sum(axis=0), pcnt), "what does not fit a list goes to the pool, nothing else does"
    live = np.arange(cap)[None, None, :] < np.minimum(cnt, cap)[:, :, None]
    c, q, _ = np.nonzero(live)
    q, keys = q.astype(np.int64), lists[live]
    if pcnt.any():
        pools = idx.i8_read(E.I8_POOLS, np.uint64, bp * pcap).reshape(bp, pcap)
        plive = np.arange(pcap)[None, :] < np.minimum(pcnt, pcap)[:, None]
        pq, _ = np.nonzero(plive)
        q, keys, c = np.concatenate([q, pq]), np.concatenate([keys, pools[plive]]), np.concatenate([c, np.full(pq.size, -1)])
    return q, keys, c, set(np.nonzero(pcnt > pcap)[0].tolist())


def _check_collect(idx, L, device=None, block_tiles=256):
    from semantic_query_engine_amd import engine as E
    assert L["uncertified"] == 0, "the bf16 collect pass reused the list buffers: pick data that certifies"
    tr, dim, B = L["tile_rows"], L["dim"], L["B"]
    tiles = (L["rows"] + tr - 1) // tr
    scales = idx.i8_read(E.I8_ROW_SCALES, np.uint32, tiles * tr)
    assert np.array_equal(scales.reshape(tiles, tr), np.repeat(scales[::tr, None], tr, 1)), "one scale per tile (quant.hip)"
    q8 = idx.i8_read(E.I8_QUERIES, np.int8, L["b_pad"] * L["q_pitch"]).reshape(L["b_pad"], L["q_pitch"])[:, :dim]
    assert not q8[B:].any()
    thr = idx.i8_read(E.I8_THRESHOLDS, np.int32, L["b_pad"])
    got_q, got_key, got_chunk, over = _launch_lists(idx, L)
    # every key sits in the list of the chunk that owns its row
    got_row = (0xFFFFFFFF - (got_key & np.uint64(0xFFFFFFFF))).astype(np.int64)
    bounds = np.array([_chunk_range(tiles, L["n_chunks"], c)[0] for c in range(L["n_chunks"])] + [tiles]) * tr
    in_list = got_chunk >= 0
    assert np.array_equal((np.searchsorted(bounds, got_row, side="right") - 1)[in_list], got_chunk[in_list])
    exp_q, exp_key = [], []
    first = True
    for t0 in range(0, tiles, block_tiles):
        t1 = min(tiles, t0 + block_tiles)
        raw = idx.i8_read(E.I8_ROWS, np.int8, (t1 - t0) * L["tile_stride"], t0 * L["tile_stride"])
        x8 = _untile(raw, L, t0, t1)
        acc = _exact_acc(x8, q8, device)
        if device is not None and first:                       # the device GEMM against NumPy on the first rows
            assert np.array_equal(acc[:2048], _exact_acc(x8[:2048], q8))
            first = False
        eq, ek = _expected_keys(acc, scales[t0 * tr:t1 * tr], thr, t0 * tr, L["rows"], B)
        exp_q.append(eq)
        exp_key.append(ek)
    exp_q, exp_key = np.concatenate(exp_q), np.concatenate(exp_key)
    if over:                                                   # an overflowed pool kept an arbitrary subset: those queries are not compared
        assert len(over) <= max(1, B // 50), over
        keep_g, keep_e = ~np.isin(got_q, list(over)), ~np.isin(exp_q, list(over))
        got_q, got_key, exp_q, exp_key = got_q[keep_g], got_key[keep_g], exp_q[keep_e], exp_key[keep_e]
    go, eo = np.lexsort((got_key, got_q)), np.lexsort((exp_key, exp_q))
    assert got_q.size == exp_q.size, (got_q.size, exp_q.size)
    assert np.array_equal(got_q[go], exp_q[eo]) and np.array_equal(got_key[go], exp_key[eo])
    assert got_q.size > 0
    return got_q.size


def _check_sample(idx, L, device=None):
    """sample_i8_pp_kernel: per (chunk, query, row lane) the two best scaled scores of the lane's stream, first seen wins a tie"""
    from semantic_query_engine_amd import engine as E
    if not L["sample_int8"]:
        return
    tr, dim, B, step = L["tile_rows"], L["dim"], L["B"], L["sample_step"]
    nts, ncs, bps = L["sample_tiles"], L["sample_chunks"], L["sample_b_pad"]
    out = idx.i8_read(E.I8_SAMPLE_BEST, np.int32, ncs * bps * 16 * 2).reshape(ncs, bps, 8, 2, 2)
    tiles = (L["rows"] + tr - 1) // tr
    scales = idx.i8_read(E.I8_ROW_SCALES, np.uint32, tiles * tr)
    q8 = idx.i8_read(E.I8_QUERIES, np.int8, L["b_pad"] * L["q_pitch"]).reshape(L["b_pad"], L["q_pitch"])[:B, :dim]
    # stream position of a tile row: lane l = (row >> 7) * 4 + ((row >> 2) & 3), order inside the tile (row >> 4) & 7, row & 3
    rr = np.arange(tr)
    lane_of = (rr >> 7) * 4 + ((rr >> 2) & 3)
    order_in_tile = ((rr >> 4) & 7) * 4 + (rr & 3)
    for c in range(ncs):
        tb, te = _chunk_range(nts, ncs, c)
        if te == tb:
            continue
        sc_all = []
        for t in range(tb, te):
            raw = idx.i8_read(E.I8_ROWS, np.int8, L["tile_stride"], t * step * L["tile_stride"])
            acc = _exact_acc(_untile(raw, L, 0, 1), q8, device)
            sc_all.append(acc * scales[t * step * tr:(t * step + 1) * tr, None].astype(np.int64))
        sc = np.stack(sc_all)                                  # [tiles of the chunk, tile rows, queries]
        for lane in range(8):
            rows_l = rr[lane_of == lane]
            rows_l = rows_l[np.argsort(order_in_tile[rows_l], kind="stable")]
            stream = sc[:, rows_l, :].reshape(-1, B)                                     # stream order: tile, then (i, j)
            srow = (np.arange(tb, te)[:, None] * step * tr + rows_l[None, :]).reshape(-1)
            o = np.argsort(-stream, axis=0, kind="stable")[:2]                             # first seen wins a tie
            best = np.take_along_axis(stream, o, 0)
            assert np.array_equal(out[c, :B, lane, :, 0].T, best), (c, lane)
            assert np.array_equal(out[c, :B, lane, :, 1].T, srow[o]), (c, lane)


@pytest.mark.parametrize("n,d,b", [(200_000, 1024, 256), (150_000 + 77, 512, 700), (120_000, 256, 1024),
                                   (90_000 + 1, 1024, 100), (90_000, 512, 33)])
def test_collect_set_is_bit_exact(ctx, n, d, b):
    """The ping-pong kernel (256-query blocks: b = 256 / 700 / 1024, two of them with a partial last tile) and the staged 128- /
    64-query kernels (b = 100 / 33) against NumPy, plus the threshold pass's per-lane best-two lists."""
    rng = np.random.default_rng(n + b)
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((b, d), dtype=np.float32)
    idx = _i8_index(ctx, d, step=8, m=64)
    idx.add(x)
    ctx.stats_reset()
    idx.search(q, 10)
    L = idx.i8_last()
    assert (L["rows"], L["dim"], L["B"]) == (n, d, b)
    assert L["query_block"] == (256 if b > 128 else 128 if b > 64 else 64)
    keys = _check_collect(idx, L)
    assert keys == ctx.stats()["i8_collected"]
    _check_sample(idx, L)
    idx.close()


def test_collect_set_600k_rows_the_r03_failure_shape(ctx):
    """600 k x 1024 rows, 700 queries: the shape on which r03's schedule lost keys (GPUTEST_r03: 359,428 vs 359,432 collected
    between identical calls -- group 0 retired its DMA pieces behind the barrier that its sibling waves' reads relied on).  One
    launch, 4.2e8 scores, compared once."""
    import torch
    n, d, b = 600_000, 1024, 700
    rng = np.random.default_rng(17)
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((b, d), dtype=np.float32)
    idx = _i8_index(ctx, d, step=8, m=64)
    idx.add(x)
    ctx.stats_reset()
    idx.search(q, 10)
    L = idx.i8_last()
    keys = _check_collect(idx, L, device=torch.device("cuda", 0))
    assert keys == ctx.stats()["i8_collected"]
    _check_sample(idx, L, device=torch.device("cuda", 0))
    idx.close()


def test_collect_set_ten_million_rows(ctx):
    """BASELINE.json's headline launch, 10 M x 1024 rows x 1024 queries with the default options: 1.05e10 scores against a blocked
    exact GEMM of the launch's own operands (torch fp32 on the device, itself checked against NumPy on the first rows)."""
    import torch
    from semantic_query_engine_amd import VectorIndex
    dev = torch.device("cuda", 0)
    n, d, b = 10_000_000, 1024, 1024
    idx = VectorIndex(ctx, d)
    idx.reserve(n)
    block = 1 << 20
    for i in range((n + block - 1) // block):
        rows = min(block, n - i * block)
        g = torch.Generator(device=dev).manual_seed(991 + i)
        xb = torch.randn((rows, d), generator=g, device=dev, dtype=torch.float32)
        torch.cuda.synchronize()
        idx.add_device(xb.data_ptr(), rows)
        ctx.synchronize()
        del xb
    g = torch.Generator(device=dev).manual_seed(5)
    q = torch.randn((b, d), generator=g, device=dev).cpu().numpy()
    ctx.stats_reset()
    idx.search(q, 10)
    L = idx.i8_last()
    assert (L["rows"], L["B"], L["query_block"]) == (n, b, 256)
    keys = _check_collect(idx, L, device=dev, block_tiles=1024)
    assert keys == ctx.stats()["i8_collected"]
    idx.close()
