"""Pins the oracle restatement on the golden vectors produced by the reference's own
(ast-lifted) functions -- tests/golden/make_golden.py.  CPU only."""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import retrieval as R
from tests import golden_cases as G


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_cosine_similarity_matches_reference_vectors(golden_dir):
    exp = _load(golden_dir, "cosine_pairs.npz")["expected"]
    a, b = G.cosine_cases()
    with np.errstate(all="ignore"):
        got = np.array([R.cosine_similarity(x, y) for x, y in zip(a, b)])
    assert isinstance(R.cosine_similarity(a[0], b[0]), float)
    nan = np.isnan(exp)
    assert nan[19] and np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got[~nan], exp[~nan])          # same NumPy expression -> bit equal
    assert exp[16] == 0.0 and exp[17] == 0.0 and exp[18] == 0.0   # zero-norm rule
    assert exp[24] == 1.0


def test_cosine_c_port_matches_reference_vectors(golden_dir, oracle_c):
    exp = _load(golden_dir, "cosine_pairs.npz")["expected"]
    a, b = G.cosine_cases()
    fp = ctypes.POINTER(ctypes.c_float)
    for i in range(64):
        got = oracle_c.oracle_cosine_similarity(a[i].ctypes.data_as(fp), b[i].ctypes.data_as(fp), 1024)
        if np.isnan(exp[i]):
            assert np.isnan(got)
        elif i == 22:                     # denormal-range inputs: fp32 norm underflow differs
            assert abs(got - exp[i]) < 1e-3 or exp[i] == 0.0
        else:
            assert abs(got - exp[i]) <= 2e-6, (i, got, exp[i])


def test_normalize_matches_reference_statements(golden_dir, oracle_c):
    exp = _load(golden_dir, "normalize_rows.npz")["expected"]
    e = G.normalize_case()
    got = R.normalize_rows(e)
    assert got.dtype == np.float32 and np.array_equal(got, exp)
    assert not np.isnan(exp).any() and np.all(exp[3] == 0.0)       # zero row stays zero
    out = np.empty_like(e)
    fp = ctypes.POINTER(ctypes.c_float)
    oracle_c.oracle_normalize_rows(e.ctypes.data_as(fp), ctypes.c_int64(32), 1024, out.ctypes.data_as(fp))
    big = np.abs(exp) > 1e-30
    assert np.allclose(out[big], exp[big], rtol=3e-7, atol=0)


def test_knn_known_answer(golden_dir, oracle_c):
    g = _load(golden_dir, "knn_small.npz")
    x, q = G.knn_case()
    cos, ids = R.knn_search(x, q, 10)
    assert np.array_equal(ids, g["ids"]) and np.allclose(cos, g["cos"], rtol=0, atol=1e-12)
    # planted structure: query 0's neighbour is row 5 and its 40 duplicates at 3000.. ->
    # equal scores resolve to the lowest ids
    assert ids[0].tolist() == [5] + list(range(3000, 3009))
    assert ids[15].tolist() == list(range(10)) and np.all(cos[15] == 0.0)   # zero query
    for i in range(2, 8):
        assert ids[i, 0] == 37 * i + 5
    # row 200 = 2.5 * row 42: same cosine up to fp32 normalisation rounding (a near-tie
    # the GPU tests must treat as one)
    assert set(ids[1, :2].tolist()) == {42, 200} and abs(cos[1, 0] - cos[1, 1]) < 1e-6
    # the plain-C port agrees on ids and to 1e-9 on scores
    xn, qn = R.normalize_rows(x), R.normalize_rows(q)
    c_cos = np.empty((16, 10)); c_ids = np.empty((16, 10), dtype=np.int64)
    fp = ctypes.POINTER(ctypes.c_float)
    oracle_c.oracle_exact_topk(xn.ctypes.data_as(fp), ctypes.c_int64(4096), qn.ctypes.data_as(fp), 16, 1024, 10,
                               c_cos.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                               c_ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    assert np.array_equal(c_ids, ids) and np.allclose(c_cos, cos, atol=1e-9)


def test_knn_edge_cases():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((7, 64)).astype(np.float32)
    q = rng.standard_normal((3, 64)).astype(np.float32)
    cos, ids = R.knn_search(x, q, 10)                  # k > N: -1 padded
    assert np.all(ids[:, 7:] == -1) and np.all(np.isneginf(cos[:, 7:]))
    assert sorted(ids[0, :7].tolist()) == list(range(7))
    s = R.os_score_from_cosine(cos[:, :7])
    assert np.all(np.diff(s, axis=1) <= 0) and np.all(s <= 1.0 + 1e-12)
    assert R.recall_at_k(ids, ids) == 1.0


def test_cache_trace(golden_dir):
    t = json.load(open(os.path.join(golden_dir, "cache_trace.json")))
    base = G.cache_base(t["seed"])
    cache = R.LfuCacheOracle(max_items=t["max_items"])
    for op in t["ops"]:
        if op["op"] == "put":
            cache.put(base[op["vec"]:op["vec"] + 1], f"resp{op['vec']}")
        else:
            r = cache.get(np.array([op["query"]], dtype=np.float32))
            assert r == op["result"] and cache.last_index == op["index"]
            assert cache.last_sim == pytest.approx(op["sim"], abs=1e-7)
        assert cache.responses() == op["responses"] and cache.freqs() == op["freqs"]
    # a duplicate embedding inserted later sits at index 0 and wins the tie (strict >)
    last_get = [o for o in t["ops"] if o["op"] == "get"][-1]
    assert last_get["index"] == 0 and last_get["result"] == "resp2"


def test_cosine_best_first_strict_max():
    rng = np.random.default_rng(1)
    m = rng.standard_normal((50, 32)).astype(np.float32)
    m[30] = m[10]; m[40] = 0.0; m[41, 0] = np.nan
    sim, idx = R.cosine_best(m, m[10])
    assert idx == 10 and sim == pytest.approx(1.0, abs=1e-6)
    assert R.cosine_best(np.zeros((3, 8), np.float32), np.ones(8, np.float32)) == (0.0, 0)
    assert R.cosine_best(np.full((3, 8), np.nan, np.float32), np.ones(8, np.float32)) == (-1.0, -1)
    assert R.cosine_best(np.zeros((0, 8), np.float32), np.ones(8, np.float32)) == (-1.0, -1)


def test_chunker_small(golden_dir):
    t = json.load(open(os.path.join(golden_dir, "chunker_small.json")))
    for text, exp in zip(t["inputs"], t["expected"]):
        assert R.chunk_text(R.basic_cleaning(text), t["size"]) == exp
    assert R.chunk_text("a b  c\td\ne", 2) == ["a b", "c d", "e"]


@pytest.mark.skipif(not os.path.isdir("/root/reference/PMC"), reason="corpus only in the build container")
def test_chunker_corpus(golden_dir):
    t = json.load(open(os.path.join(golden_dir, "chunker.json")))
    assert t["total"] == 32717 and len(t["counts"]) == 3027
    docs = R.corpus_docs("/root/reference/PMC", files=sorted(t["sha256"]))
    by_file = {}
    for d in docs:
        by_file.setdefault(d["doc_id"], []).append(hashlib.sha256(d["text"].encode("utf-8")).hexdigest())
    assert by_file == t["sha256"]
    for f, hs in by_file.items():
        assert len(hs) == t["counts"][f]
