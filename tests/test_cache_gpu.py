"""Parity of the cache-scan kernel and the LFU mirror against the oracle.  GPU only."""
import json
import os

import numpy as np
import pytest

from oracle import retrieval as R
from tests import golden_cases as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from semantic_query_engine_amd import Context
    return Context(0)


def test_cosine_matches_reference_golden(ctx, golden_dir):
    exp = np.load(os.path.join(golden_dir, "cosine_pairs.npz"))["expected"]
    a, b = G.cosine_cases()
    for i in range(64):
        got = float(ctx.cosine_all(b[i:i + 1], a[i])[0])
        if np.isnan(exp[i]):
            assert np.isnan(got)
        elif i == 22:          # denormal-range inputs: fp32 sum of squares underflows either way
            assert got == 0.0 or abs(got - exp[i]) < 1e-3
        else:
            assert abs(got - exp[i]) <= 1e-6, (i, got, exp[i])
    assert float(ctx.cosine_all(b[16:17], a[16])[0]) == 0.0       # zero-norm rule


def test_cosine_best_first_strict_max(ctx):
    rng = np.random.default_rng(1)
    m = rng.standard_normal((1000, 1024)).astype(np.float32)
    m[700] = m[10]; m[40] = 0.0; m[41, 0] = np.nan
    sim, idx = ctx.cosine_best(m, m[10] * 2.0)
    rs, ri = R.cosine_best(m, m[10] * 2.0)
    assert idx == ri == 10 and abs(sim - rs) < 1e-6
    q = rng.standard_normal(1024).astype(np.float32)
    sim, idx = ctx.cosine_best(m, q)
    rs, ri = R.cosine_best(m, q)
    assert idx == ri and abs(sim - rs) < 1e-6
    sims = ctx.cosine_all(m, q)
    ref = R.cosine_all(m, q)
    ok = ~np.isnan(ref)
    assert np.isnan(sims[41]) and np.abs(sims[ok] - ref[ok]).max() < 1e-6
    assert ctx.cosine_best(np.zeros((3, 8), np.float32), np.ones(8, np.float32)) == (0.0, 0)
    assert ctx.cosine_best(np.full((3, 8), np.nan, np.float32), np.ones(8, np.float32)) == (-1.0, -1)
    assert ctx.cosine_best(np.zeros((0, 8), np.float32), np.ones(8, np.float32)) == (-1.0, -1)
    assert ctx.cosine_best(-np.ones((2, 8), np.float32), np.ones(8, np.float32)) == (-1.0, -1)  # -1.0 is not > -1.0


def test_lfu_cache_trace(ctx, golden_dir):
    from semantic_query_engine_amd.retrieval import SemanticLfuCache
    t = json.load(open(os.path.join(golden_dir, "cache_trace.json")))
    base = G.cache_base(t["seed"])
    cache = SemanticLfuCache(ctx, max_items=t["max_items"])
    for op in t["ops"]:
        if op["op"] == "put":
            cache.put(base[op["vec"]:op["vec"] + 1], f"resp{op['vec']}")
        else:
            r = cache.get(np.array([op["query"]], dtype=np.float32))
            assert r == op["result"] and cache.last_index == op["index"]
            assert cache.last_sim == pytest.approx(op["sim"], abs=1e-6)
        assert cache.responses() == op["responses"] and cache.freqs() == op["freqs"]


def test_lfu_cache_random_against_oracle(ctx):
    from semantic_query_engine_amd.retrieval import SemanticLfuCache
    rng = np.random.default_rng(9)
    gpu = SemanticLfuCache(ctx, max_items=16, dim=256)
    ora = R.LfuCacheOracle(max_items=16)
    pool = rng.standard_normal((40, 256)).astype(np.float32)
    for step in range(200):
        i = int(rng.integers(0, 40))
        v = (pool[i] + float(rng.choice([0.0, 0.05, 0.5])) * rng.standard_normal(256).astype(np.float32))[None]
        if rng.random() < 0.5:
            assert gpu.get(v) == ora.get(v)
            assert gpu.last_index == ora.last_index
        else:
            gpu.put(v, f"r{step}"); ora.put(v, f"r{step}")
        assert gpu.responses() == ora.responses() and gpu.freqs() == ora.freqs()
