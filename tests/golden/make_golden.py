#!/usr/bin/env python3
"""Generates the committed golden fixtures in tests/golden/.

Runs ONLY in the build container (it reads /root/reference); the fixtures it writes
are data (inputs / seeds and expected outputs) and are what travels to the GPU box.

What comes from the reference itself: the three pure functions
``cosine_similarity`` (app/main.py:59-64), ``basic_cleaning`` (:379-380) and
``chunk_text`` (:383-393) and the two normalisation statements of
``OpenSearchIndexer.add_embeddings`` (:315-316) are lifted from the source text
with ``ast`` and executed in a namespace that holds only NumPy and typing names
(the module as a whole is not importable here: ``ModuleNotFoundError: dotenv`` ...,
SURVEY 8c).  Everything else is produced by the oracle restatement, which the
lifted functions pin (tests/test_oracle_golden.py).
"""
from __future__ import annotations

import ast
import hashlib
import json
import os
import sys
from typing import Dict, List, Optional, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_MAIN = "/root/reference/app/main.py"
REF_PMC = "/root/reference/PMC"


def lift_reference():
    src = open(REF_MAIN, encoding="utf-8").read()
    tree = ast.parse(src)
    ns = {"np": np, "List": List, "Tuple": Tuple, "Optional": Optional, "Dict": Dict, "CHUNK_SIZE": 512}
    wanted = {"cosine_similarity", "basic_cleaning", "chunk_text"}
    norm_stmts = None
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in wanted:
            exec(compile(ast.Module([node], []), REF_MAIN, "exec"), ns)
        if isinstance(node, ast.FunctionDef) and node.name == "add_embeddings":
            norm_stmts = [s for s in node.body if isinstance(s, ast.Assign)
                          and getattr(s.targets[0], "id", "") in ("norms", "embeddings")]
    assert norm_stmts and len(norm_stmts) == 2, "normalisation statements not found"

    def ref_normalize(embeddings):
        loc = {"np": np, "embeddings": embeddings}
        exec(compile(ast.Module(norm_stmts, []), REF_MAIN, "exec"), loc)
        return loc["embeddings"]

    return ns["cosine_similarity"], ns["basic_cleaning"], ns["chunk_text"], ref_normalize


def main():
    ref_cos, ref_clean, ref_chunk, ref_norm = lift_reference()
    from oracle import retrieval as R
    from tests import golden_cases as G

    # (i) cosine_similarity pairs, outputs from the lifted reference function
    a, b = G.cosine_cases(20240901)
    with np.errstate(all="ignore"):
        cos_out = np.array([ref_cos(x, y) for x, y in zip(a, b)], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "cosine_pairs.npz"), seed=20240901, expected=cos_out)

    # (ii) normalisation, outputs from the lifted reference statements
    e = G.normalize_case(7)
    normed = ref_norm(e.copy())
    assert normed.dtype == np.float32
    np.savez_compressed(os.path.join(HERE, "normalize_rows.npz"), seed=7, expected=normed)

    # (iii) k-NN known-answer case: N=4096, D=1024, B=16, k=10, with planted ties
    xk, qk = G.knn_case(11)
    cos, ids = R.knn_search(xk, qk, 10)
    np.savez_compressed(os.path.join(HERE, "knn_small.npz"), seed=11, ids=ids, cos=cos)

    # (iv) LFU cache scenario trace (restatement driven, cosine pinned above)
    trace = cache_trace()
    json.dump(trace, open(os.path.join(HERE, "cache_trace.json"), "w"))

    # (v) chunker: per-file chunk counts for the whole corpus + sha256 of chunk texts
    # for a seeded 50-file subset, from the lifted reference functions
    names = sorted(f for f in os.listdir(REF_PMC) if f.startswith("PMC") and f.endswith(".txt"))
    counts = {}
    sub = set(np.random.default_rng(3).choice(len(names), 50, replace=False).tolist())
    hashes = {}
    for i, fname in enumerate(names):
        path = os.path.join(REF_PMC, fname)
        try:
            text = open(path, "r", encoding="utf-8").read()
        except UnicodeDecodeError:
            text = open(path, "r", encoding="latin-1").read()
        chunks = ref_chunk(ref_clean(text), 512)
        counts[fname] = len(chunks)
        if i in sub:
            hashes[fname] = [hashlib.sha256(c.encode("utf-8")).hexdigest() for c in chunks]
    json.dump({"counts": counts, "total": sum(counts.values()), "sha256": hashes},
              open(os.path.join(HERE, "chunker.json"), "w"))
    small = {"inputs": ["a b  c\td\ne", "", "   ", "one", " x\n\ny  z " * 3, "é ü 　w"], "size": 2}
    small["expected"] = [ref_chunk(ref_clean(t), 2) for t in small["inputs"]]
    json.dump(small, open(os.path.join(HERE, "chunker_small.json"), "w"))
    print("corpus chunks:", sum(counts.values()), "files:", len(names))


def cache_trace():
    from oracle.retrieval import LfuCacheOracle
    from tests import golden_cases as G
    base = G.cache_base(5)
    rng = np.random.default_rng(55)
    cache = LfuCacheOracle(max_items=8)
    ops = []
    def put(i):
        cache.put(base[i:i + 1], f"resp{i}")
        ops.append({"op": "put", "vec": i, "responses": cache.responses(), "freqs": cache.freqs()})
    def get(i, noise):
        v = (base[i] + noise * rng.standard_normal(1024).astype(np.float32))[None]
        r = cache.get(v)
        ops.append({"op": "get", "vec": i, "noise": noise, "noise_seed_pos": len(ops),
                    "result": r, "index": cache.last_index, "sim": cache.last_sim,
                    "responses": cache.responses(), "freqs": cache.freqs(), "query": v[0].tolist()})
    get(0, 0.0)                      # empty cache -> None
    for i in range(6):
        put(i)
    get(2, 0.0); get(2, 0.05); get(4, 0.1); get(1, 1.0)     # hits, hits, hit, miss
    put(6); put(7)                   # full (8)
    put(8)                           # evicts first strict-min freq
    get(8, 0.0); get(0, 0.0)
    put(9); put(10); put(2)          # duplicate embedding of 2 inserted at head: newest wins ties
    get(2, 0.0)
    put(11)
    return {"seed": 5, "max_items": 8, "ops": ops}


if __name__ == "__main__":
    main()
