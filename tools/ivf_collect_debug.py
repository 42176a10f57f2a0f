"""Hand-run check: the crowd-of-copies case of tests/test_ivf_gpu.py::test_ivf_collect_mode_and_its_fallback, with the mismatches printed.
usage (GPU box): [SQE_LIB=...libsqe_knobs.so SQE_IVF_STRIPS=1] python tools/ivf_collect_debug.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import retrieval as R
from semantic_query_engine_amd import Context, INDEX_IVF_FLAT, VectorIndex

ctx = Context(0)
n, d, k, nlist, nprobe, b = 120_000, 256, 10, 32, 8, 200
rng = np.random.default_rng(11)
cen = rng.standard_normal((100, d)).astype(np.float32)
lab = rng.integers(0, 100, n)
x = (cen[lab] + 0.3 * rng.standard_normal((n, d)).astype(np.float32)).astype(np.float32)
rng = np.random.default_rng(12)
q = (x[rng.integers(0, n, b)] + 0.2 * rng.standard_normal((b, d))).astype(np.float32)
idx = VectorIndex(ctx, d, INDEX_IVF_FLAT, nlist)
idx.train(x[:40000], iters=6, seed=13)
idx.add(x)
v = rng.standard_normal(d).astype(np.float32)
rows = rng.permutation(n)[:5000]
x[rows] = v
idx.update(rows, x[rows])
q[:10] = v + 0.01 * rng.standard_normal((10, d)).astype(np.float32)
centroids, assign = idx.ivf_export(nlist)
xn, qn = R.normalize_rows(x), R.normalize_rows(q)
cos, ids = idx.search(q, k, nprobe=nprobe)
ref_cos, ref_ids = R.ivf_search(xn, qn, centroids, assign, k, nprobe)
bad = np.nonzero(np.abs(cos - ref_cos).max(1) > 1e-3)[0]
print("queries with |dcos| > 1e-3:", bad.tolist())
for i in bad[:6]:
    print(i, "got ", np.round(cos[i], 5).tolist(), ids[i].tolist())
    print(i, "want", np.round(ref_cos[i], 5).tolist(), ref_ids[i].tolist())
    probes = np.argsort(-(qn[i] @ centroids.T), kind="stable")[:nprobe]
    print("   probed lists", probes.tolist(), "lengths", [int((assign == p).sum()) for p in probes], "copies in them", [int(np.isin(np.nonzero(assign == p)[0], rows).sum()) for p in probes])
print("list lengths:", np.bincount(assign, minlength=nlist).tolist())
