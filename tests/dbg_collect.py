import sys, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import retrieval as R
from semantic_query_engine_amd import Context, VectorIndex
ctx = Context(0)
rng = np.random.default_rng(5)
n, d, k = 60000, 1024, 10
x = rng.standard_normal((n, d)).astype(np.float32)
idx = VectorIndex(ctx, d); idx.add(x)
xn = R.normalize_rows(x)
idx.set_option("rescore_k", k)
for b in (8, 40, 64, 100, 128, 200, 300):
    q = rng.standard_normal((b, d)).astype(np.float32)
    qn = R.normalize_rows(q)
    s = qn.astype(np.float64) @ xn.astype(np.float64).T
    tot = 0
    for rep in range(5):
        cos, ids = idx.search(q, k)
        tot += sum(1 for i in range(b) if set(np.argsort(-s[i], kind="stable")[:k].tolist()) != set(ids[i].tolist()))
    print("B", b, "bad over 5 reps", tot, "unc", ctx.stats()["uncertified"], flush=True)
