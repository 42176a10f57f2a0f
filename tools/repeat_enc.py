"""Hand-run check (not a test): encode the same 64 x S ragged batch many times and compare the outputs bit for bit.
usage (GPU box): python tools/repeat_enc.py [S] [repeats]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import bert as OB
from semantic_query_engine_amd import Context
from semantic_query_engine_amd.encoder import BertEncoder

S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = 64
ctx = Context(0)
cfg = OB.BertCfg()
w = OB.random_weights(cfg, seed=0)
enc = BertEncoder(ctx)
enc.load_weights({k: v.numpy() for k, v in w.items()})
rng = np.random.default_rng(300 + S)
ids = rng.integers(1000, cfg.vocab_size, (B, S)).astype(np.int32)
lens = rng.integers(max(1, S // 4), S + 1, B).astype(np.int32)
lens[0], lens[1], lens[2] = S, 1, S - 1
ref = enc.encode_ids(ids, lens)
bad = 0
for r in range(REP):
    e = enc.encode_ids(ids, lens)
    if not np.array_equal(e, ref):
        bad += 1
        d = np.abs(e - ref)
        rows = np.nonzero(d.max(1) > 0)[0]
        print(f"rep {r}: differs in {len(rows)} rows {rows[:8].tolist()} max|d| {d.max():.3e} nan {int(np.isnan(e).sum())} "
              f"cos_min {min(float(np.dot(e[i], ref[i]) / (np.linalg.norm(e[i]) * np.linalg.norm(ref[i]) + 1e-30)) for i in rows):.6f}", flush=True)
print(f"S={S}: {bad} of {REP} repeats differ from the first", flush=True)
