"""One-layer BERT-large-geometry encode of 32 x 512 tokens against the oracle: NaN rows, max |diff|, min cosine.
usage (GPU box, repo root): [SQE_LIB=...libsqe_knobs.so SQE_ENC_GEMM=0|1] python tools/enc_one_layer.py   (the check behind
tests/test_encoder_gpu.py::test_large_batch_persistent_gemms, as a script for bisecting a GEMM kernel)"""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from oracle import bert as OB
from semantic_query_engine_amd import Context
cfg = OB.BertCfg(vocab_size=2000, hidden=1024, layers=1, heads=16, inter=4096, max_pos=512)
w = OB.random_weights(cfg, seed=11)
rng = np.random.default_rng(5)
b, s = 32, 512
ids = rng.integers(5, cfg.vocab_size, (b, s)); lens = rng.integers(300, s + 1, b); lens[0], lens[1] = s, 1
ctx = Context(0)
import tests.test_encoder_gpu as T
enc = T._encoder(ctx, cfg, w)
got = enc.encode_ids(ids, lens)
ref = OB.bert_encode(w, cfg, ids, lens)
print("nan rows", int(np.isnan(got).any(axis=1).sum()), "of", b, "max abs", float(np.nanmax(np.abs(got - ref))))
cs = [float(np.dot(got[i], ref[i]) / (np.linalg.norm(got[i]) * np.linalg.norm(ref[i]) + 1e-30)) for i in range(b)]
print("min cos", min(cs))
print("finite fraction", float(np.isfinite(got).mean()), "first row head", got[0, :4])
