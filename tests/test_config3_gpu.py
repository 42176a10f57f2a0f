"""BASELINE.json config 3: a 64-query batch is encoded ON the GPU (BERT-large geometry, 24 layers, seeded random
bf16 weights) and its embeddings go straight into the k-NN search -- `sqe_encode_device` ->
`sqe_index_search_device` on one stream, no host hop in between (main.py:134-145 -> :347-373).

Checked against the oracle at S in {16, 32, 128} (at S = 128 the forward mixes the persistent and the
ring GEMM inside one layer): every CLS embedding cosine >= 0.999 vs oracle/bert.py (bf16 activations,
fp32 accumulate; weights bit-identical on both sides), and the ids / cosines of the search equal to the
oracle's exact top-10 computed from the SAME encoded vectors (ids bit-exact where float64 scores are
separated, cosines within 1e-3 -- measured ~1e-6).  At N = 10M, where the oracle cannot finish, the
hand-off is checked through planted rows.  GPU only."""
import numpy as np
import pytest
import torch

from oracle import bert as OB
from oracle import retrieval as R
from tests.gpu_util import assert_topk_matches, exact_topk_fast

pytestmark = pytest.mark.gpu

B, D, K = 64, 1024, 10


@pytest.fixture(scope="module")
def setup():
    from semantic_query_engine_amd import Context
    from semantic_query_engine_amd.encoder import BertEncoder
    ctx = Context(0)
    cfg = OB.BertCfg()
    w = OB.random_weights(cfg, seed=0)
    enc = BertEncoder(ctx)
    enc.load_weights({k: v.numpy() for k, v in w.items()})
    return ctx, cfg, w, enc


def _batch(cfg, s, seed):
    rng = np.random.default_rng(seed)
    ids = rng.integers(1000, cfg.vocab_size, (B, s)).astype(np.int32)     # ids in [1000, 30522)
    lens = rng.integers(max(1, s // 4), s + 1, B).astype(np.int32)         # ragged, as real queries are
    lens[0], lens[1], lens[2] = s, 1, s - 1
    return ids, lens


def _encode_then_search(ctx, enc, idx, ids, lens, dev):
    """One stream, device pointers only: ids -> embeddings -> top-k.  Returns host copies."""
    ids_d = torch.from_numpy(ids).to(dev)
    lens_d = torch.from_numpy(lens).to(dev)
    emb = torch.empty((B, D), dtype=torch.float32, device=dev)
    cos = torch.empty((B, K), dtype=torch.float32, device=dev)
    nbr = torch.empty((B, K), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()                              # torch's stream produced the inputs
    enc.encode_ids_device(ids_d.data_ptr(), lens_d.data_ptr(), B, ids.shape[1], emb.data_ptr())
    idx.search_device(emb.data_ptr(), B, K, cos.data_ptr(), nbr.data_ptr())   # no synchronisation in between
    ctx.synchronize()
    return emb.cpu().numpy(), cos.cpu().numpy(), nbr.cpu().numpy()


@pytest.mark.parametrize("s", [16, 32, 128])
def test_encode_device_into_search_device_matches_oracle(setup, s):
    from semantic_query_engine_amd import VectorIndex
    ctx, cfg, w, enc = setup
    dev = torch.device("cuda", 0)
    ids, lens = _batch(cfg, s, seed=300 + s)
    ref_emb = OB.bert_encode(w, cfg, ids, lens)                            # fp32 oracle, [64, 1024]

    # index the oracle can finish: 200k random rows + rows planted near the oracle's embeddings, so that
    # the true neighbours are non-trivial for half the batch
    rng = np.random.default_rng(s)
    n = 200_000
    x = rng.standard_normal((n, D), dtype=np.float32)
    scale = float(np.linalg.norm(ref_emb, axis=1).mean() / np.sqrt(D))
    plant = (np.arange(B // 2) * (n // (B // 2)) + 5).astype(np.int64)
    x[plant] = ref_emb[: B // 2] + 0.1 * scale * rng.standard_normal((B // 2, D), dtype=np.float32)
    idx = VectorIndex(ctx, D)
    idx.add(x)

    emb, cos, nbr = _encode_then_search(ctx, enc, idx, ids, lens, dev)
    assert not np.isnan(emb).any()
    cs = [float(np.dot(emb[i], ref_emb[i]) / (np.linalg.norm(emb[i]) * np.linalg.norm(ref_emb[i]))) for i in range(B)]
    assert min(cs) >= 0.999, (s, min(cs), int(np.argmin(cs)), int(lens[int(np.argmin(cs))]))

    # the search must be the oracle's answer for the vectors it was handed
    ref_cos, ref_ids = exact_topk_fast(x, emb, K, extra=64)
    assert_topk_matches(cos, nbr, ref_cos, ref_ids, R.normalize_rows(x), R.normalize_rows(emb))
    assert np.array_equal(nbr[: B // 2, 0], plant)                         # planted rows first
    # and the host entry points give the same bits as the device hand-off
    emb_h = enc.encode_ids(ids, lens)
    assert np.array_equal(emb_h, emb)
    cos_h, ids_h = idx.search(emb_h, K)
    assert np.array_equal(ids_h, nbr) and np.array_equal(cos_h, cos)
    idx.close()


def test_config3_full_size_planted_rows(setup):
    """N = 10M x 1024 (config 3's index): rows planted at the GPU-encoded embeddings come back first,
    with the float64 cosine of the planted vector."""
    from semantic_query_engine_amd import VectorIndex
    ctx, cfg, w, enc = setup
    dev = torch.device("cuda", 0)
    n, block = 10_000_000, 1 << 20
    idx = VectorIndex(ctx, D)
    idx.reserve(n)
    for b in range((n + block - 1) // block):
        rows = min(block, n - b * block)
        g = torch.Generator(device=dev).manual_seed(9000 + b)
        xb = torch.randn((rows, D), generator=g, device=dev)
        torch.cuda.synchronize()
        idx.add_device(xb.data_ptr(), rows)
        ctx.synchronize()
        del xb
    s = 32
    ids, lens = _batch(cfg, s, seed=77)
    emb = enc.encode_ids(ids, lens)
    rng = np.random.default_rng(5)
    plant = (np.arange(B) * (n // B) + 123).astype(np.int64)
    scale = float(np.linalg.norm(emb, axis=1).mean() / np.sqrt(D))
    planted = (emb + 0.2 * scale * rng.standard_normal((B, D))).astype(np.float32)
    idx.update(plant, planted)
    emb2, cos, nbr = _encode_then_search(ctx, enc, idx, ids, lens, dev)
    assert np.array_equal(emb2, emb)
    assert np.array_equal(nbr[:, 0], plant)
    pn, qn = R.normalize_rows(planted).astype(np.float64), R.normalize_rows(emb).astype(np.float64)
    want = (pn * qn).sum(1)
    assert np.abs(cos[:, 0] - want).max() < 1e-3                           # north_star tolerance; measured ~1e-6
    assert np.all(cos[:, 1:] <= cos[:, :-1]) and np.all((nbr >= 0) & (nbr < n))
    assert all(len(set(r.tolist())) == K for r in nbr)
    idx.close()
