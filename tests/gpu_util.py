"""Helpers shared by the GPU parity tests."""
import numpy as np

from oracle import retrieval as R


def exact_topk_fast(x_raw: np.ndarray, q_raw: np.ndarray, k: int, extra: int = 16, block: int = 131072):
    """Oracle top-k for large N: fp32 BLAS shortlist of k+extra per query, then the float64
    scoring and (score desc, id asc) order of oracle.retrieval.exact_topk on the shortlist."""
    xn, qn = R.normalize_rows(x_raw), R.normalize_rows(q_raw)
    b, n = qn.shape[0], xn.shape[0]
    kk = min(n, k + extra)
    short_s = np.full((b, 0), -np.inf, np.float32)
    short_i = np.zeros((b, 0), np.int64)
    for lo in range(0, n, block):
        s = qn @ xn[lo:lo + block].T
        ids = np.broadcast_to(np.arange(lo, lo + s.shape[1], dtype=np.int64), s.shape)
        s = np.concatenate([short_s, s], 1)
        ids = np.concatenate([short_i, ids], 1)
        part = np.argpartition(-s, min(kk, s.shape[1] - 1), axis=1)[:, :kk]
        short_s, short_i = np.take_along_axis(s, part, 1), np.take_along_axis(ids, part, 1)
    cos = np.full((b, k), -np.inf)
    idx = np.full((b, k), -1, np.int64)
    for i in range(b):
        cand = np.sort(short_i[i])
        s64 = xn[cand].astype(np.float64) @ qn[i].astype(np.float64)
        order = np.argsort(-s64, kind="stable")[:k]
        cos[i, :order.size] = s64[order]
        idx[i, :order.size] = cand[order]
    return cos, idx


def assert_topk_matches(cos, ids, ref_cos, ref_ids, xn=None, qn=None, tol=2e-6, score_tol=1e-3):
    """Bit-exact ids wherever the oracle's scores are separated by more than `tol`; where two
    oracle scores are closer than fp32 rounding can resolve, either order is accepted provided
    the returned id's true (float64) score equals the expected score within `tol`."""
    assert cos.shape == ref_cos.shape and ids.shape == ref_ids.shape
    valid = ref_ids >= 0
    assert np.array_equal(ids >= 0, valid)
    assert np.all(np.abs(cos[valid] - ref_cos[valid]) < score_tol)          # north_star: 1e-3
    for b in range(cos.shape[0]):                                            # best first
        v = cos[b][valid[b]]
        assert np.all(v[1:] <= v[:-1])
    assert np.all(np.isneginf(cos[~valid]))
    bad = (ids != ref_ids) & valid
    for b, j in zip(*np.nonzero(bad)):
        assert xn is not None, f"ids differ at {(b, j)}: {ids[b]} vs {ref_ids[b]}"
        true = float(xn[ids[b, j]].astype(np.float64) @ qn[b].astype(np.float64))
        assert abs(true - ref_cos[b, j]) <= tol, (b, j, ids[b], ref_ids[b], true, ref_cos[b, j])
        assert len(set(ids[b][ids[b] >= 0].tolist())) == int((ids[b] >= 0).sum())
