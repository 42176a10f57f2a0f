"""The C++ WordPiece tokenizer (host-only entry points of libsqe) against the oracle restatement
and, through it, tokenizers.BertWordPieceTokenizer.  No GPU needed."""
import os

import numpy as np
import pytest

from oracle import wordpiece as WP
from tests.test_oracle_wordpiece import SAMPLES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tok_and_vocab():
    from semantic_query_engine_amd.tokenizer import WordPieceTokenizer
    words = SAMPLES + ["background methods results conclusions patients treatment study data analysis " * 3]
    toks = WP.synthetic_vocab(words, size=1500)
    return WordPieceTokenizer(vocab_text="\n".join(toks) + "\n"), {t: i for i, t in enumerate(toks)}


def test_matches_oracle_on_samples(tok_and_vocab):
    tok, vocab = tok_and_vocab
    for s in SAMPLES:
        assert tok.encode(s, 512) == WP.encode(s, vocab, 512), s
        assert tok.encode(s, 16) == WP.encode(s, vocab, 16), s
    assert tok.encode("", 512) == [101, 102]


def test_matches_tokenizers_library(tok_and_vocab, tmp_path):
    from tokenizers import BertWordPieceTokenizer
    tok, vocab = tok_and_vocab
    p = tmp_path / "vocab.txt"
    p.write_text("\n".join(sorted(vocab, key=vocab.get)) + "\n", encoding="utf-8")
    hf = BertWordPieceTokenizer(str(p), lowercase=True)
    for s in SAMPLES:
        assert tok.encode(s, 512) == hf.encode(s).ids, s


@pytest.mark.skipif(not os.path.isdir("/root/reference/PMC"), reason="corpus only in the build container")
def test_corpus_chunks_and_batch(tok_and_vocab):
    from oracle import retrieval as R
    tok, vocab = tok_and_vocab
    names = sorted(os.listdir("/root/reference/PMC"))[:6]
    docs = R.corpus_docs("/root/reference/PMC", files=names)[:60]
    texts = [d["text"] for d in docs]
    ids, lens = tok.encode_batch(texts + [""] * 30, 512)          # >= 64 texts: threaded path
    assert ids.shape == (len(texts) + 30, 512) and len(texts) + 30 >= 64
    for i, t in enumerate(texts):
        ref = WP.encode(t, vocab, 512)
        assert lens[i] == len(ref) and ids[i, :lens[i]].tolist() == ref
        assert np.all(ids[i, lens[i]:] == 0)
    assert lens[-1] == 2 and ids[-1, :2].tolist() == [101, 102]


def test_invalid_utf8_and_errors(tok_and_vocab):
    import ctypes as C
    from semantic_query_engine_amd import _native as N
    tok, vocab = tok_and_vocab
    lib = N.load()
    raw = b"ok \xff\xfe broken \xe2\x82 tail"
    ids = (C.c_int32 * 32)()
    n = C.c_int32()
    assert lib.sqe_tokenize(tok.handle, raw, len(raw), 32, ids, C.byref(n)) == 0
    assert list(ids[: n.value]) == WP.encode("ok  broken  tail", vocab, 32)
    assert lib.sqe_tokenize(tok.handle, raw, len(raw), 1, ids, C.byref(n)) != 0       # max_len < 2
    h = C.c_void_p()
    assert lib.sqe_tokenizer_create(b"a\nb\n", 4, C.byref(h)) != 0                   # no specials


def test_golden_ids_fixture():
    """The C++ tokenizer against the committed `tokenizers`-library id sequences (SURVEY 8c vii)."""
    import json
    from semantic_query_engine_amd.tokenizer import WordPieceTokenizer
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "wordpiece_ids.json"), encoding="utf-8"))
    tok = WordPieceTokenizer(vocab_text="\n".join(g["vocab"]) + "\n")
    for s, full, trunc in zip(g["sentences"], g["ids_512"], g["ids_16"]):
        assert tok.encode(s, 512) == full, s
        assert tok.encode(s, 16) == trunc, s
    ids, lens = tok.encode_batch(g["sentences"] * 4, 512)            # 80 texts: the threaded batch path
    for i, full in enumerate(g["ids_512"] * 4):
        assert lens[i] == len(full) and ids[i, :lens[i]].tolist() == full
