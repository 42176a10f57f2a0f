"""Cross-checks the WordPiece restatement against tokenizers.BertWordPieceTokenizer on a
synthetic local vocab (no fetch).  CPU only."""
import os

import pytest

from oracle import wordpiece as WP

SAMPLES = [
    "The quick brown fox jumps over the lazy dog.",
    "Hypertension (HTN) affects ~30% of adults; β-blockers & ACE-inhibitors are first-line.",
    "==== Front J Med Internet Res 10.2196/12345 ==== Body Background: mHealth apps...",
    "Café naïve façade ÅNGSTRÖM résumé — “quoted” text… and  double  spaces\ttabs\nnewlines",
    "中文字符 mixed with English and 日本語テキスト",
    "p<0.05, n=1,234; IL-6/TNF-α↑ (95% CI: 1.2–3.4)",
    "supercalifragilisticexpialidocious " + "x" * 120 + " end",
    "",
    "   ",
    "İstanbul ǅ ß ﬁ K",
    "zero​width and control\x07chars � replaced",
]


@pytest.fixture(scope="module")
def vocab(tmp_path_factory):
    words = SAMPLES + ["background methods results conclusions patients treatment study data analysis " * 3]
    toks = WP.synthetic_vocab(words, size=1500)
    p = tmp_path_factory.mktemp("vocab") / "vocab.txt"
    p.write_text("\n".join(toks) + "\n", encoding="utf-8")
    return toks, str(p)


def test_matches_tokenizers_library(vocab):
    from tokenizers import BertWordPieceTokenizer
    toks, path = vocab
    v = {t: i for i, t in enumerate(toks)}
    assert v["[UNK]"] == 100 and v["[CLS]"] == 101 and v["[SEP]"] == 102
    hf = BertWordPieceTokenizer(path, lowercase=True)
    for s in SAMPLES:
        assert WP.encode(s, v, 512) == hf.encode(s).ids, s
    hf.enable_truncation(max_length=16)
    for s in SAMPLES:
        assert WP.encode(s, v, 16) == hf.encode(s).ids, s


def test_specials_and_truncation(vocab):
    toks, _ = vocab
    v = {t: i for i, t in enumerate(toks)}
    assert WP.encode("", v) == [101, 102]
    long = "study " * 2000
    ids = WP.encode(long, v, 512)
    assert len(ids) == 512 and ids[0] == 101 and ids[-1] == 102


def test_golden_ids_fixture():
    """the committed id sequences (tests/golden/make_wordpiece_golden.py)."""
    import json
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wordpiece_ids.json"),
                       encoding="utf-8"))
    v = {t: i for i, t in enumerate(g["vocab"])}
    assert len(g["sentences"]) == 20
    for s, full, trunc in zip(g["sentences"], g["ids_512"], g["ids_16"]):
        assert WP.encode(s, v, 512) == full, s
        assert WP.encode(s, v, 16) == trunc, s
