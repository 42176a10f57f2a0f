#!/usr/bin/env python3
"""Golden fixture (vii) of SURVEY 8c: WordPiece id sequences from `tokenizers.BertWordPieceTokenizer`
(the library the model's own tokenizer is built with) under a local synthetic vocabulary, on 20
sentences: the edge-case samples of tests/test_oracle_wordpiece.py plus the opening words of nine
files of the bundled PMC corpus.  Runs only in the build container (reads /root/reference/PMC);
what it writes is data: vocabulary, input sentences, expected ids."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_PMC = "/root/reference/PMC"


def main():
    import tempfile
    from tokenizers import BertWordPieceTokenizer
    from oracle import wordpiece as WP
    from tests.test_oracle_wordpiece import SAMPLES
    names = sorted(f for f in os.listdir(REF_PMC) if f.startswith("PMC") and f.endswith(".txt"))
    corpus = []
    for fname in names[:: max(1, len(names) // 9)][:9]:
        try:
            text = open(os.path.join(REF_PMC, fname), encoding="utf-8").read()
        except UnicodeDecodeError:
            text = open(os.path.join(REF_PMC, fname), encoding="latin-1").read()
        corpus.append(" ".join(text.replace("\n", " ").split()[:40]))
    sentences = list(SAMPLES) + corpus
    assert len(sentences) == 20
    vocab = WP.synthetic_vocab(sentences, size=1500)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "vocab.txt")
        open(p, "w", encoding="utf-8").write("\n".join(vocab) + "\n")
        hf = BertWordPieceTokenizer(p, lowercase=True)
        full = [hf.encode(s).ids for s in sentences]
        hf.enable_truncation(max_length=16)
        trunc = [hf.encode(s).ids for s in sentences]
    json.dump({"source": "tokenizers.BertWordPieceTokenizer(lowercase=True)", "vocab": vocab, "sentences": sentences,
               "ids_512": full, "ids_16": trunc}, open(os.path.join(HERE, "wordpiece_ids.json"), "w"), ensure_ascii=True)
    print("sentences:", len(sentences), "unk share:",
          sum(i == 100 for s in full for i in s) / max(1, sum(len(s) for s in full)))


if __name__ == "__main__":
    main()
