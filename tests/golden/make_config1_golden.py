#!/usr/bin/env python3
"""Config-1 fixture (BASELINE.json configs[0]; main.py:413-456, :500-507): the corpus walk of
`build_embeddings_from_scratch` over a 50-file subset of the bundled PMC corpus, pinned so that the GPU box
-- which has neither the corpus nor the reference -- can replay it.

Runs ONLY in the build container (reads /root/reference).  What it writes is data:
  * per chunk of the 50 files (sorted file order, chunk order): sha256 of the chunk text (from the LIFTED
    reference `basic_cleaning` + `chunk_text`), token count and sha256 of the int32 token ids that
    `tokenizers.BertWordPieceTokenizer` (the library the model's tokenizer is built with) produces under
    the local synthetic vocabulary below, truncated to 512 ids;
  * the vocabulary;
  * 100 sample chunks (chunks 0 and 1 of each file, first 96 words) with their library ids at max_len 128:
    the rows the GPU test indexes;
  * 100 canned queries (words 20..31 of each sample chunk) with their library ids.
"""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF_PMC = "/root/reference/PMC"


def sha_ids(ids) -> str:
    return hashlib.sha256(np.asarray(ids, dtype=np.int32).tobytes()).hexdigest()


def main():
    from tokenizers import BertWordPieceTokenizer
    from make_golden import lift_reference
    from oracle import wordpiece as WP
    _cos, ref_clean, ref_chunk, _norm = lift_reference()
    files = sorted(json.load(open(os.path.join(HERE, "chunker.json")))["sha256"])       # the seeded 50-file subset
    assert len(files) == 50
    per_file = {}
    for fname in files:
        path = os.path.join(REF_PMC, fname)
        try:
            text = open(path, "r", encoding="utf-8").read()
        except UnicodeDecodeError:
            text = open(path, "r", encoding="latin-1").read()                          # main.py:433-438
        per_file[fname] = ref_chunk(ref_clean(text), 512)
    samples, sample_doc = [], []
    for fname in files:
        for c in per_file[fname][:2]:
            samples.append(" ".join(c.split()[:96]))
            sample_doc.append(fname)
    samples, sample_doc = samples[:100], sample_doc[:100]
    queries = [" ".join(s.split()[20:32]) for s in samples]
    vocab = WP.synthetic_vocab(samples + [c for f in files[:10] for c in per_file[f][:4]], size=4000)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "vocab.txt")
        open(p, "w", encoding="utf-8").write("\n".join(vocab) + "\n")
        hf = BertWordPieceTokenizer(p, lowercase=True)
        hf.enable_truncation(max_length=512)
        chunks = {}
        for fname in files:
            rows = []
            for c in per_file[fname]:
                ids = hf.encode(c).ids
                rows.append({"text_sha256": hashlib.sha256(c.encode("utf-8")).hexdigest(), "n_ids": len(ids), "ids_sha256": sha_ids(ids)})
            chunks[fname] = rows
        hf.enable_truncation(max_length=128)
        sample_ids = [hf.encode(s).ids for s in samples]
        query_ids = [hf.encode(q).ids for q in queries]
    out = {"source": "lifted basic_cleaning + chunk_text (main.py:379-393); tokenizers.BertWordPieceTokenizer(lowercase=True)",
           "files": files, "chunks": chunks, "vocab": vocab, "samples": samples, "sample_doc": sample_doc,
           "sample_ids_128": sample_ids, "queries": queries, "query_ids_128": query_ids}
    json.dump(out, open(os.path.join(HERE, "config1.json"), "w"), ensure_ascii=True)
    n_chunks = sum(len(v) for v in chunks.values())
    unk = sum(i == vocab.index("[UNK]") for s in sample_ids for i in s) / sum(len(s) for s in sample_ids)
    print("files", len(files), "chunks", n_chunks, "samples", len(samples), "unk share of samples", round(unk, 4),
          "bytes", os.path.getsize(os.path.join(HERE, "config1.json")))


if __name__ == "__main__":
    main()
