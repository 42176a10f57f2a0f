"""Config 1 plumbing on CPU: corpus walk (sorted file names, PMC*.txt filter, utf-8 -> latin-1 fallback,
main.py:427-443) -> chunker -> the C++ WordPiece tokenizer of libsqe (host-only entry points, no GPU) must
reproduce the committed fixture tests/golden/config1.json chunk by chunk: text sha256 (pinned by the LIFTED
reference chunker), token count and sha256 of the ids (pinned by the `tokenizers` library).  The corpus walk
runs only where /root/reference is present (the build container); the sample-text part runs everywhere."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import retrieval as R
from oracle import wordpiece as WP

REF_PMC = "/root/reference/PMC"


@pytest.fixture(scope="module")
def fx(golden_dir):
    return json.load(open(os.path.join(golden_dir, "config1.json")))


@pytest.fixture(scope="module")
def tok(fx):
    from semantic_query_engine_amd.tokenizer import WordPieceTokenizer
    return WordPieceTokenizer(vocab_text="\n".join(fx["vocab"]) + "\n")


def _sha(ids):
    return hashlib.sha256(np.asarray(ids, dtype=np.int32).tobytes()).hexdigest()


def test_sample_chunks_and_queries_tokenize_to_the_library_ids(fx, tok):
    v = {t: i for i, t in enumerate(fx["vocab"])}
    for texts, want in ((fx["samples"], fx["sample_ids_128"]), (fx["queries"], fx["query_ids_128"])):
        ids, lens = tok.encode_batch(texts, 128)
        for i, w in enumerate(want):
            assert ids[i, :lens[i]].tolist() == w, i
            assert WP.encode(texts[i], v, 128) == w, i             # the oracle restatement agrees as well
            assert not ids[i, lens[i]:].any()                       # [PAD] = 0 filled


@pytest.mark.skipif(not os.path.isdir(REF_PMC), reason="corpus not present (GPU box): covered by the sample-text test")
def test_corpus_walk_chunks_and_token_ids_match_fixture(fx, tok):
    docs = R.corpus_docs(REF_PMC, files=fx["files"])                # sorted subset, the reference's walk
    by_file = {}
    for d in docs:
        by_file.setdefault(d["doc_id"], []).append(d["text"])
    assert list(by_file) == fx["files"]
    texts = [t for f in fx["files"] for t in by_file[f]]
    want = [row for f in fx["files"] for row in fx["chunks"][f]]
    assert len(texts) == len(want) == 514
    ids, lens = tok.encode_batch(texts, 512)                         # multi-threaded C++ path
    for i, (t, w) in enumerate(zip(texts, want)):
        assert hashlib.sha256(t.encode("utf-8")).hexdigest() == w["text_sha256"], i
        assert int(lens[i]) == w["n_ids"], (i, int(lens[i]), w["n_ids"])
        assert _sha(ids[i, :lens[i]]) == w["ids_sha256"], i
    # ~91 % of corpus chunks are full 512-word chunks and truncate to the 512-id limit
    assert np.mean(lens == 512) > 0.8
    # row id = position in all_docs; _id rule of main.py:325
    assert [f"{d['doc_id']}_{i}" for i, d in enumerate(docs)][:2] == [f"{fx['files'][0]}_0", f"{fx['files'][0]}_1"]
