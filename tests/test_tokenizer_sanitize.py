"""csrc/tokenizer.cpp under AddressSanitizer + UndefinedBehaviorSanitizer (r02 verdict item 7).

The tokenizer is host-only C++ that takes untrusted bytes -- the reference passes raw chunk text as the prompt
(/root/reference/app/main.py:139), 89 of the bundled files are not UTF-8 (main.py:433-438 falls back to latin-1) --
and walks 19 k lines of generated Unicode tables.  `make -C semantic_query_engine_amd/csrc tokenizer_asan` builds it
with g++ -fsanitize=address,undefined beside a small driver (tests/native/); this test feeds the driver malformed
UTF-8, lone surrogates, overlong forms, a 1 MB single "word", latin-1 byte runs, NULs and random bytes, requires a
clean exit (any sanitizer report aborts with a non-zero code) and compares every id sequence with what the shipped
libsqe.so returns for the same bytes.  No GPU."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np

from oracle import wordpiece as WP
from tests.test_oracle_wordpiece import SAMPLES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "semantic_query_engine_amd", "csrc")


def _cases():
    rng = np.random.default_rng(11)
    cases = []

    def add(b, max_len=48):
        cases.append((max_len, bytes(b)))

    for s in SAMPLES[:12]:
        add(s.encode("utf-8"), 64)
    add(b"")
    add(b"\x00\x00abc\x00def\x00")
    add(b"\xff\xfe\xfd plain \x80\x81\xbf tail")                       # bytes that never start a sequence
    add(b"\xc3")                                                         # truncated 2-byte form at the very end
    add(b"abc \xe2\x82")                                                 # truncated 3-byte form at the end
    add(b"abc \xf0\x9f\x98")                                             # truncated 4-byte form at the end
    add(b"\xc0\xaf \xe0\x80\xaf \xf0\x80\x80\xaf")                       # overlong encodings
    add(b"\xed\xa0\x80 lone high \xed\xbf\xbf lone low \xed\xa0\xbd\xed\xb8\x80 pair")   # CESU surrogates
    add(b"\xf4\x90\x80\x80 above U+10FFFF \xf7\xbf\xbf\xbf \xf8\x88\x80\x80\x80")
    add("é ́́ café Å ﬁ İstanbul Σς \U0001f600 \U000e0001 ".encode("utf-8"))
    add("中文字 㐀\U00020000\U0002f800 mixed中word".encode("utf-8"))
    add("    ​‍﻿­ tabs\tand\r\nlines\x0b\x0c\x1f\x7f".encode("utf-8"))
    add(bytes(range(0xA0, 0x100)) + b" r\xe9sum\xe9 na\xefve \xb5g/ml 37\xb0C \xa9 \xbd \xd7 \xfc\xdf")   # latin-1 text as raw bytes
    add(("a" * 1_000_000).encode(), 32)                                  # a 1 MB single "word" (> 100 chars -> [UNK])
    add(("é" * 300_000).encode("utf-8"), 16)
    add(b"x" * 100 + b" " + b"y" * 101 + b" " + "ź".encode() * 100)   # at and past the 100-char word limit
    add((".,;:!?()[]{}<>" * 500).encode(), 512)                          # punctuation only: one token each
    add(b"background methods results " * 400, 512)                       # truncation in the middle of the text
    add(b"##ing ##s [CLS] [SEP] [UNK] [PAD]")
    add(b"word", 2)                                                      # max_len 2: [CLS] [SEP] only
    add(b"word word", 3)
    for n in (1, 2, 3, 7, 64, 257, 4096):
        add(rng.integers(0, 256, n, dtype=np.uint8).tobytes(), 40)
        add(rng.integers(0x80, 0x100, n, dtype=np.uint8).tobytes(), 40)  # continuation / lead bytes only
    # valid text with single bytes flipped
    base = bytearray(" ".join(SAMPLES[:6]).encode("utf-8"))
    for _ in range(24):
        b = bytearray(base)
        for p in rng.integers(0, len(b), 5):
            b[p] = int(rng.integers(0, 256))
        add(b, 96)
    return cases


def test_tokenizer_under_asan_ubsan_matches_the_library(tmp_path):
    subprocess.check_call(["make", "-C", CSRC, "tokenizer_asan"], stdout=subprocess.DEVNULL)
    exe = os.path.join(CSRC, "build_san", "tokenizer_asan")
    words = SAMPLES + ["background methods results conclusions patients treatment study data analysis " * 3]
    toks = WP.synthetic_vocab(words, size=1500)
    # vocab file with the oddities a hand-edited vocab.txt has: CRLF line ends, an empty line in the middle (it
    # takes an id), a line with invalid UTF-8, a duplicate, no newline at the end
    lines = [t.encode("utf-8") for t in toks]
    lines[700:700] = [b"", b"\xff\xfebroken", lines[10]]
    vocab = b"\r\n".join(lines)
    (tmp_path / "vocab.txt").write_bytes(vocab)
    cases = _cases()
    with open(tmp_path / "cases.bin", "wb") as f:
        for max_len, b in cases:
            f.write(struct.pack("<iq", max_len, len(b)))
            f.write(b)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([exe, str(tmp_path / "vocab.txt"), str(tmp_path / "cases.bin"), str(tmp_path / "ids.bin")],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-6000:]
    assert r.stdout.strip() == f"ok {len(cases)} cases"

    # the same bytes through the shipped library (ctypes, raw bytes: no Python-side decoding in between)
    from semantic_query_engine_amd import _native as N
    lib = N.load()
    h = C.c_void_p()
    N.check(lib.sqe_tokenizer_create(vocab, len(vocab), C.byref(h)))
    got = np.fromfile(tmp_path / "ids.bin", dtype=np.int32)
    pos = 0
    try:
        for max_len, b in cases:
            ids = (C.c_int32 * max_len)()
            n = C.c_int32()
            N.check(lib.sqe_tokenize(h, b, len(b), max_len, ids, C.byref(n)))
            want = [int(got[pos])] + got[pos + 1: pos + 1 + int(got[pos])].tolist()
            pos += 1 + int(got[pos])
            assert [n.value] + list(ids[: n.value]) == want
            assert ids[0] == toks.index("[CLS]") and ids[n.value - 1] == toks.index("[SEP]") and 2 <= n.value <= max_len
    finally:
        lib.sqe_tokenizer_destroy(h)
    assert pos == got.size
