// Sanitizer driver for csrc/tokenizer.cpp (test infrastructure; SURVEY section 5: host C++ under
// -fsanitize=address,undefined).  Built by `make -C semantic_query_engine_amd/csrc tokenizer_asan` with g++ -- no HIP
// code is involved: the tokenizer is host-only and takes untrusted UTF-8 (the reference passes raw chunk text,
// /root/reference/app/main.py:139).
//
//   tokenizer_asan <vocab file> <cases file> <out file>
// cases file: repeated { int32 max_len; int64 n_bytes; bytes }.  Every case is tokenised alone and, in groups of 70, through
// sqe_tokenize_batch (the threaded path); the driver checks the ABI's invariants and writes the ids, which the
// Python test compares with what the shipped libsqe.so returns for the same bytes.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "sqe.h"

namespace sqe {
static std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
}  // namespace sqe

static std::vector<char> slurp(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    std::vector<char> b;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + n);
    fclose(f);
    return b;
}

#define REQUIRE(c)                                                        \
    do {                                                                  \
        if (!(c)) { fprintf(stderr, "REQUIRE failed: %s (line %d)\n", #c, __LINE__); exit(3); } \
    } while (0)

int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: %s vocab cases out\n", argv[0]); return 2; }
    const std::vector<char> vocab = slurp(argv[1]);
    const std::vector<char> cases = slurp(argv[2]);

    // argument checking of the ABI itself
    sqe_tokenizer* bad = nullptr;
    REQUIRE(sqe_tokenizer_create(nullptr, 0, &bad) == SQE_ERR_INVALID && bad == nullptr);
    REQUIRE(sqe_tokenizer_create("a\nb\n", 4, &bad) == SQE_ERR_INVALID && bad == nullptr);       // no [UNK]/[CLS]/[SEP]
    REQUIRE(sqe_tokenizer_create("\n\n\n", 3, &bad) == SQE_ERR_INVALID && bad == nullptr);        // empty lines only

    sqe_tokenizer* tok = nullptr;
    REQUIRE(sqe_tokenizer_create(vocab.data(), (int64_t)vocab.size(), &tok) == SQE_OK && tok);
    int32_t one[4];
    int len1 = 0;
    REQUIRE(sqe_tokenize(tok, "x", 1, 1, one, &len1) == SQE_ERR_INVALID);                         // max_len < 2
    REQUIRE(sqe_tokenize(tok, nullptr, 5, 4, one, &len1) == SQE_ERR_INVALID);
    REQUIRE(sqe_tokenize(tok, nullptr, 0, 4, one, &len1) == SQE_OK && len1 == 2);                 // empty text: [CLS] [SEP]

    FILE* out = fopen(argv[3], "wb");
    REQUIRE(out);
    std::vector<const char*> texts;
    std::vector<int64_t> sizes;
    std::vector<int32_t> maxlens;
    size_t pos = 0;
    while (pos + 12 <= cases.size()) {
        int32_t max_len;
        int64_t n;
        memcpy(&max_len, cases.data() + pos, 4);
        memcpy(&n, cases.data() + pos + 4, 8);
        pos += 12;
        REQUIRE(n >= 0 && pos + (size_t)n <= cases.size());
        // an exact-size heap copy: a read one byte past the text is a heap-buffer-overflow ASan reports
        char* copy = (char*)malloc(n > 0 ? (size_t)n : 1);
        memcpy(copy, cases.data() + pos, (size_t)n);
        pos += (size_t)n;
        std::vector<int32_t> ids((size_t)max_len);
        int len = -1;
        REQUIRE(sqe_tokenize(tok, copy, n, max_len, ids.data(), &len) == SQE_OK);
        REQUIRE(len >= 2 && len <= max_len);
        fwrite(&len, 4, 1, out);
        fwrite(ids.data(), 4, (size_t)len, out);
        texts.push_back(copy);
        sizes.push_back(n);
        maxlens.push_back(max_len);
    }
    // the batch entry point (threads from 64 texts on): same ids as the single calls, rows padded with 0
    const int ML = 48;
    for (size_t b = 0; b < texts.size(); b += 70) {
        const int n = (int)std::min<size_t>(70, texts.size() - b);
        std::vector<int32_t> ids((size_t)n * ML, -7), lens((size_t)n, -7);
        REQUIRE(sqe_tokenize_batch(tok, texts.data() + b, sizes.data() + b, n, ML, ids.data(), lens.data()) == SQE_OK);
        for (int i = 0; i < n; ++i) {
            std::vector<int32_t> ref(ML);
            int len = 0;
            REQUIRE(sqe_tokenize(tok, texts[b + i], sizes[b + i], ML, ref.data(), &len) == SQE_OK);
            REQUIRE(lens[i] == len);
            for (int j = 0; j < ML; ++j) REQUIRE(ids[(size_t)i * ML + j] == (j < len ? ref[j] : 0));
        }
    }
    fclose(out);
    for (const char* t : texts) free((void*)t);
    sqe_tokenizer_destroy(tok);
    printf("ok %zu cases\n", texts.size());
    return 0;
}
