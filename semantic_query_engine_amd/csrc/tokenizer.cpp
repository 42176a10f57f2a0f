// tokenizer.cpp -- host-only BERT (uncased) WordPiece tokenizer behind sqe_tokenizer_*.
//
// Stands where llama.cpp's WordPiece tokenizer stood inside Ollama: the reference sends raw
// text as "prompt" (main.py:139-142).  Algorithm (the published BERT one, as implemented by
// tokenizers' BertNormalizer / BertPreTokenizer / WordPiece):
//   clean (drop NUL, U+FFFD, control chars; whitespace -> ' ') -> spaces around CJK ->
//   NFD + strip Mn -> lower-case -> split on whitespace and punctuation ->
//   greedy longest-match WordPiece with "##" continuations (words > 100 chars -> [UNK]) ->
//   [CLS] ... [SEP], truncated to max_len ids.
// Unicode data come from the generated unicode_tables.h (tools/gen_unicode_tables.py).
// Known deviation: lower-casing is per code point (no final-sigma context rule).
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "host_errors.h"
#include "unicode_tables.h"

struct sqe_tokenizer {
    std::unordered_map<std::string, int32_t> vocab;
    int32_t unk = 100, cls = 101, sep = 102;
};

namespace {

using sqe::fail;

uint8_t cp_flags(uint32_t cp) {
    if (cp >= sqe::uni::kMaxCp) return cp >= 0xE0000 ? 1 : 0;   // tags / private planes: control-like
    int lo = 0, hi = sqe::uni::kNumRanges - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const auto& r = sqe::uni::kRanges[mid];
        if (cp < r.lo) hi = mid - 1;
        else if (cp > r.hi) lo = mid + 1;
        else return r.flags;
    }
    return 0;
}

// appends lower(strip_Mn(NFD(cp)))
void append_normalized(uint32_t cp, std::vector<uint32_t>& out) {
    if (cp < sqe::uni::kMaxCp) {
        int lo = 0, hi = sqe::uni::kNumMaps - 1;
        while (lo <= hi) {
            const int mid = (lo + hi) >> 1;
            const auto& m = sqe::uni::kMaps[mid];
            if (cp < m.cp) hi = mid - 1;
            else if (cp > m.cp) lo = mid + 1;
            else {
                for (int i = 0; i < m.len; ++i) out.push_back(sqe::uni::kPool[m.off + i]);
                return;
            }
        }
    }
    out.push_back(cp);
}

bool is_cjk(uint32_t cp) {
    return (cp >= 0x4E00 && cp <= 0x9FFF) || (cp >= 0x3400 && cp <= 0x4DBF) || (cp >= 0x20000 && cp <= 0x2A6DF) ||
           (cp >= 0x2A700 && cp <= 0x2B73F) || (cp >= 0x2B740 && cp <= 0x2B81F) || (cp >= 0x2B820 && cp <= 0x2CEAF) ||
           (cp >= 0xF900 && cp <= 0xFAFF) || (cp >= 0x2F800 && cp <= 0x2FA1F);
}

// lenient UTF-8 decode: malformed bytes become U+FFFD (which the cleaner then drops)
void decode_utf8(const char* s, int64_t n, std::vector<uint32_t>& out) {
    const unsigned char* p = reinterpret_cast<const unsigned char*>(s);
    int64_t i = 0;
    while (i < n) {
        const unsigned char c = p[i];
        uint32_t cp = 0xFFFD;
        int len = 1;
        if (c < 0x80) cp = c;
        else if ((c >> 5) == 6 && i + 1 < n && (p[i + 1] & 0xC0) == 0x80) {
            cp = ((c & 0x1F) << 6) | (p[i + 1] & 0x3F); len = 2;
            if (cp < 0x80) cp = 0xFFFD;
        } else if ((c >> 4) == 14 && i + 2 < n && (p[i + 1] & 0xC0) == 0x80 && (p[i + 2] & 0xC0) == 0x80) {
            cp = ((c & 0x0F) << 12) | ((p[i + 1] & 0x3F) << 6) | (p[i + 2] & 0x3F); len = 3;
            if (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF)) cp = 0xFFFD;
        } else if ((c >> 3) == 30 && i + 3 < n && (p[i + 1] & 0xC0) == 0x80 && (p[i + 2] & 0xC0) == 0x80 &&
                   (p[i + 3] & 0xC0) == 0x80) {
            cp = ((c & 0x07) << 18) | ((p[i + 1] & 0x3F) << 12) | ((p[i + 2] & 0x3F) << 6) | (p[i + 3] & 0x3F); len = 4;
            if (cp < 0x10000 || cp > 0x10FFFF) cp = 0xFFFD;
        }
        out.push_back(cp);
        i += len;
    }
}

void encode_utf8(uint32_t cp, std::string& out) {
    if (cp < 0x80) out.push_back((char)cp);
    else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) {
        out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
        out.push_back((char)(0x80 | (cp & 0x3F)));
    } else {
        out.push_back((char)(0xF0 | (cp >> 18))); out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
        out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F)));
    }
}

void wordpiece(const sqe_tokenizer& t, const std::vector<uint32_t>& word, std::vector<int32_t>& ids, size_t limit) {
    if (word.size() > 100) { ids.push_back(t.unk); return; }
    // byte offsets of each code point in the UTF-8 form of the word
    std::string utf8;
    std::vector<int> off(word.size() + 1, 0);
    for (size_t i = 0; i < word.size(); ++i) { off[i] = (int)utf8.size(); encode_utf8(word[i], utf8); }
    off[word.size()] = (int)utf8.size();
    const size_t first = ids.size();
    size_t start = 0;
    std::string piece;
    while (start < word.size()) {
        size_t end = word.size();
        int32_t found = -1;
        while (start < end) {
            piece.clear();
            if (start > 0) piece = "##";
            piece.append(utf8, off[start], off[end] - off[start]);
            auto it = t.vocab.find(piece);
            if (it != t.vocab.end()) { found = it->second; break; }
            --end;
        }
        if (found < 0) { ids.resize(first); ids.push_back(t.unk); return; }
        ids.push_back(found);
        start = end;
        if (ids.size() > limit + 8) return;     // already past the truncation point
    }
}

int tokenize_one(const sqe_tokenizer& t, const char* text, int64_t bytes, int max_len, int32_t* ids_out, int* len_out) {
    std::vector<uint32_t> cps, norm;
    decode_utf8(text, bytes, cps);
    norm.reserve(cps.size() + 16);
    for (uint32_t cp : cps) {
        if (cp == 0 || cp == 0xFFFD) continue;
        const uint8_t fl = cp_flags(cp);
        if (fl & 2) { norm.push_back(' '); continue; }      // whitespace first: \t \n \r are Cc too
        if (fl & 1) continue;                                // control
        if (is_cjk(cp)) { norm.push_back(' '); norm.push_back(cp); norm.push_back(' '); continue; }
        if (fl & 4) continue;                                // a bare combining mark
        append_normalized(cp, norm);
    }
    const size_t body = max_len > 2 ? (size_t)max_len - 2 : 0;
    std::vector<int32_t> ids;
    ids.reserve(body + 16);
    std::vector<uint32_t> word;
    auto flush = [&]() {
        if (!word.empty() && ids.size() < body) wordpiece(t, word, ids, body);
        word.clear();
    };
    for (uint32_t cp : norm) {
        if (ids.size() >= body) break;
        if (cp == ' ') { flush(); continue; }
        const uint8_t fl = cp_flags(cp);
        if (fl & 8) { flush(); word.push_back(cp); flush(); continue; }
        word.push_back(cp);
    }
    flush();
    if (ids.size() > body) ids.resize(body);
    int n = 0;
    if (max_len >= 1) ids_out[n++] = t.cls;
    for (size_t i = 0; i < ids.size() && n < max_len - 1; ++i) ids_out[n++] = ids[i];
    if (n < max_len) ids_out[n++] = t.sep;
    *len_out = n;
    return SQE_OK;
}

}  // namespace

extern "C" {

int sqe_tokenizer_create(const char* vocab_utf8, int64_t vocab_bytes, sqe_tokenizer** out) {
    if (!out) return fail(SQE_ERR_INVALID, "sqe_tokenizer_create: out is null");
    *out = nullptr;
    if (!vocab_utf8 || vocab_bytes <= 0) return fail(SQE_ERR_INVALID, "sqe_tokenizer_create: empty vocab");
    sqe_tokenizer* t = new (std::nothrow) sqe_tokenizer;
    if (!t) return fail(SQE_ERR_OOM, "sqe_tokenizer_create: host allocation failed");
    int32_t id = 0;
    int64_t lo = 0;
    for (int64_t i = 0; i <= vocab_bytes; ++i) {
        if (i == vocab_bytes || vocab_utf8[i] == '\n') {
            int64_t hi = i;
            if (hi > lo && vocab_utf8[hi - 1] == '\r') --hi;
            if (hi > lo || i < vocab_bytes) t->vocab.emplace(std::string(vocab_utf8 + lo, (size_t)(hi - lo)), id++);
            lo = i + 1;
        }
    }
    auto need = [&](const char* tok, int32_t* dst) {
        auto it = t->vocab.find(tok);
        if (it == t->vocab.end()) return false;
        *dst = it->second;
        return true;
    };
    if (!need("[UNK]", &t->unk) || !need("[CLS]", &t->cls) || !need("[SEP]", &t->sep)) {
        delete t;
        return fail(SQE_ERR_INVALID, "sqe_tokenizer_create: vocab lacks [UNK]/[CLS]/[SEP]");
    }
    *out = t;
    return SQE_OK;
}

void sqe_tokenizer_destroy(sqe_tokenizer* t) { delete t; }

int sqe_tokenize(const sqe_tokenizer* t, const char* text_utf8, int64_t text_bytes, int max_len,
                 int32_t* ids_out, int* len_out) {
    if (!t || !ids_out || !len_out || max_len < 2 || text_bytes < 0 || (text_bytes > 0 && !text_utf8))
        return fail(SQE_ERR_INVALID, "sqe_tokenize: bad arguments (max_len >= 2)");
    return tokenize_one(*t, text_utf8, text_bytes, max_len, ids_out, len_out);
}

int sqe_tokenize_batch(const sqe_tokenizer* t, const char* const* texts, const int64_t* text_bytes, int n,
                       int max_len, int32_t* ids_out, int32_t* lens_out) {
    if (!t || n < 0 || max_len < 2 || (n > 0 && (!texts || !text_bytes || !ids_out || !lens_out)))
        return fail(SQE_ERR_INVALID, "sqe_tokenize_batch: bad arguments");
    const int nthreads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), n >= 64 ? 16u : 1u);
    auto work = [&](int w) {
        for (int i = w; i < n; i += nthreads) {
            int len = 0;
            int32_t* row = ids_out + (size_t)i * max_len;
            tokenize_one(*t, texts[i], text_bytes[i], max_len, row, &len);
            for (int j = len; j < max_len; ++j) row[j] = 0;     // [PAD]
            lens_out[i] = len;
        }
    };
    if (nthreads == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int w = 0; w < nthreads; ++w) th.emplace_back(work, w);
        for (auto& x : th) x.join();
    }
    return SQE_OK;
}

}  // extern "C"
