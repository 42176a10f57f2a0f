// ThreadSanitizer driver for oracle/hnsw.cpp (test infrastructure): several multi-threaded builds over seeded
// Gaussian rows, every query a stored row (a planted exact match the index must return first), searched on 4 threads.
// Exit code 0 = every planted match found and no race reported (TSAN_OPTIONS=halt_on_error=1 makes a report fatal).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <random>
#include <vector>

extern "C" {
void* hnsw_build(const float* x_normalised, int64_t n, int dim, int M, int ef_construction, uint64_t seed, int threads);
void hnsw_search(void* handle, const float* q_normalised, int nq, int k, int ef, int threads, int64_t* ids_out, float* cos_out);
void hnsw_free(void* handle);
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 3000;
    const int dim = 32, M = 8, efc = 60, threads = argc > 2 ? atoi(argv[2]) : 4, rounds = argc > 3 ? atoi(argv[3]) : 3;
    std::mt19937_64 rng(5);
    std::normal_distribution<float> g(0.f, 1.f);
    std::vector<float> x((size_t)n * dim);
    for (int64_t i = 0; i < n; ++i) {
        double s = 0;
        for (int d = 0; d < dim; ++d) { x[i * dim + d] = g(rng); s += (double)x[i * dim + d] * x[i * dim + d]; }
        const float inv = (float)(1.0 / sqrt(s));
        for (int d = 0; d < dim; ++d) x[i * dim + d] *= inv;
    }
    const int nq = 400, k = 5;
    int64_t missed = 0;
    for (int r = 0; r < rounds; ++r) {
        void* h = hnsw_build(x.data(), n, dim, M, efc, 100 + r, threads);
        std::vector<float> q((size_t)nq * dim);
        std::vector<int64_t> want(nq);
        for (int i = 0; i < nq; ++i) {
            want[i] = (int64_t)(rng() % (uint64_t)n);
            for (int d = 0; d < dim; ++d) q[(size_t)i * dim + d] = x[want[i] * dim + d];
        }
        std::vector<int64_t> ids((size_t)nq * k);
        std::vector<float> cos((size_t)nq * k);
        hnsw_search(h, q.data(), nq, k, 64, threads, ids.data(), cos.data());
        for (int i = 0; i < nq; ++i) missed += ids[(size_t)i * k] != want[i];
        hnsw_free(h);
    }
    printf("missed %lld of %d planted matches\n", (long long)missed, nq * rounds);
    return missed == 0 ? 0 : 1;
}
