/* Plain-C restatement of the retrieval arithmetic of the reference hot path.
 *
 * TEST INFRASTRUCTURE ONLY: built into oracle/_build/liboracle.so and loaded by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the
 * product library (libsqe.so does not link or dlopen it).
 *
 * Each function cites the lines of /root/reference/app/main.py it follows.
 * Parity status: see oracle/__init__.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* main.py:315-316 / 353-354: e / (||e|| + 1e-9) in float32.  NumPy's float32 norm is
 * sqrt(sum(x*x)) with pairwise summation; a double accumulator rounded once to
 * float32 is within 1 ulp of it, which is what the tests allow. */
void oracle_normalize_rows(const float* x, int64_t n, int dim, float* out) {
    for (int64_t i = 0; i < n; ++i) {
        const float* r = x + i * (int64_t)dim;
        double ss = 0.0;
        for (int d = 0; d < dim; ++d) ss += (double)r[d] * (double)r[d];
        float nrm = (float)sqrt(ss);
        float den = nrm + 1e-9f;
        float* o = out + i * (int64_t)dim;
        for (int d = 0; d < dim; ++d) o[d] = r[d] / den;
    }
}

/* main.py:59-64: cosine with the zero-norm rule, float32 inputs, widened result. */
double oracle_cosine_similarity(const float* a, const float* b, int dim) {
    double na = 0.0, nb = 0.0, dot = 0.0;
    for (int d = 0; d < dim; ++d) {
        na += (double)a[d] * a[d];
        nb += (double)b[d] * b[d];
        dot += (double)a[d] * b[d];
    }
    float fa = (float)sqrt(na), fb = (float)sqrt(nb);
    if (fa == 0.0f || fb == 0.0f) return 0.0;
    return (double)((float)dot / (fa * fb));
}

/* main.py:73-87: first strict maximum from (-1.0, -1); NaN never wins. */
void oracle_cosine_best(const float* mat, int m, int dim, const float* q,
                        double* best_sim, int* best_idx) {
    double bs = -1.0;
    int bi = -1;
    for (int i = 0; i < m; ++i) {
        double s = oracle_cosine_similarity(q, mat + (int64_t)i * dim, dim);
        if (s > bs) { bs = s; bi = i; }
    }
    *best_sim = bs;
    *best_idx = bi;
}

/* Exact cosine top-k over normalised rows (the semantics restated for
 * main.py:356-367): double dot products, best first, ties -> lowest id.
 * cos_out[B,k] (double), id_out[B,k] (int64, -1 padded).  Single thread. */
void oracle_exact_topk(const float* xn, int64_t n, const float* qn, int b, int dim, int k,
                       double* cos_out, int64_t* id_out) {
    for (int qi = 0; qi < b; ++qi) {
        double* cs = cos_out + (int64_t)qi * k;
        int64_t* is = id_out + (int64_t)qi * k;
        int cnt = 0;
        for (int j = 0; j < k; ++j) { cs[j] = -INFINITY; is[j] = -1; }
        const float* q = qn + (int64_t)qi * dim;
        for (int64_t i = 0; i < n; ++i) {
            const float* r = xn + i * (int64_t)dim;
            double s = 0.0;
            for (int d = 0; d < dim; ++d) s += (double)r[d] * (double)q[d];
            if (s != s) continue;                         /* NaN rows never rank */
            if (cnt == k && !(s > cs[k - 1])) continue;   /* equal score: lower id stays */
            int pos = cnt < k ? cnt : k - 1;
            while (pos > 0 && s > cs[pos - 1]) {
                cs[pos] = cs[pos - 1];
                is[pos] = is[pos - 1];
                --pos;
            }
            cs[pos] = s;
            is[pos] = i;
            if (cnt < k) ++cnt;
        }
    }
}
