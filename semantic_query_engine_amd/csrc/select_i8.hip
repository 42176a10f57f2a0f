// select_i8.hip -- after the int8 collect scan (scan_i8.hip): per query, gather the collected keys, re-score in fp32 what can
// matter, select the exact top-k and prove that no uncollected row can belong to it.  One workgroup per query.
//
//   keys   (int8 score, row) of every row whose scaled int8 score reached the query's threshold, from the (chunk, query)
//          lists, gathered into LDS (<= 8192 keys).  Nothing is sorted: the stages below are threshold filters, and a k-th largest
//          or a final order over a few dozen to a few hundred keys is found by rank counting (r03a sorted the keys and the
//          estimates with bitonic networks: ~110 barriers per query, half of the kernel's 140 us);
//   stage 1  the keys above a histogram cut that leaves ~64-100 of the best int8 scores get a bf16 ESTIMATE (the row of the bf16 scan copy against the fp32 query: 2 KiB per row,
//          error <= eps16 = scan_eps(0, largest bf16 residual of the index) ~ 0.002): t1 = (k-th best estimate) - eps16 is a
//          lower bound of the final k-th cosine;
//   stage 2  a row whose int8 estimate is below t1 - eps8 has a true cosine below t1: it cannot enter the top-k.  Every other
//          collected row -- a prefix of the sorted keys, ~300 of ~2,000 on 10 M Gaussian rows -- gets its bf16 estimate too;
//   stage 3  with t2 = (k-th best bf16 estimate of all of them) - eps16, only a row whose estimate reaches t2 - eps16 can have a
//          true cosine >= t2: those (~15 rows) are re-scored against the fp32 master, 4 KiB per row.  (r03a re-scored all ~300
//          in fp32: 1.2 MB per query, the HBM-bound part of this kernel; now ~0.7 MB.)
//   result  the k best of the fp32 re-scored rows by (cosine desc, row asc), exact fp32 cosines;
//   proof   an uncollected row has an estimated score below thr_eff (scan_i8.hip), hence a true cosine below
//          thr_eff + eps; the query is certified when that is below the k-th cosine found, and no list or buffer overflowed.
//          eps = scan_eps(int8 residual of the query, largest int8 residual of the index): kernels.h, the same bound the
//          bf16 certificate uses with the bf16 residuals.
//   else   collect_thr[q] = (k-th cosine found) - bf16 eps: the query joins the bf16 collect pass (exact.hip), which
//          gathers every row that can still reach that cosine and overwrites the result.
#include "kernels.h"

namespace sqe {

namespace {

constexpr int KEY_CAP = 12288;     // collected keys per query held in LDS (~2,000-3,000 expected; the anchored threshold admits up to the key budget, 6,144
                                   // PREDICTED keys -- a count of ~60 sample values, +-13 %: twice that fits)
constexpr int RS_CAP = 4096;       // rows that get a bf16 estimate per query (a whole cluster of near-identical rows: a few thousand)
constexpr int STAGE1 = 64;

struct SelArgs {
    I8SelectArgs a;
    float unit0;                   // S0^2
};

__device__ __forceinline__ int key_score_i32(uint64_t key) { return (int)((uint32_t)(key >> 32) ^ 0x80000000u); }

// true cosines of rows in[lo .. hi) -> out[lo .. hi) as (cosine, row) keys; one wave per row, two rows in flight
__device__ __forceinline__ void rescore_range(const uint64_t* in, uint64_t* out, int lo, int hi, const float* master, const float* qrow, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int nvec = K >> 2;
    const float4* qv = reinterpret_cast<const float4*>(qrow);
    for (int e = lo + wave * 2; e < hi; e += nw * 2) {
        const uint32_t r0 = key_row(in[e]);
        const bool two = e + 1 < hi;
        const uint32_t r1 = two ? key_row(in[e + 1]) : r0;
        const float4* a0 = reinterpret_cast<const float4*>(master + (size_t)r0 * K);
        const float4* a1 = reinterpret_cast<const float4*>(master + (size_t)r1 * K);
        float s0 = 0.f, s1 = 0.f;
        for (int v = lane; v < nvec; v += 64) {
            const float4 x0 = a0[v], x1 = a1[v], b = qv[v];
            s0 = fmaf(x0.x, b.x, s0); s0 = fmaf(x0.y, b.y, s0); s0 = fmaf(x0.z, b.z, s0); s0 = fmaf(x0.w, b.w, s0);
            s1 = fmaf(x1.x, b.x, s1); s1 = fmaf(x1.y, b.y, s1); s1 = fmaf(x1.z, b.z, s1); s1 = fmaf(x1.w, b.w, s1);
        }
        s0 = wave_sum(s0) + 0.0f;
        s1 = wave_sum(s1) + 0.0f;
        if (lane == 0) {
            out[e] = make_key(s0, r0);
            if (two) out[e + 1] = make_key(s1, r1);
        }
    }
}

// bf16 estimates of rows in[lo .. hi) -> out[lo .. hi) as (estimate, row) keys: the row of the bf16 scan copy (K bf16 at
// `pitch` bytes) against the fp32 query, fp32 accumulation; one wave per row, two rows in flight, 16 bytes per lane and load
__device__ __forceinline__ float dot8_bf16(const uint4 x, const float4 b0, const float4 b1, float s) {
    s = fmaf(__uint_as_float(x.x << 16), b0.x, s); s = fmaf(__uint_as_float(x.x & 0xffff0000u), b0.y, s);
    s = fmaf(__uint_as_float(x.y << 16), b0.z, s); s = fmaf(__uint_as_float(x.y & 0xffff0000u), b0.w, s);
    s = fmaf(__uint_as_float(x.z << 16), b1.x, s); s = fmaf(__uint_as_float(x.z & 0xffff0000u), b1.y, s);
    s = fmaf(__uint_as_float(x.w << 16), b1.z, s); s = fmaf(__uint_as_float(x.w & 0xffff0000u), b1.w, s);
    return s;
}
__device__ __forceinline__ void estimate_range_bf16(const uint64_t* in, uint64_t* out, int lo, int hi, const char* scan16, size_t pitch,
                                                    const float* qrow, int K) {
    constexpr int NR = 4;                      // rows in flight per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int nvec = K >> 3;
    const float4* qv = reinterpret_cast<const float4*>(qrow);
    for (int e = lo + wave * NR; e < hi; e += nw * NR) {
        uint32_t r[NR];
        const uint4* a[NR];
        float acc[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            r[i] = key_row(in[min(e + i, hi - 1)]);
            a[i] = reinterpret_cast<const uint4*>(scan16 + (size_t)r[i] * pitch);
            acc[i] = 0.f;
        }
        for (int v = lane; v < nvec; v += 64) {
            uint4 x[NR];
#pragma unroll
            for (int i = 0; i < NR; ++i) x[i] = a[i][v];
            const float4 b0 = qv[2 * v], b1 = qv[2 * v + 1];
#pragma unroll
            for (int i = 0; i < NR; ++i) acc[i] = dot8_bf16(x[i], b0, b1, acc[i]);
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) acc[i] = wave_sum(acc[i]) + 0.0f;
        if (lane == 0) {                       // (in and out may be the same array: this wave's slots only, read above)
#pragma unroll
            for (int i = 0; i < NR; ++i)
                if (e + i < hi) out[e + i] = make_key(acc[i], r[i]);
        }
    }
}

// k-th largest (1-based) of v[0 .. n) by rank counting (keys are distinct: the row is part of them) -> *out; all threads call it
__device__ __forceinline__ void kth_largest(const uint64_t* v, int n, int k, uint64_t* out) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const uint64_t me = v[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += v[j] > me ? 1 : 0;
        if (rank == k - 1) *out = me;
    }
}

constexpr int S1_TARGET = 64, S1_CAP = 256, NBIN = 1024;

__global__ __launch_bounds__(1024) void select_i8_kernel(SelArgs sa) {
    const I8SelectArgs& p = sa.a;
    __shared__ uint64_t keys[KEY_CAP];     // the gathered (int8 score, row) keys; from stage 3 on the (fp32 cosine, row) keys
    __shared__ uint64_t tk[RS_CAP];        // the histogram, then the (bf16 estimate, row) keys
    __shared__ int s_total, s_over, s_min, s_max, s_cutbin, s_n1, s_n2, s_n3, s_wtot[16];
    __shared__ uint64_t s_kth;
    const int q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        s_total = 0; s_over = 0; s_min = 0x7fffffff; s_max = (int)0x80000000; s_cutbin = 0; s_n1 = 0; s_n2 = 0; s_n3 = 0; s_kth = 0ull;
    }
    __syncthreads();
    // ---- gather: one wave per chunk list; "chunk" n_chunks is the query's overflow pool (what found its list full)
    for (int c = wave; c <= p.n_chunks; c += (int)(blockDim.x >> 6)) {
        const bool pool = c == p.n_chunks;
        int n = pool ? p.ovf_cnt[q] : min(p.cand_cnt[(size_t)c * p.b_pad + q], CAND_CAP);
        if (pool && n > I8_OVF_CAP) { n = I8_OVF_CAP; if (lane == 0) s_over = 1; }       // the pool overflowed too: rows were dropped
        if (n <= 0) continue;
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_total, n);
        base = __shfl(base, 0, 64);
        const uint64_t* list = pool ? p.ovf + (size_t)q * I8_OVF_CAP : p.cand + ((size_t)c * p.b_pad + q) * CAND_CAP;
        for (int i = lane; i < n; i += 64)
            if (base + i < KEY_CAP) keys[base + i] = list[i];
    }
    __syncthreads();
    int N = s_total;
    bool overflow = s_over != 0;
    if (N > KEY_CAP) { N = KEY_CAP; overflow = true; }
    // ---- histogram cut: the bin boundary above which ~S1_TARGET keys lie (1,024 linear bins over the scores present)
    {
        int lo = 0x7fffffff, hi = (int)0x80000000;
        for (int i = tid; i < N; i += blockDim.x) {
            const int sc = key_score_i32(keys[i]);
            lo = min(lo, sc); hi = max(hi, sc);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo = min(lo, __shfl_xor(lo, off, 64));
            hi = max(hi, __shfl_xor(hi, off, 64));
        }
        if (lane == 0 && lo <= hi) { atomicMin(&s_min, lo); atomicMax(&s_max, hi); }
    }
    int* hist = reinterpret_cast<int*>(tk);
    for (int i = tid; i < NBIN; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const long long smin = s_min, range = (long long)s_max - smin + 1;
    auto bin_of = [&](int sc) { return (int)((((long long)sc - smin) * NBIN) / range); };
    for (int i = tid; i < N; i += blockDim.x) atomicAdd(&hist[bin_of(key_score_i32(keys[i]))], 1);
    __syncthreads();
    {
        // suffix sums over the bins: thread t takes bin NBIN - 1 - t (a prefix scan in t); blockDim.x == NBIN
        const int mine = hist[NBIN - 1 - tid];
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) s_wtot[wave] = incl;
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave; ++w) before += s_wtot[w];
        incl += before;                                                       // keys in bins >= NBIN - 1 - tid
        if (incl >= S1_TARGET && incl - mine < S1_TARGET) s_cutbin = NBIN - 1 - tid;
    }
    __syncthreads();
    // ---- stage 1: bf16 estimates of the keys at or above the cut (any k rows give a valid lower bound: a capped subset will do)
    const int cutbin = s_cutbin;
    for (int i = tid; i < N; i += blockDim.x) {
        const uint64_t key = keys[i];
        if (bin_of(key_score_i32(key)) >= cutbin) {
            const int at = atomicAdd(&s_n1, 1);
            if (at < S1_CAP) {
                tk[at] = key;                                                 // (the histogram is dead: every thread is past the barrier)
                keys[i] = 0ull;                                               // consumed (a real key is never 0)
            }
        }
    }
    __syncthreads();
    const float* qrow = p.qn + (size_t)q * p.K;
    const char* scan16 = reinterpret_cast<const char*>(p.scan16);
    const float eps16r = scan_eps(0.f, __uint_as_float(*p.db_resid16_max), p.K);   // |bf16 row . fp32 query - true cosine|
    const int R1 = min(s_n1, S1_CAP);
    estimate_range_bf16(tk, tk, 0, R1, scan16, (size_t)p.pitch16, qrow, p.K);
    __syncthreads();
    if (R1 >= p.k) kth_largest(tk, R1, p.k, &s_kth);
    __syncthreads();
    const float eps8 = scan_eps(p.q_resid8[q], __uint_as_float(*p.db_resid8_max), p.K);
    const double unit = (double)sa.unit0 * (double)p.sqi[q];
    int thr2 = -0x7fffffff;                                                   // fewer than k rows in stage 1: everything goes on
    if (R1 >= p.k) {
        const float t1 = key_score(s_kth) - eps16r;                           // k rows have a true cosine >= t1
        // rows with score_int < thr2 have an estimated cosine below t1 - eps8, a true cosine below t1
        const double v = floor(((double)t1 - (double)eps8) / unit) - 1.0;
        thr2 = v < -2.0e9 ? -0x7fffffff : v > 2.0e9 ? 0x7fffffff : (int)v;
    }
    // ---- stage 2: bf16 estimates of every other key that reaches thr2
    if (tid == 0) s_n2 = R1;
    __syncthreads();
    for (int i = tid; i < N; i += blockDim.x) {
        const uint64_t key = keys[i];
        if (key != 0ull && key_score_i32(key) >= thr2) {
            const int at = atomicAdd(&s_n2, 1);
            if (at < RS_CAP) tk[at] = key;
        }
    }
    __syncthreads();
    int R2 = s_n2;
    if (R2 > RS_CAP) { R2 = RS_CAP; overflow = true; }
    estimate_range_bf16(tk, tk, R1, R2, scan16, (size_t)p.pitch16, qrow, p.K);
    __syncthreads();
    // ---- stage 3: fp32 re-score of the rows whose estimate reaches (k-th best estimate) - 2 eps16; `keys` is free by now
    uint64_t* fk = keys;
    int R3 = R2;
    if (R2 > p.k) {
        kth_largest(tk, R2, p.k, &s_kth);
        __syncthreads();
        const uint64_t cut_key = make_key(key_score(s_kth) - 2.0f * eps16r, 0xFFFFFFFFu);   // (lowest key of that score)
        for (int i = tid; i < R2; i += blockDim.x) {
            const uint64_t key = tk[i];
            if (key >= cut_key) fk[atomicAdd(&s_n3, 1)] = key;
        }
        __syncthreads();
        R3 = s_n3;
    } else {
        for (int i = tid; i < R2; i += blockDim.x) fk[i] = tk[i];
        __syncthreads();
    }
    rescore_range(fk, fk, 0, R3, p.master, qrow, p.K);
    __syncthreads();
    // ---- result: the k best by (cosine desc, row asc), placed by rank
    const int m = min(R3, p.k);
    float* cos_out = p.cos_out + (size_t)q * p.k;
    int64_t* id_out = p.id_out + (size_t)q * p.k;
    // Fewer than k rows collected (the threshold sat inside a band of near-identical rows and the int8 estimates fell
    // on the wrong side of it): the k best rows of the SAMPLE -- real rows with true cosines -- stand in until the bf16
    // pass, which this query now takes, overwrites them.
    const bool from_sample = m < p.k;
    const float* cs = p.sample_cos + (size_t)q * p.sample_m;
    const int64_t* is = p.sample_ids + (size_t)q * p.sample_m;
    if (from_sample) {
        for (int i = tid; i < p.k; i += blockDim.x) {
            cos_out[i] = cs[i];
            id_out[i] = is[i] >= 0 ? is[i] + p.id_base : -1;
        }
    } else {
        for (int i = tid; i < R3; i += blockDim.x) {
            const uint64_t me = fk[i];
            int rank = 0;
            for (int j = 0; j < R3; ++j) rank += fk[j] > me ? 1 : 0;
            if (rank < p.k) {
                cos_out[rank] = key_score(me);
                id_out[rank] = (int64_t)key_row(me) + p.id_base;
                if (rank == p.k - 1) s_kth = me;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        const float kth = m >= p.k ? key_score(s_kth) : -INFINITY;
        const bool certified = !overflow && m >= p.k && p.thr_eff[q] + eps8 < kth;
        float thr = INFINITY;
        if (!certified) {
            const float eps16 = scan_eps(p.q_resid16[q], __uint_as_float(*p.db_resid16_max), p.K);
            // lower bounds of the true k-th cosine: the k-th found here, the k-th of the sample (both are real rows)
            const float lb = fmaxf(kth, cs[p.k - 1]);
            thr = (lb > -INFINITY ? lb : -1.0f) - eps16;
            atomicAdd(p.unc_count, 1);
        }
        p.collect_thr[q] = thr;
        if (p.stats) {
            atomicAdd(&p.stats[0], (unsigned long long)N);
            atomicAdd(&p.stats[1], (unsigned long long)R3);                  // rows re-scored in fp32 (R2 got a bf16 estimate)
            if (overflow) atomicAdd(&p.stats[2], 1ull);
            if (!certified) atomicAdd(&p.stats[3], 1ull);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// After the int8 threshold pass (scan_i8.hip: sample_i8_pp_kernel): per query, the m-th largest of its n_chunks x 16 sample
// scores becomes the collect threshold (same units as the collect predicate: acc x tile scale), and the k best sample rows
// are re-scored in fp32: real rows with true cosines, the lower bounds select_i8_kernel falls back on.  One workgroup of
// 1,024 threads per query; a 1,024-bin histogram finds the cut, rank counting orders the few values above it.
constexpr int SS_CAP = 4096;       // sample values per query (256 chunks x 16)
constexpr int SS_TOP = 256;        // values kept above the cut

__global__ __launch_bounds__(1024) void i8_sample_select_kernel(I8SampleSelectArgs p, float unit0) {
    __shared__ uint64_t vals[SS_CAP];
    __shared__ uint64_t top[SS_TOP];
    __shared__ uint64_t fk[64];
    __shared__ int hist[1024];
    __shared__ int s_min, s_max, s_cutbin, s_ntop, s_wtot[16];
    __shared__ uint64_t s_mth;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = min(p.n_chunks * 16, SS_CAP);
    if (tid == 0) { s_min = 0x7fffffff; s_max = (int)0x80000000; s_cutbin = 0; s_ntop = 0; s_mth = 0ull; }
    hist[tid] = 0;
    __syncthreads();
    // ---- gather (empty slots carry row -1: key 0, which sorts last)
    const int2* cand = reinterpret_cast<const int2*>(p.cand);
    int lo = 0x7fffffff, hi = (int)0x80000000;
    for (int i = tid; i < N; i += 1024) {
        const int2 v = cand[((size_t)(i >> 4) * p.b_pad_s + q) * 16 + (i & 15)];
        const bool ok = v.y >= 0;
        vals[i] = ok ? (((uint64_t)((uint32_t)v.x ^ 0x80000000u) << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)v.y)) : 0ull;
        if (ok) { lo = min(lo, v.x); hi = max(hi, v.x); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, __shfl_xor(lo, off, 64));
        hi = max(hi, __shfl_xor(hi, off, 64));
    }
    if (lane == 0 && lo <= hi) { atomicMin(&s_min, lo); atomicMax(&s_max, hi); }
    __syncthreads();
    const long long smin = s_min, range = (long long)s_max - smin + 1;
    auto bin_of = [&](uint64_t key) { return (int)((((long long)key_score_i32(key) - smin) * 1024) / range); };
    for (int i = tid; i < N; i += 1024)
        if (vals[i] != 0ull) atomicAdd(&hist[bin_of(vals[i])], 1);
    __syncthreads();
    const int want = min(p.m + 8, SS_TOP);             // the m-th largest lies among the values at or above the cut
    {
        const int mine = hist[1023 - tid];
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) s_wtot[wave] = incl;
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave; ++w) before += s_wtot[w];
        incl += before;
        if (incl >= want && incl - mine < want) s_cutbin = 1023 - tid;
    }
    __syncthreads();
    const int cutbin = s_cutbin;
    for (int i = tid; i < N; i += 1024) {
        const uint64_t key = vals[i];
        if (key != 0ull && bin_of(key) >= cutbin) {
            const int at = atomicAdd(&s_ntop, 1);
            if (at < SS_TOP) top[at] = key;
        }
    }
    __syncthreads();
    const int T = min(s_ntop, SS_TOP);
    // ---- the m-th largest (fewer than m values: the smallest one -- a short sample collects more, never less)
    const int mth = min(p.m, T);
    if (T > 0) kth_largest(top, T, mth, &s_mth);
    __syncthreads();
    // ---- the k best sample rows, re-scored in fp32, best true cosine first
    const int kk = min(p.k, T);
    for (int i = tid; i < T; i += 1024) {
        const uint64_t me = top[i];
        int rank = 0;
        for (int j = 0; j < T; ++j) rank += top[j] > me ? 1 : 0;
        if (rank < kk) fk[rank] = me;
    }
    __syncthreads();
    rescore_range(fk, fk, 0, kk, p.master, p.qn + (size_t)q * p.K, p.K);
    __syncthreads();
    float* cs = p.sample_cos + (size_t)q * p.m;
    int64_t* is = p.sample_ids + (size_t)q * p.m;
    for (int i = tid; i < p.m; i += 1024) {
        if (i < kk) {
            const uint64_t me = fk[i];
            int rank = 0;
            for (int j = 0; j < kk; ++j) rank += fk[j] > me ? 1 : 0;
            cs[rank] = key_score(me);
            is[rank] = (int64_t)key_row(me);
        } else {
            cs[i] = -INFINITY;
            is[i] = -1;
        }
    }
    __syncthreads();
    // ---- the threshold.  Default: the m-th largest sample score (~ step x m rows collected).  The proof select_i8_kernel runs
    // afterwards is thr_eff + eps < (k-th cosine found); on rows that crowd together -- a cluster whose members are all within
    // eps of each other: what text embeddings look like -- the m-th sample score lies INSIDE the crowd and the proof fails whatever
    // the scan collects (r03: every query of the clustered 10 M set took the bf16 pass on top of the int8 one).  The sample's
    // best row is a real row with a TRUE cosine c0, and the k-th cosine of the index is >= c0 unless the sample caught one of
    // the top k - 1 rows; so the threshold is ANCHORED: never above c0 - eps (1 + margin).  Then thr_eff + eps <= c0 - margin
    // eps < k-th cosine: the proof holds by construction, provided the crowd fits the lists.  Whether it fits is estimated from
    // the sample too (values at or above the threshold x step): where the anchored threshold would collect more than the key
    // budget, the m-th sample score stands as it did (the best ~ step x m rows by estimate; the proof then fails and the bf16 pass
    // starts from the k-th cosine found among them).
    if (tid == 0) {
        const double unit = (double)unit0 * (double)p.sqi[q];
        const int t = T > 0 ? key_score_i32(s_mth) : -0x7fffffff;      // no sample: collect everything (the bf16 pass answers)
        int ta = t;
        if (kk > 0 && p.q_resid8) {
            const float eps8 = scan_eps(p.q_resid8[q], __uint_as_float(*p.db_resid8_max), p.K);
            const float c0 = cs[0];                                    // best true cosine of the sample (written above by this block)
            const double v = floor(((double)c0 - (double)eps8 * (1.0 + (double)p.margin)) / unit) - 1.0;
            ta = min(t, v < -2.0e9 ? -0x7fffffff : v > 2.0e9 ? 0x7fffffff : (int)v);
        }
        s_min = t;                                                     // (reused: the two thresholds, for the count below)
        s_cutbin = ta;
        s_ntop = 0;
    }
    __syncthreads();
    if (s_cutbin < s_min) {
        const int t = s_cutbin;
        int mine = 0;
        for (int i = tid; i < N; i += 1024) mine += (vals[i] != 0ull && key_score_i32(vals[i]) >= t) ? 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
        if (lane == 0 && mine) atomicAdd(&s_ntop, mine);
    }
    __syncthreads();
    if (tid == 0) {
        const double unit = (double)unit0 * (double)p.sqi[q];
        const bool too_many = p.key_budget > 0 && (long long)s_ntop * p.step > (long long)p.key_budget;
        const int t = too_many ? s_min : s_cutbin;
        p.thr_int[q] = t;
        p.thr_eff[q] = (float)((double)t * unit * (1.0 + 1e-6) + 1e-7);   // rounded up (quant.hip: i8_thresholds_kernel)
    }
}

__global__ void i8_pad_thresholds_kernel(int B, int b_pad, int* __restrict__ thr_int, float* __restrict__ thr_eff) {
    const int q = B + blockIdx.x * blockDim.x + threadIdx.x;
    if (q < b_pad) { thr_int[q] = 0x7fffffff; thr_eff[q] = INFINITY; }       // padding queries collect nothing
}

}  // namespace

int launch_select_i8(const I8SelectArgs& a, hipStream_t stream) {
    if (a.B <= 0) return SQE_OK;
    if (a.k < 1 || a.k > STAGE1 || a.k > a.sample_m) return fail(SQE_ERR_INVALID, "int8 select: k must be in [1, min(64, sample depth)]");
    if (a.K % 8 != 0 || a.pitch16 % 16 != 0 || !a.scan16) return fail(SQE_ERR_INVALID, "int8 select: dim must be a multiple of 8, the bf16 copy 16-byte aligned rows");
    SelArgs sa;
    sa.a = a;
    const float s0 = i8_scale_unit(a.K);
    sa.unit0 = s0 * s0;
    // 16 waves per query: the 80 KiB of LDS admit two workgroups per CU, the re-score wants as many row fetches in flight as it can get
    hipLaunchKernelGGL(select_i8_kernel, dim3(a.B), dim3(1024), 0, stream, sa);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

int launch_i8_sample_select(const I8SampleSelectArgs& a, hipStream_t stream) {
    if (a.B <= 0) return SQE_OK;
    if (a.m < 1 || a.m > 64 || a.k < 1 || a.k > a.m || a.K % 4 != 0 || a.n_chunks < 1)
        return fail(SQE_ERR_INVALID, "int8 sample select: 1 <= k <= m <= 64");
    const float s0 = i8_scale_unit(a.K);
    hipLaunchKernelGGL(i8_sample_select_kernel, dim3(a.B), dim3(1024), 0, stream, a, s0 * s0);
    if (a.b_pad > a.B)
        hipLaunchKernelGGL(i8_pad_thresholds_kernel, dim3((a.b_pad - a.B + 255) / 256), dim3(256), 0, stream, a.B, a.b_pad, a.thr_int, a.thr_eff);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

}  // namespace sqe
