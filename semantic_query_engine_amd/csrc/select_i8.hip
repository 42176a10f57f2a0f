// select_i8.hip -- after the int8 collect scan (scan_i8.hip): per query, gather the collected keys, re-score in fp32 what can
// matter, select the exact top-k and prove that no uncollected row can belong to it.  One workgroup per query.
//
//   keys   (int8 score, row) of every row whose scaled int8 score reached the query's threshold, from the (chunk, query)
//          lists; sorted best first in LDS (bitonic, <= 8192 keys);
//   stage 1  the 64 best by int8 score get a bf16 ESTIMATE (the row of the bf16 scan copy against the fp32 query: 2 KiB per row,
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

constexpr int KEY_CAP = 8192;      // collected keys per query held in LDS (~2,000 expected: the sample's threshold is a noisy estimate)
constexpr int RS_CAP = 2048;       // rows re-scored per query
constexpr int STAGE1 = 64;

struct SelArgs {
    I8SelectArgs a;
    float unit0;                   // S0^2
};

// descending bitonic sort of n (power of two) keys in LDS, all threads of the block
__device__ __forceinline__ void bitonic_desc(uint64_t* v, int n) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < (n >> 1); t += nt) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const uint64_t x = v[lo], y = v[hi];
                if ((x < y) == desc) { v[lo] = y; v[hi] = x; }
            }
        }
    }
    __syncthreads();
}

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
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int nvec = K >> 3;
    const float4* qv = reinterpret_cast<const float4*>(qrow);
    for (int e = lo + wave * 2; e < hi; e += nw * 2) {
        const uint32_t r0 = key_row(in[e]);
        const bool two = e + 1 < hi;
        const uint32_t r1 = two ? key_row(in[e + 1]) : r0;
        const uint4* a0 = reinterpret_cast<const uint4*>(scan16 + (size_t)r0 * pitch);
        const uint4* a1 = reinterpret_cast<const uint4*>(scan16 + (size_t)r1 * pitch);
        float s0 = 0.f, s1 = 0.f;
        for (int v = lane; v < nvec; v += 64) {
            const uint4 x0 = a0[v], x1 = a1[v];
            const float4 b0 = qv[2 * v], b1 = qv[2 * v + 1];
            s0 = dot8_bf16(x0, b0, b1, s0);
            s1 = dot8_bf16(x1, b0, b1, s1);
        }
        s0 = wave_sum(s0) + 0.0f;
        s1 = wave_sum(s1) + 0.0f;
        if (lane == 0) {
            out[e] = make_key(s0, r0);
            if (two) out[e + 1] = make_key(s1, r1);
        }
    }
}

__global__ __launch_bounds__(1024) void select_i8_kernel(SelArgs sa) {
    const I8SelectArgs& p = sa.a;
    __shared__ uint64_t keys[KEY_CAP];
    __shared__ uint64_t tk[RS_CAP];
    __shared__ int s_total, s_over, s_cut;
    __shared__ uint64_t s_t1;
    const int q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { s_total = 0; s_over = 0; s_cut = 0; s_t1 = 0ull; }
    __syncthreads();
    // ---- gather: one wave per chunk list
    for (int c = wave; c < p.n_chunks; c += (int)(blockDim.x >> 6)) {
        int n = p.cand_cnt[(size_t)c * p.b_pad + q];
        if (n > CAND_CAP) { n = CAND_CAP; if (lane == 0) s_over = 1; }       // the list overflowed: rows were dropped
        if (n <= 0) continue;
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_total, n);
        base = __shfl(base, 0, 64);
        const uint64_t* list = p.cand + ((size_t)c * p.b_pad + q) * CAND_CAP;
        for (int i = lane; i < n; i += 64)
            if (base + i < KEY_CAP) keys[base + i] = list[i];
    }
    __syncthreads();
    int N = s_total;
    bool overflow = s_over != 0;
    if (N > KEY_CAP) { N = KEY_CAP; overflow = true; }
    int np2 = 1;
    while (np2 < N) np2 <<= 1;
    for (int i = N + tid; i < np2; i += blockDim.x) keys[i] = 0ull;           // key 0 sorts last (a real key is never 0)
    bitonic_desc(keys, np2);
    // ---- stage 1: bf16 estimates of the 64 best by int8 score
    const float* qrow = p.qn + (size_t)q * p.K;
    const char* scan16 = reinterpret_cast<const char*>(p.scan16);
    const float eps16r = scan_eps(0.f, __uint_as_float(*p.db_resid16_max), p.K);   // |bf16 row . fp32 query - true cosine|
    const int R1 = min(N, STAGE1);
    estimate_range_bf16(keys, tk, 0, R1, scan16, (size_t)p.pitch16, qrow, p.K);
    __syncthreads();
    if (tid < R1) {                                                           // k-th largest of the first R1 estimates by rank counting
        const uint64_t me = tk[tid];
        int rank = 0;
        for (int j = 0; j < R1; ++j) rank += tk[j] > me ? 1 : 0;
        if (rank == p.k - 1) s_t1 = me;
    }
    __syncthreads();
    const float eps8 = scan_eps(p.q_resid8[q], __uint_as_float(*p.db_resid8_max), p.K);
    const double unit = (double)sa.unit0 * (double)p.sqi[q];
    int R2 = N;                                                               // fewer than k rows in stage 1: everything goes on
    if (R1 >= p.k && N > R1) {
        const float t1 = key_score(s_t1) - eps16r;                            // k rows have a true cosine >= t1
        // rows with score_int < thr2 have an estimated cosine below t1 - eps8, a true cosine below t1
        const double v = floor(((double)t1 - (double)eps8) / unit) - 1.0;
        const int thr2 = v < -2.0e9 ? -0x7fffffff : v > 2.0e9 ? 0x7fffffff : (int)v;
        // keys are sorted by score: the cut is the first position whose score is below thr2
        for (int i = tid; i < N; i += blockDim.x) {
            const bool below = key_score_i32(keys[i]) < thr2;
            const bool prev_below = i > 0 && key_score_i32(keys[i - 1]) < thr2;
            if (below && !prev_below) s_cut = i + 1;                          // + 1: 0 means "no position is below"
        }
        __syncthreads();
        R2 = s_cut > 0 ? s_cut - 1 : N;
        if (R2 < R1) R2 = R1;
    }
    if (R2 > RS_CAP) { R2 = RS_CAP; overflow = true; }
    // ---- stage 2: bf16 estimates of the rest of the prefix, all estimates in order
    estimate_range_bf16(keys, tk, R1, R2, scan16, (size_t)p.pitch16, qrow, p.K);
    __syncthreads();
    int rp2 = 1;
    while (rp2 < R2) rp2 <<= 1;
    for (int i = R2 + tid; i < rp2; i += blockDim.x) tk[i] = 0ull;
    bitonic_desc(tk, rp2);
    // ---- stage 3: fp32 re-score of the rows whose estimate reaches (k-th best estimate) - 2 eps16 (a prefix of tk); `keys`
    // is free by now and takes the (cosine, row) keys
    if (tid == 0) s_cut = 0;
    __syncthreads();
    int R3 = R2;
    if (R2 > p.k) {
        const uint64_t cut_key = make_key(key_score(tk[p.k - 1]) - 2.0f * eps16r, 0xFFFFFFFFu);   // (lowest key of that score)
        for (int i = tid; i < R2; i += blockDim.x) {
            const bool below = tk[i] < cut_key;
            const bool prev_below = i > 0 && tk[i - 1] < cut_key;
            if (below && !prev_below) s_cut = i + 1;
        }
        __syncthreads();
        R3 = s_cut > 0 ? s_cut - 1 : R2;
        if (R3 < p.k) R3 = p.k;
    }
    uint64_t* fk = keys;
    rescore_range(tk, fk, 0, R3, p.master, qrow, p.K);
    __syncthreads();
    int fp2 = 1;
    while (fp2 < R3) fp2 <<= 1;
    for (int i = R3 + tid; i < fp2; i += blockDim.x) fk[i] = 0ull;
    bitonic_desc(fk, fp2);
    const int m = min(R3, p.k);
    float* cos_out = p.cos_out + (size_t)q * p.k;
    int64_t* id_out = p.id_out + (size_t)q * p.k;
    // Fewer than k rows collected (the threshold sat inside a band of near-identical rows and the int8 estimates fell
    // on the wrong side of it): the k best rows of the SAMPLE -- real rows with true cosines -- stand in until the bf16
    // pass, which this query now takes, overwrites them.
    const bool from_sample = m < p.k;
    const float* cs = p.sample_cos + (size_t)q * p.sample_m;
    const int64_t* is = p.sample_ids + (size_t)q * p.sample_m;
    for (int i = tid; i < p.k; i += blockDim.x) {
        if (from_sample) {
            cos_out[i] = cs[i];
            id_out[i] = is[i] >= 0 ? is[i] + p.id_base : -1;
        } else {
            cos_out[i] = key_score(fk[i]);
            id_out[i] = (int64_t)key_row(fk[i]) + p.id_base;
        }
    }
    if (tid == 0) {
        const float kth = m >= p.k ? key_score(fk[p.k - 1]) : -INFINITY;
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

}  // namespace sqe
