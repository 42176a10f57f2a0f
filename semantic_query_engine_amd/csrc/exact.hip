// exact.hip -- certified fallback, second half: fp32 re-scoring of the rows the collect pass
// gathered for a query whose certificate failed (select.hip, scan.hip COLLECT mode).
//
// Every row that can belong to the exact top-k has a bf16 scan score >= (t - eps) and was appended
// to the query's buffer; here each collected row is re-scored in fp32 against the master, the k
// best keys (cosine desc, row id asc) are selected and written over the uncertified result.
// One workgroup per query; it returns at once for certified queries (collect_thr == +inf).
#include "kernels.h"

namespace sqe {

namespace {

struct RescoreArgs {
    const float* master;
    const float* qn;
    int K, B, k;
    const int* unc_ids;       // compact index -> query
    const int* unc_count;
    const float* collect_thr;
    uint64_t* keys;
    const int* key_cnt;
    float* cos_out;
    int64_t* id_out;
    int64_t id_base;
};

__global__ __launch_bounds__(256) void collect_rescore_kernel(RescoreArgs p) {
    __shared__ int hist[256];
    __shared__ int scratch[4];
    __shared__ uint64_t top[MAX_KP];
    const int ci = blockIdx.x;                            // index in the compacted list of uncertified queries
    if (ci >= *p.unc_count) return;
    const int q = p.unc_ids[ci];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // More rows inside the error band than the buffer holds (thousands of near-identical vectors):
    // the collected set is incomplete, so the first-pass result stands -- it is still ordered by the
    // scan's own (score, lowest id) keys, which is the right answer when those rows are duplicates.
    if (p.key_cnt[ci] > EXACT_CAP) return;
    const int n = p.key_cnt[ci];
    uint64_t* keys = p.keys + (size_t)ci * EXACT_CAP;
    // fp32 re-score, one wave per collected row
    const float4* qv = reinterpret_cast<const float4*>(p.qn + (size_t)q * p.K);
    const int nvec = p.K >> 2;
    for (int e = wave; e < n; e += 4) {
        const uint32_t row = key_row(keys[e]);
        const float4* rv = reinterpret_cast<const float4*>(p.master + (size_t)row * p.K);
        float s = 0.f;
        for (int v = lane; v < nvec; v += 64) {
            const float4 a = rv[v], b = qv[v];
            s = fmaf(a.x, b.x, s); s = fmaf(a.y, b.y, s); s = fmaf(a.z, b.z, s); s = fmaf(a.w, b.w, s);
        }
        s = wave_sum(s) + 0.0f;
        if (lane == 0) keys[e] = make_key(s, row);
    }
    __syncthreads();
    // k-th largest key by MSB-first byte-wise radix select (keys are unique)
    uint64_t prefix = 0;
    int remaining = p.k;
    const bool all = n <= p.k;
    for (int byte = 7; byte >= 0 && !all; --byte) {
        hist[tid] = 0;
        __syncthreads();
        const int shift = byte * 8;
        for (int e = tid; e < n; e += 256) {
            const uint64_t key = keys[e];
            if (byte == 7 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(int)((key >> shift) & 0xff)], 1);
        }
        __syncthreads();
        {
            int hb, hr;
            hist_locate(hist, remaining, hb, hr);
            if (tid == 0) { scratch[0] = hb < 0 ? 0 : hb; scratch[1] = hr; }
        }
        __syncthreads();
        prefix |= ((uint64_t)scratch[0] << shift);
        remaining = scratch[1];
        __syncthreads();
    }
    const uint64_t T = all ? 0ull : prefix;
    if (tid == 0) scratch[2] = 0;
    __syncthreads();
    for (int e = tid; e < n; e += 256) {
        const uint64_t key = keys[e];
        if (key >= T) {
            const int slot = atomicAdd(&scratch[2], 1);
            if (slot < MAX_KP) top[slot] = key;
        }
    }
    __syncthreads();
    const int m = min(scratch[2], p.k);
    float* cos_out = p.cos_out + (size_t)q * p.k;
    int64_t* id_out = p.id_out + (size_t)q * p.k;
    for (int i = tid; i < m; i += 256) {
        const uint64_t ki = top[i];
        int rank = 0;
        for (int j = 0; j < m; ++j) rank += top[j] > ki ? 1 : 0;
        cos_out[rank] = key_score(ki);
        id_out[rank] = (int64_t)key_row(ki) + p.id_base;
    }
    for (int i = m + tid; i < p.k; i += 256) {
        cos_out[i] = -INFINITY;
        id_out[i] = -1;
    }
}

// Compaction of the uncertified queries (collect_thr != +inf) into a dense batch, in query order: their ids,
// thresholds and bf16 rows.  The collect scan then costs what a batch of that size costs (a handful of
// failures -> one HBM-bound pass of the 64-query kernel) instead of a second full-batch scan.
// One workgroup; thr_out is padded with +inf up to `thr_cap` entries.
__global__ __launch_bounds__(1024) void compact_uncertified_kernel(const float* __restrict__ collect_thr, int B,
                                                                   const bf16_t* __restrict__ qb, int pitch_bytes, int row_bytes,
                                                                   int* __restrict__ unc_ids, float* __restrict__ thr_out,
                                                                   int thr_cap, bf16_t* __restrict__ qb_out,
                                                                   int* __restrict__ unc_count) {
    __shared__ int wave_tot[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int q0 = 0; q0 < B; q0 += 1024) {
        const int q = q0 + tid;
        const bool unc = q < B && collect_thr[q] != INFINITY;
        const uint64_t m = __ballot(unc);
        if (lane == 0) wave_tot[wave] = __popcll(m);
        __syncthreads();
        int before = base;
        for (int w = 0; w < wave; ++w) before += wave_tot[w];
        if (unc) {
            const int ci = before + __popcll(m & ((1ull << lane) - 1ull));
            unc_ids[ci] = q;
            thr_out[ci] = collect_thr[q];
        }
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < 16; ++w) t += wave_tot[w];
            base += t;
        }
        __syncthreads();
    }
    const int count = base;
    if (tid == 0) *unc_count = count;
    for (int i = count + tid; i < thr_cap; i += 1024) thr_out[i] = INFINITY;
    __syncthreads();
    // gather the bf16 query rows (16 bytes per thread-step)
    const int vec_per_row = row_bytes / 16;
    for (int64_t e = tid; e < (int64_t)count * vec_per_row; e += 1024) {
        const int ci = (int)(e / vec_per_row), v = (int)(e - (int64_t)ci * vec_per_row);
        const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(qb) + (size_t)unc_ids[ci] * pitch_bytes);
        uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<char*>(qb_out) + (size_t)ci * pitch_bytes);
        dst[v] = src[v];
    }
}

}  // namespace

int launch_compact_uncertified(const float* collect_thr, int B, const bf16_t* qb, int pitch_bytes, int row_bytes, int* unc_ids,
                               float* thr_out, int thr_cap, bf16_t* qb_out, int* unc_count, hipStream_t stream) {
    if (B <= 0) return SQE_OK;
    hipLaunchKernelGGL(compact_uncertified_kernel, dim3(1), dim3(1024), 0, stream, collect_thr, B, qb, pitch_bytes, row_bytes,
                       unc_ids, thr_out, thr_cap, qb_out, unc_count);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

int launch_collect_rescore(const ExactArgs& a, hipStream_t stream) {
    if (a.B <= 0) return SQE_OK;
    if (a.K % 4 != 0) return fail(SQE_ERR_INVALID, "collect rescore: dim must be a multiple of 4");
    RescoreArgs p;
    p.master = a.master; p.qn = a.qn; p.K = a.K; p.B = a.B; p.k = a.k; p.collect_thr = a.collect_thr;
    p.unc_ids = a.unc_ids; p.unc_count = a.unc_count;
    p.keys = a.keys; p.key_cnt = a.key_cnt; p.cos_out = a.cos_out; p.id_out = a.id_out; p.id_base = a.id_base;
    hipLaunchKernelGGL(collect_rescore_kernel, dim3(a.B), dim3(256), 0, stream, p);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

}  // namespace sqe
