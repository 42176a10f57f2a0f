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
    const int q = blockIdx.x;
    if (p.collect_thr[q] == INFINITY) return;            // certified: result already final
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // More rows inside the error band than the buffer holds (thousands of near-identical vectors):
    // the collected set is incomplete, so the first-pass result stands -- it is still ordered by the
    // scan's own (score, lowest id) keys, which is the right answer when those rows are duplicates.
    if (p.key_cnt[q] > EXACT_CAP) return;
    const int n = p.key_cnt[q];
    uint64_t* keys = p.keys + (size_t)q * EXACT_CAP;
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
        if (tid == 0) {
            int cum = 0, bin = 255;
            for (; bin >= 0; --bin) {
                if (cum + hist[bin] >= remaining) break;
                cum += hist[bin];
            }
            scratch[0] = bin < 0 ? 0 : bin;
            scratch[1] = bin < 0 ? remaining : remaining - cum;
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

}  // namespace

int launch_collect_rescore(const ExactArgs& a, hipStream_t stream) {
    if (a.B <= 0) return SQE_OK;
    if (a.K % 4 != 0) return fail(SQE_ERR_INVALID, "collect rescore: dim must be a multiple of 4");
    RescoreArgs p;
    p.master = a.master; p.qn = a.qn; p.K = a.K; p.B = a.B; p.k = a.k; p.collect_thr = a.collect_thr;
    p.keys = a.keys; p.key_cnt = a.key_cnt; p.cos_out = a.cos_out; p.id_out = a.id_out; p.id_base = a.id_base;
    hipLaunchKernelGGL(collect_rescore_kernel, dim3(a.B), dim3(256), 0, stream, p);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

}  // namespace sqe
