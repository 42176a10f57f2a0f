// exact.hip -- exact fp32 rescan for queries the bf16 scan could not certify (select.hip).
//
// For a queued query the re-scored candidates give a LOWER BOUND t of its true k-th best cosine
// (k real rows reach it).  The exact top-k is therefore among the rows whose fp32 cosine is >= t:
// one streaming pass over the fp32 master computes every row's cosine against up to 8 queued
// queries at a time and appends the rows that reach the bound; the few collected keys are then
// ordered exactly (cosine desc, row id asc).  HBM-bound (N * D * 4 bytes per group of 8 queries);
// runs only when the certificate fails -- the kernel exits at once when nothing is queued.
#include "kernels.h"

namespace sqe {

namespace {

constexpr int QG_MAX = 8;                 // queries per pass
constexpr int QLDS_FLOATS = 8192;         // 32 KiB of LDS for the query group

struct ExactKernelArgs {
    const float* master;
    const float* qn;
    int64_t n_rows;
    int K, B, k;
    const int* unc_count;
    const int* unc_list;
    const float* unc_thr;
    uint64_t* keys;
    int* key_cnt;
    float* cos_out;
    int64_t* id_out;
    int64_t id_base;
};

__global__ __launch_bounds__(256) void exact_rescan_kernel(ExactKernelArgs p) {
    __shared__ __attribute__((aligned(16))) float sq[QLDS_FLOATS];
    __shared__ float sthr[QG_MAX];
    const int n_unc = min(*p.unc_count, p.B);
    if (n_unc <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qg = min(QG_MAX, QLDS_FLOATS / p.K);             // queries per pass (>= 1: K <= 8192)
    const int nvec = p.K >> 2;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + wave;
    const int64_t wave_stride = (int64_t)gridDim.x * 4;
    for (int g0 = 0; g0 < n_unc; g0 += qg) {
        const int gq = min(qg, n_unc - g0);
        __syncthreads();
        for (int i = tid; i < gq * p.K; i += 256) {
            const int j = i / p.K, d = i - j * p.K;
            sq[i] = p.qn[(size_t)p.unc_list[g0 + j] * p.K + d];
        }
        if (tid < gq) sthr[tid] = p.unc_thr[g0 + tid];
        __syncthreads();
        for (int64_t row = wave_id; row < p.n_rows; row += wave_stride) {
            const float4* rv = reinterpret_cast<const float4*>(p.master + (size_t)row * p.K);
            float acc[QG_MAX];
#pragma unroll
            for (int j = 0; j < QG_MAX; ++j) acc[j] = 0.f;
            for (int v = lane; v < nvec; v += 64) {
                const float4 a = rv[v];
#pragma unroll
                for (int j = 0; j < QG_MAX; ++j) {
                    if (j < gq) {
                        const float4 b = *reinterpret_cast<const float4*>(&sq[j * p.K + v * 4]);
                        acc[j] = fmaf(a.x, b.x, acc[j]); acc[j] = fmaf(a.y, b.y, acc[j]);
                        acc[j] = fmaf(a.z, b.z, acc[j]); acc[j] = fmaf(a.w, b.w, acc[j]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < QG_MAX; ++j) {
                if (j < gq) {
                    const float s = wave_sum(acc[j]) + 0.0f;
                    if (lane == 0 && s >= sthr[j]) {
                        const int slot = atomicAdd(&p.key_cnt[g0 + j], 1);
                        if (slot < EXACT_CAP) p.keys[(size_t)(g0 + j) * EXACT_CAP + slot] = make_key(s, (uint32_t)row);
                    }
                }
            }
        }
    }
}

// one workgroup per queue position: exact order of the collected keys, top-k written over the
// uncertified result of that query
__global__ __launch_bounds__(256) void exact_finalize_kernel(ExactKernelArgs p) {
    __shared__ int hist[256];
    __shared__ int scratch[4];
    __shared__ uint64_t top[MAX_KP];
    const int pos = blockIdx.x;
    const int n_unc = min(*p.unc_count, p.B);
    if (pos >= n_unc) return;
    const int tid = threadIdx.x;
    const int q = p.unc_list[pos];
    const int n = min(p.key_cnt[pos], EXACT_CAP);
    const uint64_t* keys = p.keys + (size_t)pos * EXACT_CAP;
    // k-th largest key by MSB-first byte-wise radix select (keys are unique)
    uint64_t prefix = 0;
    int remaining = p.k;
    bool all = n <= p.k;
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
            scratch[0] = bin;
            scratch[1] = remaining - cum;
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

int launch_exact_rescan(const ExactArgs& a, int cu_count, hipStream_t stream) {
    if (a.B <= 0 || a.n_rows <= 0) return SQE_OK;
    if (a.K % 4 != 0 || a.K > QLDS_FLOATS) return fail(SQE_ERR_INVALID, "exact rescan: dim must be a multiple of 4, <= 8192");
    ExactKernelArgs p;
    p.master = a.master; p.qn = a.qn; p.n_rows = a.n_rows; p.K = a.K; p.B = a.B; p.k = a.k;
    p.unc_count = a.unc_count; p.unc_list = a.unc_list; p.unc_thr = a.unc_thr;
    p.keys = a.keys; p.key_cnt = a.key_cnt; p.cos_out = a.cos_out; p.id_out = a.id_out; p.id_base = a.id_base;
    int grid = cu_count * 4;
    const int64_t need = (a.n_rows + 3) / 4;
    if (grid > need) grid = (int)need;
    hipLaunchKernelGGL(exact_rescan_kernel, dim3(grid), dim3(256), 0, stream, p);
    SQE_HIP(hipGetLastError());
    hipLaunchKernelGGL(exact_finalize_kernel, dim3(a.B), dim3(256), 0, stream, p);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

}  // namespace sqe
