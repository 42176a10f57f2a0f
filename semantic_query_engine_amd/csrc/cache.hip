// cache.hip -- S8: the cosine scan of lfu_cache_get (main.py:73-87) over a small cache
// matrix of raw (un-normalised) embeddings.
//
//   sim_i = dot(q, m_i) / (||q|| * ||m_i||), 0.0 if either norm is 0   (main.py:59-64, fp32)
//   best  = first strict maximum from (-1.0, -1); NaN never wins        (main.py:74-87)
//
// <= 1000 x 1024 fp32 = 4 MB: launch-latency bound; reported as latency, not roofline.
#include "kernels.h"

namespace sqe {

namespace {

__global__ __launch_bounds__(256) void cosine_rows_kernel(const float* __restrict__ mat,
                                                          const int32_t* __restrict__ order,
                                                          int m, int dim,
                                                          const float* __restrict__ q,
                                                          float* __restrict__ sims) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= m) return;
    const int slot = order ? order[i] : i;
    const float4* a = reinterpret_cast<const float4*>(q);
    const float4* b = reinterpret_cast<const float4*>(mat + (size_t)slot * dim);
    float dot = 0.f, na = 0.f, nb = 0.f;
    for (int v = lane; v < (dim >> 2); v += 64) {
        const float4 x = a[v], y = b[v];
        dot += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
        na += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
        nb += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
    }
    dot = wave_sum(dot); na = wave_sum(na); nb = wave_sum(nb);
    if (lane == 0) {
        const float norm_a = sqrtf(na), norm_b = sqrtf(nb);
        sims[i] = (norm_a == 0.0f || norm_b == 0.0f) ? 0.0f : dot / (norm_a * norm_b);
    }
}

// first strict maximum == largest value, lowest index among equals; only values > -1.0 qualify
__global__ __launch_bounds__(256) void first_max_kernel(const float* __restrict__ sims, int m,
                                                        float* __restrict__ best_sim,
                                                        int32_t* __restrict__ best_idx) {
    __shared__ float s_val[256];
    __shared__ int s_idx[256];
    const int tid = threadIdx.x;
    float bv = -1.0f;
    int bi = -1;
    for (int i = tid; i < m; i += 256) {
        const float v = sims[i];
        if (v > bv) { bv = v; bi = i; }      // ascending i per thread: first max kept
    }
    s_val[tid] = bv; s_idx[tid] = bi;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) {
            const float ov = s_val[tid + off];
            const int oi = s_idx[tid + off];
            const bool take = oi >= 0 && (s_idx[tid] < 0 || ov > s_val[tid] || (ov == s_val[tid] && oi < s_idx[tid]));
            if (take) { s_val[tid] = ov; s_idx[tid] = oi; }
        }
        __syncthreads();
    }
    if (tid == 0) { *best_sim = s_val[0]; *best_idx = s_idx[0]; }
}

}  // namespace

int launch_cosine_scan(const float* mat, const int32_t* order, int m, int dim, const float* q,
                       float* sims, float* best_sim, int32_t* best_idx, hipStream_t stream) {
    if (dim % 4 != 0) return fail(SQE_ERR_INVALID, "cosine scan: dim must be a multiple of 4");
    if (m > 0) {
        hipLaunchKernelGGL(cosine_rows_kernel, dim3((m + 3) / 4), dim3(256), 0, stream, mat, order, m, dim, q, sims);
        SQE_HIP(hipGetLastError());
    }
    if (best_sim && best_idx) {
        hipLaunchKernelGGL(first_max_kernel, dim3(1), dim3(256), 0, stream, sims, m, best_sim, best_idx);
        SQE_HIP(hipGetLastError());
    }
    return SQE_OK;
}

}  // namespace sqe
