// normalize.hip -- S1: row L2-normalisation x / (||x|| + 1e-9) in fp32, with an optional
// bf16 copy.  Restates main.py:315-316 (index side) and main.py:353-354 (query side).
//
// Also measures what the bf16 copy loses: resid = || x_hat - bf16(x_hat) ||_2 per row.  The index
// keeps the maximum over its rows, a query batch keeps it per query; together they give the
// deterministic error bound of a bf16 scan score that the search certificate uses (select.hip).
//
// Roofline: pure HBM stream.  Algorithmic bytes per row = dim * (4 read + 4 fp32 write +
// 2 bf16 write).  One wave per row, float4 per lane per step (1 KiB per wave-instruction).
#include "kernels.h"

namespace sqe {

template <bool SCATTER>
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ x,
                                                             const int64_t* __restrict__ rows,
                                                             int64_t n, int dim,
                                                             float* __restrict__ out_f32,
                                                             bf16_t* __restrict__ out_bf16, int bf16_pitch,
                                                             float* __restrict__ resid_rows,
                                                             uint32_t* __restrict__ resid_max, int restore) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float4* src = reinterpret_cast<const float4*>(x + row * dim);
    const int nvec = dim >> 2;
    float ss = 0.f;
    for (int i = lane; i < nvec; i += 64) {
        float4 v = src[i];
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    ss = wave_sum(ss);
    // np.linalg.norm -> sqrt(sum of squares); then e / (norm + 1e-9) as a true division
    // restore: rows are stored normalised rows read back from a saved index; they pass through bit
    // for bit (a division by 1.0f is exact) and only the bf16 copy and the residual are rebuilt
    const float den = restore ? 1.0f : sqrtf(ss) + 1e-9f;
    const int64_t orow = SCATTER ? rows[row] : row;
    float4* dst = out_f32 ? reinterpret_cast<float4*>(out_f32 + orow * dim) : nullptr;
    uint2* dstb = out_bf16 ? reinterpret_cast<uint2*>(out_bf16 + orow * bf16_pitch) : nullptr;
    float rs = 0.f;
    for (int i = lane; i < nvec; i += 64) {
        float4 v = src[i];   // second read is an L1/L2 hit (row = 4 KiB at dim 1024)
        v.x /= den; v.y /= den; v.z /= den; v.w /= den;
        if (dst) dst[i] = v;
        const bf16_t b0 = f32_to_bf16(v.x), b1 = f32_to_bf16(v.y), b2 = f32_to_bf16(v.z), b3 = f32_to_bf16(v.w);
        if (dstb) {
            uint2 p;
            p.x = (uint32_t)b0 | ((uint32_t)b1 << 16);
            p.y = (uint32_t)b2 | ((uint32_t)b3 << 16);
            dstb[i] = p;
        }
        const float e0 = v.x - __uint_as_float((uint32_t)b0 << 16), e1 = v.y - __uint_as_float((uint32_t)b1 << 16);
        const float e2 = v.z - __uint_as_float((uint32_t)b2 << 16), e3 = v.w - __uint_as_float((uint32_t)b3 << 16);
        rs += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
    }
    if (resid_rows || resid_max) {
        rs = wave_sum(rs);
        // round up a little: the bound must not be under-estimated by this sum's own rounding
        const float resid = sqrtf(rs) * 1.0001f;
        if (lane == 0) {
            if (resid_rows) resid_rows[orow] = resid;
            if (resid_max && resid == resid) atomicMax(resid_max, __float_as_uint(resid));   // >= 0: uint order = float order
        }
    }
}

int launch_normalize_rows(const float* x, int64_t n, int dim, float* out_f32, bf16_t* out_bf16,
                          int bf16_pitch, float* resid_rows, uint32_t* resid_max, hipStream_t stream) {
    if (n <= 0) return SQE_OK;
    if (dim % 4 != 0 || bf16_pitch % 4 != 0 || bf16_pitch < dim)
        return fail(SQE_ERR_INVALID, "normalize: dim and pitch must be multiples of 4, pitch >= dim");
    const int64_t blocks = (n + 3) / 4;
    if (blocks > 0x7fffffffLL) return fail(SQE_ERR_INVALID, "normalize: too many rows for one launch");
    hipLaunchKernelGGL(normalize_rows_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream,
                       x, (const int64_t*)nullptr, n, dim, out_f32, out_bf16, bf16_pitch, resid_rows, resid_max, 0);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

int launch_restore_rows(const float* x, int64_t n, int dim, float* out_f32, bf16_t* out_bf16, int bf16_pitch,
                        uint32_t* resid_max, hipStream_t stream) {
    if (n <= 0) return SQE_OK;
    if (dim % 4 != 0 || bf16_pitch % 4 != 0 || bf16_pitch < dim)
        return fail(SQE_ERR_INVALID, "restore: dim and pitch must be multiples of 4, pitch >= dim");
    const int64_t blocks = (n + 3) / 4;
    if (blocks > 0x7fffffffLL) return fail(SQE_ERR_INVALID, "restore: too many rows for one launch");
    hipLaunchKernelGGL(normalize_rows_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream,
                       x, (const int64_t*)nullptr, n, dim, out_f32, out_bf16, bf16_pitch, (float*)nullptr, resid_max, 1);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

int launch_normalize_rows_scatter(const float* x, const int64_t* rows, int64_t n, int dim,
                                  float* out_f32, bf16_t* out_bf16, int bf16_pitch, uint32_t* resid_max,
                                  hipStream_t stream) {
    if (n <= 0) return SQE_OK;
    if (dim % 4 != 0) return fail(SQE_ERR_INVALID, "normalize: dim must be a multiple of 4");
    const int64_t blocks = (n + 3) / 4;
    hipLaunchKernelGGL(normalize_rows_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream,
                       x, rows, n, dim, out_f32, out_bf16, bf16_pitch, (float*)nullptr, resid_max, 0);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

}  // namespace sqe
