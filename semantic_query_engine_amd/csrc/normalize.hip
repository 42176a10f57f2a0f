// normalize.hip -- S1: row L2-normalisation x / (||x|| + 1e-9) in fp32, with an optional
// bf16 copy.  Restates main.py:315-316 (index side) and main.py:353-354 (query side).
//
// Also measures what the bf16 copy loses: resid = || x_hat - bf16(x_hat) ||_2 per row.  The index
// keeps the maximum over its rows, a query batch keeps it per query; together they give the
// deterministic error bound of a bf16 scan score that the search certificate uses (select.hip).
//
// Roofline: pure HBM stream.  Algorithmic bytes per row = dim * (4 read + 4 fp32 write +
// 2 bf16 write).  One wave per row, float4 per lane per step (1 KiB per wave-instruction).
#include <algorithm>

#include "kernels.h"

namespace sqe {

// Row arithmetic shared by both kernel forms: v / den, fp32 + bf16 stores, squared bf16 rounding error.
__device__ __forceinline__ float emit_vec(float4 v, float den, float4* dst, uint2* dstb, int i) {
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
    return e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
}

// Persistent grid: wave w of W takes rows w, w + W, ...  NV > 0: the row (dim <= NV * 256) is read ONCE
// into registers, all NV loads issued before the first use (NV KiB in flight per wave, 8 waves per SIMD);
// NV == 0: any dim, second read from L1/L2.  The index-wide residual maximum is kept per wave and
// published once at the end, behind a read of the current value: a launch makes at most one atomic per
// wave (r01 made one per ROW to a single address, which ran the 1 M-row add at 0.88 TB/s).
template <int NV, bool SCATTER>
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ x,
                                                             const int64_t* __restrict__ rows,
                                                             int64_t n, int dim, int64_t x_stride,
                                                             float* __restrict__ out_f32,
                                                             bf16_t* __restrict__ out_bf16, int bf16_pitch,
                                                             float* __restrict__ resid_rows,
                                                             uint32_t* __restrict__ resid_max, int restore) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int nvec = dim >> 2;
    float wave_resid = 0.f;
    for (int64_t row = wave0; row < n; row += nwaves) {
        const float4* src = reinterpret_cast<const float4*>(x + row * x_stride);
        float4 v[NV > 0 ? NV : 1];
        float ss = 0.f;
        if constexpr (NV > 0) {
#pragma unroll
            for (int t = 0; t < NV; ++t) {
                const int i = lane + 64 * t;
                v[t] = float4{0.f, 0.f, 0.f, 0.f};
                if (i < nvec) {   // read once: streaming load
                    const f32x4 u = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src) + i);
                    v[t] = float4{u[0], u[1], u[2], u[3]};
                }
            }
#pragma unroll
            for (int t = 0; t < NV; ++t) ss += v[t].x * v[t].x + v[t].y * v[t].y + v[t].z * v[t].z + v[t].w * v[t].w;
        } else {
            for (int i = lane; i < nvec; i += 64) {
                const float4 u = src[i];
                ss += u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w;
            }
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
        if constexpr (NV > 0) {
#pragma unroll
            for (int t = 0; t < NV; ++t) {
                const int i = lane + 64 * t;
                if (i < nvec) rs += emit_vec(v[t], den, dst, dstb, i);
            }
        } else {
            for (int i = lane; i < nvec; i += 64) rs += emit_vec(src[i], den, dst, dstb, i);
        }
        if (resid_rows || resid_max) {
            rs = wave_sum(rs);
            // round up a little: the bound must not be under-estimated by this sum's own rounding
            const float resid = sqrtf(rs) * 1.0001f;
            if (resid_rows && lane == 0) resid_rows[orow] = resid;
            if (resid == resid) wave_resid = fmaxf(wave_resid, resid);
        }
    }
    if (resid_max && lane == 0 && wave_resid > 0.f) {
        const uint32_t bits = __float_as_uint(wave_resid);             // >= 0: uint order = float order
        if (bits > __hip_atomic_load(resid_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(resid_max, bits);
    }
}

namespace {
template <bool SCATTER>
int launch_rows(const float* x, const int64_t* rows, int64_t n, int dim, int64_t x_stride, float* out_f32, bf16_t* out_bf16, int bf16_pitch,
                float* resid_rows, uint32_t* resid_max, int restore, hipStream_t stream) {
    // 8 blocks of 4 waves per CU (32 waves per CU: full occupancy at <= 64 VGPRs), rows dealt round-robin
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int64_t blocks = std::min<int64_t>((n + 3) / 4, (int64_t)cus * 8);
    const dim3 grid((unsigned)blocks), block(256);
    const int nvec = dim >> 2;
#define SQE_NORM_LAUNCH(NVV)                                                                                   \
    hipLaunchKernelGGL((normalize_rows_kernel<NVV, SCATTER>), grid, block, 0, stream, x, rows, n, dim, x_stride, out_f32, \
                       out_bf16, bf16_pitch, resid_rows, resid_max, restore)
    if (nvec <= 64) SQE_NORM_LAUNCH(1);
    else if (nvec <= 128) SQE_NORM_LAUNCH(2);
    else if (nvec <= 256) SQE_NORM_LAUNCH(4);
    else if (nvec <= 512) SQE_NORM_LAUNCH(8);
    else SQE_NORM_LAUNCH(0);
#undef SQE_NORM_LAUNCH
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}
}  // namespace

int launch_normalize_rows(const float* x, int64_t n, int dim, int64_t x_stride, float* out_f32, bf16_t* out_bf16,
                          int bf16_pitch, float* resid_rows, uint32_t* resid_max, hipStream_t stream) {
    if (n <= 0) return SQE_OK;
    if (dim % 4 != 0 || bf16_pitch % 4 != 0 || bf16_pitch < dim)
        return fail(SQE_ERR_INVALID, "normalize: dim and pitch must be multiples of 4, pitch >= dim");
    if (x_stride < dim || x_stride % 4 != 0) return fail(SQE_ERR_INVALID, "normalize: input row stride must be a multiple of 4, >= dim");
    return launch_rows<false>(x, nullptr, n, dim, x_stride, out_f32, out_bf16, bf16_pitch, resid_rows, resid_max, 0, stream);
}

int launch_restore_rows(const float* x, int64_t n, int dim, int64_t x_stride, float* out_f32, bf16_t* out_bf16, int bf16_pitch,
                        uint32_t* resid_max, hipStream_t stream) {
    if (n <= 0) return SQE_OK;
    if (dim % 4 != 0 || bf16_pitch % 4 != 0 || bf16_pitch < dim)
        return fail(SQE_ERR_INVALID, "restore: dim and pitch must be multiples of 4, pitch >= dim");
    if (x_stride < dim || x_stride % 4 != 0) return fail(SQE_ERR_INVALID, "restore: input row stride must be a multiple of 4, >= dim");
    return launch_rows<false>(x, nullptr, n, dim, x_stride, out_f32, out_bf16, bf16_pitch, nullptr, resid_max, 1, stream);
}

int launch_normalize_rows_scatter(const float* x, const int64_t* rows, int64_t n, int dim,
                                  float* out_f32, bf16_t* out_bf16, int bf16_pitch, uint32_t* resid_max,
                                  hipStream_t stream) {
    if (n <= 0) return SQE_OK;
    if (dim % 4 != 0) return fail(SQE_ERR_INVALID, "normalize: dim must be a multiple of 4");
    if (bf16_pitch % 4 != 0 || bf16_pitch < dim) return fail(SQE_ERR_INVALID, "normalize: pitch must be a multiple of 4, >= dim");
    return launch_rows<true>(x, rows, n, dim, dim, out_f32, out_bf16, bf16_pitch, nullptr, resid_max, 0, stream);
}

}  // namespace sqe
