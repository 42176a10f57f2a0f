// quant.hip -- int8 copies for the int8 first-pass scan (scan_i8.hip): symmetric per-row quantisation of the
// L2-normalised fp32 rows, and the per-query integer thresholds of that scan.
//
//   x_hat  ~=  (sxi * S0) * x8          x8 in [-127, 127]^dim,  sxi an integer in [1, 65535]: per QUERY row, and per 256-row
//                                       TILE of the index (the largest scale its rows need: see below),
//   S0 = 4 / (127 * 160 * sqrt(dim))    the index-wide scale unit (a typical row -- largest element ~4 / sqrt(dim) --
//                                       gets sxi ~ 160: the integer grid of the row scales costs < 1 % of resolution)
// A row needs the smallest multiple of S0 that (a) maps its largest element to <= 127 and (b) keeps
// ||x8|| <= 2800 + rounding (< 2896 = 2^11.5), so that every dot product of two quantised rows is below 2^23
// (v_mul_i32_i24 in the scan).
// The rounding residual || x_hat - sxi S0 x8 ||_2 is measured per row; the index keeps the maximum over its rows,
// a query batch keeps it per query: the inputs of the deterministic error bound (kernels.h: scan_eps) that the int8
// certificate uses exactly as the bf16 one does.
//
// Index rows share ONE scale per 256-row tile, the largest their rows need (r03b).  The certificate's bound uses the index-wide
// MAXIMUM residual, which a tile-wide scale does not change (the worst row's tile carries the worst row's scale, as before),
// and a scale that is uniform over a scan tile moves from the 128 accumulators of a wave to the 4 thresholds of a lane: the
// scan's per-tile scaling pass (128 v_mul_i32_i24 + their max3 per wave, VALU-bound under 32 MFMAs) disappears.
//
// DB rows are written TILED for the scan: tile t = rows 256 t .. 256 t + 255 at t * tile_stride bytes, inside it the
// 64-element K slice h of row r at h * 16 KiB + r * 64 -- every half-step of the ping-pong scan reads one contiguous
// 16 KiB block of whole 128-B lines (r03 experiment SQE_DBG=8192: -0.5 / -2 / -2.4 % at batch 1024 / 512 / 256 against
// 64-B segments at the row pitch).  Query rows are written row-major (they are re-read from L2 for every tile).
#include <algorithm>

#include "kernels.h"

namespace sqe {

namespace {

// one wave per row; dim <= 8192, dim % 64 == 0
template <bool TILED>
__global__ __launch_bounds__(256) void quantize_rows_i8_kernel(const float* __restrict__ x, const int64_t* __restrict__ rows, int64_t first_row,
                                                               int64_t n, int dim, float s0, int8_t* __restrict__ out, int64_t tile_stride,
                                                               int q_pitch, uint32_t* __restrict__ sxi_out, float* __restrict__ resid_rows,
                                                               uint32_t* __restrict__ resid_max, const int* __restrict__ gather = nullptr,
                                                               const int* __restrict__ scatter = nullptr) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int nvec = dim >> 2;
    float wave_resid = 0.f;
    for (int64_t i = wave0; i < n; i += nwaves) {
        const int64_t in_row = rows ? rows[i] : first_row + i;
        // gather / scatter (the IVF copy in list order): input row gather[i], output row scatter[i]
        const int64_t row = scatter ? (int64_t)scatter[in_row] : in_row;
        const float4* src = reinterpret_cast<const float4*>(x + (gather ? (int64_t)gather[in_row] : in_row) * (int64_t)dim);
        float mx = 0.f, ss = 0.f;
        for (int v = lane; v < nvec; v += 64) {
            const float4 u = src[v];
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(u.x), fabsf(u.y)), fmaxf(fabsf(u.z), fabsf(u.w))));
            ss += u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        ss = wave_sum(ss);
        // NaN / inf rows: scale 1, zeros (their scores are 0; the fp32 re-score decides what they are worth)
        const bool finite = mx < 3.0e38f && ss == ss;
        const float need = finite ? fmaxf(mx / 127.0f, sqrtf(ss) / 2800.0f) : 0.f;
        int sxi = (int)ceilf(need / s0 * 1.000001f);
        sxi = min(max(sxi, 1), 65535);
        const float s = (float)sxi * s0;
        float rs = 0.f;
        for (int v = lane; v < nvec; v += 64) {
            const float4 u = finite ? src[v] : float4{0.f, 0.f, 0.f, 0.f};
            const float q0 = fminf(fmaxf(rintf(u.x / s), -127.f), 127.f), q1 = fminf(fmaxf(rintf(u.y / s), -127.f), 127.f);
            const float q2 = fminf(fmaxf(rintf(u.z / s), -127.f), 127.f), q3 = fminf(fmaxf(rintf(u.w / s), -127.f), 127.f);
            const float e0 = u.x - q0 * s, e1 = u.y - q1 * s, e2 = u.z - q2 * s, e3 = u.w - q3 * s;
            rs += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
            const uint32_t packed = ((uint32_t)(int)q0 & 0xffu) | (((uint32_t)(int)q1 & 0xffu) << 8) | (((uint32_t)(int)q2 & 0xffu) << 16) |
                                    (((uint32_t)(int)q3 & 0xffu) << 24);
            int8_t* dst;
            if constexpr (TILED) dst = out + (row >> 8) * tile_stride + (int64_t)(v >> 4) * 16384 + (row & 255) * 64 + (v & 15) * 4;
            else dst = out + row * (int64_t)q_pitch + v * 4;
            *reinterpret_cast<uint32_t*>(dst) = packed;
        }
        rs = wave_sum(rs);
        // rounded up: the bound must not be under-estimated by this sum's own rounding (nor by x / s * s)
        float resid = sqrtf(rs) * 1.0001f + 1.0e-7f;
        if (!finite) resid = 0.f;
        if (lane == 0) {
            sxi_out[row] = (uint32_t)sxi;
            if (resid_rows) resid_rows[row] = resid;
        }
        wave_resid = fmaxf(wave_resid, resid);
    }
    if (resid_max && lane == 0 && wave_resid > 0.f) {
        const uint32_t bits = __float_as_uint(wave_resid);
        if (bits > __hip_atomic_load(resid_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(resid_max, bits);
    }
}

// TILED index copy: one workgroup (4 waves) per 256-row tile.  `tile_list` (optional) names the tiles to (re)quantise as the tiles
// of the listed ROWS (sqe_index_update: duplicates are harmless, they write the same bytes); else tiles first_tile + blockIdx.x.
// Pass 1: the scale every row needs, maximum over the tile; pass 2: quantise every row with it (second read from L2 /
// Infinity Cache), residual maximum.  Rows past n_rows are written as zero vectors.
__global__ __launch_bounds__(256) void quantize_tiles_i8_kernel(const float* __restrict__ x, const int64_t* __restrict__ rows_of_tiles,
                                                                int64_t first_tile, int64_t n_tiles, int64_t n_rows, int dim, float s0,
                                                                int8_t* __restrict__ out, int64_t tile_stride, uint32_t* __restrict__ sxi_out,
                                                                uint32_t* __restrict__ resid_max) {
    __shared__ float s_need[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = dim >> 2;
    float wave_resid = 0.f;
    for (int64_t ti = blockIdx.x; ti < n_tiles; ti += gridDim.x) {
        const int64_t tile = rows_of_tiles ? (rows_of_tiles[ti] >> 8) : first_tile + ti;
        const int64_t row0 = tile * 256;
        // ---- pass 1
        float need = 0.f;
        for (int r = wave; r < 256; r += 4) {
            const int64_t row = row0 + r;
            if (row >= n_rows) break;
            const float4* src = reinterpret_cast<const float4*>(x + row * (int64_t)dim);
            float mx = 0.f, ss = 0.f;
            for (int v = lane; v < nvec; v += 64) {
                const float4 u = src[v];
                mx = fmaxf(mx, fmaxf(fmaxf(fabsf(u.x), fabsf(u.y)), fmaxf(fabsf(u.z), fabsf(u.w))));
                ss += u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
            ss = wave_sum(ss);
            if (mx < 3.0e38f && ss == ss) need = fmaxf(need, fmaxf(mx / 127.0f, sqrtf(ss) / 2800.0f));   // NaN / inf rows: zeros below
        }
        __syncthreads();                       // s_need of the previous tile has been read
        if (lane == 0) s_need[wave] = need;
        __syncthreads();
        need = fmaxf(fmaxf(s_need[0], s_need[1]), fmaxf(s_need[2], s_need[3]));
        int sxi = (int)ceilf(need / s0 * 1.000001f);
        sxi = min(max(sxi, 1), 65535);
        const float s = (float)sxi * s0;
        // ---- pass 2
        for (int r = wave; r < 256; r += 4) {
            const int64_t row = row0 + r;
            const bool live = row < n_rows;
            const float4* src = reinterpret_cast<const float4*>(x + row * (int64_t)dim);
            bool finite = live;
            if (live) {
                float mx = 0.f, ss = 0.f;
                for (int v = lane; v < nvec; v += 64) {
                    const float4 u = src[v];
                    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(u.x), fabsf(u.y)), fmaxf(fabsf(u.z), fabsf(u.w))));
                    ss += u.x + u.y + u.z + u.w;
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
                ss = wave_sum(ss);
                finite = mx < 3.0e38f && ss == ss;
            }
            float rs = 0.f;
            for (int v = lane; v < nvec; v += 64) {
                const float4 u = finite ? src[v] : float4{0.f, 0.f, 0.f, 0.f};
                const float q0 = fminf(fmaxf(rintf(u.x / s), -127.f), 127.f), q1 = fminf(fmaxf(rintf(u.y / s), -127.f), 127.f);
                const float q2 = fminf(fmaxf(rintf(u.z / s), -127.f), 127.f), q3 = fminf(fmaxf(rintf(u.w / s), -127.f), 127.f);
                const float e0 = u.x - q0 * s, e1 = u.y - q1 * s, e2 = u.z - q2 * s, e3 = u.w - q3 * s;
                rs += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
                const uint32_t packed = ((uint32_t)(int)q0 & 0xffu) | (((uint32_t)(int)q1 & 0xffu) << 8) | (((uint32_t)(int)q2 & 0xffu) << 16) |
                                        (((uint32_t)(int)q3 & 0xffu) << 24);
                *reinterpret_cast<uint32_t*>(out + tile * tile_stride + (int64_t)(v >> 4) * 16384 + r * 64 + (v & 15) * 4) = packed;
            }
            rs = wave_sum(rs);
            const float resid = finite ? sqrtf(rs) * 1.0001f + 1.0e-7f : 0.f;
            if (lane == 0) sxi_out[row] = (uint32_t)sxi;
            wave_resid = fmaxf(wave_resid, resid);
        }
    }
    if (resid_max && lane == 0 && wave_resid > 0.f) {
        const uint32_t bits = __float_as_uint(wave_resid);
        if (bits > __hip_atomic_load(resid_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(resid_max, bits);
    }
}

// Per-query integer threshold of the int8 scan from the sample pass: tau[q] = m-th best TRUE cosine of the row sample
// (cos_s[q][m - 1], -inf when the sample held fewer rows).  A row is collected iff acc * sxi_row >= thr_int[q]; a row that is
// not has an estimated score below thr_eff[q] = thr_int[q] * unit(q), unit(q) = S0^2 * sqi[q].
__global__ void i8_thresholds_kernel(const float* __restrict__ cos_s, int m, const uint32_t* __restrict__ sqi, float s0, int B, int b_pad,
                                     int* __restrict__ thr_int, float* __restrict__ thr_eff) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= b_pad) return;
    if (q >= B) { thr_int[q] = 0x7fffffff; thr_eff[q] = INFINITY; return; }        // padding queries collect nothing
    const float tau = cos_s[(size_t)q * m + (m - 1)];
    const double unit = (double)s0 * (double)s0 * (double)sqi[q];
    int t;
    if (!(tau > -INFINITY)) t = -0x7fffffff;                                       // no estimate: collect everything (the lists overflow
    else {                                                                         //   and the query takes the bf16 fallback)
        const double v = ceil((double)tau / unit);
        t = v > 2.0e9 ? 0x7ffffffe : v < -2.0e9 ? -0x7fffffff : (int)v;
    }
    thr_int[q] = t;
    thr_eff[q] = (float)((double)t * unit * (1.0 + 1e-6) + 1e-7);                  // rounded up
}

}  // namespace

float i8_scale_unit(int dim) { return 4.0f / (127.0f * 160.0f * sqrtf((float)dim)); }

int launch_quantize_rows_i8(const float* master, const int64_t* rows, int64_t first_row, int64_t n, int64_t n_rows, int dim, int8_t* out,
                            int64_t tile_stride, uint32_t* sxi, uint32_t* resid_max, hipStream_t stream) {
    if (n <= 0) return SQE_OK;
    if (dim % 64 != 0 || dim > 8192) return fail(SQE_ERR_INVALID, "int8 copy: dim must be a multiple of 64, <= 8192");
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    // whole tiles: the tiles of the listed rows, or every tile that holds a row of [first_row, first_row + n)
    const int64_t first_tile = rows ? 0 : first_row / 256;
    const int64_t n_tiles = rows ? n : (first_row + n + 255) / 256 - first_tile;
    const int64_t blocks = std::min<int64_t>(n_tiles, (int64_t)cus * 8);
    hipLaunchKernelGGL(quantize_tiles_i8_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, master, rows, first_tile, n_tiles, n_rows, dim,
                       i8_scale_unit(dim), out, tile_stride, sxi, resid_max);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

int launch_quantize_queries_i8(const float* qn, int B, int dim, int8_t* out, int q_pitch, uint32_t* sqi, float* resid_rows, hipStream_t stream) {
    if (B <= 0) return SQE_OK;
    if (dim % 64 != 0 || dim > 8192 || q_pitch < dim || q_pitch % 16 != 0) return fail(SQE_ERR_INVALID, "int8 queries: bad dim / pitch");
    hipLaunchKernelGGL((quantize_rows_i8_kernel<false>), dim3((unsigned)((B + 3) / 4)), dim3(256), 0, stream, qn, (const int64_t*)nullptr,
                       (int64_t)0, (int64_t)B, dim, i8_scale_unit(dim), out, (int64_t)0, q_pitch, sqi, resid_rows, (uint32_t*)nullptr);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

// The IVF index's int8 copy (ivf.hip): output row scatter[p] = quantised x row gather[p], p in [0, n), per-row scales, in the flat
// scan's TILED layout (256-row tiles of tile_stride bytes, the 64-B K slice h of tile row r at h * 16 KiB + r * 64)
int launch_quantize_gather_i8(const float* x, const int* gather, const int* scatter, int64_t n, int dim, int8_t* out, int64_t tile_stride,
                              uint32_t* sxi, hipStream_t stream) {
    if (n <= 0) return SQE_OK;
    if (dim % 64 != 0 || dim > 8192) return fail(SQE_ERR_INVALID, "int8 rows: bad dim");
    const unsigned blocks = (unsigned)std::min<int64_t>((n + 3) / 4, 1 << 20);
    hipLaunchKernelGGL((quantize_rows_i8_kernel<true>), dim3(blocks), dim3(256), 0, stream, x, (const int64_t*)nullptr, (int64_t)0, n, dim,
                       i8_scale_unit(dim), out, tile_stride, 0, sxi, (float*)nullptr, (uint32_t*)nullptr, gather, scatter);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

int launch_i8_thresholds(const float* cos_s, int m, const uint32_t* sqi, int dim, int B, int b_pad, int* thr_int, float* thr_eff,
                         hipStream_t stream) {
    hipLaunchKernelGGL(i8_thresholds_kernel, dim3((b_pad + 255) / 256), dim3(256), 0, stream, cos_s, m, sqi, i8_scale_unit(dim), B, b_pad,
                       thr_int, thr_eff);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

}  // namespace sqe
