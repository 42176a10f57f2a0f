// encoder.hip -- BERT encoder (mxbai-embed-large = BERT-large: 24 x {MHA 16x64, FFN 4096}, post-LN,
// erf-GELU, CLS pooling) behind sqe_encode*: stands where Ollama's /api/embeddings stood for
// ollama_embed_text (main.py:134-145).  bf16 weights and activations, fp32 MFMA accumulation,
// fp32 softmax / LayerNorm / residual sums.
//
// Kernels (gfx950, wave64):
//   E1 embed_ln_kernel      word + position + type embedding, LayerNorm            (HBM stream)
//   E2/E4/E5/E6 gemm_bf16_kernel<FM,FN,EPI>  Y = X W^T + b on v_mfma_f32_16x16x32_bf16, both
//                           operands staged global -> LDS by global_load_lds in full 128-B lines
//                           (XOR chunk swizzle on the source address and on the ds_read_b128
//                           address), double buffered; epilogues: bias | bias + erf-GELU |
//                           bias + residual (fp32 out, feeds LayerNorm)           (MFMA-bound)
//   E3 attention_kernel     flash-style per (sequence, head, 64 query rows): S^T = K Q^T so the
//                           probabilities land in registers exactly as the A operand of the P V
//                           MFMA; V fragments by ds_read_b64_tr_b16; online softmax in fp32
//   layernorm_kernel        fp32 row -> bf16 row;  E7 pool_ln_kernel: LayerNorm of the CLS rows -> fp32
//
// Token layout: padded [B, S] (row = b * S + s); keys at positions >= lens[b] are masked, padded
// query rows compute values nobody reads.  Activation buffers are padded to a multiple of the
// GEMM token tile.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <memory>
#include <string>
#include <type_traits>
#include <vector>

#include <stdlib.h>

#include "internal.h"

namespace sqe {

namespace {

constexpr int ROWB = 128;   // bytes per tile row per 64-wide bf16 K step

// ------------------------------------------------------------------ staging helpers
template <int ROWS, int NW>
__device__ __forceinline__ void stage_rows(const char* gbase, size_t ld_bytes, char* lds, int wave, int lane) {
    constexpr int NINSTR = ROWS / 8;          // one 1-KiB wave-instruction per 8 rows
    constexpr int ITERS = (NINSTR + NW - 1) / NW;
    const int r_local = lane >> 3;
    const int cprime = lane & 7;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int g = it * NW + wave;
        if (NINSTR % NW == 0 || g < NINSTR) {
            const int r = g * 8 + r_local;
            const int c = cprime ^ ((r >> 1) & 7);
            const char* src = gbase + (size_t)r * ld_bytes + c * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(lds + g * 1024), 16, 0, 0);
        }
    }
}
// same tile image, pieces issued through inline asm (invisible to hipcc's waitcnt insertion: the caller
// counts its own waits and may keep a tile in flight across LDS reads)
template <int ROWS, int NW>
__device__ __forceinline__ void stage_rows_asm(const char* gbase, size_t ld_bytes, char* lds, int wave, int lane) {
    constexpr int NINSTR = ROWS / 8;
    static_assert(NINSTR % NW == 0, "every wave issues the same number of pieces");
    const int r_local = lane >> 3;
    const int cprime = lane & 7;
#pragma unroll
    for (int it = 0; it < NINSTR / NW; ++it) {
        const int g = it * NW + wave;
        const int r = g * 8 + r_local;
        const int c = cprime ^ ((r >> 1) & 7);
        lds_dma16(gbase + (size_t)r * ld_bytes + c * 16, lds + g * 1024);
    }
}
__device__ __forceinline__ bf16x8 frag(const char* tile, int r, int c) {
    return *reinterpret_cast<const bf16x8*>(tile + r * ROWB + ((c ^ ((r >> 1) & 7)) << 4));
}
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float((uint32_t)v << 16); }

// ------------------------------------------------------------------ GEMM
enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_F32 = 3 };   // EPI_F32: plain fp32 products, no bias (IVF coarse scores)

// erf-GELU, 0.5 v (1 + erf(v / sqrt 2)), with erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below
// the bf16 rounding of the result): one v_rcp, one v_exp and a 5-term Horner chain instead of libm's
// branchy erff -- the GELU epilogue of the FFN-up GEMM was as long as its K loop.
__device__ __forceinline__ float gelu_erf(float v) {
    const float x = fabsf(v) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = 1.0f - poly * t * __expf(-x * x);          // erf(|v| / sqrt 2)
    return 0.5f * v * (1.0f + copysignf(e, v));
}

// The same on two values at once: every multiply / fma of the chain is one packed fp32 instruction
// (v_pk_mul_f32 / v_pk_fma_f32), which halves the instruction count of the GELU epilogue -- the epilogue of the
// FFN-up GEMM is VALU-issue-bound (no MFMA runs beside it).
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 v) {
    const f32x2 x = {fabsf(v[0]) * 0.70710678118654752f, fabsf(v[1]) * 0.70710678118654752f};
    const f32x2 den = __builtin_elementwise_fma(f32x2{0.3275911f, 0.3275911f}, x, f32x2{1.0f, 1.0f});
    const f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    f32x2 poly = __builtin_elementwise_fma(f32x2{1.061405429f, 1.061405429f}, t, f32x2{-1.453152027f, -1.453152027f});
    poly = __builtin_elementwise_fma(poly, t, f32x2{1.421413741f, 1.421413741f});
    poly = __builtin_elementwise_fma(poly, t, f32x2{-0.284496736f, -0.284496736f});
    poly = __builtin_elementwise_fma(poly, t, f32x2{0.254829592f, 0.254829592f});
    const f32x2 nx2 = x * x * -1.4426950408889634f;                  // exp(-x^2) = exp2(-x^2 log2 e)
    const f32x2 ex = {__builtin_amdgcn_exp2f(nx2[0]), __builtin_amdgcn_exp2f(nx2[1])};
    const f32x2 e = __builtin_elementwise_fma(poly * t, -ex, f32x2{1.0f, 1.0f});      // erf(|v| / sqrt 2)
    const f32x2 es = {copysignf(e[0], v[0]), copysignf(e[1], v[1])};
    return __builtin_elementwise_fma(es, v * 0.5f, v * 0.5f);          // 0.5 v (1 + erf)
}

// GELU for the large-batch GEMM epilogue, where it is the cost (128 values per lane, VALU-issue-bound: the erf form
// above is ~36 issue slots per pair of values, four of them transcendental): 0.5 x (1 + erf(x / sqrt 2)) =
// x (0.5 + h(x)), h odd; h(x) ~ x Q(x^2) with a degree-13 odd polynomial fitted on |x| <= 4 (weighted so that
// |x| |h error| is minimised), x clamped to [-4, 4] inside h, and the fit pinned so that this fp32 evaluation gives
// h(4) = 0.5 EXACTLY: beyond the clamp GELU is x or 0 exactly, however large |x| is.  12 issue slots per pair, no
// transcendental.  |GELU_poly - GELU_erf| <= 1.9e-4 absolute for every x (tools/gelu_fit.py makes the fit and prints
// the error of the fp32 evaluation); the output is rounded to bf16 (2^-9 relative) right after.
__device__ __forceinline__ f32x2 gelu_poly2(f32x2 v) {
    const f32x2 xc = {__builtin_amdgcn_fmed3f(v[0], -4.0f, 4.0f), __builtin_amdgcn_fmed3f(v[1], -4.0f, 4.0f)};
    const f32x2 t = xc * xc;
    f32x2 q = __builtin_elementwise_fma(f32x2{2.258820153144825e-08f, 2.258820153144825e-08f}, t, f32x2{-1.5888268762864755e-06f, -1.5888268762864755e-06f});
    q = __builtin_elementwise_fma(q, t, f32x2{4.776388232130557e-05f, 4.776388232130557e-05f});
    q = __builtin_elementwise_fma(q, t, f32x2{-0.000812187441624701f, -0.000812187441624701f});
    q = __builtin_elementwise_fma(q, t, f32x2{0.00876369047909975f, 0.00876369047909975f});
    q = __builtin_elementwise_fma(q, t, f32x2{-0.06455441564321518f, -0.06455441564321518f});
    q = __builtin_elementwise_fma(q, t, f32x2{0.39787042140960693f, 0.39787042140960693f});
    const f32x2 h = q * xc;
    return __builtin_elementwise_fma(v, h, v * 0.5f);
}

struct GemmArgs {
    const bf16_t* W;      // [N, K] row-major (torch Linear weight)
    const bf16_t* X;      // [T_pad, K]
    const float* bias;    // [N]
    const bf16_t* resid;  // [T_pad, N] (EPI_RESID)
    void* out;            // bf16 [T_pad, N] or fp32 [T_pad, N] (EPI_RESID)
    int N, K, T;          // T = valid token rows
    int n_tiles;
    int splits;           // split-K factor (ring kernel, EPI_RESID only): split s writes its partial sum to
    size_t split_stride;  //   out + s * split_stride floats; bias and residual are added by split 0
    int t_tiles;
    int xcd_patches;      // gemm_ring_kernel: XCD-aware tile walk (0: linear)
    int pp_stagger;       // gemm_pp_kernel: half of a group's waves read their operands before they issue their DMA pieces
    int pp_dbg;           // gemm_pp_kernel, knobs build, timing only: 2 = no stores (results wrong), 4 = phase stamps (STAMPS build)
};

// Block tile: (2*FM*16) output features x (4*FN*16) tokens, 8 waves as 2 (features) x 4 (tokens).
// MFMA A operand = W rows, B operand = X rows, so a lane holds 4 CONSECUTIVE features of one
// token per fragment: acc[i][j][r] = Y[t0 + j*16 + (lane&15)][n0 + i*16 + (lane>>4)*4 + r].
template <int FM, int FN, int EPI>
__global__ __launch_bounds__(512) void gemm_bf16_kernel(GemmArgs p) {
    constexpr int BNW = 2 * FM * 16, BT = 4 * FN * 16;
    constexpr int STAGE = (BNW + BT) * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nt = blockIdx.x % p.n_tiles, tt = blockIdx.x / p.n_tiles;
    const int n0 = nt * BNW, t0 = tt * BT;
    const size_t ld = (size_t)p.K * 2;
    const char* wbase = reinterpret_cast<const char*>(p.W) + (size_t)n0 * ld;
    const char* xbase = reinterpret_cast<const char*>(p.X) + (size_t)t0 * ld;
    const int KS = p.K / 64;

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage_rows<BNW, 8>(wbase, ld, smem, wave, lane);
    stage_rows<BT, 8>(xbase, ld, smem + BNW * ROWB, wave, lane);
    __syncthreads();
    for (int ks = 0; ks < KS; ++ks) {
        const char* cur = smem + (ks & 1) * STAGE;
        if (ks + 1 < KS) {
            char* nxt = smem + ((ks + 1) & 1) * STAGE;
            stage_rows<BNW, 8>(wbase + (size_t)(ks + 1) * ROWB, ld, nxt, wave, lane);
            stage_rows<BT, 8>(xbase + (size_t)(ks + 1) * ROWB, ld, nxt + BNW * ROWB, wave, lane);
        }
        const char* tA = cur;
        const char* tB = cur + BNW * ROWB;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[FM], b[FN];
            const int c = kk * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < FM; ++i) a[i] = frag(tA, wm * (FM * 16) + i * 16 + (lane & 15), c);
#pragma unroll
            for (int j = 0; j < FN; ++j) b[j] = frag(tB, wn * (FN * 16) + j * 16 + (lane & 15), c);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int n = n0 + wm * (FM * 16) + i * 16 + (lane >> 4) * 4;
        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int t = t0 + wn * (FN * 16) + j * 16 + (lane & 15);
            if (t >= p.T) continue;
            float v0 = acc[i][j][0] + b4.x, v1 = acc[i][j][1] + b4.y, v2 = acc[i][j][2] + b4.z, v3 = acc[i][j][3] + b4.w;
            const size_t o = (size_t)t * p.N + n;
            if (EPI == EPI_RESID) {
                const uint2 r2 = *reinterpret_cast<const uint2*>(p.resid + o);
                v0 += bf16_to_f32((bf16_t)(r2.x & 0xffff)); v1 += bf16_to_f32((bf16_t)(r2.x >> 16));
                v2 += bf16_to_f32((bf16_t)(r2.y & 0xffff)); v3 += bf16_to_f32((bf16_t)(r2.y >> 16));
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + o) = make_float4(v0, v1, v2, v3);
            } else {
                if (EPI == EPI_GELU) { v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3); }
                uint2 w2;
                w2.x = pack_bf16x2(v0, v1);
                w2.y = pack_bf16x2(v2, v3);
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + o) = w2;
            }
        }
    }
}

template <int FM, int FN, int EPI>
int launch_gemm_cfg(const GemmArgs& a, int t_pad, hipStream_t stream) {
    constexpr int BNW = 2 * FM * 16, BT = 4 * FN * 16;
    constexpr int LDS = 2 * (BNW + BT) * ROWB;
    GemmArgs p = a;
    p.n_tiles = a.N / BNW;
    auto kern = gemm_bf16_kernel<FM, FN, EPI>;
    SQE_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS));
    hipLaunchKernelGGL(kern, dim3(p.n_tiles * (t_pad / BT)), dim3(512), LDS, stream, p);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

// Large-batch form: 256 x 256 tiles, PERSISTENT workgroups.  One workgroup per CU walks its tiles in a
// single flattened stream of K steps (two LDS stages; the first K step of the next tile is already in
// flight while the last one of this tile is computed), the epilogue's stores are fire-and-forget: they
// are issued after the next tile's first DMA pieces and the counted wait that ends the K step leaves
// them in flight.  Stores and DMA pieces go through inline asm so that hipcc inserts no vmcnt(0) of its
// own.  Consecutive tile ids share the token tile or the weight tile, so the 256 tiles in flight at any
// time reuse each other's operands in L2.
// Hazard audit (r03, DESIGN.md "asm stores"): the data of this store always comes out of v_cvt_pk_bf16_f32 -- a VALU
// write, which the hardware interlocks against a VMEM read of the same VGPR (no software wait states; the MFMA ->
// VALU distance in front of the pack is hipcc's, both instructions being visible to it) -- and a store of at most
// 8 bytes has read its data by the time the next instruction issues.  The 16-byte form does NOT have that second
// property (its data registers must not be written for two wait states AFTER it): r02 padded in front of it instead
// and got garbage; the large-batch epilogue's 16-byte stores are plain stores hipcc pads itself.
__device__ __forceinline__ void store_b64_asm(void* p, uint2 v) {
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    u32x2 w;
    w.x = v.x; w.y = v.y;
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(w) : "memory");
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_persistent_kernel(GemmArgs p) {
    constexpr int FM = 8, FN = 4;
    constexpr int BNW = 256, BT = 256;
    constexpr int STAGE = (BNW + BT) * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const size_t ld = (size_t)p.K * 2;
    const int KS = p.K / 64;
    const int tiles = p.n_tiles * p.t_tiles;
    // Tile walk.  Workgroups b and b + 8 share an XCD (and its 4 MiB L2): round r hands XCD x the patch
    // r * 8 + x of 4 weight tiles x 8 token tiles, one tile per workgroup, so the 32 tiles an XCD has in flight
    // read 12 operand tiles between them (a K slice of all of them is 384 KiB: it stays in that L2) instead of
    // the ~27 a strided walk touches.  Patches go weight-tile-group first, so the XCDs of one round share token
    // rows through the Infinity Cache.  Needs a full grid and n_tiles % 4 == 0; otherwise tile = b + r * grid.
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3;
    const bool patched = gridDim.x == 256 && (p.n_tiles & 3) == 0;
    const int NP = p.n_tiles >> 2, TP = (p.t_tiles + 7) >> 3;
    auto tile_at = [&](int r) -> int {             // linear id of this workgroup's r-th tile, < 0: none (and none later)
        if (!patched) {
            const int t = (int)blockIdx.x + r * (int)gridDim.x;
            return t < tiles ? t : -1;
        }
        const int pidx = r * 8 + xcd;
        if (pidx >= NP * TP) return -1;
        const int nt = (pidx % NP) * 4 + (lb & 3), tt = (pidx / NP) * 8 + (lb >> 2);
        return tt < p.t_tiles ? tt * p.n_tiles + nt : -1;
    };
    int my_tiles = 0;
    while (tile_at(my_tiles) >= 0) ++my_tiles;
    const int total = my_tiles * KS;
    if (total == 0) return;

    unsigned off[4];                       // per-lane source offsets of this wave's 4 pieces per operand
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave + 8 * i) * 8 + (lane >> 3);
        off[i] = (unsigned)(r * ld) + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
    }
    // issue cursor
    int i_round = 0, i_tile = tile_at(0), i_ks = 0;
    auto issue = [&](int s) {
        const int nt = i_tile % p.n_tiles, tt = i_tile / p.n_tiles;
        const char* ws = reinterpret_cast<const char*>(p.W) + (size_t)nt * BNW * ld + (size_t)i_ks * ROWB;
        const char* xs = reinterpret_cast<const char*>(p.X) + (size_t)tt * BT * ld + (size_t)i_ks * ROWB;
        char* buf = smem + (s & 1) * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(ws + off[i], buf + (wave + 8 * i) * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(xs + off[i], buf + BNW * ROWB + (wave + 8 * i) * 1024);
        if (++i_ks == KS) { i_ks = 0; i_tile = tile_at(++i_round); }
    };

    f32x4 acc[FM][FN];
    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    int round = 0, tile = tile_at(0), ks = 0;
    for (int s = 0; s < total; ++s) {
        const bool more = s + 1 < total;
        if (more) issue(s + 1);
        if (ks == 0) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const char* tA = smem + (s & 1) * STAGE;
        const char* tB = tA + BNW * ROWB;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[FM], b[FN];
            const int c = kk * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < FM; ++i) a[i] = frag(tA, wm * (FM * 16) + i * 16 + (lane & 15), c);
#pragma unroll
            for (int j = 0; j < FN; ++j) b[j] = frag(tB, wn * (FN * 16) + j * 16 + (lane & 15), c);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        bool stored = false;
        if (++ks == KS) {
            ks = 0;
            const int nt = tile % p.n_tiles, tt = tile / p.n_tiles;
            const int n0 = nt * BNW, t0 = tt * BT;
            // every load the epilogue needs is issued before its first store: a load hipcc can see makes it
            // wait for everything older, the stores included
            uint2 res[EPI == EPI_RESID ? FM : 1][EPI == EPI_RESID ? FN : 1];
            float4 bias4[FM];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int n = n0 + wm * (FM * 16) + i * 16 + (lane >> 4) * 4;
                bias4[i] = *reinterpret_cast<const float4*>(p.bias + n);
                if (EPI == EPI_RESID) {
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        const int t = t0 + wn * (FN * 16) + j * 16 + (lane & 15);
                        res[EPI == EPI_RESID ? i : 0][EPI == EPI_RESID ? j : 0] =
                            *reinterpret_cast<const uint2*>(p.resid + (size_t)t * p.N + n);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int n = n0 + wm * (FM * 16) + i * 16 + (lane >> 4) * 4;
                const float4 b4 = bias4[i];
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int t = t0 + wn * (FN * 16) + j * 16 + (lane & 15);
                    // rows past T are padding of the activation buffers: written like the others (every
                    // wave then issues exactly FM * FN stores, which the counted wait below relies on)
                    float v0 = acc[i][j][0] + b4.x, v1 = acc[i][j][1] + b4.y, v2 = acc[i][j][2] + b4.z, v3 = acc[i][j][3] + b4.w;
                    const size_t o = (size_t)t * p.N + n;
                    if (EPI == EPI_RESID) {
                        const uint2 r2 = res[EPI == EPI_RESID ? i : 0][EPI == EPI_RESID ? j : 0];
                        v0 += bf16_to_f32((bf16_t)(r2.x & 0xffff)); v1 += bf16_to_f32((bf16_t)(r2.x >> 16));
                        v2 += bf16_to_f32((bf16_t)(r2.y & 0xffff)); v3 += bf16_to_f32((bf16_t)(r2.y >> 16));
                    } else if (EPI == EPI_GELU) {
                        const f32x2 g01 = gelu_erf2(f32x2{v0, v1}), g23 = gelu_erf2(f32x2{v2, v3});
                        v0 = g01[0]; v1 = g01[1]; v2 = g23[0]; v3 = g23[1];
                    }
                    // (the pre-LayerNorm sums of EPI_RESID leave as bf16 too: half the bytes written here and
                    // read by the LayerNorm kernel; the small-batch kernels keep fp32 split-K partial sums)
                    uint2 w2;
                    w2.x = pack_bf16x2(v0, v1);
                    w2.y = pack_bf16x2(v2, v3);
                    store_b64_asm(reinterpret_cast<bf16_t*>(p.out) + o, w2);
                }
            }
            tile = tile_at(++round);
            stored = true;
        }
        // next K step landed.  After an epilogue the FM * FN = 32 stores just issued are younger than its
        // DMA pieces and stay in flight.
        if (stored && more) asm volatile("s_waitcnt vmcnt(32) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int EPI>
int launch_gemm_persistent(const GemmArgs& a, int t_pad, int cu_count, hipStream_t stream, int* splits_out = nullptr) {
    constexpr int LDS = 2 * (256 + 256) * ROWB;
    GemmArgs p = a;
    p.n_tiles = a.N / 256;
    p.t_tiles = t_pad / 256;
    p.splits = 1; p.split_stride = 0;
    const int tiles = p.n_tiles * p.t_tiles;
    auto kern = gemm_persistent_kernel<EPI>;
    SQE_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS));
    hipLaunchKernelGGL(kern, dim3(std::min(tiles, cu_count)), dim3(512), LDS, stream, p);
    SQE_HIP(hipGetLastError());
    if (EPI == EPI_RESID && splits_out) *splits_out = 0;      // 0 partial sums: one bf16 row
    return SQE_OK;
}

// Large-batch form, ping-pong schedule (the flat scan's, scan_pp.hip, with an epilogue where the scan has its
// filter): 256 x 256 tiles, PERSISTENT workgroups, K in 32-wide half-steps through a 4-stage LDS ring, the two
// waves of a SIMD taking turns at the matrix pipe: while one computes a half-step (32 MFMAs on registers) the
// other does its memory work (4 LDS-DMA pieces, 12 ds_read_b128).
// Schedule (r04, shipped): ONE s_barrier per half-step (period T_j) instead of r02's one per phase -- the two groups no longer wait for
// each other in the middle of a period:
//     G0 (waves 0-3), T_j: compute j | [epilogue] issue pieces j + 3 | vmcnt: own pieces of j + 2 | read operands j + 1 | barrier
//     G1 (waves 4-7), T_j: [epilogue] issue pieces j + 3 | read operands j | compute j | vmcnt: own pieces of j + 2 | barrier
// Invariant, as in scan_i8.hip: a piece read in a period was retired by the wave that issued it before a barrier that precedes the
// read (every wave retires its pieces of x in T_{x-2}; x is read at the end of T_{x-1} and the head of T_x).  r03's form of this
// schedule had G0 retire its pieces of j + 1 at the head of T_j, behind the barrier -- sibling G0 waves read them unordered, the
// output of 64 x 512 tokens changed in a row or two between identical calls (7 of 30), and r03 shipped r02's schedule instead.  On
// the corrected schedule: 0 of 12 repeats at 64 x 512 tokens and 0 of 40 at 64 x 128 differ (tools/repeat_enc.py), 24.09-24.14 ->
// 23.70-23.74 ms for the encode of 64 x 512 tokens (profiles/r04_configs/enc_one_barrier_ab.log).  An epilogue always stands in
// FRONT of its period's pieces, so the plain vmcnt(4) behind them covers its stores (operations retire in issue order).
// Build 512 (tools/build_gpp_ablate.sh 512) keeps r02's schedule for A/B.
namespace gpp {
// Ablation builds of the ping-pong GEMM (tools/build_gpp_ablate.sh <bits>, timing only, results wrong): 8 = no DMA pieces,
// 16 = no operand reads, 32 = no epilogue, 64 = every tile reads the operands of tile 0 (always in L2), 128 = the pieces of a
// half-step read whole 128-B lines of 128 rows instead of 64-B halves of 256 rows (the same bytes per tile), 256 = a 5-stage
// ring (four half-steps in flight) over all 160 KiB, the bias vector read from inside it; 512 = r02's schedule, a barrier after
// every phase.  Compile-time, so
// that the shipped kernel's register allocation is the one measured (run-time switches made hipcc spill).
#ifndef SQE_GPP_ABLATE
#define SQE_GPP_ABLATE 0
#endif
constexpr int GPP_ABLATE = SQE_GPP_ABLATE;
constexpr int HALF_K = 32, LINE_BYTES = 128, OPER_BYTES = 128 * LINE_BYTES, STAGE_BYTES = 2 * OPER_BYTES;
constexpr int NSTAGE = (GPP_ABLATE & 256) ? 5 : 4;   // (ablation 256: a 5-stage ring over ALL of the LDS, the bias vector inside it)
constexpr int AHEAD = NSTAGE - 1;                     // half-steps of DMA in flight
constexpr int RING_BYTES = NSTAGE * STAGE_BYTES;      // 128 KiB
constexpr int MAX_BIAS_N = 4096;                      // the bias vector sits in LDS behind the ring
constexpr int LDS_BYTES = (GPP_ABLATE & 256) ? RING_BYTES : RING_BYTES + MAX_BIAS_N * 4;
typedef bf16x8 AOps[8];
typedef bf16x8 BOps[4];
struct Cursor { int e, h; const char* a; const char* b; };
// DMA pieces through inline asm (common.h: lds_dma16), as in scan_pp.hip: hipcc does not see them and so never puts an
// s_waitcnt vmcnt(0) of its own in front of an LDS read; every wait for a piece is an explicit counted one.
__device__ __forceinline__ void glds16(const char* src, char* lds_wave_base) { lds_dma16(src, lds_wave_base); }
#define GPP_BARRIER()                          \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)
template <bool FIRST>
__device__ __forceinline__ void cmp_phase(f32x4 (&acc)[8][4], const AOps& a, const BOps& b) {
#pragma unroll
    for (int fm = 0; fm < 8; ++fm)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn)
            acc[fm][fn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[fm], b[fn], FIRST ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[fm][fn], 0, 0, 0);
}
}  // namespace gpp

// Phase timing of the ping-pong GEMM (make KNOBS=1 STAMPS=1, SQE_GEMM_DBG bit 4): core-clock sums of workgroup 0,
// waves 0 (group 0) and 4 (group 1): [0] tiles, [1] kernel, [2] epilogue, [3] memory phase after the epilogue,
// [4] barrier after it, [5] steady compute phases, [6] barriers after them, [7] steady phases counted
#ifdef SQE_PHASE_STAMPS
__device__ unsigned long long g_gemm_clk[2][8];
#define GPP_STAMP(var)                                 \
    do {                                               \
        __builtin_amdgcn_sched_barrier(0);             \
        var = __builtin_readcyclecounter();            \
        __builtin_amdgcn_sched_barrier(0);             \
    } while (0)
#define GPP_ACC(...) \
    do {             \
        __VA_ARGS__; \
    } while (0)
#else
#define GPP_STAMP(var) \
    do {               \
    } while (0)
#define GPP_ACC(...) \
    do {             \
    } while (0)
#endif

template <int EPI>
__global__ __launch_bounds__(512) void gemm_pp_kernel(GemmArgs p) {
    using namespace gpp;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* bias_lds = reinterpret_cast<float*>(smem + ((GPP_ABLATE & 256) ? 0 : RING_BYTES));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = wave >> 2, wm = wave >> 2, wn = wave & 3;
    const size_t ld = (size_t)p.K * 2;
    const int HS = p.K / HALF_K;
    const int tiles = p.n_tiles * p.t_tiles;
    // tile walk: as gemm_persistent_kernel (XCD-aware 4 x 8 patches on a full grid)
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3;
    const bool patched = gridDim.x == 256 && (p.n_tiles & 3) == 0;
    const int NP = p.n_tiles >> 2, TP = (p.t_tiles + 7) >> 3;
    auto tile_at = [&](int r) -> int {
        if (!patched) {
            const int t = (int)blockIdx.x + r * (int)gridDim.x;
            return t < tiles ? t : -1;
        }
        const int pidx = r * 8 + xcd;
        if (pidx >= NP * TP) return -1;
        const int nt = (pidx % NP) * 4 + (lb & 3), tt = (pidx / NP) * 8 + (lb >> 2);
        return tt < p.t_tiles ? tt * p.n_tiles + nt : -1;
    };
    int my_tiles = 0;
    while (tile_at(my_tiles) >= 0) ++my_tiles;
    const int J = my_tiles * HS;
    if (J == 0) return;

    // per-lane DMA source offsets (piece t covers LDS lines 8t .. 8t+7; line L holds the slices of tile rows L
    // and L + 128, chunk positions XOR-swizzled by (L >> 1) & 7) and operand read offsets: as scan_pp.hip
    unsigned off0, off1, offA0, offA1, rdA, rdB;
    {
        const int line = wave * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((line >> 1) & 7);
        const int row = line + 128 * (c >> 2);
        off0 = (unsigned)(row * ld) + (c & 3) * 16;
        off1 = off0 + (unsigned)(64 * ld);
        // Weight rows enter the tile PERMUTED inside each 128-row half: tile row p = i * 16 + g * 4 + r (fragment i,
        // lane group g, accumulator element r) holds feature (i >> 1) * 32 + g * 8 + (i & 1) * 4 + r, so that a lane's
        // fragments 2s and 2s + 1 are EIGHT consecutive features of its token: one 16-byte store, and the four lane
        // groups of a store cover 64 contiguous bytes of a token row (32-byte segments cost the epilogue ~300 cycles
        // per store instruction).  Only the source row of the DMA changes; the LDS image and the reads do not.
        auto wrow = [](int tr) {
            const int pr = tr & 127, i = pr >> 4, g = (pr >> 2) & 3, r = pr & 3;
            return (tr & 128) + (i >> 1) * 32 + g * 8 + (i & 1) * 4 + r;
        };
        offA0 = (unsigned)(wrow(row) * ld) + (c & 3) * 16;
        offA1 = (unsigned)(wrow(row + 64) * ld) + (c & 3) * 16;
        if (GPP_ABLATE & 128) {                       // whole 128-B lines: 8 lanes per row, 128 rows per half-step
            off0 = offA0 = (unsigned)(line * ld) + c * 16;
            off1 = offA1 = off0 + (unsigned)(64 * ld);
        }
        const int r = lane & 15, cq = lane >> 4, sw = (r >> 1) & 7;
        rdA = (unsigned)(r * LINE_BYTES + (((wm * 4 + cq) ^ sw) << 4));
        rdB = (unsigned)(((wn & 1) * 64 + r) * LINE_BYTES + ((((wn >> 1) * 4 + cq) ^ sw) << 4));
    }
    auto seat = [&](Cursor& c) {                      // operand tiles of entry c.e
        const int t = tile_at(c.e);
        int nt = t % p.n_tiles, tt = t / p.n_tiles;
        if (GPP_ABLATE & 64) nt = tt = 0;
        c.a = reinterpret_cast<const char*>(p.W) + (size_t)nt * 256 * ld;
        c.b = reinterpret_cast<const char*>(p.X) + (size_t)tt * 256 * ld;
    };
    auto advance = [&](Cursor& c) {
        if (++c.h == HS) {
            c.h = 0;
            if (++c.e < my_tiles) seat(c);
        }
    };
    auto issue = [&](const Cursor& c, int stage) {
        char* st = smem + stage * STAGE_BYTES;
        size_t koff = (size_t)c.h * (HALF_K * 2);
        if (GPP_ABLATE & 128) {                       // (rows 0-127 over all of K, then rows 128-255: the tile's bytes, each once)
            const size_t hh = (size_t)c.h * 128;
            koff = hh % ld + hh / ld * 128 * ld;
        }
        const char* as = c.a + koff;
        const char* bs = c.b + koff;
        glds16(as + offA0, st + wave * 1024);
        glds16(as + offA1, st + (wave + 8) * 1024);
        glds16(bs + off0, st + OPER_BYTES + wave * 1024);
        glds16(bs + off1, st + OPER_BYTES + (wave + 8) * 1024);
    };
    Cursor rd{0, 0, nullptr, nullptr}, dm{0, 0, nullptr, nullptr};
    seat(rd);
    seat(dm);
    int post_epi = 0;                                 // memory phases that still have this wave's epilogue stores in flight

    f32x4 acc[8][4];
    AOps a;
    BOps b;
    if (GPP_ABLATE & 16) {                            // (the MFMAs then compute on these)
        for (int i = 0; i < 8; ++i) a[i] = bf16x8{};
        for (int i = 0; i < 4; ++i) b[i] = bf16x8{};
    }
    // waves 0, 1 (4, 5) of a group issue their DMA pieces first and read their operands after, waves 2, 3 (6, 7) the
    // other way round: the address unit and the LDS then work side by side (scan_pp.hip: dma_and_reads)
    const bool reads_first = p.pp_stagger && ((wave >> 1) & 1);
    auto read_operands = [&](int j) {
        const char* st = smem + (NSTAGE == 4 ? (j & 3) : j % NSTAGE) * STAGE_BYTES;
#pragma unroll
        for (int fm = 0; fm < 8; ++fm) a[fm] = *reinterpret_cast<const bf16x8*>(st + rdA + fm * 2048);
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) b[fn] = *reinterpret_cast<const bf16x8*>(st + OPER_BYTES + rdB + fn * 2048);
    };
    auto mem_phase = [&](int j) {
        const bool more = j + AHEAD < J;
        const bool do_dma = more && !(GPP_ABLATE & 8);
        constexpr bool do_reads = !(GPP_ABLATE & 16);
        if (reads_first) {
            if (do_reads) read_operands(j);
            if (do_dma) issue(dm, NSTAGE == 4 ? ((j + 3) & 3) : (j + AHEAD) % NSTAGE);
        } else {
            if (do_dma) issue(dm, NSTAGE == 4 ? ((j + 3) & 3) : (j + AHEAD) % NSTAGE);
            if (do_reads) read_operands(j);
        }
        // retire the DMA of half-step j + 1; j + 2 and j + 3 stay in flight, and so do the 16 epilogue stores for
        // the two memory phases after an epilogue (they are younger than the half-step that has to land)
        if (!more) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if (post_epi > 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(16 + 4 * (AHEAD - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * (AHEAD - 1)) : "memory");
        if (post_epi > 0) --post_epi;
        advance(rd);
        if (more) advance(dm);
    };
    auto epilogue = [&](int e) {
        if ((GPP_ABLATE & 32) && p.K > 0) return;         // (a run-time condition: the MFMAs stay)
        const int t = tile_at(e);
        const int nt = t % p.n_tiles, tt = t / p.n_tiles;
        const int n0 = nt * 256, t0 = tt * 256;
        // The loads an epilogue needs are issued before its first store (a load hipcc can see makes it wait for
        // everything older, the stores included).  EPI_RESID reads 64 VGPRs of residual per wave: it goes in two
        // halves of two fragment pairs, so that the kernel keeps its accumulators and the residual in registers.
        // A fragment pair (2s, 2s + 1) is eight consecutive features of the lane's token (see the row permutation
        // above): 16 stores of 16 bytes per wave.
        constexpr int HALVES = EPI == EPI_RESID ? 2 : 1;
        constexpr int SB = 4 / HALVES;                    // fragment pairs per half
#pragma unroll
        for (int half = 0; half < HALVES; ++half) {
            uint4 res[EPI == EPI_RESID ? SB : 1][EPI == EPI_RESID ? 4 : 1];
#pragma unroll
            for (int ss = 0; ss < SB; ++ss) {
                const int sp = half * SB + ss;
                const int n = n0 + wm * 128 + sp * 32 + (lane >> 4) * 8;
                if (EPI == EPI_RESID) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int tk = t0 + wn * 64 + j * 16 + (lane & 15);
                        res[EPI == EPI_RESID ? ss : 0][EPI == EPI_RESID ? j : 0] =
                            *reinterpret_cast<const uint4*>(p.resid + (size_t)tk * p.N + n);
                    }
                }
            }
            // bias from LDS (the whole vector was copied in at kernel start): no vector-memory load in the epilogue of the
            // bias / GELU kernels; all reads of a half up front, one LDS round trip (the operand registers are dead here)
            float4 ball[SB][2];
#pragma unroll
            for (int ss = 0; ss < SB; ++ss) {
                const int n = n0 + wm * 128 + (half * SB + ss) * 32 + (lane >> 4) * 8;
                ball[ss][0] = *reinterpret_cast<const float4*>(bias_lds + n);
                ball[ss][1] = *reinterpret_cast<const float4*>(bias_lds + n + 4);
            }
#pragma unroll
            for (int ss = 0; ss < SB; ++ss) {
                const int sp = half * SB + ss;
                const int n = n0 + wm * 128 + sp * 32 + (lane >> 4) * 8;
                const float4 bq[2] = {ball[ss][0], ball[ss][1]};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int tk = t0 + wn * 64 + j * 16 + (lane & 15);
                    const size_t o = (size_t)tk * p.N + n;
                    unsigned w[4];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {         // the two fragments of the pair
                        const int i = 2 * sp + q;
                        const float4 b4 = bq[q];
                        float v0 = acc[i][j][0] + b4.x, v1 = acc[i][j][1] + b4.y, v2 = acc[i][j][2] + b4.z, v3 = acc[i][j][3] + b4.w;
                        if (EPI == EPI_RESID) {
                            const uint4 r4 = res[EPI == EPI_RESID ? ss : 0][EPI == EPI_RESID ? j : 0];
                            const unsigned rx = q ? r4.z : r4.x, ry = q ? r4.w : r4.y;
                            v0 += bf16_to_f32((bf16_t)(rx & 0xffff)); v1 += bf16_to_f32((bf16_t)(rx >> 16));
                            v2 += bf16_to_f32((bf16_t)(ry & 0xffff)); v3 += bf16_to_f32((bf16_t)(ry >> 16));
                        } else if (EPI == EPI_GELU) {
                            const f32x2 g01 = gelu_poly2(f32x2{v0, v1}), g23 = gelu_poly2(f32x2{v2, v3});
                            v0 = g01[0]; v1 = g01[1]; v2 = g23[0]; v3 = g23[1];
                        }
                        w[2 * q] = pack_bf16x2(v0, v1);
                        w[2 * q + 1] = pack_bf16x2(v2, v3);
                    }
                    // a plain store: hipcc may see it (what it must not see are the DMA pieces).  The same 16-byte store
                    // through an asm block left garbage in the output in r02 -- with s_nop padding for the store-data
                    // hazard too; the 8-byte asm stores of the other GEMM kernels are fine -- cause not found, form not used.
                    if (!(p.pp_dbg & 2)) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + o) = uint4{w[0], w[1], w[2], w[3]};
                }
            }
        }
        post_epi = AHEAD - 1;
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- the bias vector into LDS (N <= MAX_BIAS_N: the launcher checks), then the prologue: half-steps 0, 1, 2
    if (!(GPP_ABLATE & 256))
        for (int i = tid * 4; i < p.N; i += 512 * 4) *reinterpret_cast<float4*>(bias_lds + i) = *reinterpret_cast<const float4*>(p.bias + i);
    for (int s = 0; s < AHEAD && s < J; ++s) {
        issue(dm, s);
        advance(dm);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the prologue's pieces (inline asm: the compiler does not wait for them)
    __syncthreads();                       // prologue landed

    int j = 0;
#ifdef SQE_PHASE_STAMPS
    unsigned long long k0 = 0, t0 = 0, t1 = 0, t2 = 0, t3 = 0, c0 = 0, c1 = 0, c2 = 0, clk[8] = {};
#endif
    GPP_STAMP(k0);
    static_assert(!(GPP_ABLATE & 256) || (GPP_ABLATE & 512), "the five-stage ring experiment runs on r02's schedule: build 768");
    if (!(GPP_ABLATE & 512)) {
        // SHIPPED (r04): one barrier per half-step (the schedule and its invariant: comment above the kernel, scan_i8.hip); build 512
        // keeps r02's loop, a barrier after every phase, below
        auto issue_next = [&](int jj) {
            if (jj + 3 < J && !(GPP_ABLATE & 8)) {
                issue(dm, (jj + 3) & 3);
                advance(dm);
            }
        };
        // at the end of T_jj: everything this wave has in flight but the four pieces of jj + 3 it issued LAST in this period
        // (vector-memory operations retire in issue order: epilogue stores and residual loads are older entries)
        auto wait_pieces = [&](int jj) {
            if (jj + 3 < J && !(GPP_ABLATE & 8)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        if (group == 0) {
            read_operands(0);
            for (int e = 0; e < my_tiles; ++e) {
                cmp_phase<true>(acc, a, b);
                issue_next(j);
                wait_pieces(j);
                if (j + 1 < J) read_operands(j + 1);
                GPP_BARRIER();
                ++j;
                for (int h = 1; h < HS - 1; ++h) {
                    // (stamps of this form: [3] memory part, [4] barrier, [5] compute part, per steady period)
                    GPP_STAMP(c0);
                    cmp_phase<false>(acc, a, b);
                    GPP_STAMP(c1);
                    issue_next(j);
                    wait_pieces(j);
                    if (j + 1 < J) read_operands(j + 1);
                    GPP_STAMP(c2);
                    GPP_BARRIER();
                    GPP_STAMP(t0);
                    GPP_ACC(clk[5] += c1 - c0; clk[3] += c2 - c1; clk[4] += t0 - c2; ++clk[7]; ++clk[0]);
                    ++j;
                }
                // last period of a tile: the epilogue goes in FRONT of the period's pieces, so that the plain vmcnt(4) covers its stores
                cmp_phase<false>(acc, a, b);
                epilogue(e);
                issue_next(j);
                wait_pieces(j);
                if (j + 1 < J) read_operands(j + 1);
                GPP_BARRIER();
                ++j;
            }
        } else {
            for (int e = 0; e < my_tiles; ++e) {
                if (e > 0) epilogue(e - 1);                  // beside G0's compute part; in front of this period's pieces
                issue_next(j);
                read_operands(j);
                cmp_phase<true>(acc, a, b);
                wait_pieces(j);
                GPP_BARRIER();
                ++j;
                for (int h = 1; h < HS; ++h) {
                    GPP_STAMP(c0);
                    issue_next(j);
                    read_operands(j);
                    GPP_STAMP(c1);
                    cmp_phase<false>(acc, a, b);
                    GPP_STAMP(c2);
                    wait_pieces(j);
                    GPP_BARRIER();
                    GPP_STAMP(t0);
                    GPP_ACC(clk[3] += c1 - c0; clk[5] += c2 - c1; clk[4] += t0 - c2; ++clk[7]; ++clk[0]);
                    ++j;
                }
            }
            epilogue(my_tiles - 1);
        }
    } else if (group == 0) {
        mem_phase(0);
        GPP_BARRIER();
        for (int e = 0; e < my_tiles; ++e) {
            cmp_phase<true>(acc, a, b);
            GPP_BARRIER();
            mem_phase(j + 1);
            GPP_BARRIER();
            ++j;
            for (int h = 1; h < HS - 1; ++h) {
                GPP_STAMP(c0);
                cmp_phase<false>(acc, a, b);
                GPP_STAMP(c1);
                GPP_BARRIER();
                GPP_STAMP(c2);
                GPP_ACC(clk[5] += c1 - c0; clk[6] += c2 - c1; ++clk[7]);
                mem_phase(j + 1);
                GPP_BARRIER();
                ++j;
            }
            cmp_phase<false>(acc, a, b);
            GPP_BARRIER();
            GPP_STAMP(t0);
            epilogue(e);
            GPP_STAMP(t1);
            if (j + 1 < J) mem_phase(j + 1);
            GPP_STAMP(t2);
            GPP_BARRIER();
            GPP_STAMP(t3);
            GPP_ACC(clk[2] += t1 - t0; clk[3] += t2 - t1; clk[4] += t3 - t2; ++clk[0]);
            ++j;
        }
    } else {
        GPP_BARRIER();
        for (int e = 0; e < my_tiles; ++e) {
            GPP_STAMP(t0);
            if (e > 0) epilogue(e - 1);
            GPP_STAMP(t1);
            mem_phase(j);
            GPP_STAMP(t2);
            GPP_BARRIER();
            GPP_STAMP(t3);
            GPP_ACC(if (e > 0) { clk[2] += t1 - t0; clk[3] += t2 - t1; clk[4] += t3 - t2; ++clk[0]; });
            cmp_phase<true>(acc, a, b);
            GPP_BARRIER();
            ++j;
            for (int h = 1; h < HS; ++h) {
                mem_phase(j);
                GPP_BARRIER();
                GPP_STAMP(c0);
                cmp_phase<false>(acc, a, b);
                GPP_STAMP(c1);
                GPP_BARRIER();
                GPP_STAMP(c2);
                GPP_ACC(clk[5] += c1 - c0; clk[6] += c2 - c1; ++clk[7]);
                ++j;
            }
        }
        epilogue(my_tiles - 1);
    }
#ifdef SQE_PHASE_STAMPS
    if ((p.pp_dbg & 4) && blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0) {
        clk[1] = __builtin_readcyclecounter() - k0;
        for (int i = 0; i < 8; ++i) g_gemm_clk[wave >> 2][i] = clk[i];
    }
#endif
}

template <int EPI>
int launch_gemm_pp(const GemmArgs& a, int t_pad, int cu_count, hipStream_t stream, int* splits_out = nullptr) {
    GemmArgs p = a;
    p.n_tiles = a.N / 256;
    p.t_tiles = t_pad / 256;
    p.splits = 1; p.split_stride = 0;
    static const int stagger = [] { const char* e = knob_env("SQE_GEMM_STAGGER"); return e ? atoi(e) : 1; }();   // knobs build: A/B
    p.pp_stagger = stagger;
    static const int ppdbg = [] { const char* e = knob_env("SQE_GEMM_DBG"); return e ? atoi(e) : 0; }();
    p.pp_dbg = ppdbg;
    const int tiles = p.n_tiles * p.t_tiles;
    auto kern = gemm_pp_kernel<EPI>;
    SQE_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), gpp::LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(std::min(tiles, cu_count)), dim3(512), gpp::LDS_BYTES, stream, p);
    SQE_HIP(hipGetLastError());
#ifdef SQE_PHASE_STAMPS
    if (p.pp_dbg & 4) {
        static int printed[3] = {0, 0, 0};
        if (printed[EPI]++ == 30) {                      // one launch well after warm-up, per epilogue kind
            unsigned long long h[2][8];
            SQE_HIP(hipStreamSynchronize(stream));
            SQE_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gemm_clk), sizeof(h)));
            for (int g = 0; g < 2; ++g) {
                const double t = (double)std::max<unsigned long long>(h[g][0], 1), n = (double)std::max<unsigned long long>(h[g][7], 1);
                fprintf(stderr, "[sqe dbg] gemm_pp<%d> N=%d K=%d group %d: kernel %llu cycles, %llu tile epilogues: epilogue %.0f, memory phase after it %.0f, "
                                "barrier %.0f | steady compute phase %.0f + barrier %.0f (%llu phases)\n",
                        EPI, p.N, p.K, g, h[g][1], h[g][0], h[g][2] / t, h[g][3] / t, h[g][4] / t, h[g][5] / n, h[g][6] / n, h[g][7]);
            }
        }
    }
#endif
    if (EPI == EPI_RESID && splits_out) *splits_out = 0;      // 0 partial sums: one bf16 row
    return SQE_OK;
}

// Small-batch form: LDS ring with counted waits (pieces issued through lds_dma16 so hipcc adds no vmcnt(0) of its own),
// optional split-K.  With a few hundred to a few thousand tokens a GEMM has about as many tiles as the chip has CUs:
// what decides its time is how many ROUNDS of tiles the grid needs and what one tile's K step costs, so the tile is
// picked per call from a menu (r03; r02 had 128 x 128 only: at 64 x 32 tokens the QKV GEMM made 384 tiles = two
// rounds, the second half empty, and ran at 0.46 PFLOP/s):
//     features x tokens   stage     ring   pieces per wave
//     (2 FM 16) x (4 FN 16)
//      64 x  64           16 KiB     4        2
//     128 x  64           24 KiB     4        3
//     192 x  64           32 KiB     4        4
//     256 x  64           40 KiB     3        5
//     128 x 128           32 KiB     4        4        (the r02 tile)
//     192 x 128           40 KiB     3        5
//     256 x 128           48 KiB     3        6
// ring_pick() takes the shape with the smallest  rounds x K steps x max(MFMA cycles, stage bytes / 28 B per cycle)
// (a CU takes in ~55 GB/s of L2-resident operands through LDS-DMA: MI355X_MICROARCH.md, indexed rows table).  The K
// loop of the two N = hidden GEMMs may also be cut into `splits` workgroups whose fp32 partial sums the LayerNorm
// kernels add up (deterministic, no atomics).
template <int EPI, int FM, int FN, int NST>
__global__ __launch_bounds__(512) void gemm_ring_kernel(GemmArgs p) {
    constexpr int BNW = 2 * FM * 16, BT = 4 * FN * 16;
    constexpr int STAGE = (BNW + BT) * ROWB;
    constexpr int PW = BNW / 8, PX = BT / 8;              // 1-KiB pieces (8 rows) of a stage: weights, then tokens
    constexpr int PIECES = (PW + PX) / 8;                 // per wave per stage
    static_assert((PW + PX) % 8 == 0, "every wave issues the same number of pieces");
    static_assert(NST * STAGE <= 160 * 1024, "LDS budget");
    constexpr int IN_FLIGHT = PIECES * (NST - 2);         // pieces the wait at the end of a K step leaves in flight
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int tiles = p.n_tiles * p.t_tiles;
    // Workgroups b and b + 8 share an XCD and its 4 MiB L2 (speed only).  When the tile grid cuts into 4 x 2 patches, XCD x
    // takes patch (x % 4, x / 4) -- a quarter of the weight tiles against half of the token tiles, all K slices of a tile
    // together -- so that what its 32 workgroups stream is a few MB that stay in its L2; a linear walk gives every XCD two
    // weight tiles against ALL tokens (at 64 x 32 tokens: 4.8 MB per XCD, served from the Infinity Cache at half the rate).
    int split, nt, tt;
    {
        const int G = gridDim.x;
        const int nq = p.n_tiles / 4, tq = p.t_tiles / 2;
        if (p.xcd_patches && (p.n_tiles & 3) == 0 && (p.t_tiles & 1) == 0 && (G & 7) == 0) {
            const int x = blockIdx.x & 7, j = blockIdx.x >> 3;          // j < nq * tq * splits
            const int per = nq * tq;
            split = j / per;
            const int jj = j - split * per;
            nt = (x & 3) * nq + jj % nq;
            tt = (x >> 2) * tq + jj / nq;
        } else {
            split = blockIdx.x / tiles;
            const int tile = blockIdx.x % tiles;
            nt = tile % p.n_tiles;
            tt = tile / p.n_tiles;
        }
    }
    const int n0 = nt * BNW, t0 = tt * BT;
    const size_t ld = (size_t)p.K * 2;
    const int KS = p.K / 64 / p.splits;
    const char* wbase = reinterpret_cast<const char*>(p.W) + (size_t)n0 * ld + (size_t)split * KS * ROWB;
    const char* xbase = reinterpret_cast<const char*>(p.X) + (size_t)t0 * ld + (size_t)split * KS * ROWB;

    // this wave's pieces: piece q = wave + 8 i of the stage; q < PW: weight rows 8 q .., else token rows 8 (q - PW) ..
    // (per-lane source: row 8 q' + (lane >> 3), 16-byte chunk (lane & 7) ^ swizzle(row))
    const char* src[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int q = wave + 8 * i;
        const bool is_w = q < PW;
        const int r = (is_w ? q : q - PW) * 8 + (lane >> 3);
        src[i] = (is_w ? wbase : xbase) + (size_t)r * ld + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
    }
    auto issue = [&](int ks) {
        char* buf = smem + (ks % NST) * STAGE;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) lds_dma16(src[i] + (size_t)ks * ROWB, buf + (wave + 8 * i) * 1024);
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int s = 0; s < NST - 1 && s < KS; ++s) issue(s);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    for (int ks = 0; ks < KS; ++ks) {
        const bool more = ks + NST - 1 < KS;
        if (more) issue(ks + NST - 1);        // into the buffer K step ks - 1 used (all waves passed its barrier)
        const char* tA = smem + (ks % NST) * STAGE;
        const char* tB = tA + BNW * ROWB;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[FM], b[FN];
            const int c = kk * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < FM; ++i) a[i] = frag(tA, wm * (FM * 16) + i * 16 + (lane & 15), c);
#pragma unroll
            for (int j = 0; j < FN; ++j) b[j] = frag(tB, wn * (FN * 16) + j * 16 + (lane & 15), c);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        // K step ks + 1 landed; the younger stages stay in flight
        if (!more) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if (IN_FLIGHT == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else if (IN_FLIGHT == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
        else if (IN_FLIGHT == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else if (IN_FLIGHT == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        static_assert(IN_FLIGHT == 4 || IN_FLIGHT == 5 || IN_FLIGHT == 6 || IN_FLIGHT == 8, "a counted wait exists for this shape");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue (as gemm_bf16_kernel; split-K partial sums for EPI_RESID)
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int n = n0 + wm * (FM * 16) + i * 16 + (lane >> 4) * 4;
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (EPI != EPI_F32 && !(EPI == EPI_RESID && split != 0)) b4 = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int t = t0 + wn * (FN * 16) + j * 16 + (lane & 15);
            if (t >= p.T) continue;
            float v0 = acc[i][j][0] + b4.x, v1 = acc[i][j][1] + b4.y, v2 = acc[i][j][2] + b4.z, v3 = acc[i][j][3] + b4.w;
            const size_t o = (size_t)t * p.N + n;
            if (EPI == EPI_F32) {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + o) = make_float4(v0, v1, v2, v3);
            } else if (EPI == EPI_RESID) {
                if (split == 0) {
                    const uint2 r2 = *reinterpret_cast<const uint2*>(p.resid + o);
                    v0 += bf16_to_f32((bf16_t)(r2.x & 0xffff)); v1 += bf16_to_f32((bf16_t)(r2.x >> 16));
                    v2 += bf16_to_f32((bf16_t)(r2.y & 0xffff)); v3 += bf16_to_f32((bf16_t)(r2.y >> 16));
                }
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (size_t)split * p.split_stride + o) =
                    make_float4(v0, v1, v2, v3);
            } else {
                if (EPI == EPI_GELU) { v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3); }
                uint2 w2;
                w2.x = pack_bf16x2(v0, v1);
                w2.y = pack_bf16x2(v2, v3);
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + o) = w2;
            }
        }
    }
}

struct RingShape { int fm, fn, nst; };
constexpr RingShape RING_MENU[7] = {{2, 1, 4}, {4, 1, 4}, {6, 1, 4}, {8, 1, 3}, {4, 2, 4}, {6, 2, 3}, {8, 2, 3}};

// estimated cycles of the whole GEMM with shape s and `splits` K slices (see the table above)
inline double ring_cost(const RingShape& s, int N, int K, int t_pad, int splits, int cu_count) {
    const int bnw = 2 * s.fm * 16, bt = 4 * s.fn * 16;
    if (N % bnw != 0 || t_pad % bt != 0) return 1e30;
    const long tiles = (long)(N / bnw) * (t_pad / bt) * splits;
    const long rounds = (tiles + cu_count - 1) / cu_count;
    const double mfma = s.fm * s.fn * 2 * 16.0;                       // cycles of one K step on a wave's SIMD
    const double mem = (bnw + bt) * 128.0 / 28.0;                     // LDS-DMA intake of a CU: ~28 B per cycle
    const double step = (mfma > mem ? mfma : mem) + 150.0;            // + barrier and issue
    const double fill = 2500.0 + (s.nst - 1) * 0.0;                   // first stages: one memory round trip
    return rounds * ((double)(K / 64 / splits) * step + fill + (splits > 1 ? 600.0 : 0.0));
}

template <int EPI, int FM, int FN, int NST>
int launch_gemm_ring_shape(const GemmArgs& p, int tiles, hipStream_t stream) {
    constexpr int LDS = NST * (2 * FM * 16 + 4 * FN * 16) * ROWB;
    auto kern = gemm_ring_kernel<EPI, FM, FN, NST>;
    SQE_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS));
    hipLaunchKernelGGL(kern, dim3(tiles * p.splits), dim3(512), LDS, stream, p);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

template <int EPI>
int launch_gemm_ring(const GemmArgs& a, int t_pad, int cu_count, size_t split_stride, int* splits_out, hipStream_t stream) {
    GemmArgs p = a;
    // knobs build: SQE_ENC_RING=0 pins the r02 tile (128 x 128), A/B of the shape menu
    static const bool menu_on = [] { const char* e = knob_env("SQE_ENC_RING"); return !(e && e[0] == '0'); }();
    int best = 4, best_splits = 1;
    double best_cost = 1e30;
    const int ks = a.K / 64;
    for (int m = 0; m < 7; ++m) {
        if (!menu_on && m != 4) continue;
        for (int splits = 1; splits <= (EPI == EPI_RESID ? 4 : 1); splits *= 2) {
            if (splits > 1 && (ks % splits != 0 || ks / splits < 4)) continue;
            const double c = ring_cost(RING_MENU[m], a.N, a.K, t_pad, splits, cu_count);
            if (c < best_cost) { best_cost = c; best = m; best_splits = splits; }
        }
    }
    if (best_cost >= 1e30) return fail(SQE_ERR_INVALID, "encoder gemm: no ring tile fits N / padded token count");
    if (EPI == EPI_RESID) {
        // The two N = hidden GEMMs write fp32 partial sums that the LayerNorm kernel adds up: every extra K slice is
        // another T x N x 4 bytes written and read, which the cycle model above does not see.  Measured (tools/ring_tune.py,
        // whole encoder, 64 x 32 and 64 x 16 tokens, profiles/r03_configs/ring_tune_*.jsonl): out-proj (K = hidden) is
        // fastest on 64 x 64 tiles with NO split (3.03 -> 2.77 ms and 2.06 -> 1.92 ms against the model's 256 x 128 / 4
        // slices), FFN-down (K = 4 x hidden) on 128 x 128 / 2 slices from 2,048 tokens on and 128 x 64 / 2 slices below.
        // These rules cover 1,024-2,048 padded tokens (what was measured); elsewhere the model decides.
        if (t_pad >= 1024 && t_pad <= 2048 && a.N % 128 == 0) {
            if (a.K <= a.N) { best = 0; best_splits = 1; }
            else if (t_pad >= 2048) { best = 4; best_splits = ks % 2 == 0 && ks / 2 >= 4 ? 2 : 1; }
            else { best = 1; best_splits = ks % 2 == 0 && ks / 2 >= 4 ? 2 : 1; }
        }   // other token counts: the model's choice (1 x 128 tokens: 1.84 -> 1.28 ms against the r02 rule; 64 x 128: 7.63 -> 7.34)
    }
    {
        // knobs build, tuning runs (tools/ring_tune.sh): SQE_RING_FORCE_<EPI>_K<K>="menu index:splits" pins the choice for
        // the GEMMs of that epilogue and depth; SQE_RING_XCD=0 switches the XCD-aware tile walk off
        char name[64];
        snprintf(name, sizeof name, "SQE_RING_FORCE_%d_K%d", (int)EPI, a.K);
        const char* e = knob_env(name);
        int m = -1, sp = 1;
        if (e && sscanf(e, "%d:%d", &m, &sp) >= 1 && m >= 0 && m < 7 && ring_cost(RING_MENU[m], a.N, a.K, t_pad, 1, cu_count) < 1e30 &&
            (sp == 1 || (EPI == EPI_RESID && (sp == 2 || sp == 4) && ks % sp == 0))) {
            best = m;
            best_splits = sp;
        }
    }
    const RingShape sh = RING_MENU[best];
    p.n_tiles = a.N / (2 * sh.fm * 16);
    p.t_tiles = t_pad / (4 * sh.fn * 16);
    p.splits = best_splits;
    p.split_stride = split_stride;
    {
        static const bool xcd_off = [] { const char* e = knob_env("SQE_RING_XCD"); return e && e[0] == '0'; }();
        p.xcd_patches = xcd_off ? 0 : 1;
    }
    if (splits_out) *splits_out = best_splits;
    const int tiles = p.n_tiles * p.t_tiles;
    switch (best) {
        case 0: return launch_gemm_ring_shape<EPI, 2, 1, 4>(p, tiles, stream);
        case 1: return launch_gemm_ring_shape<EPI, 4, 1, 4>(p, tiles, stream);
        case 2: return launch_gemm_ring_shape<EPI, 6, 1, 4>(p, tiles, stream);
        case 3: return launch_gemm_ring_shape<EPI, 8, 1, 3>(p, tiles, stream);
        case 4: return launch_gemm_ring_shape<EPI, 4, 2, 4>(p, tiles, stream);
        case 5: return launch_gemm_ring_shape<EPI, 6, 2, 3>(p, tiles, stream);
        default: return launch_gemm_ring_shape<EPI, 8, 2, 3>(p, tiles, stream);
    }
}

// A handful of tokens (T <= 64: one query, as the reference issues it): every GEMM of the layer is a few MB of
// weights against at most four 16-token groups, i.e. one memory round trip if the whole chip pulls on it at once,
// and the 128-row tiles above give it 8-32 workgroups.  Here a workgroup owns 16 output features: its 4 waves
// split K, each lane reads its MFMA fragments straight from global memory (W row n0 + (lane & 15), 16 B at
// k + 32 i + 8 (lane >> 4): 64 contiguous bytes per row per instruction; X rows the same way, L2-resident), no
// LDS staging, the four partial sums meet in LDS and wave j finishes token group j.  N / 16 workgroups (x the
// split-K factor for the two N = hidden GEMMs, whose partial sums the LayerNorm kernels already add).
template <int EPI>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs p) {
    __shared__ float4 red[4][4][64];          // [K-split wave][token group][lane]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int groups = p.N >> 4;
    const int split = blockIdx.x / groups;
    const int n0 = (blockIdx.x % groups) << 4;
    const int nt = (p.T + 15) >> 4;           // 16-token groups, <= 4
    const int kwave = p.K / p.splits / 4;     // K elements per wave, a multiple of 128
    const int kbeg = split * (p.K / p.splits) + wave * kwave;
    const int r = lane & 15, c = lane >> 4;
    const bf16_t* wrow = p.W + (size_t)(n0 + r) * p.K + kbeg + c * 8;
    const bf16_t* xrow = p.X + (size_t)r * p.K + kbeg + c * 8;
    const size_t xgroup = (size_t)16 * p.K;

    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // two 128-wide steps per pass, the loads of both issued before the first MFMA (kwave is 128 or 256 for the
    // BERT-large shapes: one pass, one memory round trip)
    for (int k = 0; k < kwave; k += 256) {
        bf16x8 a[2][4], b[2][4][4];
        const bool two = k + 128 < kwave;
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (h == 0 || two) {
#pragma unroll
                for (int i = 0; i < 4; ++i) a[h][i] = *reinterpret_cast<const bf16x8*>(wrow + k + h * 128 + i * 32);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < nt) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            b[h][j][i] = *reinterpret_cast<const bf16x8*>(xrow + j * xgroup + k + h * 128 + i * 32);
                    }
            }
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (h == 0 || two) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < nt) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[h][i], b[h][j][i], acc[j], 0, 0, 0);
                    }
            }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < nt) red[wave][j][lane] = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
    __syncthreads();
    if (wave >= nt) return;
    // wave j: token group j, partial sums added in wave order (deterministic)
    const int j = wave;
    float4 v = red[0][j][lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const float4 u = red[w][j][lane];
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    const int t = j * 16 + r;
    if (t >= p.T) return;
    const int n = n0 + c * 4;
    if (!(EPI == EPI_RESID && split != 0)) {
        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
        v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
    }
    const size_t o = (size_t)t * p.N + n;
    if (EPI == EPI_RESID) {
        if (split == 0) {
            const uint2 r2 = *reinterpret_cast<const uint2*>(p.resid + o);
            v.x += bf16_to_f32((bf16_t)(r2.x & 0xffff)); v.y += bf16_to_f32((bf16_t)(r2.x >> 16));
            v.z += bf16_to_f32((bf16_t)(r2.y & 0xffff)); v.w += bf16_to_f32((bf16_t)(r2.y >> 16));
        }
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (size_t)split * p.split_stride + o) = v;
    } else {
        if (EPI == EPI_GELU) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
        uint2 w2;
        w2.x = pack_bf16x2(v.x, v.y);
        w2.y = pack_bf16x2(v.z, v.w);
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + o) = w2;
    }
}

template <int EPI>
int launch_gemm_skinny(const GemmArgs& a, int cu_count, size_t split_stride, int* splits_out, hipStream_t stream) {
    GemmArgs p = a;
    int splits = 1;
    if (EPI == EPI_RESID)
        while (splits < 4 && a.K % (512 * splits * 2) == 0 && (a.N / 16) * splits * 2 <= cu_count) splits *= 2;
    p.splits = splits;
    p.split_stride = split_stride;
    if (splits_out) *splits_out = splits;
    hipLaunchKernelGGL(gemm_skinny_kernel<EPI>, dim3((a.N / 16) * splits), dim3(256), 0, stream, p);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

// 256x256 tiles when there are enough tokens to fill the chip with them, the 128x128 ring kernel otherwise.
// *splits_out = number of fp32 partial sums written (EPI_RESID), `split_stride` floats apart; 0 = the output is
// ONE bf16 row per token (the persistent kernel).
template <int EPI>
int launch_gemm(const GemmArgs& a, int t_pad, int cu_count, hipStream_t stream, size_t split_stride = 0,
                int* splits_out = nullptr) {
    if (a.N % 128 != 0 || a.K % 64 != 0) return fail(SQE_ERR_INVALID, "encoder gemm: N % 128 or K % 64");
    if (splits_out) *splits_out = 1;
    if (EPI != EPI_F32 && a.T <= 64 && a.K % 512 == 0) {
        static const bool off = [] { const char* e = knob_env("SQE_ENC_SKINNY"); return e && e[0] == '0'; }();
        if (!off) return launch_gemm_skinny<EPI>(a, cu_count, split_stride, splits_out, stream);
    }
    const bool big = a.N % 256 == 0 && t_pad % 256 == 0 && (int64_t)(a.N / 256) * (t_pad / 256) >= cu_count;
    if (big) {
        static const bool old_form = [] { const char* e = knob_env("SQE_ENC_GEMM_V0"); return e && e[0] == '1'; }();
        // the ping-pong kernel (r02: QKV 220 -> 207 us, out-proj + FFN-down 178 -> 172 us per call at 64 x 512 tokens,
        // FFN-up + GELU equal); SQE_ENC_GEMM=0 in a knobs build picks the two-stage persistent kernel it replaced
        static const int form = [] { const char* e = knob_env("SQE_ENC_GEMM"); return e ? atoi(e) : 1; }();
        if (!old_form && form == 1 && a.N <= gpp::MAX_BIAS_N) return launch_gemm_pp<EPI>(a, t_pad, cu_count, stream, splits_out);
        if (!old_form) return launch_gemm_persistent<EPI>(a, t_pad, cu_count, stream, splits_out);
        GemmArgs p = a;
        p.splits = 1; p.split_stride = 0; p.t_tiles = 0;
        return launch_gemm_cfg<8, 4, EPI>(p, t_pad, stream);
    }
    return launch_gemm_ring<EPI>(a, t_pad, cu_count, split_stride, splits_out, stream);
}

// scores[t][n] = <X[t], W[n]> in fp32 (bf16 operands, rows K elements apart): the IVF coarse quantiser
int scores_gemm(const bf16_t* W, const bf16_t* X, float* out, int N, int K, int T, int t_pad, int cu_count, hipStream_t stream) {
    GemmArgs a;
    a.W = W; a.X = X; a.bias = nullptr; a.resid = nullptr; a.out = out; a.N = N; a.K = K; a.T = T; a.n_tiles = 0;
    return launch_gemm_ring<EPI_F32>(a, t_pad, cu_count, 0, nullptr, stream);
}

// ------------------------------------------------------------------ LayerNorm family
// one wave per row; H % 4 == 0; two passes over registers-or-cache (row <= 16 KiB)
// row element i = sum over the nsplit split-K partial sums (stride floats apart); nsplit = 1: plain fp32 row;
// nsplit = 0: the row is bf16 (`row` then points at bf16 data: ln_row() does the addressing)
__device__ __forceinline__ float4 ln_load(const float* row, int i, int nsplit, size_t stride) {
    if (nsplit == 0) {
        const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(row) + i);
        return make_float4(bf16_to_f32((bf16_t)(w.x & 0xffff)), bf16_to_f32((bf16_t)(w.x >> 16)),
                           bf16_to_f32((bf16_t)(w.y & 0xffff)), bf16_to_f32((bf16_t)(w.y >> 16)));
    }
    float4 v = *reinterpret_cast<const float4*>(row + i);
    if (nsplit == 1) return v;
    if (nsplit <= 4) {
        // (r04c) up to four partial sums, their loads issued together: a loop of `nsplit` trips is `nsplit` dependent round trips (the
        // LayerNorm of 16 tokens took 6.8 us, profiles/r04_configs/enc_1x16_forward_trace.txt).  A partial sum that does not exist is
        // read from partial sum 0 (a hit) and not added: no branch around the loads, the same additions in the same order.
        float4 w[3];
#pragma unroll
        for (int s = 1; s < 4; ++s) w[s - 1] = *reinterpret_cast<const float4*>(row + (size_t)(s < nsplit ? s : 0) * stride + i);
#pragma unroll
        for (int s = 1; s < 4; ++s)
            if (s < nsplit) { v.x += w[s - 1].x; v.y += w[s - 1].y; v.z += w[s - 1].z; v.w += w[s - 1].w; }
        return v;
    }
    for (int s = 1; s < nsplit; ++s) {
        const float4 w = *reinterpret_cast<const float4*>(row + (size_t)s * stride + i);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    return v;
}
// first element of row `r` of the pre-LayerNorm buffer in either form
__device__ __forceinline__ const float* ln_row(const float* in, size_t r, int H, int nsplit) {
    return nsplit == 0 ? reinterpret_cast<const float*>(reinterpret_cast<const bf16_t*>(in) + r * H) : in + r * H;
}

__device__ __forceinline__ void ln_stats(const float* row, int H, int lane, float& mean, float& rstd, float eps,
                                         int nsplit = 1, size_t stride = 0) {
    float s = 0.f;
    for (int i = lane * 4; i < H; i += 256) {
        const float4 v = ln_load(row, i, nsplit, stride);
        s += v.x + v.y + v.z + v.w;
    }
    mean = wave_sum(s) / (float)H;
    float q = 0.f;
    for (int i = lane * 4; i < H; i += 256) {
        const float4 v = ln_load(row, i, nsplit, stride);
        const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
        q += a * a + b * b + c * c + d * d;
    }
    rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
}

// Rows of at most 1024 elements (H % 256 == 0) are read ONCE into registers -- one memory round trip instead of
// three; sums run in the order ln_stats uses, so both forms give the same bits.
__device__ __forceinline__ bool ln_row_regs(const float* row, int H, int lane, float eps, int nsplit, size_t stride,
                                            float4 (&v)[4], float& mean, float& rstd) {
    const int nit = H >> 8;
    if ((H & 255) != 0 || nit > 4) return false;
#pragma unroll
    for (int it = 0; it < 4; ++it)
        if (it < nit) v[it] = ln_load(row, lane * 4 + it * 256, nsplit, stride);
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it)
        if (it < nit) s += v[it].x + v[it].y + v[it].z + v[it].w;
    mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it)
        if (it < nit) {
            const float a = v[it].x - mean, b = v[it].y - mean, c = v[it].z - mean, d = v[it].w - mean;
            q += a * a + b * b + c * c + d * d;
        }
    rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
    return true;
}

__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ in, const float* __restrict__ g,
                                                        const float* __restrict__ b, bf16_t* __restrict__ out,
                                                        int T, int H, float eps, int nsplit, size_t stride) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const float* x = ln_row(in, (size_t)row, H, nsplit);
    float mean, rstd;
    float4 vr[4];
    if (ln_row_regs(x, H, lane, eps, nsplit, stride, vr, mean, rstd)) {
#pragma unroll
        for (int it = 0; it < 4; ++it)
            if (it < (H >> 8)) {
                const int i = lane * 4 + it * 256;
                const float4 v = vr[it];
                const float4 gg = *reinterpret_cast<const float4*>(g + i);
                const float4 bb = *reinterpret_cast<const float4*>(b + i);
                uint2 w;
                w.x = pack_bf16x2((v.x - mean) * rstd * gg.x + bb.x, (v.y - mean) * rstd * gg.y + bb.y);
                w.y = pack_bf16x2((v.z - mean) * rstd * gg.z + bb.z, (v.w - mean) * rstd * gg.w + bb.w);
                *reinterpret_cast<uint2*>(out + (size_t)row * H + i) = w;
            }
        return;
    }
    ln_stats(x, H, lane, mean, rstd, eps, nsplit, stride);
    for (int i = lane * 4; i < H; i += 256) {
        const float4 v = ln_load(x, i, nsplit, stride);
        const float4 gg = *reinterpret_cast<const float4*>(g + i);
        const float4 bb = *reinterpret_cast<const float4*>(b + i);
        uint2 w;
        w.x = pack_bf16x2((v.x - mean) * rstd * gg.x + bb.x, (v.y - mean) * rstd * gg.y + bb.y);
        w.y = pack_bf16x2((v.z - mean) * rstd * gg.z + bb.z, (v.w - mean) * rstd * gg.w + bb.w);
        *reinterpret_cast<uint2*>(out + (size_t)row * H + i) = w;
    }
}

// E7: LayerNorm of the CLS row of every sequence, fp32 out [B, H]
__global__ __launch_bounds__(256) void pool_ln_kernel(const float* __restrict__ in, const float* __restrict__ g,
                                                      const float* __restrict__ b, float* __restrict__ out,
                                                      int B, int S, int H, float eps, int nsplit, size_t stride) {
    const int lane = threadIdx.x & 63;
    const int seq = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seq >= B) return;
    const float* x = ln_row(in, (size_t)seq * S, H, nsplit);
    float mean, rstd;
    float4 vr[4];
    if (ln_row_regs(x, H, lane, eps, nsplit, stride, vr, mean, rstd)) {
#pragma unroll
        for (int it = 0; it < 4; ++it)
            if (it < (H >> 8)) {
                const int i = lane * 4 + it * 256;
                const float4 v = vr[it];
                const float4 gg = *reinterpret_cast<const float4*>(g + i);
                const float4 bb = *reinterpret_cast<const float4*>(b + i);
                *reinterpret_cast<float4*>(out + (size_t)seq * H + i) =
                    make_float4((v.x - mean) * rstd * gg.x + bb.x, (v.y - mean) * rstd * gg.y + bb.y,
                                (v.z - mean) * rstd * gg.z + bb.z, (v.w - mean) * rstd * gg.w + bb.w);
            }
        return;
    }
    ln_stats(x, H, lane, mean, rstd, eps, nsplit, stride);
    for (int i = lane * 4; i < H; i += 256) {
        const float4 v = ln_load(x, i, nsplit, stride);
        const float4 gg = *reinterpret_cast<const float4*>(g + i);
        const float4 bb = *reinterpret_cast<const float4*>(b + i);
        *reinterpret_cast<float4*>(out + (size_t)seq * H + i) =
            make_float4((v.x - mean) * rstd * gg.x + bb.x, (v.y - mean) * rstd * gg.y + bb.y,
                        (v.z - mean) * rstd * gg.z + bb.z, (v.w - mean) * rstd * gg.w + bb.w);
    }
}

// E1: x = LN(word[id] + pos[s] + type[0]); the fp32 sum goes through `scratch` (one row per wave slot)
__global__ __launch_bounds__(256) void embed_ln_kernel(const int32_t* __restrict__ ids, const bf16_t* __restrict__ word,
                                                       const bf16_t* __restrict__ pos, const bf16_t* __restrict__ type,
                                                       const float* __restrict__ g, const float* __restrict__ b,
                                                       float* __restrict__ scratch, bf16_t* __restrict__ out,
                                                       int T, int S, int H, int vocab, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    int id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const int s = row % S;
    float* tmp = scratch + (size_t)row * H;
    for (int i = lane * 4; i < H; i += 256) {
        const uint2 w = *reinterpret_cast<const uint2*>(word + (size_t)id * H + i);
        const uint2 p = *reinterpret_cast<const uint2*>(pos + (size_t)s * H + i);
        const uint2 t = *reinterpret_cast<const uint2*>(type + i);
        float4 v;
        v.x = bf16_to_f32((bf16_t)(w.x & 0xffff)) + bf16_to_f32((bf16_t)(p.x & 0xffff)) + bf16_to_f32((bf16_t)(t.x & 0xffff));
        v.y = bf16_to_f32((bf16_t)(w.x >> 16)) + bf16_to_f32((bf16_t)(p.x >> 16)) + bf16_to_f32((bf16_t)(t.x >> 16));
        v.z = bf16_to_f32((bf16_t)(w.y & 0xffff)) + bf16_to_f32((bf16_t)(p.y & 0xffff)) + bf16_to_f32((bf16_t)(t.y & 0xffff));
        v.w = bf16_to_f32((bf16_t)(w.y >> 16)) + bf16_to_f32((bf16_t)(p.y >> 16)) + bf16_to_f32((bf16_t)(t.y >> 16));
        *reinterpret_cast<float4*>(tmp + i) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    float mean, rstd;
    ln_stats(tmp, H, lane, mean, rstd, eps);
    for (int i = lane * 4; i < H; i += 256) {
        const float4 v = *reinterpret_cast<const float4*>(tmp + i);
        const float4 gg = *reinterpret_cast<const float4*>(g + i);
        const float4 bb = *reinterpret_cast<const float4*>(b + i);
        uint2 w;
        w.x = pack_bf16x2((v.x - mean) * rstd * gg.x + bb.x, (v.y - mean) * rstd * gg.y + bb.y);
        w.y = pack_bf16x2((v.z - mean) * rstd * gg.z + bb.z, (v.w - mean) * rstd * gg.w + bb.w);
        *reinterpret_cast<uint2*>(out + (size_t)row * H + i) = w;
    }
}

// ------------------------------------------------------------------ attention (head_dim = 64)
// grid = B * heads * ceil(S / 64); 256 threads = 4 waves x 16 query rows.
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
// issue only: the caller waits once for a batch of these (lds_tr_wait8)
__device__ __forceinline__ u32x2 lds_tr_b64_issue(const char* addr) {
    u32x2 v;
    const uint32_t a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)addr;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=&v"(v) : "v"(a) : "memory");
    return v;
}
// the same read at a compile-time byte offset from a per-lane LDS address (the offset field of the DS instruction)
template <int OFF>
__device__ __forceinline__ u32x2 lds_tr_b64_issue_at(uint32_t lds_addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(v) : "v"(lds_addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ void lds_tr_wait8(u32x2 (&x)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
                 :
                 : "memory");
}

// reductions over the four lane groups g of a query column c (lanes c, c + 16, c + 32, c + 48): two lane-swap
// instructions of gfx950 (v_permlane16_swap / v_permlane32_swap, VALU) instead of two trips through the LDS crossbar
__device__ __forceinline__ float xg_max(float v) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
    u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xg_sum(float v) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
    u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// NW waves x NQ blocks of 16 query rows per wave = NW * NQ * 16 query rows per workgroup: 64 for short sequences,
// 128 from 128 tokens on (8 waves x 1 block; from 256 tokens on 4 waves x 2 blocks).  A wave's NQ blocks share every K
// fragment and every transposed V fragment it reads from LDS, so LDS traffic per MFMA halves with NQ = 2.
//
// Per 64-key tile and query block: S^T[key][q] = K Q^T (8 MFMAs) leaves the scores of query c in lane (g, c);
// the softmax runs there in base 2 -- t = s * (log2 e / 8), p = exp2(t - m), one v_mul, one v_sub and one
// v_exp per score; keys are masked only in the last, partial tile -- and its probabilities are, as they stand
// in the accumulators, the B operand of O^T[d][q] += V^T P^T (8 MFMAs, A operand = the transposed V reads).
// With O transposed a lane holds output columns of ITS OWN query: the rescale by alpha and the final 1 / l
// need no cross-lane traffic, and a lane stores 4 consecutive output features at a time.
template <int NW, int NQ, int MINW = 1>
__global__ __launch_bounds__(NW * 64, MINW) void attention_kernel(const bf16_t* __restrict__ qkv, const int32_t* __restrict__ lens,
                                                           bf16_t* __restrict__ ctx, int S, int H, int heads) {
    constexpr int QB = NW * NQ * 16;
    constexpr float SCALE_LOG2E = 0.125f * 1.4426950408889634f;        // 1 / sqrt(64) folded into the base-2 exponent
    __shared__ __attribute__((aligned(16))) char sKV[2][2 * 64 * ROWB];    // [buffer][K tile | V tile]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qblocks = (S + QB - 1) / QB;
    const int qb = blockIdx.x % qblocks;
    const int head = (blockIdx.x / qblocks) % heads;
    const int seq = blockIdx.x / (qblocks * heads);
    const int len = min(lens[seq], S);
    if (qb * QB >= len) return;                      // only padded query rows here
    const int g = lane >> 4, c = lane & 15;
    const size_t ld = (size_t)3 * H * 2;             // bytes per token row of qkv
    const char* base = reinterpret_cast<const char*>(qkv) + (size_t)seq * S * ld + (size_t)head * 64 * 2;

    // Q fragments of this wave's NQ x 16 query rows (lane (g, c) holds Q[q = c][d = kk*32 + g*8 .. +8))
    const int q_base = qb * QB + wave * (NQ * 16);
    bf16x8 qf[NQ][2];
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        const int q_row = q_base + n * 16 + c;
        const int qr = q_row < S ? q_row : S - 1;
        const char* qp = base + (size_t)qr * ld;
        qf[n][0] = *reinterpret_cast<const bf16x8*>(qp + (0 * 32 + g * 8) * 2);
        qf[n][1] = *reinterpret_cast<const bf16x8*>(qp + (1 * 32 + g * 8) * 2);
    }
    float m[NQ], l[NQ];                              // running max (scaled, base-2 domain) / sum of query c
    f32x4 o[NQ][4];                                  // O^T: row d = dj*16 + g*4 + r, column = query c
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        m[n] = -INFINITY;
        l[n] = 0.f;
#pragma unroll
        for (int dj = 0; dj < 4; ++dj) o[n][dj] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // K/V tiles are double buffered: tile t + 1 is fetched while tile t is used (one barrier per tile)
    auto issue_kv = [&](int kv0, int buf) {
        // (the builtin form: hipcc knows these are loads; through inline asm it has to assume a store that still
        // reads its address registers and waits on vmcnt before it reuses them -- in the middle of the MFMAs)
        stage_rows<64, NW>(base + (size_t)kv0 * ld + (size_t)H * 2, ld, sKV[buf], wave, lane);                   // K part
        stage_rows<64, NW>(base + (size_t)kv0 * ld + (size_t)2 * H * 2, ld, sKV[buf] + 64 * ROWB, wave, lane);   // V part
    };
    // Per-lane LDS addresses, computed ONCE: every read of the loop is one of these plus a compile-time offset
    // (buffer, key block), so no address arithmetic is left in the loop.  The XOR swizzle of a tile row depends
    // on (row >> 1) & 7 only, which a step of 16 rows (K fragments) or 16 / 32 keys (V blocks) does not change.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)&sKV[0][0];
    const int ksw = (c >> 1) & 7;
    const uint32_t kaddr0 = lds0 + c * ROWB + ((g ^ ksw) << 4);              // K fragment, d = g*8 .. (first 32 of 64)
    const uint32_t kaddr1 = lds0 + c * ROWB + (((4 + g) ^ ksw) << 4);        // d = 32 + g*8 ..
    uint32_t vaddr[4];                                                       // V^T fragment of feature block dj, key 4g + qq
    {
        const int qq = c >> 2, pp = c & 3, key0 = 4 * g + qq, vsw = (key0 >> 1) & 7;
#pragma unroll
        for (int dj = 0; dj < 4; ++dj)
            vaddr[dj] = lds0 + 64 * ROWB + key0 * ROWB + (((dj * 2 + (pp >> 1)) ^ vsw) << 4) + 8 * (pp & 1);
    }
    constexpr int BUFB = 2 * 64 * ROWB;              // bytes per K|V buffer

    // one 64-key tile out of buffer BUF (a compile-time constant: it selects the immediate offsets)
    auto tile = [&](int kv0, auto buf_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        // S^T[key][q] = sum_d K[key][d] Q[q][d]: every K fragment feeds all NQ query blocks
        f32x4 st[NQ][4];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) {
            const bf16x8 k0 = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>((uintptr_t)(kaddr0 + BUF * BUFB + kf * 16 * ROWB));
            const bf16x8 k1 = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>((uintptr_t)(kaddr1 + BUF * BUFB + kf * 16 * ROWB));
#pragma unroll
            for (int n = 0; n < NQ; ++n) {
                st[n][kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[n][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                st[n][kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[n][1], st[n][kf], 0, 0, 0);
            }
        }
        const bool partial = kv0 + 64 > len;         // wave-uniform: only the last tile of a sequence masks keys
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            // scaled scores of query c (keys kf*16 + g*4 + r); the products are canonical numbers, so the maxima
            // need no NaN quieting and pair up into v_max3
            f32x4 t[4];
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) t[kf] = st[n][kf] * SCALE_LOG2E;
            if (partial) {
                int lenx = len;
                asm volatile("; tail tile: mask keys >= len" : "+s"(lenx));   // opaque: nothing of this block is hoisted
#pragma unroll
                for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kv0 + kf * 16 + g * 4 + r >= lenx) t[kf][r] = -INFINITY;
            }
            float mx = fmaxf(fmaxf(t[0][0], t[0][1]), t[0][2]);
            mx = fmaxf(fmaxf(mx, t[0][3]), t[1][0]);
            mx = fmaxf(fmaxf(mx, t[1][1]), t[1][2]);
            mx = fmaxf(fmaxf(mx, t[1][3]), t[2][0]);
            mx = fmaxf(fmaxf(mx, t[2][1]), t[2][2]);
            mx = fmaxf(fmaxf(mx, t[2][3]), t[3][0]);
            mx = fmaxf(fmaxf(mx, t[3][1]), t[3][2]);
            mx = xg_max(fmaxf(mx, t[3][3]));
            const float m_new = fmaxf(m[n], mx);     // finite: key kv0 < len is never masked
            const float alpha = __builtin_amdgcn_exp2f(m[n] - m_new);
            // p = exp2(s * c - m): one packed fma on the raw scores (masked ones come from t: they are -inf there)
            const f32x4 cv = {SCALE_LOG2E, SCALE_LOG2E, SCALE_LOG2E, SCALE_LOG2E};
            const f32x4 nm = {-m_new, -m_new, -m_new, -m_new};
            if (partial) {
                asm volatile("; tail tile: exponents from the masked scores" ::);
#pragma unroll
                for (int kf = 0; kf < 4; ++kf) t[kf] = t[kf] + nm;
            } else {
#pragma unroll
                for (int kf = 0; kf < 4; ++kf) t[kf] = __builtin_elementwise_fma(st[n][kf], cv, nm);
            }
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) {
#pragma unroll
                for (int r = 0; r < 4; ++r) t[kf][r] = __builtin_amdgcn_exp2f(t[kf][r]);
                st[n][kf] = t[kf];
            }
            const f32x4 sv = (t[0] + t[1]) + (t[2] + t[3]);
            const float psum = xg_sum((sv[0] + sv[1]) + (sv[2] + sv[3]));
            l[n] = l[n] * alpha + psum;
            m[n] = m_new;
            // O^T columns are this lane's own query: rescale in place
#pragma unroll
            for (int dj = 0; dj < 4; ++dj) o[n][dj] *= alpha;
        }
        // O^T += V^T P^T : A operand = V by transposed LDS reads (k slot j of lane group g = key
        // 16*(2*kk2 + (j>>2)) + 4*g + (j&3), feature dj*16 + c), B operand = P straight from the S^T accumulators
        auto pv_half = [&](auto kk2_tag) {
            constexpr int KK2 = decltype(kk2_tag)::value;
            constexpr int LO = BUF * BUFB + 32 * KK2 * ROWB, HI = LO + 16 * ROWB;
            // the eight transposed reads of this half go out together and are waited for once
            u32x2 vt[8];
            vt[0] = lds_tr_b64_issue_at<LO>(vaddr[0]); vt[1] = lds_tr_b64_issue_at<HI>(vaddr[0]);
            vt[2] = lds_tr_b64_issue_at<LO>(vaddr[1]); vt[3] = lds_tr_b64_issue_at<HI>(vaddr[1]);
            vt[4] = lds_tr_b64_issue_at<LO>(vaddr[2]); vt[5] = lds_tr_b64_issue_at<HI>(vaddr[2]);
            vt[6] = lds_tr_b64_issue_at<LO>(vaddr[3]); vt[7] = lds_tr_b64_issue_at<HI>(vaddr[3]);
            bf16x8 pb[NQ];
#pragma unroll
            for (int n = 0; n < NQ; ++n) {
                typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
                const u32x4 pk = {pack_bf16x2(st[n][2 * KK2][0], st[n][2 * KK2][1]), pack_bf16x2(st[n][2 * KK2][2], st[n][2 * KK2][3]),
                                  pack_bf16x2(st[n][2 * KK2 + 1][0], st[n][2 * KK2 + 1][1]),
                                  pack_bf16x2(st[n][2 * KK2 + 1][2], st[n][2 * KK2 + 1][3])};
                pb[n] = __builtin_bit_cast(bf16x8, pk);
            }
            lds_tr_wait8(vt);
#pragma unroll
            for (int dj = 0; dj < 4; ++dj) {
                typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
                const u32x4 packed = {vt[2 * dj].x, vt[2 * dj].y, vt[2 * dj + 1].x, vt[2 * dj + 1].y};
                const bf16x8 va = __builtin_bit_cast(bf16x8, packed);
#pragma unroll
                for (int n = 0; n < NQ; ++n) o[n][dj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb[n], o[n][dj], 0, 0, 0);
            }
        };
        pv_half(std::integral_constant<int, 0>{});
        pv_half(std::integral_constant<int, 1>{});
    };
    auto tile_sync = [&]() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    // my pieces of this tile landed, my LDS reads are done
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                                  // everyone's landed; everyone left the other buffer
        __builtin_amdgcn_sched_barrier(0);
    };
    issue_kv(0, 0);
    for (int kv0 = 0; kv0 < len; kv0 += 128) {       // two tiles per trip: the buffer index is a compile-time constant
        tile_sync();
        if (kv0 + 64 < len) issue_kv(kv0 + 64, 1);
        tile(kv0, std::integral_constant<int, 0>{});
        if (kv0 + 64 >= len) break;
        tile_sync();
        if (kv0 + 128 < len) issue_kv(kv0 + 128, 0);
        tile(kv0 + 64, std::integral_constant<int, 1>{});
    }
    // normalise and store: lane (g, c) holds O[q = c][d = dj*16 + g*4 + r], four consecutive features per dj
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        const int q = q_base + n * 16 + c;
        if (q < len) {
            const float inv = 1.0f / l[n];
            bf16_t* dst = ctx + ((size_t)seq * S + q) * H + head * 64 + g * 4;
#pragma unroll
            for (int dj = 0; dj < 4; ++dj) {
                uint2 w;
                w.x = pack_bf16x2(o[n][dj][0] * inv, o[n][dj][1] * inv);
                w.y = pack_bf16x2(o[n][dj][2] * inv, o[n][dj][3] * inv);
                *reinterpret_cast<uint2*>(dst + dj * 16) = w;
            }
        }
    }
}

// ------------------------------------------------------------------ host side
struct DevMem {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevMem() { if (p) (void)hipFree(p); }
    int alloc(size_t n) {
        if (n <= bytes) return SQE_OK;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) return fail(SQE_ERR_OOM, std::string("encoder hipMalloc: ") + hipGetErrorString(e));
        bytes = n;
        return SQE_OK;
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

bf16_t host_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

}  // namespace

int launch_scores_gemm(const bf16_t* W, const bf16_t* X, float* out, int N, int K, int T, int t_pad, int cu_count, hipStream_t stream) {
    if (N % 128 != 0 || K % 64 != 0 || t_pad % 128 != 0 || T > t_pad) return fail(SQE_ERR_INVALID, "scores gemm: N % 128, K % 64, t_pad % 128");
    return scores_gemm(W, X, out, N, K, T, t_pad, cu_count, stream);
}
}  // namespace sqe

using namespace sqe;

struct sqe_layer {
    DevMem w_qkv, b_qkv, w_o, b_o, ln1_g, ln1_b, w_1, b_1, w_2, b_2, ln2_g, ln2_b;
};

struct sqe_encoder {
    sqe_ctx* ctx = nullptr;
    OpOrder ord;                     // own mutex + stream (internal.h): the encoder never blocks an index or the cache
    sqe_bert_cfg cfg;
    DevMem word, pos, type, emb_g, emb_b;
    std::vector<std::unique_ptr<sqe_layer>> layers;
    std::map<std::string, bool> loaded;
    bool finalized = false;
    // workspace
    DevMem x, x1, qkv, att, hbuf, pre, ids, lens, out;
    int t_cap = 0;
    // Launch-bound regime (a query batch is ~170 short kernels): the forward pass of a given
    // (B, S, buffers) is captured once into a hipGraph and replayed.  A key is captured the second
    // time it is seen, so callers that pass fresh buffers every time just run eagerly.
    struct GraphEntry {
        int B = 0, S = 0;
        const void* ids = nullptr; const void* lens = nullptr; void* out = nullptr;
        int seen = 0;
        uint64_t last_use = 0;
        hipGraphExec_t exec = nullptr;
    };
    std::vector<GraphEntry> graphs;
    uint64_t graph_clock = 0;
    bool use_graphs = true;
    ~sqe_encoder() {
        for (auto& g : graphs)
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
    }
};

namespace {

int upload(DevMem& dst, const float* src, size_t n, bool as_bf16, size_t offset_elems, size_t total_elems, hipStream_t st) {
    SQE_TRY(dst.alloc(total_elems * (as_bf16 ? 2 : 4)));
    if (as_bf16) {
        std::vector<bf16_t> tmp(n);
        for (size_t i = 0; i < n; ++i) tmp[i] = host_bf16(src[i]);
        SQE_HIP(hipMemcpyAsync(dst.as<bf16_t>() + offset_elems, tmp.data(), n * 2, hipMemcpyHostToDevice, st));
        SQE_HIP(hipStreamSynchronize(st));
    } else {
        SQE_HIP(hipMemcpyAsync(dst.as<float>() + offset_elems, src, n * 4, hipMemcpyHostToDevice, st));
        SQE_HIP(hipStreamSynchronize(st));
    }
    return SQE_OK;
}

}  // namespace

extern "C" {

int sqe_encoder_create(sqe_ctx* ctx, const sqe_bert_cfg* cfg, sqe_encoder** out) {
    if (!ctx || !cfg || !out) return fail(SQE_ERR_INVALID, "sqe_encoder_create: null argument");
    *out = nullptr;
    if (cfg->hidden % 128 != 0 || cfg->inter % 128 != 0 || cfg->heads <= 0 || cfg->hidden / cfg->heads != 64 ||
        cfg->layers <= 0 || cfg->max_pos <= 0 || cfg->vocab_size <= 0)
        return fail(SQE_ERR_INVALID, "sqe_encoder_create: need hidden % 128 == 0, inter % 128 == 0, head_dim == 64");
    std::unique_ptr<sqe_encoder> e(new (std::nothrow) sqe_encoder);
    if (!e) return fail(SQE_ERR_OOM, "sqe_encoder_create: host allocation failed");
    e->ctx = ctx;
    e->cfg = *cfg;
    SQE_HIP(hipSetDevice(ctx->device));
    SQE_TRY(e->ord.init());
    for (int l = 0; l < cfg->layers; ++l) e->layers.emplace_back(new sqe_layer);
    {
        const char* g = knob_env("SQE_ENC_GRAPH");          // SQE_ENC_GRAPH=0: always launch kernel by kernel
        e->use_graphs = !(g && g[0] == '0');
    }
    *out = e.release();
    return SQE_OK;
}

void sqe_encoder_destroy(sqe_encoder* enc) {
    if (!enc) return;
    (void)hipSetDevice(enc->ctx->device);
    {
        std::lock_guard<std::mutex> lk(enc->ord.mu);
        enc->ord.quiesce();
    }
    enc->ord.destroy();
    delete enc;
}

int sqe_encoder_load_tensor(sqe_encoder* enc, const char* name, const float* data_host, const int64_t* shape, int ndim) {
    if (!enc || !name || !data_host || !shape || ndim < 1 || ndim > 2) return fail(SQE_ERR_INVALID, "sqe_encoder_load_tensor: bad arguments");
    OpScope op(enc->ctx, enc->ord, true);
    hipStream_t st = op.s;
    const sqe_bert_cfg& c = enc->cfg;
    const size_t H = c.hidden, I = c.inter;
    const std::string n(name);
    size_t count = 1;
    for (int i = 0; i < ndim; ++i) count *= (size_t)shape[i];
    auto expect = [&](size_t want) -> int {
        if (count != want) return fail(SQE_ERR_INVALID, "sqe_encoder_load_tensor: wrong size for " + n);
        return SQE_OK;
    };
    int rc = SQE_OK;
    if (n == "embeddings.word_embeddings.weight") { SQE_TRY(expect((size_t)c.vocab_size * H)); rc = upload(enc->word, data_host, count, true, 0, count, st); }
    else if (n == "embeddings.position_embeddings.weight") { SQE_TRY(expect((size_t)c.max_pos * H)); rc = upload(enc->pos, data_host, count, true, 0, count, st); }
    else if (n == "embeddings.token_type_embeddings.weight") { SQE_TRY(expect((size_t)c.type_vocab * H)); rc = upload(enc->type, data_host, count, true, 0, count, st); }
    else if (n == "embeddings.LayerNorm.weight") { SQE_TRY(expect(H)); rc = upload(enc->emb_g, data_host, count, false, 0, count, st); }
    else if (n == "embeddings.LayerNorm.bias") { SQE_TRY(expect(H)); rc = upload(enc->emb_b, data_host, count, false, 0, count, st); }
    else if (n.rfind("encoder.layer.", 0) == 0) {
        const size_t dot = n.find('.', 14);
        if (dot == std::string::npos) return fail(SQE_ERR_INVALID, "unknown tensor " + n);
        const int l = atoi(n.substr(14, dot - 14).c_str());
        if (l < 0 || l >= c.layers) return fail(SQE_ERR_INVALID, "layer index out of range in " + n);
        sqe_layer& L = *enc->layers[l];
        const std::string s = n.substr(dot + 1);
        if (s == "attention.self.query.weight") { SQE_TRY(expect(H * H)); rc = upload(L.w_qkv, data_host, count, true, 0, 3 * H * H, st); }
        else if (s == "attention.self.key.weight") { SQE_TRY(expect(H * H)); rc = upload(L.w_qkv, data_host, count, true, H * H, 3 * H * H, st); }
        else if (s == "attention.self.value.weight") { SQE_TRY(expect(H * H)); rc = upload(L.w_qkv, data_host, count, true, 2 * H * H, 3 * H * H, st); }
        else if (s == "attention.self.query.bias") { SQE_TRY(expect(H)); rc = upload(L.b_qkv, data_host, count, false, 0, 3 * H, st); }
        else if (s == "attention.self.key.bias") { SQE_TRY(expect(H)); rc = upload(L.b_qkv, data_host, count, false, H, 3 * H, st); }
        else if (s == "attention.self.value.bias") { SQE_TRY(expect(H)); rc = upload(L.b_qkv, data_host, count, false, 2 * H, 3 * H, st); }
        else if (s == "attention.output.dense.weight") { SQE_TRY(expect(H * H)); rc = upload(L.w_o, data_host, count, true, 0, count, st); }
        else if (s == "attention.output.dense.bias") { SQE_TRY(expect(H)); rc = upload(L.b_o, data_host, count, false, 0, count, st); }
        else if (s == "attention.output.LayerNorm.weight") { SQE_TRY(expect(H)); rc = upload(L.ln1_g, data_host, count, false, 0, count, st); }
        else if (s == "attention.output.LayerNorm.bias") { SQE_TRY(expect(H)); rc = upload(L.ln1_b, data_host, count, false, 0, count, st); }
        else if (s == "intermediate.dense.weight") { SQE_TRY(expect(I * H)); rc = upload(L.w_1, data_host, count, true, 0, count, st); }
        else if (s == "intermediate.dense.bias") { SQE_TRY(expect(I)); rc = upload(L.b_1, data_host, count, false, 0, count, st); }
        else if (s == "output.dense.weight") { SQE_TRY(expect(H * I)); rc = upload(L.w_2, data_host, count, true, 0, count, st); }
        else if (s == "output.dense.bias") { SQE_TRY(expect(H)); rc = upload(L.b_2, data_host, count, false, 0, count, st); }
        else if (s == "output.LayerNorm.weight") { SQE_TRY(expect(H)); rc = upload(L.ln2_g, data_host, count, false, 0, count, st); }
        else if (s == "output.LayerNorm.bias") { SQE_TRY(expect(H)); rc = upload(L.ln2_b, data_host, count, false, 0, count, st); }
        else return fail(SQE_ERR_INVALID, "unknown tensor " + n);
    } else {
        return fail(SQE_ERR_INVALID, "unknown tensor " + n);
    }
    if (rc == SQE_OK) { enc->loaded[n] = true; enc->finalized = false; }
    return rc;
}

int sqe_encoder_finalize(sqe_encoder* enc) {
    if (!enc) return fail(SQE_ERR_INVALID, "null encoder");
    std::lock_guard<std::mutex> lk(enc->ord.mu);
    const size_t want = 5 + (size_t)enc->cfg.layers * 16;
    if (enc->loaded.size() != want)
        return fail(SQE_ERR_STATE, "sqe_encoder_finalize: " + std::to_string(enc->loaded.size()) + " of " + std::to_string(want) + " tensors loaded");
    enc->finalized = true;
    return SQE_OK;
}

static int encode_enqueue(sqe_encoder* enc, const int32_t* ids_dev, const int32_t* lens_dev, int B, int S, float* out_dev,
                          int t_pad, hipStream_t st);

// forward pass on stream st (caller holds the encoder's lock)
static int encode_impl(sqe_encoder* enc, const int32_t* ids_dev, const int32_t* lens_dev, int B, int S, float* out_dev, hipStream_t st) {
    if (!enc->finalized) return fail(SQE_ERR_STATE, "sqe_encode: encoder weights not finalized");
    if (S > enc->cfg.max_pos) return fail(SQE_ERR_INVALID, "sqe_encode: need 1 <= S <= max_pos");
    StageTimer timer(enc->ctx->prof, st, ST_ENCODE);
    const sqe_bert_cfg& c = enc->cfg;
    const int H = c.hidden, I = c.inter;
    const int T = B * S;
    const int t_pad = (T + 255) / 256 * 256;
    // +64 rows: the attention kernel stages whole 64-key tiles, which may run past the last sequence
    if (t_pad > enc->t_cap) {
        for (auto& g : enc->graphs)                       // recorded launches point into the old workspace
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
        enc->graphs.clear();
        SQE_TRY(enc->x.alloc((size_t)(t_pad + 64) * H * 2));
        SQE_TRY(enc->x1.alloc((size_t)(t_pad + 64) * H * 2));
        SQE_TRY(enc->att.alloc((size_t)(t_pad + 64) * H * 2));
        SQE_TRY(enc->qkv.alloc((size_t)(t_pad + 64) * 3 * H * 2));
        SQE_TRY(enc->hbuf.alloc((size_t)(t_pad + 64) * I * 2));
        SQE_TRY(enc->pre.alloc((size_t)t_pad * H * 4 * 4));     // up to 4 split-K partial sums
        SQE_HIP(hipMemsetAsync(enc->x.p, 0, (size_t)(t_pad + 64) * H * 2, st));
        SQE_HIP(hipMemsetAsync(enc->x1.p, 0, (size_t)(t_pad + 64) * H * 2, st));
        SQE_HIP(hipMemsetAsync(enc->att.p, 0, (size_t)(t_pad + 64) * H * 2, st));
        SQE_HIP(hipMemsetAsync(enc->qkv.p, 0, (size_t)(t_pad + 64) * 3 * H * 2, st));
        SQE_HIP(hipMemsetAsync(enc->hbuf.p, 0, (size_t)(t_pad + 64) * I * 2, st));
        enc->t_cap = t_pad;
    }
    // ---- replay / capture (small batches only: above ~8k tokens the kernels are long enough)
    constexpr int GRAPH_MAX_TOKENS = 8192;
    constexpr size_t GRAPH_CACHE = 8;
    sqe_encoder::GraphEntry* ge = nullptr;
    if (enc->use_graphs && t_pad <= GRAPH_MAX_TOKENS) {
        for (auto& e : enc->graphs)
            if (e.B == B && e.S == S && e.ids == ids_dev && e.lens == lens_dev && e.out == out_dev) ge = &e;   // replays on any stream
        if (!ge) {
            if (enc->graphs.size() >= GRAPH_CACHE) {              // evict the least recently used entry
                size_t victim = 0;
                for (size_t i = 1; i < enc->graphs.size(); ++i)
                    if (enc->graphs[i].last_use < enc->graphs[victim].last_use) victim = i;
                if (enc->graphs[victim].exec) (void)hipGraphExecDestroy(enc->graphs[victim].exec);
                enc->graphs.erase(enc->graphs.begin() + victim);
            }
            sqe_encoder::GraphEntry fresh;
            fresh.B = B; fresh.S = S; fresh.ids = ids_dev; fresh.lens = lens_dev; fresh.out = out_dev;
            enc->graphs.push_back(fresh);
            ge = &enc->graphs.back();
        }
        ge->last_use = ++enc->graph_clock;
        ++ge->seen;
        if (ge->exec) {
            SQE_HIP(hipGraphLaunch(ge->exec, st));
            return SQE_OK;
        }
    }
    const bool capture = ge && ge->seen >= 2;
    if (capture) SQE_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int rc = encode_enqueue(enc, ids_dev, lens_dev, B, S, out_dev, t_pad, st);
    if (capture) {
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamEndCapture(st, &graph);
        if (rc != SQE_OK) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc;
        }
        if (e != hipSuccess || !graph) return fail(SQE_ERR_HIP, std::string("encoder graph capture: ") + hipGetErrorString(e));
        e = hipGraphInstantiate(&ge->exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { ge->exec = nullptr; return fail(SQE_ERR_HIP, std::string("encoder graph instantiate: ") + hipGetErrorString(e)); }
        SQE_HIP(hipGraphLaunch(ge->exec, st));
    }
    return rc;
}

// the forward pass as a sequence of launches on `st` (run directly, or recorded by a stream capture)
static int encode_enqueue(sqe_encoder* enc, const int32_t* ids_dev, const int32_t* lens_dev, int B, int S, float* out_dev,
                          int t_pad, hipStream_t st) {
    const sqe_bert_cfg& c = enc->cfg;
    const int H = c.hidden, I = c.inter;
    const int T = B * S;
    const int cus = enc->ctx->cu_count;
    const int rows4 = (T + 3) / 4;
    const size_t pstride = (size_t)t_pad * H;             // floats between split-K partial sums in `pre`
    hipLaunchKernelGGL(embed_ln_kernel, dim3(rows4), dim3(256), 0, st, ids_dev, enc->word.as<bf16_t>(), enc->pos.as<bf16_t>(),
                       enc->type.as<bf16_t>(), enc->emb_g.as<float>(), enc->emb_b.as<float>(), enc->pre.as<float>(),
                       enc->x.as<bf16_t>(), T, S, H, c.vocab_size, c.ln_eps);
    SQE_HIP(hipGetLastError());
    const int att_qb = S >= 256 ? 256 : S >= 128 ? 128 : 64;      // query rows per attention workgroup
    const int qblocks = (S + att_qb - 1) / att_qb;
    for (int l = 0; l < c.layers; ++l) {
        sqe_layer& L = *enc->layers[l];
        GemmArgs a;
        a.T = T; a.resid = nullptr; a.n_tiles = 0;
        // E2: QKV projection
        a.W = L.w_qkv.as<bf16_t>(); a.X = enc->x.as<bf16_t>(); a.bias = L.b_qkv.as<float>(); a.out = enc->qkv.p; a.N = 3 * H; a.K = H;
        SQE_TRY(launch_gemm<EPI_BIAS>(a, t_pad, cus, st));
        // E3: attention
        static const int att_form = [] { const char* e = knob_env("SQE_ATT_FORM"); return e ? atoi(e) : 0; }();   // knobs build: A/B
        const bf16_t* qkvp = enc->qkv.as<bf16_t>();
        bf16_t* attp = enc->att.as<bf16_t>();
        const int nsh = B * c.heads;
        if (att_qb == 256 && att_form == 1)
            hipLaunchKernelGGL((attention_kernel<8, 2, 4>), dim3(nsh * qblocks), dim3(512), 0, st, qkvp, lens_dev, attp, S, H, c.heads);
        else if (att_qb == 256 && att_form == 2)
            hipLaunchKernelGGL((attention_kernel<8, 1>), dim3(nsh * ((S + 127) / 128)), dim3(512), 0, st, qkvp, lens_dev, attp, S, H, c.heads);
        else if (att_qb == 256 && att_form == 3)
            hipLaunchKernelGGL((attention_kernel<4, 1>), dim3(nsh * ((S + 63) / 64)), dim3(256), 0, st, qkvp, lens_dev, attp, S, H, c.heads);
        else if (att_qb == 256)      // 4 waves x 2 blocks: 128 query rows per workgroup, three workgroups per CU
            hipLaunchKernelGGL((attention_kernel<4, 2, 3>), dim3(nsh * ((S + 127) / 128)), dim3(256), 0, st, qkvp, lens_dev, attp, S, H, c.heads);
        else if (att_qb == 128)
            hipLaunchKernelGGL((attention_kernel<8, 1>), dim3(nsh * qblocks), dim3(512), 0, st, qkvp, lens_dev, attp, S, H, c.heads);
        else
            hipLaunchKernelGGL((attention_kernel<4, 1>), dim3(nsh * qblocks), dim3(256), 0, st, qkvp, lens_dev, attp, S, H, c.heads);
        SQE_HIP(hipGetLastError());
        // E4: output projection + residual, LayerNorm
        a.W = L.w_o.as<bf16_t>(); a.X = enc->att.as<bf16_t>(); a.bias = L.b_o.as<float>(); a.resid = enc->x.as<bf16_t>();
        a.out = enc->pre.p; a.N = H; a.K = H;
        int ns = 1;
        SQE_TRY(launch_gemm<EPI_RESID>(a, t_pad, cus, st, pstride, &ns));
        hipLaunchKernelGGL(layernorm_kernel, dim3(rows4), dim3(256), 0, st, enc->pre.as<float>(), L.ln1_g.as<float>(),
                           L.ln1_b.as<float>(), enc->x1.as<bf16_t>(), T, H, c.ln_eps, ns, pstride);
        // E5: FFN up + GELU
        a.W = L.w_1.as<bf16_t>(); a.X = enc->x1.as<bf16_t>(); a.bias = L.b_1.as<float>(); a.resid = nullptr;
        a.out = enc->hbuf.p; a.N = I; a.K = H;
        SQE_TRY(launch_gemm<EPI_GELU>(a, t_pad, cus, st));
        // E6: FFN down + residual, LayerNorm (the last layer's LayerNorm is done by the pooling kernel in fp32)
        a.W = L.w_2.as<bf16_t>(); a.X = enc->hbuf.as<bf16_t>(); a.bias = L.b_2.as<float>(); a.resid = enc->x1.as<bf16_t>();
        a.out = enc->pre.p; a.N = H; a.K = I;
        SQE_TRY(launch_gemm<EPI_RESID>(a, t_pad, cus, st, pstride, &ns));
        if (l + 1 < c.layers) {
            hipLaunchKernelGGL(layernorm_kernel, dim3(rows4), dim3(256), 0, st, enc->pre.as<float>(), L.ln2_g.as<float>(),
                               L.ln2_b.as<float>(), enc->x.as<bf16_t>(), T, H, c.ln_eps, ns, pstride);
        } else {
            hipLaunchKernelGGL(pool_ln_kernel, dim3((B + 3) / 4), dim3(256), 0, st, enc->pre.as<float>(), L.ln2_g.as<float>(),
                               L.ln2_b.as<float>(), out_dev, B, S, H, c.ln_eps, ns, pstride);
        }
        SQE_HIP(hipGetLastError());
    }
    return SQE_OK;
}

int sqe_encode_device(sqe_encoder* enc, const int32_t* ids_dev, const int32_t* lens_dev, int B, int S, float* out_dev) {
    if (!enc) return fail(SQE_ERR_INVALID, "null encoder");
    if (B < 0 || S < 1) return fail(SQE_ERR_INVALID, "sqe_encode: need 1 <= S <= max_pos");
    if (B == 0) return SQE_OK;
    if (!ids_dev || !lens_dev || !out_dev) return fail(SQE_ERR_INVALID, "sqe_encode: null buffer");
    OpScope op(enc->ctx, enc->ord, false);
    return encode_impl(enc, ids_dev, lens_dev, B, S, out_dev, op.s);
}

int sqe_encode(sqe_encoder* enc, const int32_t* ids_host, const int32_t* lens_host, int B, int S, float* out_host) {
    if (!enc) return fail(SQE_ERR_INVALID, "null encoder");
    if (B < 0 || S < 1) return fail(SQE_ERR_INVALID, "sqe_encode: bad shape");
    if (B == 0) return SQE_OK;
    if (!ids_host || !lens_host || !out_host) return fail(SQE_ERR_INVALID, "sqe_encode: null buffer");
    OpScope op(enc->ctx, enc->ord, true);
    hipStream_t st = op.s;
    SQE_TRY(enc->ids.alloc((size_t)B * S * 4));
    SQE_TRY(enc->lens.alloc((size_t)B * 4));
    SQE_TRY(enc->out.alloc((size_t)B * enc->cfg.hidden * 4));
    SQE_HIP(hipMemcpyAsync(enc->ids.p, ids_host, (size_t)B * S * 4, hipMemcpyHostToDevice, st));
    SQE_HIP(hipMemcpyAsync(enc->lens.p, lens_host, (size_t)B * 4, hipMemcpyHostToDevice, st));
    SQE_TRY(encode_impl(enc, enc->ids.as<int32_t>(), enc->lens.as<int32_t>(), B, S, enc->out.as<float>(), st));
    SQE_HIP(hipMemcpyAsync(out_host, enc->out.p, (size_t)B * enc->cfg.hidden * 4, hipMemcpyDeviceToHost, st));
    SQE_HIP(hipStreamSynchronize(st));
    return SQE_OK;
}

}  // extern "C"
