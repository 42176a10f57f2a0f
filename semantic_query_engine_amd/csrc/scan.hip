// scan.hip -- S2: brute-force cosine scan with a fused top-k filter (gfx950, wave64, MFMA).
//
// scores[row, query] = <db[row, :], q[query, :]> over bf16 copies of the L2-normalised
// vectors, fp32 accumulation on v_mfma_f32_16x16x32_bf16.  The [rows, B] score matrix
// never reaches HBM: every accumulator is compared in registers against its query's
// running threshold (the kp-th best key seen so far by this workgroup) and only the
// rare survivors are appended to a per-(chunk, query) candidate list.
//
// Work decomposition
//   * the DB is cut into `n_chunks` contiguous runs of 256-row tiles; one PERSISTENT
//     workgroup owns (chunk, query block) and streams its tiles, so thresholds tighten
//     as it goes and the expected number of survivors per query is O(kp * log(rows/kp));
//   * blockIdx is remapped so the `qblocks` workgroups that share a chunk sit on ONE XCD
//     (blocks b and b+8 share an XCD): the DB tile is fetched from HBM once and re-read
//     from that XCD's L2 by the other query blocks.
//
// Pipeline (per 64-wide K step): both operand tiles go global -> LDS with
// global_load_lds_dwordx4 (1 KiB per wave-instruction, full 128-B lines), double buffered;
// the XOR chunk swizzle c' = c ^ ((row >> 1) & 7) is applied on the per-lane SOURCE
// address and on the ds_read_b128 address (the LDS image itself stays lane-linear), which
// makes every fragment read bank-conflict free.
//
// Candidate keys are (orderable(score) << 32) | (0xFFFFFFFF - row): one total order,
// higher score first, lower row id first among equals, so results are deterministic.
#include <stdlib.h>

#include "scan_common.h"

namespace sqe {

namespace {

constexpr int THREADS = SCAN_THREADS;
constexpr int NWAVES = SCAN_NWAVES;
constexpr int ROW_BYTES = SCAN_ROW_BYTES;

// Issue the global->LDS copy of `rows8 * 8` tile rows x 64 k (128 B per row) spread over
// the 8 waves.  `gbase` points at (tile_row0, k0); `ld_bytes` is the global row pitch.
template <int ROWS>
__device__ __forceinline__ void stage_tile(const char* gbase, size_t ld_bytes, char* lds, int wave, int lane) {
    constexpr int NINSTR = ROWS / 8;          // one 1-KiB wave-instruction per 8 rows
    constexpr int ITERS = (NINSTR + NWAVES - 1) / NWAVES;
    const int r_local = lane >> 3;
    const int cprime = lane & 7;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int g = it * NWAVES + wave;     // wave-uniform
        if (NINSTR % NWAVES == 0 || g < NINSTR) {
            const int r = g * 8 + r_local;
            const int c = cprime ^ ((r >> 1) & 7);
            const char* src = gbase + (size_t)r * ld_bytes + c * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(lds + g * 1024),
                                             16, 0, 0);
        }
    }
}

__device__ __forceinline__ bf16x8 lds_frag(const char* tile, int r, int c) {
    return *reinterpret_cast<const bf16x8*>(tile + r * ROW_BYTES + ((c ^ ((r >> 1) & 7)) << 4));
}

template <int WM, int WN, int FM, int FN>
__global__ __launch_bounds__(THREADS) void scan_bf16_kernel(ScanKernelArgs p) {
    constexpr int BM = WM * FM * 16;
    constexpr int BN = WN * FN * 16;
    static_assert(BM == SCAN_BM, "DB tile must be 256 rows");
    static_assert(WM * WN == NWAVES, "8 waves");
    constexpr int STAGE_BYTES = (BM + BN) * ROW_BYTES;
    constexpr int OFF_THR_KEY = 2 * STAGE_BYTES;          // uint64 [BN]
    constexpr int OFF_THR_S = OFF_THR_KEY + BN * 8;       // float  [BN]
    constexpr int OFF_CNT = OFF_THR_S + BN * 4;           // int    [BN]

    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* thr_key = reinterpret_cast<uint64_t*>(smem + OFF_THR_KEY);
    float* thr_s = reinterpret_cast<float*>(smem + OFF_THR_S);
    int* cnt = reinterpret_cast<int*>(smem + OFF_CNT);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN;
    const int wn = wave % WN;

    // XCD-aware remap: blocks b and b+8 share an XCD; give each XCD a contiguous run of
    // logical ids so the query blocks of one chunk share an L2 (speed only).
    int logical = blockIdx.x;
    const int G = gridDim.x;
    if ((G & 7) == 0) logical = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int chunk = logical / p.qblocks;
    const int qb = logical % p.qblocks;
    const int q0 = qb * BN;

    const int tile_begin = chunk * p.tiles_per_chunk;
    const int tile_end = min(p.n_tiles, tile_begin + p.tiles_per_chunk);
    const int KS = p.K / SCAN_BK;
    const size_t ld_bytes = (size_t)p.K * 2;

    // per-query running state
    for (int i = tid; i < BN; i += THREADS) {
        const bool live = (q0 + i) < p.B;
        thr_key[i] = live ? 0ull : ~0ull;
        thr_s[i] = live ? -INFINITY : INFINITY;
        cnt[i] = 0;
    }
    uint64_t* cand_base = p.cand + ((size_t)chunk * p.b_pad + q0) * CAND_CAP;

    const int total_stages = (tile_end - tile_begin) * KS;
    const char* qbase = reinterpret_cast<const char*>(p.q) + (size_t)q0 * ld_bytes;
    const char* dbbase = reinterpret_cast<const char*>(p.db);

    f32x4 acc[FM][FN];

    if (total_stages > 0) {
        // prologue: stage 0
        stage_tile<BM>(dbbase + (size_t)tile_begin * BM * ld_bytes, ld_bytes, smem, wave, lane);
        stage_tile<BN>(qbase, ld_bytes, smem + BM * ROW_BYTES, wave, lane);
    }
    __syncthreads();   // vmcnt(0) + barrier: stage 0 landed, state initialised

    int tile = tile_begin;
    int ks = 0;
    for (int s = 0; s < total_stages; ++s) {
        char* cur = smem + (s & 1) * STAGE_BYTES;
        // prefetch the next stage into the other buffer (its readers finished before the
        // barrier that ended the previous iteration)
        if (s + 1 < total_stages) {
            int ntile = tile, nks = ks + 1;
            if (nks == KS) { nks = 0; ++ntile; }
            char* nxt = smem + ((s + 1) & 1) * STAGE_BYTES;
            stage_tile<BM>(dbbase + (size_t)ntile * BM * ld_bytes + (size_t)nks * ROW_BYTES, ld_bytes, nxt, wave, lane);
            stage_tile<BN>(qbase + (size_t)nks * ROW_BYTES, ld_bytes, nxt + BM * ROW_BYTES, wave, lane);
        }
        if (ks == 0) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const char* tA = cur;
        const char* tB = cur + BM * ROW_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[FM], b[FN];
            const int c = kk * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < FM; ++i) a[i] = lds_frag(tA, wm * (FM * 16) + i * 16 + (lane & 15), c);
#pragma unroll
            for (int j = 0; j < FN; ++j) b[j] = lds_frag(tB, wn * (FN * 16) + j * 16 + (lane & 15), c);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }

        if (ks == KS - 1) {
            // ---------------- fused top-k filter on the finished 256 x BN tile
            float thr[FN];
            bool hit = false;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                thr[j] = thr_s[wn * (FN * 16) + j * 16 + (lane & 15)];
                float mx = acc[0][j][0];
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc[i][j][r]);
                hit |= (mx >= thr[j]);
            }
            if (__any(hit)) {
                const int64_t row_base = (int64_t)tile * BM + wm * (FM * 16) + (lane >> 4) * 4;
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int qcol = wn * (FN * 16) + j * 16 + (lane & 15);
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float sc = acc[i][j][r];
                            if (sc >= thr[j]) {
                                const int64_t row = row_base + i * 16 + r;
                                if (row < p.n_rows && (q0 + qcol) < p.B) {
                                    const uint64_t key = make_key(sc + 0.0f, (uint32_t)row);
                                    if (key > thr_key[qcol]) {
                                        const int slot = atomicAdd(&cnt[qcol], 1);
                                        cand_base[(size_t)qcol * CAND_CAP + slot] = key;
                                    }
                                }
                            }
                        }
                    }
                }
            }
            __syncthreads();   // appends of this tile visible workgroup-wide
            // lists that could overflow on the next tile are cut back to their best kp
            compact_owned(cand_base, wave * (BN / NWAVES), BN / NWAVES, p.trig, p.kp, lane, cnt, thr_s, thr_key);
        }
        ++ks;
        if (ks == KS) { ks = 0; ++tile; }
        __syncthreads();   // next stage landed (vmcnt(0)); everyone done with `cur`; state settled
    }

    // final: every list down to <= kp entries, counts published
    compact_owned(cand_base, wave * (BN / NWAVES), BN / NWAVES, p.kp + 1, p.kp, lane, cnt, thr_s, thr_key);
    __syncthreads();
    for (int i = tid; i < BN; i += THREADS)
        p.cand_cnt[(size_t)chunk * p.b_pad + q0 + i] = cnt[i];
}

template <int WM, int WN, int FM, int FN>
int launch_cfg(const ScanPlan& plan, const ScanArgs& a, hipStream_t stream) {
    constexpr int BN = WN * FN * 16;
    constexpr int LDS = 2 * (SCAN_BM + BN) * ROW_BYTES + BN * 16;
    ScanKernelArgs k;
    k.db = a.db; k.q = a.q; k.n_rows = a.n_rows; k.K = a.K; k.B = a.B; k.b_pad = plan.b_pad;
    k.n_tiles = plan.n_tiles; k.tiles_per_chunk = plan.tiles_per_chunk; k.n_chunks = plan.n_chunks;
    k.qblocks = plan.qblocks; k.kp = plan.kp;
    k.trig = plan.kp > 128 ? plan.kp : 128;
    k.cand = a.cand; k.cand_cnt = a.cand_cnt;
    auto kern = scan_bf16_kernel<WM, WN, FM, FN>;
    static bool attr_set = false;
    if (!attr_set) {
        SQE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(plan.n_chunks * plan.qblocks), dim3(THREADS), LDS, stream, k);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

}  // namespace

ScanPlan make_scan_plan(int64_t n_rows, int B, int kp, int cu_count) {
    ScanPlan p;
    p.bn = B > 64 ? 256 : 64;
    p.qblocks = (B + p.bn - 1) / p.bn;
    p.b_pad = p.qblocks * p.bn;
    p.n_tiles = (int)((n_rows + SCAN_BM - 1) / SCAN_BM);
    int chunks = cu_count / p.qblocks;                 // one persistent workgroup per CU
    if (chunks < 1) chunks = 1;
    // keep n_chunks * qblocks a multiple of 8 when possible so the XCD remap applies
    if (chunks > p.n_tiles) chunks = p.n_tiles > 0 ? p.n_tiles : 1;
    p.tiles_per_chunk = p.n_tiles > 0 ? (p.n_tiles + chunks - 1) / chunks : 0;
    p.n_chunks = p.tiles_per_chunk > 0 ? (p.n_tiles + p.tiles_per_chunk - 1) / p.tiles_per_chunk : 1;
    p.kp = kp;
    return p;
}

int launch_scan_bf16(const ScanPlan& plan, const ScanArgs& a, hipStream_t stream) {
    if (a.K % SCAN_BK != 0) return fail(SQE_ERR_INVALID, "scan: dim must be a multiple of 64");
    if (plan.kp < 1 || plan.kp > MAX_KP) return fail(SQE_ERR_INVALID, "scan: kp out of range");
    if (plan.bn == 256) {
        static const bool use_v0 = [] { const char* e = getenv("SQE_SCAN_V0"); return e && e[0] == '1'; }();
        if (!use_v0) return launch_scan_bf16_p8(plan, a, stream);
        return launch_cfg<2, 4, 8, 4>(plan, a, stream);
    }
    return launch_cfg<8, 1, 2, 4>(plan, a, stream);
}

}  // namespace sqe
