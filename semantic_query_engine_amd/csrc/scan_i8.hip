// scan_i8.hip -- int8 first-pass scan for query blocks of 256 (gfx950): v_mfma_i32_16x16x64_i8 at twice the bf16 MFMA rate
// (measured, tools/micro/mfma_power.hip: 4.2-4.3 POP/s on Gaussian int8 rows against 2.07-2.11 PFLOP/s bf16, same clock).
//
// What the scan computes.  scores = X8 . Q8^T in exact integer arithmetic over the int8 copies of the L2-normalised rows
// (quant.hip: symmetric per-row scales); acc * sxi[row] * (S0^2 * sqi[query]) estimates the cosine to within the
// deterministic bound scan_eps(dq8, dx8max) -- ~0.02 on 1024-d Gaussian rows, eight times the bf16 bound.  With an error
// band that wide an adaptive top-k filter cannot be selective (every row within 2 eps of the k-th place would have to be
// kept), so this kernel does not rank: it COLLECTS.  Every query carries a fixed integer threshold, derived before the
// launch from the true cosines of a 2 % row sample (api.hip: the m-th best of the sample sits ~1 sigma below the k-th best of
// the index), and every row whose scaled score reaches it is appended to the (chunk, query) list -- the candidate lists and
// LDS slot counters of the bf16 scan, without compaction, bound exchange or boot pass.  select_i8.hip then re-scores the
// collected rows in fp32 and proves per query that no uncollected row can reach the k-th cosine:
// thr_eff + eps < k-th true cosine.  Queries whose proof fails (or whose lists overflowed) take the bf16 collect pass that
// already backs the bf16 certificate.  Results are therefore the exact fp32 top-k whatever the int8 rounding did.
//
// Schedule: the ping-pong schedule of scan_pp.hip byte for byte -- a half-step is 64 int8 elements = 64 B per row, the same
// LDS image, swizzle, DMA pieces and ds_read_b128 addresses as 32 bf16 elements -- with these differences:
//   * DB tiles are read from the TILED int8 copy: a half-step is one contiguous 16 KiB block (quant.hip);
//   * every row of a tile carries the same scale (quant.hip), so the scale is applied to the lane's four thresholds, once per
//     tile, not to the accumulators: the last half-step folds the RAW accumulators into four running maxima under its MFMAs
//     exactly as the bf16 kernel does; a wave whose maxima reach a (conservative) threshold marks the tile, and the append
//     path applies the exact predicate acc * s >= thr (the first r03 version scaled row by row: 128 v_mul_i32_i24 per wave
//     and tile in a VALU-bound last phase);
//   * a wave whose tile holds a survivor runs the append path on its way into its next phase: no cross-wave step follows
//     (no compaction exists), so the appends need neither a phase nor a barrier of their own (measured: tile_end);
//   * the 256 row scales of a tile (1 KiB) arrive through one extra LDS-DMA piece per tile.
#include <stdlib.h>

#include <type_traits>

#include "scan_common.h"

namespace sqe {

namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4;

constexpr int HALF_BYTES = 64;                     // bytes per row per half-step (64 int8 elements)
constexpr int LINE_BYTES = 128;                    // LDS line: the slices of tile rows L and L + 128
constexpr int OPER_BYTES = 128 * LINE_BYTES;       // 16 KiB: one operand of one half-step
constexpr int NSTAGE = 4;
constexpr int BNQ = 256;
#ifndef SQE_I8_VARIANT
#define SQE_I8_VARIANT 1
#endif
constexpr int I8V = SQE_I8_VARIANT;                // schedule variants of the ping-pong kernel (A/B builds; see g0_head)
// LDS: a ring of NSTA stages for the DB rows, then a ring of NSTAGE stages for the queries.  I8V & 64 (A/B build): FIVE row stages --
// the rows of half-step x are issued in T_{x-4} instead of T_{x-3}, one more period of HBM latency cover, the queries (L2 hits) as before
constexpr int NSTA = (I8V & 64) ? 5 : NSTAGE;
constexpr int OFF_B = NSTA * OPER_BYTES;
constexpr int OFF_SCALES = OFF_B + NSTAGE * OPER_BYTES;   // 2 x [256] u32 row scales (tile parity)
constexpr int OFF_CNT = OFF_SCALES + 2 * 1024;     // int [256] list lengths of this workgroup's queries
constexpr int OFF_FLAGS = OFF_CNT + BNQ * 4;       // int [16]
constexpr int LDS_BYTES = OFF_FLAGS + 64;
constexpr int BLOCK_BYTES = SCAN_BM * HALF_BYTES;  // 16 KiB: one half-step of a tiled DB tile
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

struct I8KernelArgs {
    const int8_t* db8;        // tiled int8 copy
    long long tile_stride;    // bytes between tiles
    const uint32_t* sxi;      // [rows] row scales (multiples of S0)
    const int8_t* q8;         // [b_pad] query rows, row-major
    int q_pitch;
    const int* thr_int;       // [b_pad] collect thresholds on acc * sxi
    int64_t n_rows;
    int K, B, b_pad, n_tiles, n_chunks, qblocks;
    uint64_t* cand;           // [n_chunks, b_pad, CAND_CAP]
    int* cand_cnt;            // [n_chunks, b_pad]: entries APPENDED (beyond CAND_CAP: into the query's overflow pool)
    uint64_t* ovf;            // [b_pad, I8_OVF_CAP]
    int* ovf_cnt;             // [b_pad]
    unsigned long long* stamps;   // knobs build, SQE_I8_STAMPS=1: (constant-rate clock, core clock) of workgroups 0 and 100 at tiles 0, 1, 2, 4, 8, ...
    int dbg;
};

#define I8_BARRIER()                           \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)
#define I8_WAIT(imm)                               \
    do {                                           \
        __builtin_amdgcn_sched_barrier(0);         \
        __builtin_amdgcn_s_waitcnt(imm);           \
        __builtin_amdgcn_sched_barrier(0);         \
    } while (0)
// gfx9 encoding: vmcnt in [3:0] and [15:14], expcnt [6:4] left at its maximum, lgkmcnt [11:8]
#define I8_WAIT_VM8_LGKM0() I8_WAIT(0x0078)
#define I8_WAIT_VM7_LGKM0() I8_WAIT(0x0077)
#define I8_WAIT_VM0_LGKM0() I8_WAIT(0x0070)

typedef i32x4 AOps[8];    // [fm]: 128 rows x 64 k (16 int8 per lane and fragment)
typedef i32x4 BOps[4];    // [fn]:  64 queries x 64 k

struct Cur {              // a half-step: (tile entry, 64-wide slice inside it)
    int e, h;
    const char* tile;
};

struct S8 {
    const char* qbase;
    const char* scales_src;                // sxi of tile 0 of the chunk, as bytes
    long long scales_stride = 1024;        // bytes between the scale blocks of consecutive entries (a sampled scan skips tiles)
    char* smem;
    unsigned offA0, offA1, offB0, offB1;   // per-lane source offsets of this wave's DMA pieces
    unsigned rdA, rdB;                     // per-lane LDS offsets of the operand reads inside a stage
    int wave, wm, wn, lane;
    int nt, HS, J;
    long long tile_bytes;
    Cur rd, dm;
    int order;
    int pend_h, pend_stage;
    bool defer_on;

    __device__ __forceinline__ void advance(Cur& c) const {
        if (++c.h == HS) {
            c.h = 0;
            ++c.e;
            c.tile += tile_bytes;
        }
    }
    __device__ __forceinline__ void issue_a(const Cur& c, int stage) const {
        char* st = smem + stage * OPER_BYTES;
        const char* as = c.tile + (long long)c.h * BLOCK_BYTES;
        lds_dma16(as + offA0, st + wave * 1024);
        lds_dma16(as + offA1, st + (wave + 8) * 1024);
    }
    __device__ __forceinline__ void issue_b(int h, int stage, int piece) const {
        char* st = smem + OFF_B + stage * OPER_BYTES;
        const char* bs = qbase + h * HALF_BYTES;
        if (piece == 0) lds_dma16(bs + offB0, st + wave * 1024);
        else lds_dma16(bs + offB1, st + (wave + 8) * 1024);
    }
    // this wave's pieces of half-step c (+ the row scales of c's tile, once per tile, from wave 0)
    __device__ __forceinline__ void issue(const Cur& c, int stage, bool all) const {
        if (wave == 0 && c.h == 0) lds_dma16(scales_src + (long long)c.e * scales_stride + lane * 16, smem + OFF_SCALES + (c.e & 1) * 1024);
        issue_a(c, stage);
        issue_b(c.h, stage, 0);
        if (all) issue_b(c.h, stage, 1);
    }
};

template <bool FIRST>
__device__ __forceinline__ void cmp_phase(i32x4 (&acc)[8][4], const AOps& a, const BOps& b) {
#pragma unroll
    for (int fm = 0; fm < 8; ++fm)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn)
            acc[fm][fn] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[fm], b[fn], FIRST ? i32x4{0, 0, 0, 0} : acc[fm][fn], 0, 0, 0);
}

// middle compute phase that also issues the one DMA piece its wave's preceding memory phase left over (scan_pp.hip)
__device__ __forceinline__ void cmp_phase_mid(S8& P, i32x4 (&acc)[8][4], const AOps& a, const BOps& b) {
#pragma unroll
    for (int fm = 0; fm < 8; ++fm) {
        if (fm == 2) {
            __builtin_amdgcn_sched_barrier(0);
            if (P.pend_h >= 0) {
                P.issue_b(P.pend_h, P.pend_stage, 1);
                P.pend_h = -1;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) acc[fm][fn] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[fm], b[fn], acc[fm][fn], 0, 0, 0);
    }
}

// running max over the four scores of one accumulator fragment: 2 v_max3_i32
__device__ __forceinline__ int fold_i32(int m, const i32x4& v) { return max(max(max(max(m, v[0]), v[1]), v[2]), v[3]); }

// Last half-step of a tile, as scan_pp.hip's: MFMAs column group by column group; the MFMA of fragment (fm, fn) is followed
// by the two v_max3_i32 that fold fragment (fm, fn - 1) -- final for 8 MFMAs by then -- into that group's running maximum
// (order requested with sched_group_barrier), so the test rides under the MFMAs.  The accumulators stay RAW: every row of
// the tile has the same scale s (quant.hip), so "acc * s >= thr" is tested as "acc >= tlo" with tlo <= ceil(thr / s), computed
// once per tile and lane (tile_thresholds); the append path applies the exact predicate.  Returns the wave-uniform mask of
// the column groups whose maximum reaches its (conservative) threshold.
__device__ __forceinline__ unsigned cmp_phase_last(i32x4 (&acc)[8][4], const AOps& a, const BOps& b, const int (&tlo)[4]) {
    int mx[4] = {(int)0x80000000, (int)0x80000000, (int)0x80000000, (int)0x80000000};
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) {
        if (fn > 1) asm("" : "+v"(mx[fn - 1]) : "v"(mx[fn - 2]));       // group fn - 1's chain of folds starts after group fn - 2's (scan_pp.hip)
#pragma unroll
        for (int fm = 0; fm < 8; ++fm) {
            acc[fm][fn] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[fm], b[fn], acc[fm][fn], 0, 0, 0);
            if (fn > 0) mx[fn - 1] = fold_i32(mx[fn - 1], acc[fm][fn - 1]);
        }
    }
    asm("" : "+v"(mx[3]) : "v"(mx[2]));
#pragma unroll
    for (int fm = 0; fm < 8; ++fm) mx[3] = fold_i32(mx[3], acc[fm][3]);
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
#pragma unroll
    for (int t = 0; t < 24; ++t) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
    }
    unsigned mask = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) mask |= __any(mx[j] >= tlo[j]) ? (1u << j) : 0u;
    return __builtin_amdgcn_readfirstlane(mask);
}

// conservative per-tile thresholds on the raw accumulators: tlo <= ceil(thr / s) for the tile's scale s >= 1 (float division, error
// of a few ulp covered by the margin; |acc| < 2^23, so a threshold beyond +-1.5e7 behaves as "never" / "always")
__device__ __forceinline__ void tile_thresholds(const int (&thr)[4], int s, int (&tlo)[4]) {
    const float inv = 1.0f / (float)s;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float f = (float)thr[c] * inv;
        tlo[c] = (int)floorf(f - fabsf(f) * 9.5367431640625e-07f) - 1;
    }
}

// operands of half-step j: rows from stage sa of their ring, queries from stage j & 3 of theirs
__device__ __forceinline__ void read_operands(const S8& P, AOps& a, BOps& b, int j, int sa) {
    const char* sta = P.smem + sa * OPER_BYTES;
    const char* stb = P.smem + OFF_B + (j & 3) * OPER_BYTES;
#pragma unroll
    for (int fm = 0; fm < 8; ++fm) a[fm] = *reinterpret_cast<const i32x4*>(sta + P.rdA + fm * 2048);
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) b[fn] = *reinterpret_cast<const i32x4*>(stb + P.rdB + fn * 2048);
}
__device__ __forceinline__ void read_operands(const S8& P, AOps& a, BOps& b, int j) { read_operands(P, a, b, j, j & 3); }

// MEMORY phase of half-step j (= P.rd): DMA of half-step j + 3 into the stage half-step j - 1 used, operand reads of
// half-step j, counted wait that retires this wave's pieces of half-step j + 1 (scan_pp.hip: mem_lean / mem_phase).
__device__ __forceinline__ void mem_phase(S8& P, AOps& a, BOps& b, int j, bool defer) {
    const bool more = j + 3 < P.J;
    const int stage = (j + 3) & 3;
    const bool all = !(defer && more);
    if (P.order == 0) {
        if (more) P.issue(P.dm, stage, all);
        read_operands(P, a, b, j);
    } else {
        read_operands(P, a, b, j);
        if (more) P.issue(P.dm, stage, all);
    }
    if (more && !all) {
        P.pend_h = P.dm.h;
        P.pend_stage = stage;
    }
    if (!more) I8_WAIT_VM0_LGKM0();
    else if (all) I8_WAIT_VM8_LGKM0();
    else I8_WAIT_VM7_LGKM0();
    P.advance(P.rd);
    if (more) P.advance(P.dm);
}

__device__ __forceinline__ uint64_t make_key_i32(int score, uint32_t row) {
    return ((uint64_t)((uint32_t)score ^ 0x80000000u) << 32) | (uint64_t)(0xFFFFFFFFu - row);
}

// Append path of one column group of a finished tile (integer scores, no compaction).  The exact predicate acc * s >= thr
// (|acc| < 2^23, s < 2^16, product < 2^31: quant.hip) is applied fragment by fragment: s >= 1, so the largest of a fragment's four
// accumulators passes iff any of them does -- one v_max3 + v_max, one multiply and one compare per fragment, and the four values are
// looked at only where a lane's fragment holds a survivor (a wave-uniform branch; ~13 survivors per 256 x 256 tile at 10 M rows).
// r03 built a 32-bit mask per lane first (32 x multiply, compare, add-with-carry: ~100 VALU instructions per flagged group) and walked
// its set bits through a 32-way switch: at the key density of a 1.25 M-row shard (8 x the survivors per tile) that was a quarter of
// the scan (profiles/r04_search/i8_tile_stamps_1p25m.txt: 34.7 k cycles per tile against 26.5 k at 10 M rows).
template <int J>
__device__ __forceinline__ void collect_group(const i32x4 (&acc)[8][4], int64_t row_base, int64_t n_rows, bool partial, int qcol, bool live,
                                              int thr, int scale, int* cnt, uint64_t* cand_base, uint64_t* ovf_base, int* ovf_cnt, bool no_store = false) {
    uint64_t* list = cand_base + (size_t)qcol * CAND_CAP;
#pragma unroll
    for (int fm = 0; fm < 8; ++fm) {
        const i32x4 v = acc[fm][J];
        const int m4 = max(max(max(v[0], v[1]), v[2]), v[3]);
        const bool hit = live && __mul24(m4, scale) >= thr;
        if (__any(hit)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int sc = __mul24(v[e], scale);
                const int64_t row = row_base + fm * 16 + e;
                const bool ok = hit && sc >= thr && (!partial || row < n_rows);   // (last tile of the index: rows past the end are not rows)
                if (!__any(ok)) continue;
                if (no_store) continue;                     // (knobs build, SQE_I8_DBG = 32, timing only: survivors found, nothing appended --
                                                            //  every query then takes the bf16 pass; read the kernel's time under rocprofv3)
                if (ok) {
                    const int slot = atomicAdd(&cnt[qcol], 1);
                    if (slot < CAND_CAP) list[slot] = make_key_i32(sc, (uint32_t)row);
                    else {                                  // the list is full: the query's pool (kernels.h: I8_OVF_CAP)
                        const int o = atomicAdd(&ovf_cnt[qcol], 1);
                        if (o < I8_OVF_CAP) ovf_base[(size_t)qcol * I8_OVF_CAP + o] = make_key_i32(sc, (uint32_t)row);
                    }
                }
            }
        }
    }
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_i8_pp_kernel(I8KernelArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    S8 P;
    P.lane = lane;
    P.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = P.wave >> 2;           // waves w and w + 4 share a SIMD
    P.wm = P.wave >> 2;
    P.wn = P.wave & 3;
    P.order = 0;
    P.pend_h = -1;
    P.pend_stage = 0;
    P.defer_on = true;
    P.smem = smem;

    int logical = blockIdx.x;
    const int G = gridDim.x;
    if ((G & 7) == 0) logical = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int chunk = __builtin_amdgcn_readfirstlane(logical / p.qblocks);
    const int qb = __builtin_amdgcn_readfirstlane(logical % p.qblocks);
    const int q0 = qb * BNQ;

    int tile_begin, tile_end;
    chunk_tile_range(p.n_tiles, p.n_chunks, chunk, tile_begin, tile_end);
    P.nt = tile_end - tile_begin;
    P.HS = p.K / 64;
    P.J = P.nt * P.HS;
    P.tile_bytes = (I8V & 32) ? 0 : p.tile_stride;     // (timing build 32: every tile of a chunk reads the chunk's first tile -- L2 hits only, results wrong)
    const size_t ldB = (size_t)p.q_pitch;

    int* cnt = reinterpret_cast<int*>(smem + OFF_CNT);
    int* flags = reinterpret_cast<int*>(smem + OFF_FLAGS);
    for (int i = tid; i < BNQ; i += SCAN_THREADS) cnt[i] = 0;
    if (tid < 16) flags[tid] = 0;
    uint64_t* cand_base = p.cand + ((size_t)chunk * p.b_pad + q0) * CAND_CAP;
    uint64_t* ovf_base = p.ovf + (size_t)q0 * I8_OVF_CAP;
    int* ovf_cnt = p.ovf_cnt + q0;
    const int q_live = min(BNQ, p.B - q0);

    // ---- per-lane DMA source offsets (scan_pp.hip): piece t covers LDS lines 8t .. 8t+7; lane l writes chunk position
    // l & 7 of line 8t + (l >> 3), which holds logical chunk c = pos ^ ((line >> 1) & 7): bytes (c & 3) * 16 of the slice of
    // tile row line + 128 * (c >> 2).  DB rows of a half-step block are 64 B apart (tiled copy), query rows q_pitch apart.
    {
        const int line = P.wave * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((line >> 1) & 7);
        const int row = line + 128 * (c >> 2);
        P.offA0 = (unsigned)(row * HALF_BYTES) + (c & 3) * 16;
        P.offB0 = (unsigned)(row * ldB) + (c & 3) * 16;
        P.offA1 = P.offA0 + (unsigned)(64 * HALF_BYTES);
        P.offB1 = P.offB0 + (unsigned)(64 * ldB);
    }
    {
        const int r = lane & 15, cq = lane >> 4, sw = (r >> 1) & 7;
        P.rdA = (unsigned)(r * LINE_BYTES + (((P.wm * 4 + cq) ^ sw) << 4));
        P.rdB = (unsigned)(((P.wn & 1) * 64 + r) * LINE_BYTES + ((((P.wn >> 1) * 4 + cq) ^ sw) << 4));
    }
    P.qbase = reinterpret_cast<const char*>(p.q8) + (size_t)q0 * ldB;
    P.scales_src = reinterpret_cast<const char*>(p.sxi + (size_t)tile_begin * SCAN_BM);
    const char* tile0 = reinterpret_cast<const char*>(p.db8) + (long long)tile_begin * p.tile_stride;
    P.rd = Cur{0, 0, tile0};
    P.dm = Cur{0, 0, tile0};

    // this lane's four collect thresholds (fixed for the whole scan) and whether its queries exist
    int thr[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) thr[c] = p.thr_int[q0 + P.wn * 64 + c * 16 + (lane & 15)];

    i32x4 acc[8][4];
    AOps a;
    BOps b;

    // ---- prologue: half-steps 0, 1, 2 (the launcher admits dim >= 256 only: HS >= 4, so J >= 4 whenever J > 0, and the
    // row scales of tile e + 1 -- fetched three half-steps ahead -- never land in the buffer tile e is still using)
    constexpr bool DEEP = (I8V & 64) != 0;
    int dmb_h = 0;                              // DEEP: K slice of the next QUERY pieces (half-step j + 3); P.dm is the row cursor (j + 4)
    int ia = 0, ra = 0;                         // DEEP: row-ring stage of the next issue / the next read (half-step mod 5)
    if (P.J > 0) {
        for (int s = 0; s < 3 && s < P.J; ++s) {
            P.issue(P.dm, s, true);
            P.advance(P.dm);
        }
        if (DEEP) {
            dmb_h = 3 % P.HS;
            ia = 3;
            if (3 < P.J) {                      // rows of half-step 3 as well (the launcher admits HS >= 4: h = 3, no scale piece)
                P.issue_a(P.dm, 3);
                P.advance(P.dm);
                ia = 4;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the prologue's pieces (inline asm: the compiler does not wait for them)
    __syncthreads();

    if (P.J > 0) {
        const int HS = P.HS;
        int j = 0;
        unsigned cols = 0;
        // knobs build, SQE_I8_SYNC=1: appends in a common phase behind a barrier, the r03a form (A/B; tile_end)
        const bool sync_appends = (p.dbg & 7) == 2;
        int* any_cols = flags + 8;
        int tile_scale = 1;                // the finished tile's scale (every row of a tile has the same: quant.hip)
        auto last_phase = [&](int e) {
            tile_scale = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(smem + OFF_SCALES + (e & 1) * 1024));
            int tlo[4];
            tile_thresholds(thr, tile_scale, tlo);
            cols = cmp_phase_last(acc, a, b, tlo);
            if (sync_appends && cols != 0 && fresh_lane() == 0) *any_cols = 1;
        };
        // after the barrier that ends G1's last compute phase of entry e: every wave takes the same path
        // Appends of a finished tile.  No cross-wave step follows them (no compaction, no shared flags), so they need no phase
        // and no barrier of their own: a wave with survivors simply enters its next phase late.  Measured against the common
        // phase behind a barrier that the bf16 kernel needs for its compaction (profiles/r03_search/ab_append_barrier.log):
        // batch 256 -3 %, 768 -1.2 %, 512 and 1024 within +-0.4 %.
#ifdef SQE_DEBUG_KNOBS
        const bool dbg_no_store = ((p.dbg >> 3) & 32) != 0;      // SQE_I8_DBG = 32 (timing only): survivors found, not appended
#else
        constexpr bool dbg_no_store = false;
#endif
        auto tile_end = [&](int e) {
#ifdef SQE_DEBUG_KNOBS
            // where a launch's time goes along the chunk: two clocks at tiles 0, 1, 2, 4, 8, ... and at the last one (api.hip prints
            // microseconds and core MHz per tile for each interval)
            if (p.stamps && tid == 0 && (blockIdx.x == 0 || blockIdx.x == 100) && ((e & (e - 1)) == 0 || e == P.nt - 1)) {
                const int slot = e == P.nt - 1 ? 31 : (e == 0 ? 0 : 1 + (31 - __builtin_clz(e)));
                unsigned long long* o = p.stamps + (blockIdx.x ? 128 : 0) + slot * 3;
                o[0] = (unsigned long long)e + 1; o[1] = wall_clock64(); o[2] = clock64();
            }
#endif
            if (sync_appends) {
                const bool any = __builtin_amdgcn_readfirstlane(*any_cols) != 0;
                if (!any) return;
            }
            if (cols) {
                const int fl = fresh_lane();
                const int64_t tile_row0 = (int64_t)(tile_begin + e) * SCAN_BM;
                const int64_t row_base = tile_row0 + P.wm * 128 + (fl >> 4) * 4;
                const bool partial = tile_row0 + SCAN_BM > p.n_rows;
                const int qc0 = P.wn * 64 + (fl & 15);
                if (cols & 1u) collect_group<0>(acc, row_base, p.n_rows, partial, qc0, qc0 < q_live, thr[0], tile_scale, cnt, cand_base, ovf_base, ovf_cnt, dbg_no_store);
                if (cols & 2u) collect_group<1>(acc, row_base, p.n_rows, partial, qc0 + 16, qc0 + 16 < q_live, thr[1], tile_scale, cnt, cand_base, ovf_base, ovf_cnt, dbg_no_store);
                if (cols & 4u) collect_group<2>(acc, row_base, p.n_rows, partial, qc0 + 32, qc0 + 32 < q_live, thr[2], tile_scale, cnt, cand_base, ovf_base, ovf_cnt, dbg_no_store);
                if (cols & 8u) collect_group<3>(acc, row_base, p.n_rows, partial, qc0 + 48, qc0 + 48 < q_live, thr[3], tile_scale, cnt, cand_base, ovf_base, ovf_cnt, dbg_no_store);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (sync_appends) {
                I8_BARRIER();
                if (tid == 0) *any_cols = 0;       // read again a whole tile later
            }
        };
#ifndef SQE_I8_TWO_BARRIERS
        // ONE barrier per half-step.  Period T_j is what lies between barrier B_{j-1} and barrier B_j:
        //     G0, T_j: [appends] | compute j | issue pieces j + 3 | vmcnt: own pieces of j + 2 | read operands j + 1 | B_j
        //     G1, T_j: [appends] | issue pieces j + 3 | read operands j | compute j | vmcnt: own pieces of j + 2 | B_j
        // INVARIANT (every wave reads lines that all eight waves filled, and nothing but the issuing wave's vmcnt plus a barrier
        // orders a ds_read behind another wave's LDS-DMA): a piece read in period T was retired by the wave that ISSUED it
        // before a barrier that precedes the read.  Here every wave retires its pieces of half-step x in T_{x-2}, in front of
        // B_{x-2}; half-step x is read at the end of T_{x-1} (G0) and at the head of T_x (G1), both behind B_{x-2}.  The stage
        // pieces j + 3 go to held half-step j - 1, last read in T_{j-2} (G0) and at the head of T_{j-1} (G1, consumed by that
        // period's MFMAs), both in front of B_{j-1}.  (r03 had G0 wait for its pieces of j + 1 at the HEAD of T_j, behind B_{j-1}:
        // a sibling G0 wave's reads of j + 1 later in T_j were ordered against that wait by nothing but time.)
        // A wave keeps at most two half-steps of pieces in flight and vmcnt(4) retires the older one: vector-memory operations
        // retire in issue order, so wave 0's scale piece and a wave's appended keys -- extra entries of the queue, older than
        // the four pieces of j + 3 or among them -- only make a wait retire more.
        auto issue_next = [&](int jj) {
            if (DEEP) {
                // queries of jj + 3 FIRST, then the rows of jj + 4 (the wait below leaves the rows of jj + 3 in flight: they must be
                // younger than the queries of jj + 2, which it retires)
                if (jj + 3 < P.J) {
                    P.issue_b(dmb_h, (jj + 3) & 3, 0);
                    P.issue_b(dmb_h, (jj + 3) & 3, 1);
                    dmb_h = dmb_h + 1 == P.HS ? 0 : dmb_h + 1;
                }
                if (jj + 4 < P.J) {
                    if (P.wave == 0 && P.dm.h == 0)
                        lds_dma16(P.scales_src + (long long)P.dm.e * P.scales_stride + P.lane * 16, P.smem + OFF_SCALES + (P.dm.e & 1) * 1024);
                    P.issue_a(P.dm, ia);
                    P.advance(P.dm);
                    ia = ia + 1 == NSTA ? 0 : ia + 1;
                }
                return;
            }
            if (jj + 3 < P.J) {
                P.issue(P.dm, (jj + 3) & 3, true);
                P.advance(P.dm);
            }
        };
        // at the end of T_jj: this wave's pieces of jj + 2 (and everything older); the pieces of jj + 3 stay in flight
        // (DEEP: its queue ends ... Q(jj+2) | A(jj+3) x 2 | Q(jj+3) x 2 | A(jj+4) x 2: six entries stay)
        auto wait_pieces = [&](int jj) {
            if (DEEP) {
                if (jj + 4 < P.J) I8_WAIT(0x0F76);           // vmcnt(6)
                else if (jj + 3 < P.J) I8_WAIT(0x0F74);      // vmcnt(4): A(jj+3), Q(jj+3)
                else I8_WAIT(0x0F70);
                return;
            }
            if (jj + 3 < P.J) I8_WAIT(0x0F74);               // vmcnt(4)
            else I8_WAIT(0x0F70);                            // vmcnt(0): nothing was issued in this period
        };
        // every wave reads the half-steps in order, once each: ra is the row stage of its next read
        auto read_next = [&](int x) {
            read_operands(P, a, b, x, DEEP ? ra : (x & 3));
            if (DEEP) ra = ra + 1 == NSTA ? 0 : ra + 1;
        };
#ifdef SQE_DEBUG_KNOBS
        // SQE_I8_DBG (knobs build, run-time A/B): 4 NO raised priority while computing, 16 appends before the barrier, 32 survivors are found
        // but not appended (timing only)
        const int xdbg = p.dbg >> 3;
#define I8_PRIO(n) do { if (!(xdbg & 4)) __builtin_amdgcn_s_setprio(n); } while (0)
#else
        constexpr int xdbg = 0;
        // a wave's compute part runs at raised priority: when both waves of a SIMD have instructions ready, the MFMAs go first
#define I8_PRIO(n) __builtin_amdgcn_s_setprio(n)
#endif
        // G0's half of a period around its compute part.  I8V (-DSQE_I8_VARIANT=<bits>, tools/build_variant.sh; compile-time: run-time
        // switches made hipcc spill; the invariant holds in every form): 0 G0 issues its pieces at the HEAD of the period as G1 does
        // (two periods of latency cover, but all eight waves queue at the address unit at once), 1 (shipped) BEHIND its compute
        // part (compute | pieces | vmcnt | reads: the two groups' pieces leave at different times; one period of cover is enough),
        // 2 as 1 with the reads in front of the pieces, 8 G1 reads before it issues, 64 a five-stage row ring (scan_i8_deep.hip ships
        // it for single-block batches).  (r04 also tried the appends as the YOUNGEST entries of the queue, so that the first wait of a
        // tile need not retire them: no change at 10 M or 1.25 M rows, profiles/r04_search/ab_young_stores.log; not kept.)  Measured at 10 M x 1024
        // (profiles/r04_search/ab_schedule_variants.log), batch 1024 / 256: 0: 9.10-9.16 / 2.62-2.63 ms, 1: 8.44-8.46 / 2.52-2.62,
        // 2: 8.47-8.52 / 2.53, 9: 8.45-8.49 / 2.57-2.64, 10: 8.40-8.45 / 2.56-2.57.
        auto g0_head = [&](int jj) {
            if (!(I8V & 3)) issue_next(jj);
        };
        auto g0_tail = [&](int jj) {
            if (I8V & 2) {
                if (jj + 1 < P.J) read_next(jj + 1);
                issue_next(jj);
                wait_pieces(jj);
                return;
            }
            if (I8V & 1) issue_next(jj);
            wait_pieces(jj);
            if (jj + 1 < P.J) read_next(jj + 1);
        };
        auto g1_head = [&](int jj) {
            if (I8V & 8) {
                read_next(jj);
                issue_next(jj);
                return;
            }
            issue_next(jj);
            read_next(jj);
        };
        if (group == 0) {
            read_next(0);                                    // (the prologue's pieces: retired by every wave before __syncthreads)
            for (int e = 0; e < P.nt; ++e) {
                g0_head(j);
                I8_PRIO(2);
                cmp_phase<true>(acc, a, b);
                I8_PRIO(0);
                g0_tail(j);
                I8_BARRIER();
                ++j;
                for (int h = 1; h < HS - 1; ++h) {
                    g0_head(j);
                    I8_PRIO(2);
                    cmp_phase<false>(acc, a, b);
                    I8_PRIO(0);
                    g0_tail(j);
                    I8_BARRIER();
                    ++j;
                }
                g0_head(j);
                I8_PRIO(2);
                last_phase(e);
                I8_PRIO(0);
                g0_tail(j);
                if (xdbg & 16) tile_end(e);
                I8_BARRIER();
                ++j;
                if (!(xdbg & 16)) tile_end(e);               // (the accumulators are the finished tile's until the next compute part)
            }
        } else {
            for (int e = 0; e < P.nt; ++e) {
                g1_head(j);
                I8_PRIO(2);
                cmp_phase<true>(acc, a, b);
                I8_PRIO(0);
                wait_pieces(j);
                I8_BARRIER();
                ++j;
                for (int h = 1; h < HS - 1; ++h) {
                    g1_head(j);
                    I8_PRIO(2);
                    cmp_phase<false>(acc, a, b);
                    I8_PRIO(0);
                    wait_pieces(j);
                    I8_BARRIER();
                    ++j;
                }
                g1_head(j);
                I8_PRIO(2);
                last_phase(e);
                I8_PRIO(0);
                wait_pieces(j);
                if (xdbg & 16) tile_end(e);
                I8_BARRIER();
                ++j;
                if (!(xdbg & 16)) tile_end(e);
            }
        }
#undef I8_PRIO
#else
        // r03a schedule, a barrier after every phase (A/B: tools/build_variant.sh scan_i8 -DSQE_I8_TWO_BARRIERS):
        //     G0: .. CMP_LAST(e) | MEM(e+1,0) | [append phase] | CMP(e+1,0) | MEM(e+1,1) ..
        //     G1: .. MEM(e,last) | CMP_LAST(e)| [append phase] | MEM(e+1,0) | CMP(e+1,0) ..
        if (group == 0) {
            mem_phase(P, a, b, 0, false);
            I8_BARRIER();
            for (int e = 0; e < P.nt; ++e) {
                cmp_phase<true>(acc, a, b);
                I8_BARRIER();
                mem_phase(P, a, b, j + 1, P.defer_on && HS > 2);
                I8_BARRIER();
                ++j;
                for (int h = 1; h < HS - 1; ++h) {
                    cmp_phase_mid(P, acc, a, b);
                    I8_BARRIER();
                    mem_phase(P, a, b, j + 1, P.defer_on && h < HS - 2);
                    I8_BARRIER();
                    ++j;
                }
                last_phase(e);
                I8_BARRIER();
                if (j + 1 < P.J) mem_phase(P, a, b, j + 1, false);
                I8_BARRIER();
                ++j;
                tile_end(e);
            }
        } else {
            I8_BARRIER();
            for (int e = 0; e < P.nt; ++e) {
                mem_phase(P, a, b, j, false);
                I8_BARRIER();
                cmp_phase<true>(acc, a, b);
                I8_BARRIER();
                ++j;
                for (int h = 1; h < HS - 1; ++h) {
                    mem_phase(P, a, b, j, P.defer_on);
                    I8_BARRIER();
                    cmp_phase_mid(P, acc, a, b);
                    I8_BARRIER();
                    ++j;
                }
                mem_phase(P, a, b, j, false);
                I8_BARRIER();
                last_phase(e);
                I8_BARRIER();
                ++j;
                tile_end(e);
            }
        }
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < BNQ; i += SCAN_THREADS) p.cand_cnt[(size_t)chunk * p.b_pad + q0 + i] = cnt[i];
}

// ---------------------------------------------------------------------------------------------------------------
// Threshold pass in int8 (r03c): the same tile loop over every step-th tile of the index, no thresholds, no lists.  Each lane
// keeps the TWO best scaled scores (and their rows) it has seen per column group -- a lane owns 32 rows of one query per tile
// -- and writes them out at the end: [chunk][query][8 row lanes][2] (score, row).  select_i8.hip takes the m-th largest of a
// query's n_chunks x 16 values as its collect threshold: an order statistic of the SAME quantity the collect scan compares
// (acc x tile scale), so no bf16 scan, no fp32 re-score and no conversion stand between the sample and the threshold.  (The
// m best of the sample are all among the candidates unless three of them fell into one lane's 32-row-per-tile stream; then the
// threshold is the (m+1)-th best or so -- it only places the collection.)
struct I8SampleKernelArgs {
    const int8_t* db8; long long tile_stride; const uint32_t* sxi;
    const int8_t* q8; int q_pitch;
    int K, b_pad, n_tiles_s, step, n_chunks, qblocks;
    int2* out;                // [n_chunks][b_pad][16]
};

__global__ __launch_bounds__(SCAN_THREADS) void sample_i8_pp_kernel(I8SampleKernelArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    S8 P;
    P.lane = lane;
    P.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = P.wave >> 2;
    P.wm = P.wave >> 2;
    P.wn = P.wave & 3;
    P.order = 0;
    P.pend_h = -1;
    P.pend_stage = 0;
    P.defer_on = false;
    P.smem = smem;
    int logical = blockIdx.x;
    const int G = gridDim.x;
    if ((G & 7) == 0) logical = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int chunk = __builtin_amdgcn_readfirstlane(logical / p.qblocks);
    const int qb = __builtin_amdgcn_readfirstlane(logical % p.qblocks);
    const int q0 = qb * BNQ;
    int tile_begin, tile_end;
    chunk_tile_range(p.n_tiles_s, p.n_chunks, chunk, tile_begin, tile_end);
    P.nt = tile_end - tile_begin;
    P.HS = p.K / 64;
    P.J = P.nt * P.HS;
    P.tile_bytes = p.tile_stride * p.step;
    P.scales_stride = 1024ll * p.step;
    const size_t ldB = (size_t)p.q_pitch;
    {
        const int line = P.wave * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((line >> 1) & 7);
        const int row = line + 128 * (c >> 2);
        P.offA0 = (unsigned)(row * HALF_BYTES) + (c & 3) * 16;
        P.offB0 = (unsigned)(row * ldB) + (c & 3) * 16;
        P.offA1 = P.offA0 + (unsigned)(64 * HALF_BYTES);
        P.offB1 = P.offB0 + (unsigned)(64 * ldB);
    }
    {
        const int r = lane & 15, cq = lane >> 4, sw = (r >> 1) & 7;
        P.rdA = (unsigned)(r * LINE_BYTES + (((P.wm * 4 + cq) ^ sw) << 4));
        P.rdB = (unsigned)(((P.wn & 1) * 64 + r) * LINE_BYTES + ((((P.wn >> 1) * 4 + cq) ^ sw) << 4));
    }
    P.qbase = reinterpret_cast<const char*>(p.q8) + (size_t)q0 * ldB;
    P.scales_src = reinterpret_cast<const char*>(p.sxi + (size_t)tile_begin * p.step * SCAN_BM);
    const char* tile0 = reinterpret_cast<const char*>(p.db8) + (long long)tile_begin * p.step * p.tile_stride;
    P.rd = Cur{0, 0, tile0};
    P.dm = Cur{0, 0, tile0};

    int best[4][2], brow[4][2];                        // per column group: the two best scaled scores of this lane and their rows
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        best[c][0] = best[c][1] = (int)0x80000000;
        brow[c][0] = brow[c][1] = -1;
    }
    i32x4 acc[8][4];
    AOps a;
    BOps b;
    if (P.J > 0) {
        for (int s = 0; s < 3 && s < P.J; ++s) {
            P.issue(P.dm, s, true);
            P.advance(P.dm);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (P.J > 0) {
        const int HS = P.HS;
        int j = 0;
        auto issue_next = [&](int jj) {
            if (jj + 3 < P.J) {
                P.issue(P.dm, (jj + 3) & 3, true);
                P.advance(P.dm);
            }
        };
        // scan_i8_pp_kernel's schedule and invariant: every wave retires its pieces of half-step x in T_{x-2}
        auto wait_pieces = [&](int jj) {
            if (jj + 3 < P.J) I8_WAIT(0x0F74);
            else I8_WAIT(0x0F70);
        };
        auto g0_tail = [&](int jj) {                         // behind G0's compute part (scan_i8_pp_kernel: I8V = 1)
            issue_next(jj);
            wait_pieces(jj);
            if (jj + 1 < P.J) read_operands(P, a, b, jj + 1);
        };
        auto g1_head = [&](int jj) {
            issue_next(jj);
            read_operands(P, a, b, jj);
        };
        // the finished tile into the lane's best-two lists (its scale is one per tile: quant.hip)
        auto tile_end = [&](int e) {
            const int s = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(smem + OFF_SCALES + (e & 1) * 1024));
            const int row0 = (tile_begin + e) * p.step * SCAN_BM + P.wm * 128 + (lane >> 4) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                int mx = (int)0x80000000;
#pragma unroll
                for (int fm = 0; fm < 8; ++fm) mx = fold_i32(mx, acc[fm][c]);
                if (__mul24(mx, s) > best[c][1]) {           // (s >= 1: the order of the raw scores is the order of the scaled ones)
#pragma unroll
                    for (int fm = 0; fm < 8; ++fm)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int v = __mul24(acc[fm][c][r], s);
                            const int row = row0 + fm * 16 + r;
                            if (v > best[c][0]) {
                                best[c][1] = best[c][0]; brow[c][1] = brow[c][0];
                                best[c][0] = v; brow[c][0] = row;
                            } else if (v > best[c][1]) {
                                best[c][1] = v; brow[c][1] = row;
                            }
                        }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        if (group == 0) {
            read_operands(P, a, b, 0);
            for (int e = 0; e < P.nt; ++e) {
                cmp_phase<true>(acc, a, b);
                g0_tail(j);
                I8_BARRIER();
                ++j;
                for (int h = 1; h < HS; ++h) {
                    cmp_phase<false>(acc, a, b);
                    g0_tail(j);
                    I8_BARRIER();
                    ++j;
                }
                tile_end(e);
            }
        } else {
            for (int e = 0; e < P.nt; ++e) {
                g1_head(j);
                cmp_phase<true>(acc, a, b);
                wait_pieces(j);
                I8_BARRIER();
                ++j;
                for (int h = 1; h < HS; ++h) {
                    g1_head(j);
                    cmp_phase<false>(acc, a, b);
                    wait_pieces(j);
                    I8_BARRIER();
                    ++j;
                }
                tile_end(e);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- this lane's candidates: query q0 + wn * 64 + c * 16 + (lane & 15), row lane wm * 4 + (lane >> 4)
    int2* out = p.out + ((size_t)chunk * p.b_pad + q0) * 16;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int2* o = out + (size_t)(P.wn * 64 + c * 16 + (lane & 15)) * 16 + (P.wm * 4 + (lane >> 4)) * 2;
        o[0] = make_int2(best[c][0], brow[c][0]);
        o[1] = make_int2(best[c][1], brow[c][1]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Small batches (<= 128 queries): the staged form of scan.hip -- one barrier per 128-element K step, 3-stage LDS ring for
// the DB operand (two K steps of DMA in flight per CU), one 1-KiB row-scale piece per tile -- HBM-bound like its bf16
// twin, at half the bytes per row.  Same collect semantics as the ping-pong kernel above.
template <int WM, int WN, int FM, int FN, int NST, int NSTB>
__global__ __launch_bounds__(SCAN_THREADS) void scan_i8_small_kernel(I8KernelArgs p) {
    constexpr int BM = WM * FM * 16;
    constexpr int BN = WN * FN * 16;
    static_assert(BM == SCAN_BM && WM * WN == SCAN_NWAVES, "256-row DB tile, 8 waves");
    constexpr int ROWB_ = 128;                               // bytes per row per K step (128 int8 elements)
    constexpr int A_BYTES = BM * ROWB_, B_BYTES = BN * ROWB_;
    constexpr int OFF_B = NST * A_BYTES;
    constexpr int OFF_SC = OFF_B + NSTB * B_BYTES;           // 2 x [256] u32 row scales
    constexpr int OFF_C = OFF_SC + 2048;                     // int [BN] list lengths
    constexpr int PIECES_A = BM / 8 / SCAN_NWAVES, PIECES_B = BN / 8 / SCAN_NWAVES;
    static_assert((BN / 8) % SCAN_NWAVES == 0, "every wave issues the same number of query pieces");
    constexpr int IN_FLIGHT = PIECES_A * (NST - 2) + PIECES_B * (NSTB - 2);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int logical = blockIdx.x;
    const int G = gridDim.x;
    if ((G & 7) == 0) logical = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int chunk = __builtin_amdgcn_readfirstlane(logical / p.qblocks);
    const int qb = __builtin_amdgcn_readfirstlane(logical % p.qblocks);
    const int q0 = qb * BN;
    int tile_begin, tile_end;
    chunk_tile_range(p.n_tiles, p.n_chunks, chunk, tile_begin, tile_end);
    const int nt = tile_end - tile_begin;
    const int KS = p.K / 128;
    const size_t ldB = (size_t)p.q_pitch;
    int* cnt = reinterpret_cast<int*>(smem + OFF_C);
    for (int i = tid; i < BN; i += SCAN_THREADS) cnt[i] = 0;
    uint64_t* cand_base = p.cand + ((size_t)chunk * p.b_pad + q0) * CAND_CAP;
    const int q_live = min(BN, p.B - q0);
    const char* qbase = reinterpret_cast<const char*>(p.q8) + (size_t)q0 * ldB;
    const char* dbbase = reinterpret_cast<const char*>(p.db8) + (long long)tile_begin * p.tile_stride;
    const char* scales_src = reinterpret_cast<const char*>(p.sxi + (size_t)tile_begin * SCAN_BM);
    const int total = nt * KS;

    int thr[FN];
#pragma unroll
    for (int c = 0; c < FN; ++c) thr[c] = p.thr_int[q0 + wn * (FN * 16) + c * 16 + (lane & 15)];

    // DMA of one stage: DB rows from the TILED copy (K step s = blocks 2 s and 2 s + 1 of the tile: chunk c < 4 of row r
    // at block 2 s + r * 64 + c * 16, chunks 4-7 in the next block), query rows row-major.  Piece g = 8 rows of 128 B;
    // lane l writes chunk position l & 7 of row 8 g + (l >> 3), which holds logical chunk (l & 7) ^ ((row >> 1) & 7).
    int d_entry = 0, d_ks = 0, q_ks = 0;
    auto issue_db = [&](int s_idx) {
        char* buf = smem + (s_idx % NST) * A_BYTES;
        const char* base = dbbase + (long long)d_entry * p.tile_stride + (long long)d_ks * (2 * BLOCK_BYTES);
        if (wave == 0 && d_ks == 0) lds_dma16(scales_src + (long long)d_entry * 1024 + lane * 16, smem + OFF_SC + (d_entry & 1) * 1024);
#pragma unroll
        for (int it = 0; it < PIECES_A; ++it) {
            const int g = it * SCAN_NWAVES + wave;
            const int r = g * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            lds_dma16(base + (c >> 2) * BLOCK_BYTES + r * HALF_BYTES + (c & 3) * 16, buf + g * 1024);
        }
        if (++d_ks == KS) { d_ks = 0; ++d_entry; }
    };
    auto issue_q = [&](int s_idx) {
        char* buf = smem + OFF_B + (s_idx % NSTB) * B_BYTES;
#pragma unroll
        for (int it = 0; it < PIECES_B; ++it) {
            const int g = it * SCAN_NWAVES + wave;
            const int r = g * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            lds_dma16(qbase + (size_t)r * ldB + (size_t)q_ks * ROWB_ + c * 16, buf + g * 1024);
        }
        if (++q_ks == KS) q_ks = 0;
    };
    for (int s = 0; s < NST - 1 && s < total; ++s) issue_db(s);
    for (int s = 0; s < NSTB - 1 && s < total; ++s) issue_q(s);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    i32x4 acc[FM][FN];
    int entry = 0, ks = 0;
    for (int s = 0; s < total; ++s) {
        const char* tA = smem + (s % NST) * A_BYTES;
        const char* tB = smem + OFF_B + (s % NSTB) * B_BYTES;
        const bool more = s + NST - 1 < total;
        if (NSTB < NST && s + NSTB - 1 < total) issue_q(s + NSTB - 1);
        if (more) issue_db(s + NST - 1);
        if (NSTB == NST && more) issue_q(s + NSTB - 1);
        if (ks == 0) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = i32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            i32x4 a[FM], b[FN];
            const int c = kk * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int r = wm * (FM * 16) + i * 16 + (lane & 15);
                a[i] = *reinterpret_cast<const i32x4*>(tA + r * ROWB_ + ((c ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int r = wn * (FN * 16) + j * 16 + (lane & 15);
                b[j] = *reinterpret_cast<const i32x4*>(tB + r * ROWB_ + ((c ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (ks == KS - 1) {
            // finished tile: scale by the rows' scales, test against the thresholds, append the survivors
            const i32x4* sc = reinterpret_cast<const i32x4*>(smem + OFF_SC + (entry & 1) * 1024 + (wm * (FM * 16) + (lane >> 4) * 4) * 4);
            int mx[FN];
#pragma unroll
            for (int j = 0; j < FN; ++j) mx[j] = (int)0x80000000;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const i32x4 sv = sc[i * 4];
#pragma unroll
                for (int j = 0; j < FN; ++j) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = __mul24(acc[i][j][r], sv[r]);
                    mx[j] = max(max(mx[j], acc[i][j][0]), acc[i][j][1]);
                    mx[j] = max(max(mx[j], acc[i][j][2]), acc[i][j][3]);
                }
            }
            const int64_t tile_row0 = (int64_t)(tile_begin + entry) * SCAN_BM;
            const bool partial = tile_row0 + SCAN_BM > p.n_rows;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                if (!__any(mx[j] >= thr[j])) continue;
                const int qcol = wn * (FN * 16) + j * 16 + (lane & 15);
                const bool live = qcol < q_live;
                uint64_t* list = cand_base + (size_t)qcol * CAND_CAP;
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int64_t row = tile_row0 + wm * (FM * 16) + i * 16 + (lane >> 4) * 4 + r;
                        if (live && acc[i][j][r] >= thr[j] && (!partial || row < p.n_rows)) {
                            const int slot = atomicAdd(&cnt[qcol], 1);
                            if (slot < CAND_CAP) list[slot] = make_key_i32(acc[i][j][r], (uint32_t)row);
                            else {
                                const int o = atomicAdd(&p.ovf_cnt[q0 + qcol], 1);
                                if (o < I8_OVF_CAP) p.ovf[(size_t)(q0 + qcol) * I8_OVF_CAP + o] = make_key_i32(acc[i][j][r], (uint32_t)row);
                            }
                        }
                    }
            }
        }
        ++ks;
        if (ks == KS) { ks = 0; ++entry; }
        if (!more) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if (IN_FLIGHT == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else if (IN_FLIGHT == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        static_assert(IN_FLIGHT == 4 || IN_FLIGHT == 5, "counted wait");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < BN; i += SCAN_THREADS) p.cand_cnt[(size_t)chunk * p.b_pad + q0 + i] = cnt[i];
}

// ---------------------------------------------------------------------------------------------------------------
// Small batches (<= 128 queries), r04: the STREAMING form.  The staged kernel above is HBM-bound at 0.73-0.77 of the peak: every K step
// of a tile passes through a three-stage LDS ring behind one workgroup barrier, two stages (64 KiB) in flight per CU.  But the tiled
// copy holds the A operand of v_mfma_i32_16x16x64_i8 for 16 consecutive rows and one 64-byte K slice as ONE contiguous KiB (row r of
// slice h at h x 16 KiB + r x 64; lane l wants row l & 15, bytes (l >> 4) * 16), so a wave loads a fragment with a single
// global_load_dwordx4 ... nt straight into the operand registers (ivf.hip has the same stream over the list-ordered copy).  Here:
//   * the workgroup's BN queries sit in LDS for the whole chunk (BN x (K + 128) bytes, swizzled as the staged images are);
//   * wave w streams rows 32 w .. 32 w + 31 of every tile through a register ring of RING fragments (RING KiB in flight per wave:
//     128-256 KiB per CU) -- plain loads, hipcc counts the vmcnt waits; the refill is unconditional inside the steady loop and the
//     chunk's last group of slices is code of its own (a branch around the loads would make the compiler drain the ring at every join);
//   * no barrier in the loop: the waves of a workgroup drift apart as the memory system lets them;
//   * survivors of a finished tile go to a wave-private LDS buffer (ballot + prefix: no atomics) and are appended to the (chunk, query)
//     lists only when it fills up and at the end of the chunk: vector-memory operations retire in issue order, so a row fragment loaded
//     behind a key store could not be consumed before that store was acknowledged (ivf.hip measured that stall: 20 % of its kernel).
// Same collect semantics and the same lists as every other kernel of this file (tests/test_i8_exact_gpu.py compares the key sets).
template <int BN, int RING>
__global__ __launch_bounds__(SCAN_THREADS) void scan_i8_stream_kernel(I8KernelArgs p) {
    constexpr int FN = BN / 16;                       // query fragments: every wave scores all BN queries against its 32 rows
    constexpr int SL = RING / 2;                      // K slices the ring holds (two 16-row fragments per slice)
    constexpr int WB = 192;                           // keys of a wave's buffer
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int qrow = p.K + 128;                       // LDS pitch of a query row: rows r and r + 1 start 32 banks apart
    int* cnt = reinterpret_cast<int*>(smem + BN * qrow);                                  // [BN] list lengths
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint64_t* wkey = reinterpret_cast<uint64_t*>(cnt + BN) + wave * WB;                   // [8][WB] keys waiting for their stores
    unsigned char* wcol = reinterpret_cast<unsigned char*>(reinterpret_cast<uint64_t*>(cnt + BN) + SCAN_NWAVES * WB) + wave * WB;
    int logical = blockIdx.x;
    const int G = gridDim.x;
    if ((G & 7) == 0) logical = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int chunk = __builtin_amdgcn_readfirstlane(logical / p.qblocks);
    const int qb = __builtin_amdgcn_readfirstlane(logical % p.qblocks);
    const int q0 = qb * BN;
    int tile_begin, tile_end;
    chunk_tile_range(p.n_tiles, p.n_chunks, chunk, tile_begin, tile_end);
    const int nt = tile_end - tile_begin;
    const int HS = p.K >> 6;                          // (a multiple of SL: the launcher checks)
    const int GPT = HS / SL;                          // groups of slices per tile
    uint64_t* cand_base = p.cand + ((size_t)chunk * p.b_pad + q0) * CAND_CAP;
    const int q_live = min(BN, p.B - q0);

    // ---- the queries into LDS: chunk cc of row r at r * qrow + ((cc ^ ((r >> 1) & 7)) << 4)
    {
        const int cpr = p.K >> 4;
        const char* qsrc = reinterpret_cast<const char*>(p.q8) + (size_t)q0 * p.q_pitch;
        for (int c = tid; c < BN * cpr; c += SCAN_THREADS) {
            const int r = c / cpr, cc = c - r * cpr;
            const i32x4 v = *reinterpret_cast<const i32x4*>(qsrc + (size_t)r * p.q_pitch + cc * 16);
            *reinterpret_cast<i32x4*>(smem + r * qrow + ((cc ^ ((r >> 1) & 7)) << 4)) = v;
        }
        for (int i = tid; i < BN; i += SCAN_THREADS) cnt[i] = 0;
    }
    int thr[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) thr[j] = p.thr_int[q0 + j * 16 + (lane & 15)];
    __syncthreads();
    if (nt > 0) {
        const char* abase = reinterpret_cast<const char*>(p.db8) + (long long)tile_begin * p.tile_stride + wave * 2048;
        const unsigned aoff = (unsigned)((lane & 15) * 64 + (lane >> 4) * 16);
        const int br = lane & 15, bsw = (br >> 1) & 7, bcq = lane >> 4;
        const char* bbase = smem + br * qrow;
        // issue cursor: the next K slice (both 16-row halves of the wave's rows)
        const char* ctile = abase;
        int ch = 0;
        auto issue_slice = [&](i32x4* r) {
            const char* ptr = ctile + (size_t)ch * 16384 + aoff;
            r[0] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(ptr));
            r[1] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(ptr + 1024));
            if (++ch == HS) { ch = 0; ctile += p.tile_stride; }
        };
        i32x4 ring[RING];
#pragma unroll
        for (int e = 0; e < SL; ++e) issue_slice(&ring[2 * e]);
        i32x4 acc[2][FN];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = i32x4{0, 0, 0, 0};
        // SL slices against the resident queries; ISSUE: a slice's registers are refilled as soon as they have been read
        auto group = [&](int hb, auto issue_tag) {
            constexpr bool ISSUE = decltype(issue_tag)::value;
            const char* bq = bbase + hb * 64;
#pragma unroll
            for (int e = 0; e < SL; ++e) {
                const int boff = ((e * 4 + bcq) ^ bsw) << 4;
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const i32x4 b = *reinterpret_cast<const i32x4*>(bq + j * 16 * qrow + boff);
                    acc[0][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[2 * e], b, acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[2 * e + 1], b, acc[1][j], 0, 0, 0);
                }
                if (ISSUE) issue_slice(&ring[2 * e]);
            }
        };
        int wn = 0;                                   // keys in this wave's buffer (wave-uniform)
        auto flush = [&]() {
            for (int i = lane; i < wn; i += 64) {
                const uint64_t key = wkey[i];
                const int qcol = wcol[i];
                const int slot = atomicAdd(&cnt[qcol], 1);
                if (slot < CAND_CAP) cand_base[(size_t)qcol * CAND_CAP + slot] = key;
                else {                                // the list is full: the query's pool (kernels.h: I8_OVF_CAP)
                    const int o = atomicAdd(&p.ovf_cnt[q0 + qcol], 1);
                    if (o < I8_OVF_CAP) p.ovf[(size_t)(q0 + qcol) * I8_OVF_CAP + o] = key;
                }
            }
            wn = 0;
        };
        // survivors of a finished tile -> the wave's buffer; the exact predicate acc * s >= thr, fragment by fragment (collect_group)
        auto tile_done = [&](int t) {
            const int scale = (int)p.sxi[(size_t)(tile_begin + t) * SCAN_BM];          // one scale per tile (quant.hip); a scalar load
            const int64_t row0 = (int64_t)(tile_begin + t) * SCAN_BM + wave * 32 + (lane >> 4) * 4;
            const bool partial = (int64_t)(tile_begin + t + 1) * SCAN_BM > p.n_rows;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int qcol = j * 16 + (lane & 15);
                const bool live = qcol < q_live;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const i32x4 v = acc[i][j];
                    const int m4 = max(max(max(v[0], v[1]), v[2]), v[3]);
                    const bool hit = live && __mul24(m4, scale) >= thr[j];
                    if (__any(hit)) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int sc = __mul24(v[e], scale);
                            const int64_t row = row0 + i * 16 + e;
                            const bool ok = hit && sc >= thr[j] && (!partial || row < p.n_rows);
                            const unsigned long long mask = __ballot(ok);
                            if (mask == 0ull) continue;
                            if (wn + 64 > WB) flush();                              // (wave-uniform)
                            if (ok) {
                                const int at = wn + __popcll(mask & ((1ull << lane) - 1ull));
                                wkey[at] = make_key_i32(sc, (uint32_t)row);
                                wcol[at] = (unsigned char)qcol;
                            }
                            wn += __popcll(mask);
                        }
                    }
                    acc[i][j] = i32x4{0, 0, 0, 0};
                }
            }
        };
        const int n_grp = nt * GPT;
        int gt = 0, t = 0;
        for (int g = 0; g + 1 < n_grp; ++g) {
            group(gt * SL, std::true_type{});
            if (++gt == GPT) {
                tile_done(t);
                gt = 0;
                ++t;
            }
        }
        group(gt * SL, std::false_type{});
        tile_done(t);
        flush();
    }
    __syncthreads();
    for (int i = tid; i < BN; i += SCAN_THREADS) p.cand_cnt[(size_t)chunk * p.b_pad + q0 + i] = cnt[i];
}

template <int BN, int RING>
int launch_stream(const I8KernelArgs& k, hipStream_t stream) {
    const int lds = BN * (k.K + 128) + BN * 4 + SCAN_NWAVES * 192 * 9;
    auto kern = scan_i8_stream_kernel<BN, RING>;
    SQE_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds));
    hipLaunchKernelGGL(kern, dim3(k.n_chunks * k.qblocks), dim3(SCAN_THREADS), lds, stream, k);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

template <int WM, int WN, int FM, int FN, int NST, int NSTB>
int launch_small(const I8KernelArgs& k, hipStream_t stream) {
    constexpr int BN = WN * FN * 16;
    constexpr int LDS = (NST * SCAN_BM + NSTB * BN) * 128 + 2048 + BN * 4;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    auto kern = scan_i8_small_kernel<WM, WN, FM, FN, NST, NSTB>;
    SQE_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS));
    hipLaunchKernelGGL(kern, dim3(k.n_chunks * k.qblocks), dim3(SCAN_THREADS), LDS, stream, k);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

}  // namespace

int launch_scan_i8(const I8ScanArgs& a, hipStream_t stream) {
    if (a.K % 128 != 0 || a.K < 256) return fail(SQE_ERR_INVALID, "int8 scan: dim must be a multiple of 128, >= 256");
    if (a.bn != 64 && a.bn != 128 && a.bn != BNQ) return fail(SQE_ERR_INVALID, "int8 scan: query block must be 64, 128 or 256");
    if (a.qblocks * a.bn != a.b_pad) return fail(SQE_ERR_INVALID, "int8 scan: padded batch must be a whole number of query blocks");
    I8KernelArgs k;
    k.db8 = a.db8; k.tile_stride = a.tile_stride; k.sxi = a.sxi; k.q8 = a.q8; k.q_pitch = a.q_pitch; k.thr_int = a.thr_int;
    k.n_rows = a.n_rows; k.K = a.K; k.B = a.B; k.b_pad = a.b_pad; k.n_tiles = a.n_tiles; k.n_chunks = a.n_chunks; k.qblocks = a.qblocks;
    k.cand = a.cand; k.cand_cnt = a.cand_cnt; k.ovf = a.ovf; k.ovf_cnt = a.ovf_cnt; k.stamps = a.stamps;
    if (!a.ovf || !a.ovf_cnt) return fail(SQE_ERR_INVALID, "int8 scan: no overflow pool");
    {
        static const int force = [] { const char* e = knob_env("SQE_I8_SYNC"); return e ? (e[0] == '0' ? 1 : 2) : 0; }();   // knobs build only
        static const int xdbg = [] { const char* e = knob_env("SQE_I8_DBG"); return e ? atoi(e) : 0; }();
        k.dbg = force | (xdbg << 3);
    }
    {
        // the streaming kernels wherever the queries fit in LDS beside the buffers (dim 1024: both; up to 2,048 for 64 queries) and the
        // slices of a row divide into ring groups; the staged kernels otherwise (knobs build, SQE_I8_STAGED=1: always, for A/B)
        static const bool staged = [] { const char* e = knob_env("SQE_I8_STAGED"); return e && e[0] == '1'; }();
        const int lds = a.bn * (a.K + 128) + a.bn * 4 + SCAN_NWAVES * 192 * 9;
        if (!staged && a.bn <= 128 && lds <= 160 * 1024) {
            if (a.bn == 64 && a.K % 1024 == 0) return launch_stream<64, 32>(k, stream);
            if (a.bn == 64 && a.K % 512 == 0) return launch_stream<64, 16>(k, stream);
            if (a.bn == 128 && a.K % 512 == 0) return launch_stream<128, 16>(k, stream);     // (a 32-fragment ring spills here: 256 VGPRs)
        }
    }
    if (a.bn == 64) return launch_small<8, 1, 2, 4, 3, 3>(k, stream);          // the tilings of scan.hip's 64- / 128-query kernels
    if (a.bn == 128) return launch_small<4, 2, 4, 4, 3, 2>(k, stream);
    auto kern = scan_i8_pp_kernel;
    SQE_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(a.n_chunks * a.qblocks), dim3(SCAN_THREADS), LDS_BYTES, stream, k);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

int launch_sample_i8(const I8SampleArgs& a, hipStream_t stream) {
    if (a.K % 128 != 0 || a.K < 256) return fail(SQE_ERR_INVALID, "int8 sample: dim must be a multiple of 128, >= 256");
    if (a.b_pad % BNQ != 0 || a.n_tiles_s < 1 || a.step < 1 || a.n_chunks < 1 || a.n_chunks > a.n_tiles_s)
        return fail(SQE_ERR_INVALID, "int8 sample: bad plan");
    I8SampleKernelArgs k;
    k.db8 = a.db8; k.tile_stride = a.tile_stride; k.sxi = a.sxi; k.q8 = a.q8; k.q_pitch = a.q_pitch;
    k.K = a.K; k.b_pad = a.b_pad; k.n_tiles_s = a.n_tiles_s; k.step = a.step; k.n_chunks = a.n_chunks; k.qblocks = a.b_pad / BNQ;
    k.out = reinterpret_cast<int2*>(a.out);
    auto kern = sample_i8_pp_kernel;
    SQE_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(a.n_chunks * k.qblocks), dim3(SCAN_THREADS), LDS_BYTES, stream, k);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

}  // namespace sqe
