// scan.hip -- S2: brute-force cosine scan with a fused top-k filter (gfx950, wave64, MFMA).
//
// scores[row, query] = <db[row, :], q[query, :]> over bf16 copies of the L2-normalised
// vectors, fp32 accumulation on v_mfma_f32_16x16x32_bf16.  The [rows, B] score matrix
// never reaches HBM: every accumulator is compared in registers against its query's
// running threshold and only the rare survivors are appended to a per-(chunk, query)
// candidate list (scan_common.h describes the filter).
//
// Work decomposition
//   * the DB is cut into `n_chunks` contiguous runs of 256-row tiles; one PERSISTENT
//     workgroup owns (chunk, query block) and streams its tiles;
//   * blockIdx is remapped so the `qblocks` workgroups that share a chunk sit on ONE XCD
//     (blocks b and b+8 share an XCD): a DB tile fetched by one of them is an L2 hit for
//     the others while they run in step.
//
// This file holds the generic staged form (one barrier per 64-wide K step; LDS ring of 2 stages for the
// 256-query tile, 3 for the HBM-bound 64-query tile, DB stages 3 deep and query stages 2 deep for the
// 128-query tile -- what fits in 160 KiB): the kernel of batches <= 128 and of the
// collect pass, and the A/B baseline for 256-query blocks, whose default is the ping-pong schedule of
// scan_pp.hip.  Both operand tiles go global -> LDS in full 128-B
// lines; the XOR chunk swizzle c' = c ^ ((row >> 1) & 7) is applied on the per-lane SOURCE
// address and on the ds_read_b128 address (the LDS image itself stays lane-linear), which
// makes every fragment read bank-conflict free.
//
// Candidate keys are (orderable(score) << 32) | (0xFFFFFFFF - row): one total order,
// higher score first, lower row id first among equals, so results are deterministic.
#include <stdlib.h>

#include <algorithm>

#include "scan_common.h"

namespace sqe {

namespace {

constexpr int THREADS = SCAN_THREADS;
constexpr int NWAVES = SCAN_NWAVES;
constexpr int ROW_BYTES = SCAN_ROW_BYTES;

// Issue the global->LDS copy of ROWS tile rows x 64 k (128 B per row) spread over the 8
// waves.  `gbase` points at (tile_row0, k0); `ld_bytes` is the global row pitch.
// barrier that orders LDS traffic only (no vmcnt drain: DMA stays in flight)
#define LDS_BARRIER()                                          \
    do {                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_sched_barrier(0);                     \
        __builtin_amdgcn_s_barrier();                          \
        __builtin_amdgcn_sched_barrier(0);                     \
    } while (0)

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
            lds_dma16(src, lds + g * 1024);
        }
    }
}

__device__ __forceinline__ bf16x8 lds_frag(const char* tile, int r, int c) {
    return *reinterpret_cast<const bf16x8*>(tile + r * ROW_BYTES + ((c ^ ((r >> 1) & 7)) << 4));
}

// NST = LDS ring depth: NST - 1 K steps of DMA are in flight while one is computed.  2 for the
// 256-query block (MFMA-bound); 3 for the 64-query block, which is HBM-bound and needs the extra
// 40 KiB per CU in flight to cover the memory latency.  NSTB = ring depth of the query operand alone: the
// 128-query block keeps DB tiles three deep and query tiles (L2-resident, short latency) two deep, which is
// what fits in 160 KiB next to the filter state; a shallower query ring is issued first in every step so that
// the counted wait retires it together with the older DB stage.
template <int WM, int WN, int FM, int FN, bool COLLECT, int NST, int NSTB = NST>
__global__ __launch_bounds__(THREADS) void scan_bf16_kernel(ScanKernelArgs p) {
    constexpr int BM = WM * FM * 16;
    constexpr int BN = WN * FN * 16;
    static_assert(BM == SCAN_BM, "DB tile must be 256 rows");
    static_assert(WM * WN == NWAVES, "8 waves");
    static_assert(BN % GSLICE_Q == 0, "query block is a whole number of refresh slices");
    static_assert(NSTB >= 2 && NSTB <= NST, "query ring no deeper than the DB ring");
    constexpr int A_BYTES = BM * ROW_BYTES;              // one DB stage
    constexpr int B_BYTES = BN * ROW_BYTES;              // one query stage
    constexpr int OFF_B = NST * A_BYTES;
    constexpr int OFF_F = OFF_B + NSTB * B_BYTES;
    constexpr int PIECES_A = BM / 8 / NWAVES;            // DMA wave-instructions per wave per stage
    constexpr int PIECES_B = BN / 8 / NWAVES;
    constexpr int IN_FLIGHT = PIECES_A * (NST - 2) + PIECES_B * (NSTB - 2);   // pieces the end-of-step wait leaves
    static_assert((BM / 8) % NWAVES == 0 && (BN / 8) % NWAVES == 0, "every wave issues the same number of pieces");
    using FL = FilterLds<BN>;
    constexpr int PER_WAVE = BN / NWAVES;
    constexpr int NSLICE = BN / GSLICE_Q;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* gstage = smem + OFF_F + FL::OFF_GSTAGE;

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
    // integer division runs on the VALU: pull the (uniform) results back into SGPRs so that
    // everything derived from them (tile range, base pointers) stays scalar
    const int chunk = __builtin_amdgcn_readfirstlane(logical / p.qblocks);
    const int qb = __builtin_amdgcn_readfirstlane(logical % p.qblocks);
    const int q0 = qb * BN;
    // COLLECT mode: the batch is the compacted list of uncertified queries, whose length only the device
    // knows.  Both plans are always enqueued; the one that does not fit the count, and query blocks
    // past it, return at once.
    int batch = p.B;
    if (COLLECT) {
        batch = *p.unc_count;
        if (batch <= 0 || q0 >= batch) return;
        if (batch < p.collect_lo || batch > p.collect_hi) return;
    }

    int tile_begin, tile_end;
    chunk_tile_range(p.n_tiles, p.n_chunks, chunk, tile_begin, tile_end);
    const int nt = tile_end - tile_begin;
    const int KS = p.K / SCAN_BK;
    const size_t ldA = (size_t)p.db_pitch, ldB = (size_t)p.q_pitch;

    Filter f;
    f.cand_base = p.cand + ((size_t)chunk * p.b_pad + q0) * CAND_CAP;
    f.gstride = p.ngroups * GMAX_COLS * 64;
    const uint32_t* gmax_group;
    bound_rows(p, chunk, q0, gmax_group, f.gmax_mine);
    f.thr_key = reinterpret_cast<uint64_t*>(smem + OFF_F + FL::OFF_THR_KEY);
    f.thr_s = reinterpret_cast<float*>(smem + OFF_F + FL::OFF_THR_S);
    f.cnt = reinterpret_cast<int*>(smem + OFF_F + FL::OFF_CNT);
    f.cmax = reinterpret_cast<uint32_t*>(smem + OFF_F + FL::OFF_CMAX);
    f.flags = reinterpret_cast<int*>(smem + OFF_F + FL::OFF_FLAGS);
    f.n_rows = p.n_rows;
    f.q_live = min(BN, batch - q0);
    f.trig = p.trig;
    f.per_wave = PER_WAVE;
    f.dbg_no_slow = (SQE_DBG_BITS(p) & 16) != 0;
    f.dbg_counters = (SQE_DBG_BITS(p) & 32) ? p.dbg_counters : nullptr;
    constexpr bool collect = COLLECT;                    // second pass for uncertified queries
    f.collect_keys = collect ? p.collect_keys + (size_t)q0 * EXACT_CAP : nullptr;
    f.collect_cnt = collect ? p.collect_cnt + q0 : nullptr;
    float* slack = reinterpret_cast<float*>(smem + OFF_F + FL::OFF_SLACK);
    f.slack = slack;
    filter_init<BN>(p, f, slack, q0, batch, collect ? p.collect_thr : nullptr, tid);

    // Tile sequence: entry 0 = first tile in BOOT mode, entries 1..nt-1 the other tiles, entry
    // nt = the first tile again, normally.  (nt == 0: nothing.)
    const int n_entries = nt > 0 ? nt + (collect ? 0 : 1) : 0;   // collect mode: every tile once, no boot entry
    const int total_stages = n_entries * KS;
    const char* qbase = reinterpret_cast<const char*>(p.q) + (size_t)q0 * ldB;
    const char* dbbase = reinterpret_cast<const char*>(p.db);
    auto tile_of = [&](int e) { return (e < nt ? tile_begin + e : tile_begin) * p.tile_step; };
    // K step ks covers k slice (ks + rot) mod KS: query blocks sharing a DB tile walk K in rotated
    // order and touch the same DB lines one K step apart (the dot product is order independent)
    const int rot = (logical * p.krot) % KS;
    auto kslice = [&](int k) { const int r = k + rot; return r >= KS ? r - KS : r; };

    f32x4 acc[FM][FN];

    // stage index -> (entry, K step) of the stage each DMA cursor points at
    int d_entry = 0, d_ks = 0, q_ks = 0;
    auto issue_db = [&](int s_idx) {
        stage_tile<BM>(dbbase + (size_t)tile_of(d_entry) * BM * ldA + (size_t)kslice(d_ks) * ROW_BYTES, ldA,
                       smem + (s_idx % NST) * A_BYTES, wave, lane);
        if (++d_ks == KS) { d_ks = 0; ++d_entry; }
    };
    auto issue_q = [&](int s_idx) {
        stage_tile<BN>(qbase + (size_t)kslice(q_ks) * ROW_BYTES, ldB, smem + OFF_B + (s_idx % NSTB) * B_BYTES, wave, lane);
        if (++q_ks == KS) q_ks = 0;
    };
    for (int s = 0; s < NST - 1 && s < total_stages; ++s) issue_db(s);
    for (int s = 0; s < NSTB - 1 && s < total_stages; ++s) issue_q(s);
    // The DMA pieces are inline asm, invisible to hipcc: __syncthreads() alone would NOT wait for them (the
    // first K step would read LDS before its tile landed -- harmless-looking in the normal pass, whose first
    // tile only feeds the boot maxima, but rows of the first tile were lost in COLLECT mode).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // prologue stages landed, filter state initialised

    int entry = 0;
    int ks = 0;
    int refresh_pending = -1;      // slice whose fetch is in flight
    int refresh_age = 0;           // K steps since it was issued
    int refresh_ctr = 0;
    for (int s = 0; s < total_stages; ++s) {
        const char* tA = smem + (s % NST) * A_BYTES;
        const char* tB = smem + OFF_B + (s % NSTB) * B_BYTES;
        // the bound rows fetched during the previous K step have landed (barrier below)
        // The fetch is younger than the DMA pieces of its own K step, so the counted wait that ends that
        // step leaves it in flight; the wait of the NEXT step retires it (NST = 2 drains everything every step).
        if (refresh_pending >= 0 && ++refresh_age >= (NST == 2 ? 1 : 2)) {
            if (wave == (refresh_ctr & 7)) refresh_apply<BN == 256 ? 16 : GMAX_COLS>(f, gstage, refresh_pending, p.gshift, p.gshift_k, BN == 256 ? 0 : p.k_rows, lane);   // (256, the two-stage A/B form: no registers for 64 reads in flight or for the sort)
            refresh_pending = -1;
        }
        // prefetch stage s + NST - 1 (queries: s + NSTB - 1) into the buffer stage s - 1 used (its readers
        // finished before the barrier that ended the previous iteration)
        const bool more = s + NST - 1 < total_stages;
        if (NSTB < NST && s + NSTB - 1 < total_stages) issue_q(s + NSTB - 1);
        if (more) issue_db(s + NST - 1);
        if (NSTB == NST && more) issue_q(s + NSTB - 1);
        {
            // Bound refresh schedule: entry 1 fetches every slice back to back from K step KS/4 on
            // (after every chunk has published its boot maxima); later one slice per tile.
            // (... for 128 queries and more every fourth tile from tile 32 of the chunk on: the bound moves by 1 / t per tile
            // by then; measured -2 % at batch 128, +1.5 % at batch 64, whose one slice is its whole query block)
            const bool want = entry == 1 ? (ks >= KS / 4 && refresh_ctr < NSLICE)
                                         : (entry > 1 && ks == 0 && (BN < 128 || entry <= 32 || (entry & 3) == 0));
            if (want && (p.gshift >= 0 || p.gshift_k >= 0) && !collect && refresh_pending < 0) {
                refresh_age = 0;
                refresh_pending = refresh_ctr % NSLICE;
                ++refresh_ctr;
                refresh_issue<true>(gmax_group, f.gstride, refresh_pending, gstage, wave, lane);
            }
        }
        if (ks == 0) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
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

        if (ks == KS - 1 && !(SQE_DBG_BITS(p) & 4)) {                 // SQE_DBG bit 4: timing experiment, no filter
            const int64_t row0 = (int64_t)tile_of(entry) * BM;
            if (entry == 0 && !collect) {
                filter_boot<FM, FN>(acc, f, row0, wm * (FM * 16), wn * (FN * 16), lane);
                LDS_BARRIER();
                publish_cmax(f, wave * PER_WAVE, PER_WAVE, lane);
                wait_vm0_visible();
            } else {
                // The DMA pieces are invisible to hipcc (lds_dma16), so any wait it inserts for a pending
                // store of its own is a vmcnt(0) that drains the ring.  Waves that stored keys retire
                // them here, inside the rare branch, and the compiler's state is clean on every path
                // back to the K loop.
                if (filter_tile<FM, FN, COLLECT>(acc, f, row0, wm * (FM * 16), wn * (FN * 16), lane)) wait_vm0_visible();
                LDS_BARRIER();       // every wave's flags are set
                // Appended keys only have to be in memory when a list is compacted; every wave reads
                // the same flag words, so the branch and its barriers are uniform.
                if (__builtin_amdgcn_readfirstlane(__any(f.flags[lane & 7] != 0))) {
                    __syncthreads();                                  // vmcnt(0): keys visible workgroup-wide
                    if (__builtin_amdgcn_readfirstlane(f.flags[wave]) != 0) {
                        compact_owned(f, wave * PER_WAVE, PER_WAVE, p.trig, p.kp, lane);
                        wait_vm0_visible();
                    }
                    LDS_BARRIER();                                    // flags read before they are cleared
                    if (lane == 0) f.flags[wave] = 0;
                }
            }
        }
        ++ks;
        if (ks == KS) { ks = 0; ++entry; }
        // stage s + 1 landed (the NST - 2 younger stages stay in flight; anything else this wave issued
        // in between only makes the wait retire more); everyone done with this stage; filter state settled
        if (NST == 2 || !more) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if (IN_FLIGHT == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else if (IN_FLIGHT == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
        else if (IN_FLIGHT == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    if (collect) return;                                 // keys went straight to the per-query buffers
    // final: every list down to <= kp entries, counts published
    compact_owned(f, wave * PER_WAVE, PER_WAVE, p.kp + 1, p.kp, lane);
    __syncthreads();
    for (int i = tid; i < BN; i += THREADS)
        p.cand_cnt[(size_t)chunk * p.b_pad + q0 + i] = f.cnt[i];
}

template <int WM, int WN, int FM, int FN, int NST, int NSTB = NST>
int launch_cfg(const ScanPlan& plan, const ScanArgs& a, hipStream_t stream, bool collect = false) {
    constexpr int BN = WN * FN * 16;
    constexpr int LDS = (NST * SCAN_BM + NSTB * BN) * ROW_BYTES + FilterLds<BN>::BYTES;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    ScanKernelArgs k = make_kernel_args(plan, a);
    if (collect) {
        k.collect_thr = a.collect_thr; k.collect_keys = a.collect_keys; k.collect_cnt = a.collect_cnt;
        k.unc_count = a.unc_count;
        k.collect_lo = a.collect_lo; k.collect_hi = a.collect_hi;
    }
    auto kern = collect ? scan_bf16_kernel<WM, WN, FM, FN, true, NST, NSTB> : scan_bf16_kernel<WM, WN, FM, FN, false, NST, NSTB>;
    SQE_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS));
    hipLaunchKernelGGL(kern, dim3(plan.n_chunks * plan.qblocks), dim3(THREADS), LDS, stream, k);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

}  // namespace

ScanKernelArgs make_kernel_args(const ScanPlan& plan, const ScanArgs& a) {
    ScanKernelArgs k;
    k.db = a.db; k.q = a.q; k.n_rows = a.n_rows; k.K = a.K; k.B = a.B; k.b_pad = plan.b_pad;
    k.db_pitch = a.db_pitch; k.q_pitch = a.q_pitch;
    k.n_tiles = plan.n_tiles; k.tiles_per_chunk = plan.tiles_per_chunk; k.n_chunks = plan.n_chunks;
    k.qblocks = plan.qblocks; k.kp = plan.kp;
    // a list is cut back to its best kp when it reaches `trig`; the window trig - kp (>= 64) is what one
    // compaction buys, and a tile can add SCAN_BM entries on top before the next check
    k.trig = std::min(CAND_CAP - SCAN_BM, std::max(2 * plan.kp, 128));
    k.ngroups = plan.ngroups; k.gshift = plan.gshift;
    k.q_resid = a.q_resid; k.db_resid_max = a.db_resid_max;
    k.gshift_k = (a.q_resid && a.db_resid_max) ? plan.gshift_k : -1;
    k.k_rows = k.gshift_k == 2 ? plan.k_rows : 0;
    {
        static const int krot = [] { const char* e = knob_env("SQE_KROT"); return e ? atoi(e) : 0; }();
        static const int dbg = [] { const char* e = knob_env("SQE_DBG"); return e ? atoi(e) : 0; }();
        k.krot = krot;
        k.dbg = dbg;
    }
    k.tile_step = a.tile_step > 0 ? a.tile_step : 1;
    k.cand = a.cand; k.cand_cnt = a.cand_cnt; k.gmax = a.gmax; k.dbg_counters = a.dbg_counters;
    k.collect_thr = nullptr; k.collect_keys = nullptr; k.collect_cnt = nullptr; k.unc_count = nullptr;
    k.collect_lo = 0; k.collect_hi = 0;
    return k;
}

ScanPlan make_scan_plan(int64_t n_rows, int B, int kp, int cu_count, int k) {
    ScanPlan p;
    p.bn = B > 128 ? 256 : B > 64 ? 128 : 64;          // 128: HBM-bound like 64, half the padding of a 256 block
    p.qblocks = (B + p.bn - 1) / p.bn;
    p.b_pad = p.qblocks * p.bn;
    p.n_tiles = (int)((n_rows + SCAN_BM - 1) / SCAN_BM);
    int chunks = cu_count / p.qblocks;                 // one persistent workgroup per CU
    if (chunks < 1) chunks = 1;
    // chunk c folds its maxima into column c % 64 of the global-bound table (one row per query slice), so the
    // chunk count need not be a multiple of 64
    if (chunks > p.n_tiles) chunks = p.n_tiles > 0 ? p.n_tiles : 1;
    // tiles are dealt out evenly (chunk_tile_range): exactly `chunks` chunks, none empty
    p.n_chunks = chunks;
    p.tiles_per_chunk = p.n_tiles > 0 ? (p.n_tiles + chunks - 1) / chunks : 0;
    p.kp = kp;
    p.ngroups = 1;
    // global bound: 64 >> gshift groups, each contributing one distinct row, must be >= kp
    p.gshift = kp <= 16 ? 2 : kp <= 32 ? 1 : kp <= 64 ? 0 : -1;
    // k-row bound (scan_common.h: refresh_apply): fewer, larger groups; pointless where it names the same groups
    p.gshift_k = k < 1 ? -1 : k <= 16 ? 2 : k <= 32 ? 1 : k <= 64 ? 0 : -1;
    if (p.gshift_k <= p.gshift) p.gshift_k = -1;
    p.k_rows = (p.gshift_k == 2 && k <= 16) ? k : 0;
    if (p.n_chunks < GMAX_COLS) { p.gshift = p.gshift_k = -1; p.k_rows = 0; }   // a column without a chunk: no cross-chunk bound
    return p;
}

// SQE_SCAN128=2: the older two-stage ring of the 128-query tile (A/B comparisons)
static bool bn128_two_stage() {
    static const bool v = [] { const char* e = knob_env("SQE_SCAN128"); return e && e[0] == '2'; }();
    return v;
}

int launch_scan_collect(const ScanPlan& plan, const ScanArgs& a, hipStream_t stream) {
    if (!a.collect_thr || !a.collect_keys || !a.collect_cnt || !a.unc_count)
        return fail(SQE_ERR_INVALID, "scan collect: missing buffers");
    if (plan.bn == 256) return launch_cfg<2, 4, 8, 4, 2>(plan, a, stream, true);
    if (plan.bn == 128) return bn128_two_stage() ? launch_cfg<4, 2, 4, 4, 2>(plan, a, stream, true)
                                                 : launch_cfg<4, 2, 4, 4, 3, 2>(plan, a, stream, true);
    return launch_cfg<8, 1, 2, 4, 3>(plan, a, stream, true);
}

int launch_scan_bf16(const ScanPlan& plan, const ScanArgs& a, hipStream_t stream) {
    if (a.K % SCAN_BK != 0) return fail(SQE_ERR_INVALID, "scan: dim must be a multiple of 64");
    if (plan.kp < 1 || plan.kp > MAX_KP) return fail(SQE_ERR_INVALID, "scan: kp out of range");
    if (plan.bn == 256) {
        // the ping-pong schedule (scan_pp.hip); SQE_SCAN=v0 in a knobs build picks the two-stage form below
        static const bool two_stage = [] { const char* e = knob_env("SQE_SCAN"); return e && e[0] == 'v'; }();
        if (two_stage) return launch_cfg<2, 4, 8, 4, 2>(plan, a, stream);
        return launch_scan_bf16_pp(plan, a, stream);
    }
    if (plan.bn == 128) return bn128_two_stage() ? launch_cfg<4, 2, 4, 4, 2>(plan, a, stream)
                                                 : launch_cfg<4, 2, 4, 4, 3, 2>(plan, a, stream);
    return launch_cfg<8, 1, 2, 4, 3>(plan, a, stream);
}

}  // namespace sqe
