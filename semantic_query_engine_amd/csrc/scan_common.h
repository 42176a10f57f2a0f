// scan_common.h -- device code shared by the scan kernels: kernel argument block and the
// fused top-k filter (thresholds, candidate lists, cross-chunk bound, compaction).
//
// Filter design
//   * Every persistent workgroup (chunk c, query block) keeps per query: a threshold key
//     (LDS), a candidate list (global, CAND_CAP slots) and the best score it has seen.
//   * A row survives when its key exceeds the threshold.  The threshold is the maximum of
//       - the LOCAL bound: the kp-th best key of this chunk so far (set by list compaction);
//       - the GLOBAL bound: every chunk folds its per-query maximum score (atomic max) into
//         column c % 64 of a small table gmax[query slice of 64][64 columns][64 queries]; a
//         column holds the maximum over the chunks that share it -- disjoint sets of rows, so
//         the 64 columns of a query name 64 distinct real rows -- and min over g-sized groups of
//         max-in-group is a score that at least 64/g >= kp distinct rows reach, hence a lower
//         bound of the final kp-th best score.  With all chunks streaming in parallel this
//         bound tracks the whole index, not one chunk, so survivors become rare after the
//         first few tiles; with more than 64 chunks (batches <= 768: up to 256 chunks) every
//         column is the maximum of up to four chunks, which puts the bound 4 x deeper into the
//         tail than a table row per 64 chunks did (r01), for the same one-row fetch.
//         (Columns-major inside a slice: a lane owns a query and walks the 64 columns with
//         conflict-free LDS reads, no cross-lane traffic.)
//   * The first tile of a chunk runs in BOOT mode: it only folds per-lane maxima into the
//     chunk maximum (no appends, so no first-tile append storm); that tile is scanned again,
//     normally, at the end of the chunk.
//   * The gmax rows a workgroup needs are fetched with global_load_lds (no VGPR loads in the
//     pipelined loops: a VGPR load would make hipcc drain the DMA queue).
#pragma once

#include <type_traits>

#include "kernels.h"

namespace sqe {

constexpr int SCAN_THREADS = 512;
constexpr int SCAN_NWAVES = 8;
constexpr int SCAN_ROW_BYTES = SCAN_BK * 2;   // 128 B per tile row per 64-wide bf16 K step
constexpr int GSLICE_Q = 64;                  // queries refreshed per gmax fetch
constexpr int GSTAGE_BYTES = GSLICE_Q * GMAX_COLS * 4;   // 16 KiB

struct ScanKernelArgs {
    const bf16_t* db;
    const bf16_t* q;
    int64_t n_rows;
    int K;
    int db_pitch;        // bytes between DB rows of the scanned copy
    int q_pitch;         // bytes between query rows
    int B;
    int b_pad;
    int n_tiles;
    int tiles_per_chunk;
    int n_chunks;
    int qblocks;
    int kp;
    int trig;            // compaction trigger (kp <= trig <= CAND_CAP - SCAN_BM)
    int ngroups;         // table rows per query slice (1: chunk c folds its maxima into column c % 64)
    int gshift;          // log2 of the group size g used by the global bound; < 0: bound off
    int gshift_k;        // log2 of the group size of the k-row bound (threshold = bound - slack); < 0: off
    int k_rows;          // > 0 (then gshift_k == 2, k <= 16): the k-row bound is the k-th largest of the 16 quad maxima, not their minimum
    const float* q_resid;            // [B] error-bound inputs of the k-row bound (kernels.h: scan_eps)
    const uint32_t* db_resid_max;
    int tile_step;       // scanned tile t is DB tile t * tile_step (1: every tile; > 1: the row sample of the int8 path's threshold pass)
    int krot;            // workgroup w walks K rotated by w * krot steps
    int dbg;             // timing experiments only (SQE_DBG): 1 = no MFMA / LDS reads, 2 = no DMA in the loop, 4 = no filter,
                         //   8 = no global-bound refresh, 16 = filter fast path only
    uint64_t* cand;
    int* cand_cnt;
    uint32_t* gmax;      // [b_pad/64][GMAX_COLS][64] orderable scores, 0 = nothing yet
    unsigned long long* dbg_counters;   // [8] or null: 0 appends, 1 slow-path wave entries, 2 compactions
    // COLLECT mode (second pass for queries whose certificate failed): fixed per-query thresholds,
    // every row at or above them is appended to a per-query global buffer; no lists, no bound exchange
    const float* collect_thr;    // [b_pad] or null (normal mode); +inf = query not collected
    uint64_t* collect_keys;      // [B][EXACT_CAP]
    int* collect_cnt;            // [B]
    const int* unc_count;        // number of collected (uncertified) queries: the batch size of this pass, on the device
    int collect_lo, collect_hi;  // this launch runs when collect_lo <= count <= collect_hi (one plan per range of counts)
};

// LDS block of the filter state for a query block of BN queries (after the staging area).
template <int BN>
struct FilterLds {
    static constexpr int OFF_THR_KEY = 0;                       // uint64 [BN]
    static constexpr int OFF_THR_S = OFF_THR_KEY + BN * 8;      // float  [BN]
    static constexpr int OFF_CNT = OFF_THR_S + BN * 4;          // int    [BN]
    static constexpr int OFF_CMAX = OFF_CNT + BN * 4;           // uint32 [BN] chunk max (orderable)
    static constexpr int OFF_SLACK = OFF_CMAX + BN * 4;         // float  [BN] 2 eps + margin of the k-row bound
    static constexpr int OFF_FLAGS = OFF_SLACK + BN * 4;        // int    [16]
    static constexpr int OFF_GSTAGE = OFF_FLAGS + 64;           // uint32 [GSLICE_Q][GMAX_COLS]
    static constexpr int BYTES = OFF_GSTAGE + GSTAGE_BYTES;
};

// Filter event counters (appends, slow-path entries, compactions) are compiled in only with
// -DSQE_FILTER_COUNTERS (make COUNTERS=1) and then switched on by SQE_DBG bit 32: a global atomic in
// the filter, even behind a run-time flag, is one more pending VMEM event hipcc has to guard against.
#ifdef SQE_FILTER_COUNTERS
#define SQE_COUNT(f, idx, cond)                                                    \
    do {                                                                           \
        if ((f).dbg_counters && (cond)) atomicAdd(&(f).dbg_counters[idx], 1ull);   \
    } while (0)
#else
#define SQE_COUNT(f, idx, cond) \
    do {                        \
    } while (0)
#endif

// Timing-experiment bits of ScanKernelArgs::dbg: a compile-time zero unless built with -DSQE_DEBUG_KNOBS,
// so the shipped kernels carry none of the no-MFMA / no-DMA / no-filter paths.
#ifdef SQE_DEBUG_KNOBS
#define SQE_DBG_BITS(p) ((p).dbg)
#else
#define SQE_DBG_BITS(p) 0
#endif

// s_waitcnt vmcnt(0) the compiler can see (expcnt / lgkmcnt fields left at their maxima)
__device__ __forceinline__ void wait_vm0_visible() { __builtin_amdgcn_s_waitcnt(0x0F70); }

struct Filter {
    uint64_t* cand_base;   // this workgroup's lists: [BN][CAND_CAP]
    uint32_t* gmax_mine;   // this chunk's column of slice 0 of the block: query q at [(q/64)*gstride + q%64]
    uint64_t* thr_key;
    float* thr_s;
    int* cnt;
    uint32_t* cmax;
    const float* slack;    // per query: what the k-row bound is lowered by
    int* flags;            // [8] per owner wave: an owned list reached the compaction trigger
    int64_t n_rows;
    int q_live;            // live queries in this block
    int trig;
    int per_wave;          // queries owned per wave (BN / 8)
    int gstride;           // ngroups * GMAX_COLS * 64: uint32 elements between consecutive query slices
    bool dbg_no_slow;      // timing experiments only: pretend no row survives
    unsigned long long* dbg_counters;
    uint64_t* collect_keys;    // COLLECT mode: per-query global buffers of this query block, else null
    int* collect_cnt;
};

// Tiles [begin, end) of a chunk: the n_tiles are dealt out as evenly as possible, so EVERY one of the
// n_chunks chunks exists (the global bound needs all 64 columns of a table row published: a plan that
// rounded the chunk length up could end with fewer chunks than columns and silently lose the bound).
__device__ __forceinline__ void chunk_tile_range(int n_tiles, int n_chunks, int chunk, int& begin, int& end) {
    const int base = n_tiles / n_chunks, rem = n_tiles - base * n_chunks;
    begin = chunk * base + min(chunk, rem);
    end = begin + base + (chunk < rem ? 1 : 0);
}

// The table row of a query block and the column chunk `chunk` folds its maxima into (shared with the chunks
// c + 64, c + 128, ...: all of them hold different rows of the index).
__device__ __forceinline__ void bound_rows(const ScanKernelArgs& p, int chunk, int q0, const uint32_t*& read_row, uint32_t*& mine) {
    uint32_t* slice0 = p.gmax + (size_t)(q0 / 64) * p.ngroups * (GMAX_COLS * 64);
    read_row = slice0;
    mine = slice0 + (chunk % GMAX_COLS) * 64;
}

// host: kernel argument block from a plan (scan.hip)
ScanKernelArgs make_kernel_args(const ScanPlan& plan, const ScanArgs& a);

// fold a chunk maximum into its (shared) table column: device-scope atomic max, no return value
__device__ __forceinline__ void publish_max_u32(uint32_t* p, uint32_t v) {
    (void)__hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------- compaction
// kth largest of the (unique, non-zero) keys held as k[j] by the wave; zero = empty slot.
template <int NREG>
__device__ __forceinline__ uint64_t wave_select_kth(const uint64_t (&k)[NREG], int nreg, int kth) {
    uint64_t prefix = 0;
    int remaining = kth;
    for (int bit = 63; bit >= 0; --bit) {
        const uint64_t trial = (prefix >> bit) | 1ull;
        int c = 0;
#pragma unroll
        for (int j = 0; j < NREG; ++j)
            if (j < nreg) c += __popcll(__ballot((k[j] >> bit) == trial));
        if (c >= remaining) prefix |= (1ull << bit);
        else remaining -= c;
    }
    return prefix;
}

// Wave-level compaction of one candidate list to its best `kp` keys; raises the threshold of
// that query to the kp-th best.  Caller guarantees n > kp and exclusive access to the list.
__device__ __forceinline__ void compact_list(uint64_t* list, int n, int kp, int lane,
                                             int* cnt_slot, float* thr_s_slot, uint64_t* thr_key_slot) {
    constexpr int NREG = CAND_CAP / 64;
    const int nreg = (n + 63) >> 6;
    uint64_t k[NREG];
#pragma unroll
    for (int j = 0; j < NREG; ++j) {
        const int i = j * 64 + lane;
        k[j] = (j < nreg && i < n) ? list[i] : 0ull;
    }
    const uint64_t T = wave_select_kth<NREG>(k, nreg, kp);
    int base = 0;
#pragma unroll
    for (int j = 0; j < NREG; ++j) {
        if (j < nreg) {
            const bool keep = k[j] >= T && k[j] != 0ull;
            const uint64_t m = __ballot(keep);
            if (keep) list[base + __popcll(m & ((1ull << lane) - 1ull))] = k[j];
            base += __popcll(m);
        }
    }
    if (lane == 0) {
        *cnt_slot = base;                // == kp
        if (T > *thr_key_slot) {
            *thr_key_slot = T;
            *thr_s_slot = key_score(T);
        }
    }
}

// Sweep over the `per_wave` queries a wave owns: any list with at least `limit` entries is
// cut back to its best kp.  Wave-uniform call.
__device__ __forceinline__ void compact_owned(const Filter& f, int first_q, int per_wave, int limit,
                                              int kp, int lane) {
    const int myq = first_q + lane;
    const bool need = lane < per_wave && f.cnt[myq] >= limit;
    uint64_t mask = __ballot(need);
    while (mask) {
        const int bq = first_q + (int)__builtin_ctzll(mask);
        mask &= mask - 1;
        SQE_COUNT(f, 2, lane == 0);
        compact_list(f.cand_base + (size_t)bq * CAND_CAP, f.cnt[bq], kp, lane, &f.cnt[bq], &f.thr_s[bq],
                     &f.thr_key[bq]);
    }
}

// ---------------------------------------------------------------- per-tile filter
// Accumulator layout of a wave: acc[i][j][r] = score(tile row  row0 + i*16 + (lane>>4)*4 + r,
//                                               query col col0 + j*16 + (lane&15)).

// BOOT mode: fold the tile's per-lane maxima into the chunk maxima (no appends).
template <int FM, int FN>
__device__ __forceinline__ void filter_boot(const f32x4 (&acc)[FM][FN], const Filter& f, int64_t tile_row0,
                                            int row0, int col0, int lane) {
    const bool partial = tile_row0 + SCAN_BM > f.n_rows;     // only the last tile of the index
    const int64_t row_base = tile_row0 + row0 + (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[i][j][r];
                if (partial && row_base + i * 16 + r >= f.n_rows) v = -INFINITY;
                mx = fmaxf(mx, v);
            }
        const int qcol = col0 + j * 16 + (lane & 15);
        if (mx > -INFINITY) atomicMax(&f.cmax[qcol], f32_orderable(mx + 0.0f));
    }
}

// Publish the chunk maxima of the queries a wave owns (after every wave's filter_boot).
__device__ __forceinline__ void publish_cmax(const Filter& f, int first_q, int per_wave, int lane) {
    if (lane < per_wave && first_q + lane < f.q_live) {
        const uint32_t m = f.cmax[first_q + lane];
        const int q = first_q + lane;
        if (m) publish_max_u32(f.gmax_mine + (size_t)(q >> 6) * f.gstride + (q & 63), m);
    }
}

// OR over the 64 lanes of a wave (all lanes active): four DPP row shifts leave the OR of a 16-lane row in its last
// lane, four v_readlane collect the rows.  Uniform result.
__device__ __forceinline__ unsigned wave_or_u32(unsigned v) {
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);    // row_shr:1, zero fill
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);    // row_shr:8
    return (unsigned)__builtin_amdgcn_readlane((int)v, 15) | (unsigned)__builtin_amdgcn_readlane((int)v, 31) |
           (unsigned)__builtin_amdgcn_readlane((int)v, 47) | (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// accumulator number t (fragment t >> 2, element t & 3) of column group J, t uniform: a switch, i.e. a short
// compare tree in front of FM * 4 one-instruction cases -- registers cannot be indexed by a run-time value
template <int FM, int FN, int J>
__device__ __forceinline__ float pick_acc(const f32x4 (&acc)[FM][FN], int t) {
    static_assert(FM == 2 || FM == 4 || FM == 8, "fragment rows per wave");
    float v = 0.f;
    // every case is an opaque one-instruction copy: without the asm hipcc turns the switch into an array in scratch
    // memory (FM * 4 stores + one indexed load per call -- vector memory traffic in the middle of the DMA ring)
#define SQE_PICK(n)                                                          \
    case n:                                                                  \
        if constexpr ((n) < FM * 4) asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "v"(acc[((n) >> 2) % FM][J][(n) & 3])); \
        break;
#define SQE_PICK4(n) SQE_PICK(n) SQE_PICK((n) + 1) SQE_PICK((n) + 2) SQE_PICK((n) + 3)
    switch (t) {
        SQE_PICK4(0) SQE_PICK4(4) SQE_PICK4(8) SQE_PICK4(12) SQE_PICK4(16) SQE_PICK4(20) SQE_PICK4(24) SQE_PICK4(28)
        default: break;
    }
#undef SQE_PICK4
#undef SQE_PICK
    return v;
}

// Slow path of one column group J of a finished tile.  Code size is what matters here: the path runs a few times per
// tile in a wave or two, from instructions that are cold by then, so the r01 form -- the whole append sequence
// unrolled at each of the FM * 4 accumulators of each group, ~180 KB of kernel -- spent most of its time fetching
// instructions.  Now: (1) one pass of v_cmp + v_addc shifts, per lane, a bit per accumulator at or above the query's
// threshold into a mask; (2) ONE LDS atomic per lane reserves its list slots; (3) a scalar loop walks the bits set
// in ANY lane, fetches that accumulator (pick_acc) and runs one copy of the append sequence -- no LDS round trip
// inside -- for the lanes that have the bit.
template <int FM, int FN, int J, bool COLLECT>
__device__ __forceinline__ bool filter_group(const f32x4 (&acc)[FM][FN], const Filter& f, int64_t row_base, bool partial,
                                             int qcol, float thr, uint32_t cmax0) {
    unsigned m = 0;
    // element t = i * 4 + r ends up in bit t: every step shifts the mask left and adds the compare bit, last element first
#pragma unroll
    for (int t = FM * 4 - 1; t >= 0; --t)
        asm volatile("v_cmp_ge_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(acc[t >> 2][J][t & 3]), "v"(thr) : "vcc");
    if (partial) {                                  // last tile of the index: rows past the end are not rows
        const int left = (int)min((int64_t)SCAN_BM, max((int64_t)0, f.n_rows - row_base));   // valid rows from row_base on
        unsigned valid = 0;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int n = min(4, max(0, left - i * 16));          // valid elements of fragment i
            valid |= ((1u << n) - 1u) << (i * 4);
        }
        m &= valid;
    }
    if (qcol >= f.q_live) m = 0;
    if (f.dbg_no_slow) m = 0;
    unsigned todo = wave_or_u32(m);
    if (todo == 0) return false;                    // the threshold has risen since the fast test
    SQE_COUNT(f, 1, (threadIdx.x & 63) == 0);
    // Keys are appended on score alone (>= threshold): at equal score a key below the threshold KEY (higher row id)
    // may come along, and the next compaction drops it again.
    const int mine = __popc(m);
    int slot = 0;
    if (mine) slot = atomicAdd(COLLECT ? &f.collect_cnt[qcol] : &f.cnt[qcol], mine);
    uint32_t best = 0;
    if constexpr (!COLLECT) {
        if (mine && slot + mine >= f.trig) f.flags[qcol / f.per_wave] = 1;
    }
    uint64_t* list = COLLECT ? f.collect_keys + (size_t)qcol * EXACT_CAP : f.cand_base + (size_t)qcol * CAND_CAP;
    while (todo) {
        const int t = __builtin_ctz(todo);          // uniform
        todo &= todo - 1;
        const float sc = pick_acc<FM, FN, J>(acc, t);
        if (m & (1u << t)) {
            const int64_t row = row_base + (t >> 2) * 16 + (t & 3);
            const uint64_t key = make_key(sc + 0.0f, (uint32_t)row);
            if (!COLLECT || slot < EXACT_CAP) list[slot] = key;
            ++slot;
            SQE_COUNT(f, 0, true);
            best = max(best, (uint32_t)(key >> 32));
        }
    }
    if constexpr (!COLLECT) {
        // the chunk maximum only rises, so the stale cmax0 only publishes a maximum that is already known
        if (best > cmax0) {
            atomicMax(&f.cmax[qcol], best);          // no return value: fire and forget
            publish_max_u32(f.gmax_mine + (size_t)(qcol >> 6) * f.gstride + (qcol & 63), best);
        }
    }
    return !COLLECT && mine != 0;
}

// Normal mode.  Returns true when this wave stored candidates (the caller drains its stores
// before the next barrier so other waves can read the lists).
// `cols`: wave-uniform mask of the column groups to look at (the ping-pong kernel finds the groups that hold a
// survivor under its MFMAs and only those come here); all groups by default, behind a max test of their own.
template <int FM, int FN, bool COLLECT = false>
__device__ __forceinline__ bool filter_tile(const f32x4 (&acc)[FM][FN], const Filter& f, int64_t tile_row0,
                                            int row0, int col0, int lane, unsigned cols = ~0u) {
    static_assert(FN == 4, "four column groups per wave");
    bool stored = false;
    const int64_t row_base = tile_row0 + row0 + (lane >> 4) * 4;
    const bool partial = tile_row0 + SCAN_BM > f.n_rows;
    const bool pretested = cols != ~0u;
    // thresholds and chunk maxima of the four groups: one LDS round trip for all of them
    float thr[FN];
    uint32_t cmax0[FN];
#pragma unroll
    for (int jj = 0; jj < FN; ++jj) {
        const int qcol = col0 + jj * 16 + (lane & 15);
        thr[jj] = f.thr_s[qcol];
        cmax0[jj] = COLLECT ? 0u : f.cmax[qcol];
    }
    auto group = [&](auto jc) {
        constexpr int J = decltype(jc)::value;
        if (!(cols & (1u << J))) return;
        const int qcol = col0 + J * 16 + (lane & 15);
        if (!pretested) {
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < FM; ++i)
                mx = fmaxf(mx, fmaxf(fmaxf(acc[i][J][0], acc[i][J][1]), fmaxf(acc[i][J][2], acc[i][J][3])));
            if (!__any(mx >= thr[J])) return;
        }
        stored |= filter_group<FM, FN, J, COLLECT>(acc, f, row_base, partial, qcol, thr[J], cmax0[J]);
    };
    group(std::integral_constant<int, 0>{});
    group(std::integral_constant<int, 1>{});
    group(std::integral_constant<int, 2>{});
    group(std::integral_constant<int, 3>{});
    return __any(stored);
}

// ---------------------------------------------------------------- global bound refresh
// Issue the LDS-DMA fetch of one query slice of the table (64 columns x 64 queries x 4 B =
// 16 KiB, contiguous): 16 pieces of 1 KiB, wave w issues pieces w and w + 8.
template <bool ASM_DMA = false>
__device__ __forceinline__ void refresh_issue(const uint32_t* gmax_block_group, int gstride, int sl,
                                              char* gstage, int wave, int lane) {
    const char* src = reinterpret_cast<const char*>(gmax_block_group + (size_t)sl * gstride) + lane * 16;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int piece = wave + 8 * t;
        if constexpr (ASM_DMA) {
            lds_dma16_sc1(src + piece * 1024, gstage + piece * 1024);
            continue;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 1024),
                                         (__attribute__((address_space(3))) void*)(gstage + piece * 1024),
                                         16, 0, /*aux: sc1*/ 16);
    }
}

// After the fetch has landed: ONE wave folds the slice, lane = query.  Walks the 64 columns
// (conflict-free ds_read_b32, no cross-lane traffic) four at a time and keeps, for group sizes 1, 2 and 4, the
// minimum over the groups of the maximum inside a group.  Two bounds come out of it:
//   * the kp-row bound (group size 2^gshift, 64 >> gshift >= kp groups): a score kp distinct rows reach;
//   * the k-row bound (group size 2^gshift_k, 64 >> gshift_k >= k groups) LOWERED BY 2 eps: L is a scan score k
//     distinct rows reach, so their true cosines are >= L - eps, and a row whose scan score is below L - 2 eps
//     has a true cosine below L - eps: it cannot be in the exact top-k.  Rows dropped under this bound need no
//     certificate; with k = 10 of kp = 64 it sits well above the kp-row bound (fewer, larger groups beat the
//     slack of ~0.16 sigma on 1024-d data).  With k <= 16 the bound is sharper still: the k-th largest of the 16
//     group maxima instead of their minimum (k_rows).
// The query's threshold becomes the larger of the two (and of what it was).
template <int READS_IN_FLIGHT = GMAX_COLS>
__device__ __forceinline__ void refresh_apply(const Filter& f, const char* gstage, int sl, int gshift, int gshift_k, int k_rows, int lane) {
    const uint32_t* st = reinterpret_cast<const uint32_t*>(gstage) + lane;
    uint32_t b1 = 0xFFFFFFFFu, b2 = 0xFFFFFFFFu, b4 = 0xFFFFFFFFu;
    uint32_t qm[GMAX_COLS / 4];                     // the 16 quad maxima: 16 distinct rows of the index
    // fully unrolled by default: all 64 reads are in flight together (ONE LDS round trip; four iterations per trip took
    // four, and the other seven waves of the workgroup wait at the phase's barrier for this one).  READS_IN_FLIGHT = 16
    // for the kernel that has no registers for that (the two-stage 256-query A/B form).
    constexpr int UNROLL = READS_IN_FLIGHT / 4;
#pragma unroll UNROLL
    for (int c = 0; c < GMAX_COLS; c += 4) {
        const uint32_t v0 = st[c * 64], v1 = st[(c + 1) * 64], v2 = st[(c + 2) * 64], v3 = st[(c + 3) * 64];
        const uint32_t p0 = max(v0, v1), p1 = max(v2, v3);
        b1 = min(b1, min(min(v0, v1), min(v2, v3)));
        b2 = min(b2, min(p0, p1));
        qm[c >> 2] = max(p0, p1);
        b4 = min(b4, qm[c >> 2]);
    }
    const uint32_t bound = gshift == 0 ? b1 : gshift == 1 ? b2 : gshift == 2 ? b4 : 0u;
    uint32_t lk = gshift_k == 0 ? b1 : gshift_k == 1 ? b2 : gshift_k == 2 ? b4 : 0u;
    if (k_rows > 0) {
        // k <= 16 of the 16 quad maxima suffice: the k-th LARGEST of them is a score k distinct rows reach -- the
        // minimum is the 16th largest.  Batcher's odd-even merge sort on 16 registers (63 compare-exchanges), then
        // the k-th by a chain of selects on the uniform k.  (Unpublished quads are 0 and sort to the bottom.)
        constexpr int N = GMAX_COLS / 4;
#pragma unroll
        for (int pp = 1; pp < N; pp <<= 1)
#pragma unroll
            for (int kk = pp; kk >= 1; kk >>= 1)
#pragma unroll
                for (int j = kk % pp; j + kk < N; j += 2 * kk)
#pragma unroll
                    for (int i = 0; i < kk; ++i)
                        if (i + j + kk < N && (i + j) / (2 * pp) == (i + j + kk) / (2 * pp)) {
                            const uint32_t hi = max(qm[i + j], qm[i + j + kk]), lo = min(qm[i + j], qm[i + j + kk]);
                            qm[i + j] = hi;                 // descending
                            qm[i + j + kk] = lo;
                        }
        uint32_t kth = 0;
#pragma unroll
        for (int t = 0; t < N; ++t) kth = (k_rows == t + 1) ? qm[t] : kth;
        lk = kth;
    }
    const int q = sl * GSLICE_Q + lane;
    if (q < f.q_live) {
        uint32_t best = bound;                                   // 0 = some column not published yet / bound off
        if (lk != 0u) best = max(best, f32_orderable(f32_from_orderable(lk) - f.slack[q]));
        if (best != 0u) {
            const uint64_t gk = (uint64_t)best << 32;
            if (gk > f.thr_key[q]) {
                f.thr_key[q] = gk;
                f.thr_s[q] = f32_from_orderable(best);
            }
        }
    }
}

// Filter state of a query block at kernel start (every thread of the workgroup; `collect_thr` non-null: COLLECT
// mode, fixed thresholds).  The slack of the k-row bound is 2 eps plus a margin for the rounding of L - slack.
template <int BN>
__device__ __forceinline__ void filter_init(const ScanKernelArgs& p, const Filter& f, float* slack, int q0, int batch,
                                            const float* collect_thr, int tid) {
    const float dx = (p.gshift_k >= 0 && p.db_resid_max) ? __uint_as_float(*p.db_resid_max) : 0.f;
    for (int i = tid; i < BN; i += SCAN_THREADS) {
        const bool live = (q0 + i) < batch;
        float ts = live ? -INFINITY : INFINITY;
        uint64_t tk = live ? 0ull : ~0ull;
        if (collect_thr && live) {
            // fixed threshold: every row whose scan score is >= collect_thr is collected
            ts = collect_thr[q0 + i];
            const uint32_t o = f32_orderable(ts);
            tk = ts == INFINITY ? ~0ull : ((uint64_t)o << 32) - 1ull;
        }
        f.thr_key[i] = tk;
        f.thr_s[i] = ts;
        f.cnt[i] = 0;
        f.cmax[i] = 0u;
        slack[i] = (live && p.gshift_k >= 0 && p.q_resid) ? 2.0f * scan_eps(p.q_resid[q0 + i], dx, p.K) * 1.000001f + 1.0e-6f : INFINITY;
    }
    if (tid < 16) f.flags[tid] = 0;
}

}  // namespace sqe
