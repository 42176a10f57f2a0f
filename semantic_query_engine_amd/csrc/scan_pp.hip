// scan_pp.hip -- S2, ping-pong form of the bf16 scan for query blocks of 256 (gfx950).
//
// Same contract as scan.hip (256-row DB tile x 256 queries per persistent workgroup,
// fused top-k filter, candidate lists).  What differs is how the two waves that share a SIMD
// are scheduled against each other:
//
//   Roles.  The 8 waves form two groups (waves w and w + 4 sit on the same SIMD): at any time one
//     group is in a COMPUTE phase -- 32 back-to-back MFMAs on operands that are already in
//     registers -- while the other is in a MEMORY phase: it issues its share of the LDS-DMA for a
//     later K slice, reads its next operands from LDS (12 ds_read_b128) and runs whatever filter
//     work is due.  One s_barrier closes every phase and the groups swap roles, so the matrix pipe
//     of each SIMD always has one wave feeding it and nothing a wave waits for (LDS latency, DMA
//     issue, filter VALU) sits between two of its own MFMAs.  Only ONE operand register set is
//     needed (a wave never loads and computes at once): 128 accumulator + 48 operand VGPRs.
//
//   Half-steps.  The unit of the pipeline is a 32-wide K slice of the 256 x 256 tile: 256 rows x
//     64 B of DB rows plus 256 x 64 B of queries = 32 KiB, one ring stage; 4 stages.  An LDS line
//     (128 B) holds the slice of tile row L (chunks 0-3) and of row L + 128 (chunks 4-7), chunk
//     positions XOR-swizzled by (L >> 1) & 7 -- the same conflict-free ds_read_b128 pattern the
//     other scan kernels use, built by the per-lane source addresses of global_load_lds.
//
//   Timeline (phase p, half-step j; G0 = waves 0-3, G1 = waves 4-7):
//         p = 2j     G0: MEM(j)      G1: CMP(j-1)
//         p = 2j+1   G0: CMP(j)      G1: MEM(j)
//     MEM(j) reads the operands of half-step j and issues the DMA of half-step j + 3 into the
//     stage half-step j - 1 used, whose last readers (both groups' MEM(j-1)) retired their reads
//     before an earlier barrier.  A counted wait at the end of MEM(j) retires this wave's DMA of
//     half-step j + 1 (issued two of its MEM phases earlier) and leaves j + 2 and j + 3 in flight.
//
//   Filter.  The fast-path test of a finished tile costs no time of its own: the last half-step of a tile issues
//     its MFMAs column group by column group (cmp_phase_last); a group's 8 accumulators are final after 8 MFMAs
//     and its test -- 16 v_max3_i32 and one compare against the query thresholds -- issues under the MFMAs of the
//     next group, leaving a wave-uniform mask of the groups that hold a survivor and, if there is one, a mark
//     in LDS.  G0 runs one phase ahead of G1, so after its last compute phase of a tile it first loads the
//     operands of the next tile's first half-step; then, ONLY IF some wave marked a survivor, both groups run
//     the slow path (appends to the candidate lists) of their marked column groups in one common phase,
//     followed by the cross-wave steps (publishing boot maxima, list compaction).  A tile without survivors
//     goes straight on.  The first compute phase of a tile takes a zero C operand, so accumulators are never
//     cleared by VALU moves.  (r01 ran fast and slow path of every tile in a common phase without MFMAs.)
//
//   This is the default schedule for 256-query blocks (scan.hip: SCAN_DEFAULT_KERNEL); measured against
//   the two-stage form it is 1-5 % faster at every index size and batch tried (profiles/r01_search).
#include <stdlib.h>

#include "scan_common.h"

namespace sqe {

namespace {

constexpr int HALF_K = 32;                         // k elements per half-step
constexpr int LINE_BYTES = 128;                    // LDS line: the slices of tile rows L and L + 128
constexpr int OPER_BYTES = 128 * LINE_BYTES;       // 16 KiB: one operand of one half-step
constexpr int STAGE_BYTES = 2 * OPER_BYTES;        // DB rows, then queries
constexpr int NSTAGE = 4;
constexpr int BNP = 256;
constexpr int OFF_F = NSTAGE * STAGE_BYTES;        // 128 KiB
using FLP = FilterLds<BNP>;
constexpr int LDS_BYTES = OFF_F + FLP::BYTES;
constexpr int NSLICEP = BNP / GSLICE_Q;

#define PP_BARRIER()                           \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)

// In-kernel phase timing (make KNOBS=1 STAMPS=1|2, SQE_DBG bit 32): core-clock stamps (s_memtime) around the
// steady-state phases of the middle half-steps, summed per wave; workgroups 0 and 100 write their sums at the end
// (api.hip prints them).  STAMPS=1 stamps the compute side only (it waits at the barrier anyway).
#ifdef SQE_PHASE_STAMPS
struct PhaseClock {
    unsigned long long cmp = 0, cmp_bar = 0, mem_issue = 0, mem_reads = 0, mem_wait = 0, mem_bar = 0, phases = 0;
    // tile boundary, group 0: [0] first compute phase, [1] barrier, [2] memory phase, [3] barrier, [4] last compute phase
    // (with the fast-path test), [5] barrier, [6] memory phase of the next tile's first half-step, [7] barrier,
    // [8] tile_end (common slow phase), [9] middle half-steps that took the general memory phase (whole half-step),
    // [10] their number, [11] tiles
    unsigned long long bnd[12] = {};
    // tile_end: [0] read of the survivor mark, [1] own slow path, [2] barrier, [3] entry_sync, [4] times this wave had a
    // flagged group, [5] times the common phase ran
    unsigned long long te[6] = {};
};
#define PP_STAMP(var)                                  \
    do {                                               \
        __builtin_amdgcn_sched_barrier(0);             \
        var = __builtin_readcyclecounter();            \
        __builtin_amdgcn_sched_barrier(0);             \
    } while (0)
#define PP_DECL(...) unsigned long long __VA_ARGS__
#define PP_ACC(...) \
    do {            \
        __VA_ARGS__; \
    } while (0)
#else
#define PP_STAMP(var) \
    do {              \
    } while (0)
#define PP_DECL(...) \
    do {             \
    } while (0)
#define PP_ACC(...) \
    do {            \
    } while (0)
#endif
// stamps inside the memory phase lengthen it (each is an s_memtime the next use waits for): only with STAMPS=2
#if defined(SQE_PHASE_STAMPS) && SQE_PHASE_STAMPS >= 2
#define PP_DECL2(...) unsigned long long __VA_ARGS__
#define PP_STAMP2(var) PP_STAMP(var)
#else
#define PP_DECL2(...) \
    do {              \
    } while (0)
#define PP_STAMP2(var) \
    do {               \
    } while (0)
#endif
#if defined(SQE_PHASE_STAMPS) && SQE_PHASE_STAMPS >= 2
#define PP_STAMP_MEM(var) PP_STAMP(var)
#elif defined(SQE_PHASE_STAMPS)
#define PP_STAMP_MEM(var) \
    do {                  \
        var = 0;          \
    } while (0)
#else
#define PP_STAMP_MEM(var) \
    do {                  \
    } while (0)
#endif

// Waits the compiler can see (gfx9 encoding: vmcnt in [3:0] and [15:14], expcnt [6:4] left at its maximum, lgkmcnt
// [11:8]): after them hipcc knows its own LDS reads have returned and adds no s_waitcnt of its own behind them.
#define PP_WAIT(imm)                               \
    do {                                           \
        __builtin_amdgcn_sched_barrier(0);         \
        __builtin_amdgcn_s_waitcnt(imm);           \
        __builtin_amdgcn_sched_barrier(0);         \
    } while (0)
#define PP_WAIT_VM8_LGKM0() PP_WAIT(0x0078)
#define PP_WAIT_VM7_LGKM0() PP_WAIT(0x0077)
#define PP_WAIT_VM0_LGKM0() PP_WAIT(0x0070)

// DMA pieces go out through inline asm (common.h: lds_dma16): hipcc does not see them, so it never guards an LDS
// read with an s_waitcnt vmcnt(0) of its own (it did, depending on the shape of the surrounding control flow --
// a drained ring every phase).  Every wait for a piece in this kernel is an explicit counted s_waitcnt.
__device__ __forceinline__ void glds16(const char* src, char* lds_wave_base) { lds_dma16(src, lds_wave_base); }

typedef bf16x8 AOps[8];   // [fm]: 128 rows x 32 k
typedef bf16x8 BOps[4];   // [fn]:  64 queries x 32 k

struct Cursor {           // a half-step: (tile entry, 32-wide slice inside it)
    int e, h;
    const char* tile;     // first byte of the DB tile of entry e
};

struct PP {
    const char* qbase;
    char* smem;
    char* gstage;
    const uint32_t* gmax_group;
    unsigned offA0, offA1, offB0, offB1;   // per-lane source offsets of this wave's DMA pieces
    unsigned rdA, rdB;                     // per-lane LDS offsets of the operand reads inside a stage
    int wave, wm, wn;
    bool tile_skew;                        // knobs build, SQE_DBG bit 8192: tiles start (tile % 16) * 2 KiB into their footprint
    int h_stride;                          // bytes between the 32-wide K slices of a DB tile row (64; knobs build: see SQE_DBG bit 8192)
    int tile_begin, nt, HS, J, tile_step;
    long long tile_bytes;
    int kp, trig, gshift, gshift_k, k_rows, refresh_mask;   // refresh_mask + 1: half-steps between bound fetches early in a chunk (power of two)
    int e_fast, e_mid, late_mask;          // bound-table fetch schedule (kernel start)
    Cursor rd;                             // half-step the next MEM phase reads
    Cursor dm;                             // half-step the next MEM phase fetches (rd + 3)
    int refresh_pending, refresh_ctr, refresh_j;
    int lean_until;                        // MEM phases of half-steps < lean_until take the lean form
    bool no_mma, no_dma, no_filter;
    bool bound_on;                         // some cross-chunk bound is in use (kp-row and / or k-row)
    int order;                             // how this wave lines up its DMA pieces and operand reads (dma_and_reads)
    int pend_h, pend_stage;                // query piece 1 of a half-step left for this wave's next compute phase (pend_h < 0: none)
    bool defer_on;
#if defined(SQE_PHASE_STAMPS) && SQE_PHASE_STAMPS >= 2
    unsigned long long mp[5] = {};         // general memory phase: bookkeeping before, pieces + reads, wait, bookkeeping after, count
#endif

    __device__ __forceinline__ void advance(Cursor& c) const {
        if (++c.h == HS) {
            c.h = 0;
            ++c.e;
            c.tile += (c.e == nt) ? -(long long)(nt - 1) * tile_bytes : tile_bytes;   // entry nt = first tile again
        }
    }
    __device__ __forceinline__ int64_t row0_of(int e) const {
        return (int64_t)(e < nt ? tile_begin + e : tile_begin) * tile_step * SCAN_BM;
    }
    // this wave's four 1-KiB pieces of a half-step: DB pieces wave, wave + 8; query pieces likewise
    __device__ __forceinline__ void issue_a(const Cursor& c, int stage) const {
        char* st = smem + stage * STAGE_BYTES;
        const char* as = c.tile + c.h * h_stride;
        // (emulated tiled layout: a row-major tile's footprint is 17 x 32 KiB, so without a skew the half-step blocks of
        // all chunks would sit at the same address modulo 32 KiB -- the same channels and L2 sets)
        if (tile_skew) as += (((c.e < nt ? c.e : 0) + tile_begin) & 15) * 2048;
        glds16(as + offA0, st + wave * 1024);
        glds16(as + offA1, st + (wave + 8) * 1024);
    }
    __device__ __forceinline__ void issue_b(int h, int stage, int piece) const {
        char* st = smem + stage * STAGE_BYTES + OPER_BYTES;
        const char* bs = qbase + h * (HALF_K * 2);
        if (piece == 0) glds16(bs + offB0, st + wave * 1024);
        else glds16(bs + offB1, st + (wave + 8) * 1024);
    }
    __device__ __forceinline__ void issue(const Cursor& c, int stage, bool all = true) const {
        issue_a(c, stage);
        issue_b(c.h, stage, 0);
        if (all) issue_b(c.h, stage, 1);
    }
};

// FIRST: first half-step of a tile -- zero C operand, so accumulators are never cleared by VALU
// moves.  The kernel's loops are nested (tile entries outside, half-steps inside, the first and
// the last half-step peeled) so that the two variants never meet at a control-flow merge: with
// `if (first) ... else ...` inside one flat loop hipcc keeps two accumulator sets and spills.
template <bool FIRST>
__device__ __forceinline__ void cmp_phase(f32x4 (&acc)[8][4], const AOps& a, const BOps& b) {
#pragma unroll
    for (int fm = 0; fm < 8; ++fm)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn)
            acc[fm][fn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                a[fm], b[fn], FIRST ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[fm][fn], 0, 0, 0);
}

// Middle compute phase that also issues the one DMA piece its wave's preceding memory phase left over (mem_lean,
// defer): the memory phase is the longer of the two (4 pieces at ~70 cycles of issue each + 12 reads against 32
// MFMAs), one piece among the MFMAs costs the matrix pipe less than it costs the memory phase.
__device__ __forceinline__ void cmp_phase_mid(PP& P, f32x4 (&acc)[8][4], const AOps& a, const BOps& b) {
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
        for (int fn = 0; fn < 4; ++fn) acc[fm][fn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[fm], b[fn], acc[fm][fn], 0, 0, 0);
    }
}

// Fast-path test in the INTEGER domain: for scores compared against a threshold t >= 0, "some score >= t" is
// "max over the float bits read as int32 >= bits(t)" -- positive floats order like their bits, every negative
// float is a negative int, and a finished accumulator is never -0.0 (the sum starts from a +0.0 C operand).
// v_max3_i32 needs no NaN quieting (fmaxf costs a canonicalising v_max per input under IEEE mode), a NaN
// score is a large int and merely flags its group for the slow path, which compares in float.  Thresholds
// below zero (the first tiles of a chunk) flag everything: int32 minimum.
__device__ __forceinline__ int pp_thr_bits(float t) { return t >= 0.f ? __float_as_int(t) : (int)0x80000000; }

// running max over the four scores of one accumulator fragment: 2 v_max3_i32
__device__ __forceinline__ int pp_fold(int m, const f32x4& v) {
    m = max(max(m, __float_as_int(v[0])), __float_as_int(v[1]));
    return max(max(m, __float_as_int(v[2])), __float_as_int(v[3]));
}

// Last half-step of a tile (never the first: K >= 64 gives two half-steps): MFMAs column group by column group;
// the MFMA of fragment (fm, fn) is followed by the two v_max3_i32 that fold fragment (fm, fn - 1) -- final for 8
// MFMAs by then -- into that group's running maximum, the order requested with sched_group_barrier, so the test
// of three of the four groups rides under the MFMAs (one MFMA = 16 cycles of the matrix pipe, the two VALU
// instructions issue beside it).  Returns the wave-uniform mask of the column groups in which some lane
// holds a score >= its query's threshold (thr: this lane's four thresholds as pp_thr_bits, read from LDS at
// the head of the phase; thresholds only rise, so an older one only flags more).
__device__ __forceinline__ unsigned cmp_phase_last(f32x4 (&acc)[8][4], const AOps& a, const BOps& b, const int (&thr)[4]) {
    int mx[4] = {(int)0x80000000, (int)0x80000000, (int)0x80000000, (int)0x80000000};
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) {
        // An empty asm makes group fn - 1's chain of folds start after group fn - 2's has ended: the scheduler
        // fills the VALU slots of the pipeline below with whatever is ready, and without this order it takes
        // the folds of the fragment the last MFMA has just written (8 wait states + the MFMA's latency each).
        if (fn > 1) asm("" : "+v"(mx[fn - 1]) : "v"(mx[fn - 2]));
#pragma unroll
        for (int fm = 0; fm < 8; ++fm) {
            acc[fm][fn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[fm], b[fn], acc[fm][fn], 0, 0, 0);
            if (fn > 0) mx[fn - 1] = pp_fold(mx[fn - 1], acc[fm][fn - 1]);
        }
    }
    asm("" : "+v"(mx[3]) : "v"(mx[2]));
#pragma unroll
    for (int fm = 0; fm < 8; ++fm) mx[3] = pp_fold(mx[3], acc[fm][3]);
    // the pipeline this region is scheduled to: 8 MFMAs, then 24 x (one MFMA, the two v_max3_i32 of the fragment
    // eight MFMAs back); the last group's 16 v_max3_i32 and the four compares follow
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
#pragma unroll
    for (int t = 0; t < 24; ++t) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
    }
    unsigned mask = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) mask |= __any(mx[j] >= thr[j]) ? (1u << j) : 0u;
    return __builtin_amdgcn_readfirstlane(mask);
}

// After the barrier that ends the common slow-path phase of a finished entry.  Every wave takes the same path
// (the flags are stable here).
__device__ __forceinline__ void entry_sync(const PP& P, const Filter& f, int finished_entry) {
    const int fl = fresh_lane();
    if (finished_entry == 0) {
        publish_cmax(f, P.wave * 32, 32, fl);
    } else {
        const int any_flag = __builtin_amdgcn_readfirstlane(__any(f.flags[fl & 7] != 0));
        if (any_flag) {
            PP_WAIT_VM0_LGKM0();                                          // my appended keys are in memory
            PP_BARRIER();
            if (__builtin_amdgcn_readfirstlane(f.flags[P.wave]) != 0) compact_owned(f, P.wave * 32, 32, P.trig, P.kp, fl);
            // a wait hipcc can see: the list loads of the sweep are retired HERE, so no path back to the loop
            // carries a pending load and no s_waitcnt vmcnt(0) appears in front of the next tile's MFMAs
            PP_WAIT_VM0_LGKM0();
            PP_BARRIER();                                                 // sweeps done before flags are cleared
            if (fl == 0) f.flags[P.wave] = 0;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// operands of half-step j: 12 ds_read_b128 from stage j & 3
__device__ __forceinline__ void read_operands(const PP& P, AOps& a, BOps& b, int j) {
    const char* st = P.smem + (j & 3) * STAGE_BYTES;
#pragma unroll
    for (int fm = 0; fm < 8; ++fm) a[fm] = *reinterpret_cast<const bf16x8*>(st + P.rdA + fm * 2048);
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) b[fn] = *reinterpret_cast<const bf16x8*>(st + OPER_BYTES + P.rdB + fn * 2048);
}

// The work of a memory phase on the two shared units of the CU: this wave's 4 DMA pieces of half-step j + 3 (the
// address unit takes 16 cycles per 1-KiB piece) and its 12 operand reads of half-step j (the LDS takes 4 cycles per
// ds_read_b128).  The four waves of a group are in this phase together; P.order picks how a wave lines the two up:
//   0  pieces, then reads        1  reads, then pieces
// With every wave in order 0 (r01) all four queue at the address unit and then at the LDS, 256 + 192 cycles in a
// row; the default gives waves 0, 1 (4, 5) of a group order 0 and waves 2, 3 (6, 7) order 1, so that the two
// units work side by side (measured at 10 M x 1024, batch 1024: 18.53 / 18.39 / 18.43 / 18.19 ms for all-0 / all-1 /
// stagger by parity / stagger by pairs; four reads then a piece, repeated, in every wave: slower than all of them).
__device__ __forceinline__ void dma_and_reads(const PP& P, AOps& a, BOps& b, int j, bool dma, bool all = true) {
    const int stage = (j + 3) & 3;
    if (P.order == 0) {
        if (dma) P.issue(P.dm, stage, all);
        read_operands(P, a, b, j);
    } else {
        read_operands(P, a, b, j);
        if (dma) P.issue(P.dm, stage, all);
    }
}

// MEMORY phase of half-step j (P.rd): bound work that is due, DMA of half-step j + 3, operand reads.
__device__ __forceinline__ void mem_phase(PP& P, const Filter& f, AOps& a, BOps& b, int j) {
    PP_DECL2(m0, m1, m2, m3, m4);
    PP_STAMP2(m0);
    const bool more = j + 3 < P.J && !P.no_dma;

    // ---- global bound: fold a slice fetched >= 3 half-steps ago (every wave's pieces have been
    // retired by its counted waits and a barrier), then maybe fetch the next one
    if (P.refresh_pending >= 0 && j >= P.refresh_j + 3) {
        if (P.wave == (P.refresh_ctr & 7)) refresh_apply(f, P.gstage, P.refresh_pending, P.gshift, P.gshift_k, P.k_rows, fresh_lane());
        P.refresh_pending = -1;
    }
    // Schedule: entry 1 fetches all slices back to back from a quarter of the tile on
    // (when every chunk has published its boot maxima), then every slice once per tile while the
    // bound still moves fast, one slice per tile later.  At least 4 half-steps between fetches.
    if (P.bound_on && P.rd.e > 0 && more && !P.no_filter && P.refresh_pending < 0 && j >= P.refresh_j + 4) {
        // (schedule: e_fast / e_mid / late_mask, kernel start)
        const bool want = P.rd.e == 1 ? (P.rd.h >= P.HS / 4 && P.refresh_ctr < NSLICEP)
                                      : (P.rd.e <= P.e_fast ? (P.rd.h & P.refresh_mask) == 0
                                                            : (P.rd.h == 0 && (P.rd.e <= P.e_mid || (P.rd.e & P.late_mask) == 0)));
        if (want) {
            P.refresh_pending = P.refresh_ctr % NSLICEP;
            ++P.refresh_ctr;
            P.refresh_j = j;
            refresh_issue<true>(P.gmax_group, f.gstride, P.refresh_pending, P.gstage, P.wave, fresh_lane());
        }
    }

    // ---- DMA of half-step j + 3 into the stage of half-step j - 1, operands of half-step j
    PP_STAMP2(m1);
    dma_and_reads(P, a, b, j, more);
    PP_STAMP2(m2);

    // ---- retire the DMA of half-step j + 1 (two MEM phases old); j + 2 and j + 3 stay in flight.
    // Anything else this wave issued in between (bound fetch, appended keys) only makes the wait
    // retire part of j + 2 as well.
    if (!more) PP_WAIT_VM0_LGKM0();
    else PP_WAIT_VM8_LGKM0();
    PP_STAMP2(m3);

    // ---- first half-step that needs this (general) form again; until then EVERY memory phase takes the lean form
    // (mem_any).  This function costs ~550 cycles of scalar bookkeeping beside its pieces and reads (SGPR state
    // spilled to VGPR lanes, branches), so it runs at events only: a fold that is due, a fetch that is due.
    // (j == rd.e * HS + rd.h: the half-step this phase handled.)
    int next;
    if (P.refresh_pending >= 0) next = max(j + 1, P.refresh_j + 3);                       // the fetch in flight is folded then
    else if (!P.bound_on || P.no_filter) next = P.J;
    else if (P.rd.e == 0) next = P.HS + P.HS / 4;                                         // entry 1, a quarter of the tile in
    else if (P.rd.e == 1) next = P.refresh_ctr < NSLICEP ? max(max(j + 1, P.refresh_j + 4), P.HS + P.HS / 4) : 2 * P.HS;
    else if (P.rd.e <= P.e_fast) next = max(j + (P.refresh_mask + 1 - (P.rd.h & P.refresh_mask)), P.refresh_j + 4);
    else {                                                                                // first half-step of the next tile that fetches
        int e2 = P.rd.e + 1;
        if (e2 > P.e_mid) e2 = (e2 + P.late_mask) & ~P.late_mask;
        next = e2 * P.HS;
    }
    P.lean_until = P.no_dma ? 0 : min(next, P.J - 3);

    P.advance(P.rd);
    P.advance(P.dm);
#if defined(SQE_PHASE_STAMPS) && SQE_PHASE_STAMPS >= 2
    PP_STAMP2(m4);
    P.mp[0] += m1 - m0; P.mp[1] += m2 - m1; P.mp[2] += m3 - m2; P.mp[3] += m4 - m3; ++P.mp[4];
#endif
}

// MEMORY phase without filter or bound work, j + 3 < J: the steady-state form.
#ifdef SQE_PHASE_STAMPS
__device__ __forceinline__ void mem_lean(PP& P, AOps& a, BOps& b, int j, bool defer, PhaseClock& pc) {
    unsigned long long t0, t1, t2, t3;
    PP_STAMP_MEM(t0);
#else
__device__ __forceinline__ void mem_lean(PP& P, AOps& a, BOps& b, int j, bool defer) {
#endif
    // defer: three pieces here, the fourth (query piece 1) from this wave's next compute phase (cmp_phase_mid).  The
    // wait still retires half-step j + 1 -- all four of its pieces are older than anything issued here -- and
    // leaves the four of j + 2 and the three of j + 3 in flight.
    if (defer) {
        dma_and_reads(P, a, b, j, true, false);
        P.pend_h = P.dm.h;
        P.pend_stage = (j + 3) & 3;
        PP_STAMP_MEM(t1);
        PP_STAMP_MEM(t2);
        PP_WAIT_VM7_LGKM0();
    } else {
        dma_and_reads(P, a, b, j, true, true);
        PP_STAMP_MEM(t1);
        PP_STAMP_MEM(t2);
        PP_WAIT_VM8_LGKM0();
    }
    PP_STAMP_MEM(t3);
#ifdef SQE_PHASE_STAMPS
    pc.mem_issue += t1 - t0; pc.mem_reads += t2 - t1; pc.mem_wait += t3 - t2; ++pc.phases;
#endif
    P.advance(P.rd);
    P.advance(P.dm);
}

#ifdef SQE_PHASE_STAMPS
#define PP_MEM_ANY(j, defer_ok, pc) mem_any(P, f, a, b, j, defer_ok, pc)
#else
#define PP_MEM_ANY(j, defer_ok, pc) mem_any(P, f, a, b, j, defer_ok)
#endif

// memory phase of half-step j: the lean form unless an event is due (mem_phase sets P.lean_until <= J - 3)
#ifdef SQE_PHASE_STAMPS
__device__ __forceinline__ void mem_any(PP& P, const Filter& f, AOps& a, BOps& b, int j, bool defer_ok, PhaseClock& pc) {
    if (j < P.lean_until) mem_lean(P, a, b, j, defer_ok && P.defer_on, pc);
    else mem_phase(P, f, a, b, j);
}
#else
__device__ __forceinline__ void mem_any(PP& P, const Filter& f, AOps& a, BOps& b, int j, bool defer_ok) {
    if (j < P.lean_until) mem_lean(P, a, b, j, defer_ok && P.defer_on);
    else mem_phase(P, f, a, b, j);
}
#endif

// ONE barrier per half-step (scan_i8.hip has the same loop and the invariant's derivation).  Period T_j lies between barriers
// B_{j-1} and B_j:
//     G0, T_j: [tile end] | compute j | pieces j + 3 | vmcnt: own pieces of j + 2 | bound work of j + 1, read operands j + 1 | B_j
//     G1, T_j: [tile end] | bound work of j, pieces j + 3 | read operands j | compute j | vmcnt: own pieces of j + 2 | B_j
// INVARIANT: a piece read in period T was retired by the wave that ISSUED it before a barrier that precedes the read -- every
// wave retires its pieces of half-step x in T_{x-2}, in front of B_{x-2}; x is read at the end of T_{x-1} (G0) and at the head
// of T_x (G1).  (r03 had G0 retire its pieces of j + 1 at the head of T_j, behind B_{j-1}: sibling G0 waves read them later in
// T_j ordered by nothing but time.)  A wave keeps at most two half-steps of pieces in flight; vmcnt(4) leaves the four youngest
// entries of its queue -- the pieces of j + 3 -- and retires everything older (operations retire in issue order): the pieces
// of j + 2, appended keys, and a bound-table fetch (two pieces per wave).  bound_work(x) is what mem_phase() does for the
// bounds, on the same schedule in x, so every wave's private copy of that state moves alike; G0 runs it behind its compute part,
// where no operand registers are live (the fold keeps 64 LDS reads in flight; in front of the compute part hipcc spilled).
//   fetch issued with x = r: G0 at the end of T_{r-1} (behind that period's wait), retired by its vmcnt(4) at the end of T_r (the
//     pieces of r + 3, issued in T_r, are the four younger entries); G1 at the head of T_r, retired at the end of T_r.  All in front of B_r.
//   fold, x = r + 3: a G0 wave at the end of T_{r+2}, a G1 wave at the head of T_{r+3}: behind B_r.
//   next fetch, x >= r + 5: from T_{r+4} at the earliest, behind the barrier that follows the fold.
// Against r02's schedule, a barrier after every phase (-DSQE_PP_TWO_BARRIERS and the STAMPS builds keep it): L2 fills 61.5 ->
// ~30 GB per launch against 20.5 GB algorithmic (pmc_traffic_bf16.json): the query-block workgroups of a chunk stay together.
#if !defined(SQE_PHASE_STAMPS) && !defined(SQE_PP_TWO_BARRIERS)
#define SQE_PP_ONE_BARRIER 1
#define PP_WAIT_VM4() PP_WAIT(0x0F74)
#define PP_WAIT_VM0() PP_WAIT(0x0F70)
// bound work due with half-step x (= P.rd): the fold of a fetched slice, the next fetch
__device__ __forceinline__ void bound_work(PP& P, const Filter& f, int x) {
    if (x < P.lean_until) return;
    const bool more = x + 2 < P.J && !P.no_dma;          // (no fetch over the last half-steps: nothing would retire it in time)
    if (P.refresh_pending >= 0 && x >= P.refresh_j + 3) {
        if (P.wave == (P.refresh_ctr & 7)) refresh_apply(f, P.gstage, P.refresh_pending, P.gshift, P.gshift_k, P.k_rows, fresh_lane());
        P.refresh_pending = -1;
    }
    if (P.bound_on && P.rd.e > 0 && more && !P.no_filter && P.refresh_pending < 0 && x >= P.refresh_j + 5) {
        const bool want = P.rd.e == 1 ? (P.rd.h >= P.HS / 4 && P.refresh_ctr < NSLICEP)
                                      : (P.rd.e <= P.e_fast ? (P.rd.h & P.refresh_mask) == 0
                                                            : (P.rd.h == 0 && (P.rd.e <= P.e_mid || (P.rd.e & P.late_mask) == 0)));
        if (want) {
            P.refresh_pending = P.refresh_ctr % NSLICEP;
            ++P.refresh_ctr;
            P.refresh_j = x;
            refresh_issue<true>(P.gmax_group, f.gstride, P.refresh_pending, P.gstage, P.wave, fresh_lane());
        }
    }
    // first half-step that needs this work again (mem_phase())
    int next;
    if (P.refresh_pending >= 0) next = max(x + 1, P.refresh_j + 3);
    else if (!P.bound_on || P.no_filter) next = P.J;
    else if (P.rd.e == 0) next = P.HS + P.HS / 4;
    else if (P.rd.e == 1) next = P.refresh_ctr < NSLICEP ? max(max(x + 1, P.refresh_j + 5), P.HS + P.HS / 4) : 2 * P.HS;
    else if (P.rd.e <= P.e_fast) next = max(x + (P.refresh_mask + 1 - (P.rd.h & P.refresh_mask)), P.refresh_j + 5);
    else {
        int e2 = P.rd.e + 1;
        if (e2 > P.e_mid) e2 = (e2 + P.late_mask) & ~P.late_mask;
        next = e2 * P.HS;
    }
    P.lean_until = P.no_dma ? 0 : next;
}
// this wave's pieces of half-step y (= P.dm), if there is one
__device__ __forceinline__ void issue_next(PP& P, int y) {
    if (y < P.J && !P.no_dma) {
        P.issue(P.dm, y & 3);
        P.advance(P.dm);
    }
}
// operands of half-step x (= P.rd)
__device__ __forceinline__ void mem_read(PP& P, AOps& a, BOps& b, int x) {
    read_operands(P, a, b, x);
    P.advance(P.rd);
}
#endif

__global__ __launch_bounds__(SCAN_THREADS) void scan_bf16_pp_kernel(ScanKernelArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifdef SQE_DEBUG_KNOBS
    const long long clk0 = clock64(), wall0 = wall_clock64();     // core-clock and constant-rate ticks (SQE_DBG bit 32)
#endif
    PP P;
    P.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = P.wave >> 2;           // waves w and w + 4 share a SIMD
    P.wm = P.wave >> 2;
    P.wn = P.wave & 3;
    P.order = (P.wave >> 1) & 1;
    P.pend_h = -1;
    P.pend_stage = 0;
    P.defer_on = true;
    P.smem = smem;
    P.gstage = smem + OFF_F + FLP::OFF_GSTAGE;

    int logical = blockIdx.x;
    const int G = gridDim.x;
    if ((G & 7) == 0) logical = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int chunk = __builtin_amdgcn_readfirstlane(logical / p.qblocks);
    const int qb = __builtin_amdgcn_readfirstlane(logical % p.qblocks);
    const int q0 = qb * BNP;

    int tile_end;
    chunk_tile_range(p.n_tiles, p.n_chunks, chunk, P.tile_begin, tile_end);
    P.nt = tile_end - P.tile_begin;
    P.HS = p.K / HALF_K;
    // entries: 0 = first tile (BOOT), 1..nt-1 = the other tiles, nt = the first tile again
    const int n_entries = P.nt > 0 ? P.nt + 1 : 0;
    P.J = n_entries * P.HS;
    const size_t ldA = (size_t)p.db_pitch, ldB = (size_t)p.q_pitch;
    P.tile_step = p.tile_step;
    P.tile_bytes = (long long)SCAN_BM * (long long)ldA * p.tile_step;
    {
        // bound-table fetches: every slice per tile for the first e_fast tiles of a chunk, one slice per tile up to e_mid,
        // then one slice every (late_mask + 1)-th tile.  A fetch costs ~2,500 cycles of the workgroup (16 KiB of DMA, two
        // general phases, the fold in one wave) and the bound moves by 1 / t per tile: measured at 10 M rows, 32 / 128 / 2
        // (r02a) against 4 / 32 / 4: batch 1024 17.31 -> 16.97 ms, 512 9.32 -> 9.09, 256 5.32 -> 5.17; 2 / 16 / 8 and
        // 1 / 8 / 16 are no better.  SQE_DBG bit 4096 (knobs build): the r02a schedule.
        const bool old_schedule = (SQE_DBG_BITS(p) & 4096) != 0;
        P.e_fast = old_schedule ? 32 : 4;
        P.e_mid = old_schedule ? 128 : 32;
        P.late_mask = old_schedule ? 1 : 3;
    }
    P.kp = p.kp; P.trig = p.trig; P.gshift = p.gshift; P.gshift_k = p.gshift_k; P.k_rows = p.k_rows;
    {
        int every = 1;
        while (every * 2 * NSLICEP <= P.HS) every *= 2;           // largest power of two <= HS / NSLICEP
        P.refresh_mask = every - 1;
    }
    P.no_mma = (SQE_DBG_BITS(p) & 1) != 0; P.no_dma = (SQE_DBG_BITS(p) & 2) != 0; P.no_filter = (SQE_DBG_BITS(p) & 4) != 0;
    if (SQE_DBG_BITS(p) & 8) P.gshift = P.gshift_k = -1;
    if (SQE_DBG_BITS(p) & 64) P.gshift_k = -1;          // kp-row bound only (the r01 filter)
    if (SQE_DBG_BITS(p) & 2048) P.k_rows = 0;           // k-row bound from the minimum of the group maxima, not their k-th largest
    if (SQE_DBG_BITS(p) & 128) P.order = 0;             // every wave: pieces, then reads (the r01 order)
    if (SQE_DBG_BITS(p) & 256) P.order = 1;             // every wave: reads, then pieces
    if (SQE_DBG_BITS(p) & 512) P.order = P.wave & 1;    // stagger by wave parity instead of pairs
    if (SQE_DBG_BITS(p) & 1024) P.defer_on = false;     // all four DMA pieces from the memory phase
    P.bound_on = P.gshift >= 0 || P.gshift_k >= 0;

    Filter f;
    f.cand_base = p.cand + ((size_t)chunk * p.b_pad + q0) * CAND_CAP;
    f.gstride = p.ngroups * GMAX_COLS * 64;
    bound_rows(p, chunk, q0, P.gmax_group, f.gmax_mine);
    f.thr_key = reinterpret_cast<uint64_t*>(smem + OFF_F + FLP::OFF_THR_KEY);
    f.thr_s = reinterpret_cast<float*>(smem + OFF_F + FLP::OFF_THR_S);
    f.cnt = reinterpret_cast<int*>(smem + OFF_F + FLP::OFF_CNT);
    f.cmax = reinterpret_cast<uint32_t*>(smem + OFF_F + FLP::OFF_CMAX);
    f.flags = reinterpret_cast<int*>(smem + OFF_F + FLP::OFF_FLAGS);
    f.n_rows = p.n_rows;
    f.q_live = min(BNP, p.B - q0);
    f.trig = p.trig;
    f.per_wave = 32;
    f.dbg_no_slow = (SQE_DBG_BITS(p) & 16) != 0;
    f.dbg_counters = (SQE_DBG_BITS(p) & 32) ? p.dbg_counters : nullptr;
    f.collect_keys = nullptr; f.collect_cnt = nullptr;
    float* slack = reinterpret_cast<float*>(smem + OFF_F + FLP::OFF_SLACK);
    f.slack = slack;
    filter_init<BNP>(p, f, slack, q0, p.B, nullptr, tid);

    // ---- per-lane DMA source offsets.  Piece t covers LDS lines 8t .. 8t+7; lane l writes chunk
    // position l & 7 of line 8t + (l >> 3), which holds logical chunk c = pos ^ ((line >> 1) & 7):
    // bytes (c & 3) * 16 of the slice of tile row line + 128 * (c >> 2).
    {
        const int line = P.wave * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((line >> 1) & 7);
        const int row = line + 128 * (c >> 2);
        P.offA0 = (unsigned)(row * ldA) + (c & 3) * 16;
        P.offB0 = (unsigned)(row * ldB) + (c & 3) * 16;
        P.offA1 = P.offA0 + (unsigned)(64 * ldA);      // piece wave + 8: 64 lines further, same swizzle
        P.offB1 = P.offB0 + (unsigned)(64 * ldB);
        P.h_stride = HALF_K * 2;
        // Timing experiment (knobs build, WRONG scores): read the DB tile as if it were stored K-slice-major
        // ([half-step][256 rows][64 B]: every half-step takes 16 KiB of whole 128-B lines, each fetched once per workgroup)
        // -- what a tiled scan copy would cost in time and L2 fills, before anything is rebuilt around it.
        if (SQE_DBG_BITS(p) & 8192) {
            P.offA0 = (unsigned)(row * 64) + (c & 3) * 16;
            P.offA1 = P.offA0 + 64 * 64;
            P.h_stride = SCAN_BM * 64;
        }
        P.tile_skew = (SQE_DBG_BITS(p) & 8192) != 0;
    }
    // ---- per-lane operand read offsets: fragment fm / fn adds fm * 2048 (16 lines)
    {
        const int r = lane & 15, cq = lane >> 4, sw = (r >> 1) & 7;
        P.rdA = (unsigned)(r * LINE_BYTES + (((P.wm * 4 + cq) ^ sw) << 4));
        P.rdB = (unsigned)(((P.wn & 1) * 64 + r) * LINE_BYTES + ((((P.wn >> 1) * 4 + cq) ^ sw) << 4));
    }
    P.qbase = reinterpret_cast<const char*>(p.q) + (size_t)q0 * ldB;
    const char* tile0 = reinterpret_cast<const char*>(p.db) + (size_t)P.tile_begin * p.tile_step * SCAN_BM * ldA;
    P.rd = Cursor{0, 0, tile0};
    P.dm = Cursor{0, 0, tile0};
    P.refresh_pending = -1;
    P.refresh_ctr = 0;
    P.refresh_j = -100;
    P.lean_until = 0;

    f32x4 acc[8][4];
    AOps a;
    BOps b;
    // Timing experiment (knobs build): static priority for the second-dispatched half of the workgroup
    // (MI355X_MICROARCH.md, "Two waves per SIMD", item 4)
    if ((SQE_DBG_BITS(p) & 16384) && group == 1) __builtin_amdgcn_s_setprio(1);
    if ((SQE_DBG_BITS(p) & 32768) && group == 0) __builtin_amdgcn_s_setprio(1);

    // ---- prologue: half-steps 0, 1, 2 (J >= 4 whenever J > 0: boot + rescan entries, K >= 64)
    if (P.J > 0) {
        for (int s = 0; s < 3; ++s) {
            P.issue(P.dm, s);
            P.advance(P.dm);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the prologue's pieces (inline asm: the compiler does not wait for them)
    __syncthreads();                       // prologue landed, state initialised

    {
        // r02's schedule, a barrier after every phase (STAMPS builds, -DSQE_PP_TWO_BARRIERS; the shipped one is above mem_part()):
        //     G0: .. CMP_LAST(e) | MEM(e+1,0) | [SLOW(e) sync] | CMP(e+1,0) | MEM(e+1,1) ..
        //     G1: .. MEM(e,last) | CMP_LAST(e)| [SLOW(e) sync] | MEM(e+1,0) | CMP(e+1,0) ..
        if (P.J > 0) {
            const int HS = P.HS;
            int j = 0;
#ifdef SQE_PHASE_STAMPS
            PhaseClock pclk, scratch_clk;          // scratch_clk: phases outside the steady-state statistics
#endif
            unsigned cols = 0;                 // column groups of the finished tile that hold a survivor
            int thr[4];
            int* any_cols = f.flags + 8;       // some wave of the workgroup marked a survivor in this tile
            auto load_thr = [&]() {            // thresholds for the fast-path test, read one phase early
                const int fl = fresh_lane();
#pragma unroll
                for (int c = 0; c < 4; ++c) thr[c] = pp_thr_bits(f.thr_s[P.wn * 64 + c * 16 + (fl & 15)]);
            };
            auto last_phase = [&]() {
                if (!P.no_mma) cols = cmp_phase_last(acc, a, b, thr);
                if (cols != 0 && fresh_lane() == 0) *any_cols = 1;
            };
            // after the barrier that ends G1's last compute phase of entry e: every wave takes the same path
            auto tile_end = [&](int e) {
                if (P.no_filter) return;
                PP_DECL(e0, e1, e2, e3, e4);
                PP_STAMP(e0);
                const bool any = e == 0 || __builtin_amdgcn_readfirstlane(*any_cols) != 0;
                if (!any) return;
                const int fl = fresh_lane();
                PP_STAMP(e1);
                if (e == 0) filter_boot<8, 4>(acc, f, P.row0_of(e), P.wm * 128, P.wn * 64, fl);
                else if (cols) filter_tile<8, 4>(acc, f, P.row0_of(e), P.wm * 128, P.wn * 64, fl, cols);
                __builtin_amdgcn_sched_barrier(0);
                PP_STAMP(e2);
                PP_BARRIER();
                PP_STAMP(e3);
                if (tid == 0) *any_cols = 0;   // read again a whole tile (2 * HS barriers) later
                entry_sync(P, f, e);
                PP_STAMP(e4);
                PP_ACC(pclk.te[0] += e1 - e0; pclk.te[1] += e2 - e1; pclk.te[2] += e3 - e2; pclk.te[3] += e4 - e3; pclk.te[4] += (cols != 0); ++pclk.te[5]);
            };
#ifdef SQE_PP_ONE_BARRIER
            // at the end of T_jj: this wave's pieces of jj + 2 and everything older; the pieces of jj + 3 stay in flight
            auto wait_pieces = [&](int jj) {
                if (jj + 3 < P.J && !P.no_dma) PP_WAIT_VM4();
                else PP_WAIT_VM0();
            };
            // G0 issues BEHIND its compute part: the two groups' pieces then leave at different times (int8 scan, A/B of the same
            // choice: -7.5 %, profiles/r04_search/ab_schedule_variants.log); one period of latency cover is enough
            auto g0_head = [&](int) {};
            auto g0_tail = [&](int jj) {
                issue_next(P, jj + 3);
                wait_pieces(jj);
                if (jj + 1 < P.J) {
                    bound_work(P, f, jj + 1);
                    mem_read(P, a, b, jj + 1);
                }
            };
            auto g1_head = [&](int jj) {
                bound_work(P, f, jj);
                issue_next(P, jj + 3);
                mem_read(P, a, b, jj);
            };
            if (group == 0) {
                mem_read(P, a, b, 0);                        // (the prologue's pieces: retired by every wave before __syncthreads)
                for (int e = 0; e < n_entries; ++e) {
                    g0_head(j);
                    if (!P.no_mma) cmp_phase<true>(acc, a, b);
                    g0_tail(j);
                    if (HS == 2) load_thr();
                    PP_BARRIER();
                    ++j;
                    for (int h = 1; h < HS - 1; ++h) {
                        g0_head(j);
                        if (!P.no_mma) cmp_phase<false>(acc, a, b);
                        g0_tail(j);
                        if (h == HS - 2) load_thr();
                        PP_BARRIER();
                        ++j;
                    }
                    g0_head(j);
                    last_phase();
                    g0_tail(j);
                    PP_BARRIER();
                    ++j;
                    if (e + 1 < n_entries) tile_end(e);      // (the accumulators are the finished tile's until the next compute part)
                }
            } else {
                for (int e = 0; e < n_entries; ++e) {
                    g1_head(j);
                    if (!P.no_mma) cmp_phase<true>(acc, a, b);
                    wait_pieces(j);
                    PP_BARRIER();
                    ++j;
                    for (int h = 1; h < HS - 1; ++h) {
                        g1_head(j);
                        if (!P.no_mma) cmp_phase<false>(acc, a, b);
                        wait_pieces(j);
                        PP_BARRIER();
                        ++j;
                    }
                    g1_head(j);
                    load_thr();
                    last_phase();
                    wait_pieces(j);
                    PP_BARRIER();
                    ++j;
                    if (e + 1 < n_entries) tile_end(e);
                }
            }
#else
            if (group == 0) {
                mem_phase(P, f, a, b, 0);
                PP_BARRIER();
                for (int e = 0; e < n_entries; ++e) {
                    PP_DECL(b0, b1, b2, b3, b4, b5, b6, b7, b8, b9);
                    PP_STAMP(b0);
                    if (!P.no_mma) cmp_phase<true>(acc, a, b);
                    PP_STAMP(b1);
                    PP_BARRIER();
                    PP_STAMP(b2);
                    PP_MEM_ANY(j + 1, HS > 2, scratch_clk);            // the next compute phase is a middle one unless HS == 2
                    if (HS == 2) load_thr();
                    PP_STAMP(b3);
                    PP_BARRIER();
                    PP_STAMP(b4);
                    PP_ACC(pclk.bnd[0] += b1 - b0; pclk.bnd[1] += b2 - b1; pclk.bnd[2] += b3 - b2; pclk.bnd[3] += b4 - b3; ++pclk.bnd[11]);
                    ++j;
                    for (int h = 1; h < HS - 1; ++h) {
                        PP_DECL(s0, s1, s2, s3, s4);
                        PP_STAMP(s0);
                        if (!P.no_mma) cmp_phase_mid(P, acc, a, b);
                        PP_STAMP(s1);
                        PP_BARRIER();
                        PP_STAMP(s2);
                        const bool lean = j + 1 < P.lean_until;
                        const bool defer = P.defer_on && h < HS - 2;      // the next compute phase is a middle one
#ifdef SQE_PHASE_STAMPS
                        if (lean) mem_lean(P, a, b, j + 1, defer, pclk);
#else
                        if (lean) mem_lean(P, a, b, j + 1, defer);
#endif
                        else mem_phase(P, f, a, b, j + 1);
                        if (h == HS - 2) load_thr();
                        PP_STAMP_MEM(s3);
                        PP_BARRIER();
                        PP_STAMP(s4);
                        PP_ACC(if (lean) { pclk.cmp += s1 - s0; pclk.cmp_bar += s2 - s1; pclk.mem_bar += s4 - s3; }
                               else { pclk.bnd[9] += s4 - s0; ++pclk.bnd[10]; });
                        ++j;
                    }
                    PP_STAMP(b4);
                    last_phase();
                    PP_STAMP(b5);
                    PP_BARRIER();
                    PP_STAMP(b6);
                    if (e + 1 < n_entries) PP_MEM_ANY(j + 1, false, scratch_clk);
                    PP_STAMP(b7);
                    PP_BARRIER();
                    PP_STAMP(b8);
                    ++j;
                    if (e + 1 < n_entries) tile_end(e);
                    // drift of the query-block workgroups that share a chunk: constant-rate clock at two tiles, per workgroup
                    PP_ACC(if (f.dbg_counters && tid == 0 && (e == 100 || e == 400)) f.dbg_counters[512 + blockIdx.x * 2 + (e == 400)] = wall_clock64());
                    PP_STAMP(b9);
                    PP_ACC(pclk.bnd[4] += b5 - b4; pclk.bnd[5] += b6 - b5; pclk.bnd[6] += b7 - b6; pclk.bnd[7] += b8 - b7; pclk.bnd[8] += b9 - b8);
                }
            } else {
                PP_BARRIER();
                for (int e = 0; e < n_entries; ++e) {
                    PP_MEM_ANY(j, false, scratch_clk);
                    PP_BARRIER();
                    if (!P.no_mma) cmp_phase<true>(acc, a, b);
                    PP_BARRIER();
                    ++j;
                    for (int h = 1; h < HS - 1; ++h) {
#ifdef SQE_PHASE_STAMPS
                        unsigned long long s0, s1, s2, s3;
                        const bool lean = j < P.lean_until;
                        if (lean) mem_lean(P, a, b, j, P.defer_on, pclk);
                        else mem_phase(P, f, a, b, j);
                        PP_STAMP_MEM(s0);
                        PP_BARRIER();
                        PP_STAMP(s1);
                        if (!P.no_mma) cmp_phase_mid(P, acc, a, b);
                        PP_STAMP(s2);
                        PP_BARRIER();
                        PP_STAMP(s3);
                        if (lean) { pclk.mem_bar += s1 - s0; pclk.cmp += s2 - s1; pclk.cmp_bar += s3 - s2; }
#else
                        if (j < P.lean_until) mem_lean(P, a, b, j, P.defer_on);
                        else mem_phase(P, f, a, b, j);
                        PP_BARRIER();
                        if (!P.no_mma) cmp_phase_mid(P, acc, a, b);
                        PP_BARRIER();
#endif
                        ++j;
                    }
#ifdef SQE_PHASE_STAMPS
                    { PhaseClock scratch; if (j < P.lean_until) mem_lean(P, a, b, j, false, scratch); else mem_phase(P, f, a, b, j); }
#else
                    if (j < P.lean_until) mem_lean(P, a, b, j, false);
                    else mem_phase(P, f, a, b, j);
#endif
                    load_thr();
                    PP_BARRIER();
                    last_phase();
                    PP_BARRIER();
                    ++j;
                    if (e + 1 < n_entries) tile_end(e);
                    // drift of the query-block workgroups that share a chunk: constant-rate clock at two tiles, per workgroup
                    PP_ACC(if (f.dbg_counters && tid == 0 && (e == 100 || e == 400)) f.dbg_counters[512 + blockIdx.x * 2 + (e == 400)] = wall_clock64());
                }
            }
#endif
#ifdef SQE_PHASE_STAMPS
            if (f.dbg_counters && (blockIdx.x == 0 || blockIdx.x == 100) && lane == 0) {
                unsigned long long* o = f.dbg_counters + 8 + (blockIdx.x ? 64 : 0) + P.wave * 8;
                o[0] = pclk.phases; o[1] = pclk.cmp; o[2] = pclk.cmp_bar; o[3] = pclk.mem_issue;
                o[4] = pclk.mem_reads; o[5] = pclk.mem_wait; o[6] = pclk.mem_bar;
                if (P.wave == 0)
                    for (int i = 0; i < 12; ++i) f.dbg_counters[8 + 128 + (blockIdx.x ? 16 : 0) + i] = pclk.bnd[i];
                for (int i = 0; i < 6; ++i) f.dbg_counters[8 + 160 + (blockIdx.x ? 48 : 0) + P.wave * 6 + i] = pclk.te[i];
#if SQE_PHASE_STAMPS >= 2
                if (blockIdx.x == 0 && (P.wave == 0 || P.wave == 4))
                    for (int i = 0; i < 5; ++i) f.dbg_counters[300 + (P.wave >> 2) * 8 + i] = P.mp[i];
#endif
            }
#endif
        }
    }

    // ---- tail: filter of the last entry (the rescan of the first tile)
    if (P.J > 0 && !P.no_filter) {
        if (filter_tile<8, 4>(acc, f, P.row0_of(0), P.wm * 128, P.wn * 64, lane))
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    compact_owned(f, P.wave * 32, 32, p.kp + 1, p.kp, lane);
    __syncthreads();
    for (int i = tid; i < BNP; i += SCAN_THREADS)
        p.cand_cnt[(size_t)chunk * p.b_pad + q0 + i] = f.cnt[i];
#ifdef SQE_DEBUG_KNOBS
    if (p.dbg_counters && (SQE_DBG_BITS(p) & 32) && blockIdx.x == 0 && tid == 0) {
        p.dbg_counters[4] = (unsigned long long)(clock64() - clk0);
        p.dbg_counters[5] = (unsigned long long)(wall_clock64() - wall0);
    }
#endif
}

}  // namespace

int launch_scan_bf16_pp(const ScanPlan& plan, const ScanArgs& a, hipStream_t stream) {
    if (plan.bn != BNP) return fail(SQE_ERR_INVALID, "scan pp: query block must be 256");
    ScanKernelArgs k = make_kernel_args(plan, a);
    auto kern = scan_bf16_pp_kernel;
    SQE_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(plan.n_chunks * plan.qblocks), dim3(SCAN_THREADS), LDS_BYTES, stream, k);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

}  // namespace sqe
