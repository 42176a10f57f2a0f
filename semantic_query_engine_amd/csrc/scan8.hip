// scan8.hip -- S2, pipelined form of the bf16 scan for query blocks of 256 (gfx950).
//
// Same contract as scan.hip (256-row DB tile x 256 queries per persistent workgroup, fused
// top-k filter, candidate lists) with a software pipeline at two levels:
//
//   LDS ring.  Every 64-wide K step is split into four 16-KiB UNITS
//         u0 = A^0  rows    {wm*128 +      0..63}   (wm = 0,1)
//         u1 = B^1  queries {wn*64  + 32 + 0..31}   (wn = 0..3)
//         u2 = A^1  rows    {wm*128 + 64 + 0..63}
//         u3 = B^0  queries {wn*64  +      0..31}
//     held in 8 unit slots (128 KiB, slot = 4 * (kstep & 1) + u).  Slot p of K step s issues the
//     global_load_lds of unit u_p of K step s+2 into the slot that unit u_p of K step s just
//     vacated, so two K steps of DMA are always in flight.
//
//   Registers.  The fragments a slot's MFMAs consume were read from LDS during the PREVIOUS
//     slot; the reads a slot issues are for the NEXT slot and complete under this slot's 16
//     MFMAs (two B register sets; ONE A set whose fragments are re-read one by one, each
//     right after its last MFMA use, in the two slots where the A set dies).  Per K step s:
//         slot 0: MFMA A^0 x B^0   | reads B^1(s)              | DMA A^0(s+2)
//         slot 1: MFMA A^0 x B^1   | reads A^1(s)              | DMA B^1(s+2)
//         slot 2: MFMA A^1 x B^1   | reads B^0(s) again        | DMA A^1(s+2), counted wait
//         slot 3: MFMA A^1 x B^0   | reads A^0(s+1), B^0(s+1)  | DMA B^0(s+2)
//     Each unit is overwritten in the slot after the one in which it was last read, and every
//     wave retires its reads (lgkmcnt(0)) before the single barrier that closes a slot.
//     ONE counted wait per K step: s_waitcnt vmcnt(6) at the end of slot 2 leaves the three
//     units issued in slots 0-2 in flight and retires all of K step s+1, which slot 3 starts
//     to read after the barrier.  The two B register sets swap roles every K step, so the
//     loop body is two K steps (8 slots).
//
// Filter placement: the filter of tile entry e runs at the top of entry e+1's first slot; what
// needs every wave's contribution (publishing the boot maxima, list compaction) runs one slot
// later (each storing wave drains its stores before the barrier in between).  Global-bound
// rows are fetched by LDS-DMA in slot 0 (older than the six youngest DMA pieces at the slot-2
// wait, hence retired by it) and folded at the next K step.
#include <stdlib.h>

#include "scan_common.h"

namespace sqe {

namespace {

constexpr int UNIT_BYTES = 128 * SCAN_ROW_BYTES;     // 16 KiB
constexpr int NSLOTS = 8;
constexpr int BN8 = 256;
constexpr int OFF_F = NSLOTS * UNIT_BYTES;
using FL8 = FilterLds<BN8>;
constexpr int LDS_BYTES = OFF_F + FL8::BYTES;
constexpr int NSLICE8 = BN8 / GSLICE_Q;

#define SQE_BARRIER()                          \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)

__device__ __forceinline__ void glds16(const char* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// one unit = 16 wave-instructions of 1 KiB; wave w issues pieces w and w + 8
__device__ __forceinline__ void issue_unit(const char* gbase, unsigned off0, unsigned off1, char* slot, int wave) {
    glds16(gbase + off0, slot + wave * 1024);
    glds16(gbase + off1, slot + (wave + 8) * 1024);
}

__device__ __forceinline__ bf16x8 unit_frag(const char* slot, int ru, int c) {
    return *reinterpret_cast<const bf16x8*>(slot + ru * SCAN_ROW_BYTES + ((c ^ ((ru >> 1) & 7)) << 4));
}

typedef bf16x8 AFrag[4][2];   // [fm][kk]: 64 rows x 64 k
typedef bf16x8 BFrag[2][2];   // [fn][kk]: 32 queries x 64 k

__device__ __forceinline__ void read_a(AFrag& a, const char* unit, int ruA, int cq) {
#pragma unroll
    for (int fm = 0; fm < 4; ++fm)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) a[fm][kk] = unit_frag(unit, ruA + fm * 16, kk * 4 + cq);
}
__device__ __forceinline__ void read_b(BFrag& b, const char* unit, int ruB, int cq) {
#pragma unroll
    for (int fn = 0; fn < 2; ++fn)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) b[fn][kk] = unit_frag(unit, ruB + fn * 16, kk * 4 + cq);
}

template <int I0, int J0>
__device__ __forceinline__ void mfma_quad(f32x4 (&acc)[8][4], const AFrag& a, const BFrag& b) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int fm = 0; fm < 4; ++fm)
#pragma unroll
            for (int fn = 0; fn < 2; ++fn)
                acc[I0 + fm][J0 + fn] =
                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[fm][kk], b[fn][kk], acc[I0 + fm][J0 + fn], 0, 0, 0);
}

// Same quadrant, but the A set dies here: each A fragment is re-read from `next_unit` right
// after its last MFMA use, so the next slot's A fragments arrive under this slot's MFMAs
// without a second A register set.
template <int I0, int J0>
__device__ __forceinline__ void mfma_quad_refill(f32x4 (&acc)[8][4], AFrag& a, const BFrag& b,
                                                 const char* next_unit, bool refill, int ruA, int cq) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int fm = 0; fm < 4; ++fm) {
#pragma unroll
            for (int fn = 0; fn < 2; ++fn)
                acc[I0 + fm][J0 + fn] =
                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[fm][kk], b[fn][kk], acc[I0 + fm][J0 + fn], 0, 0, 0);
            if (refill) a[fm][kk] = unit_frag(next_unit, ruA + fm * 16, kk * 4 + cq);
        }
}

// ---- two-slot form: 32 MFMAs per slot (one A set against both B sets)
// Slot A: the A set (A^0) dies and is refilled with A^1 of the same K step.
template <int I0>
__device__ __forceinline__ void mfma_half_refill_a(f32x4 (&acc)[8][4], AFrag& a, const BFrag& b0, const BFrag& b1,
                                                   const char* next_a, bool refill, int ruA, int cq) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int fm = 0; fm < 4; ++fm) {
#pragma unroll
            for (int fn = 0; fn < 2; ++fn) {
                acc[I0 + fm][fn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[fm][kk], b0[fn][kk], acc[I0 + fm][fn], 0, 0, 0);
                acc[I0 + fm][2 + fn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[fm][kk], b1[fn][kk], acc[I0 + fm][2 + fn], 0, 0, 0);
            }
            if (refill) a[fm][kk] = unit_frag(next_a, ruA + fm * 16, kk * 4 + cq);
        }
}
// Slot B: everything dies -- A is refilled with A^0 of the next K step fragment by fragment, the two B
// sets with B^0 / B^1 of the next K step after the half (kk) that last used them.
template <int I0>
__device__ __forceinline__ void mfma_half_refill_all(f32x4 (&acc)[8][4], AFrag& a, BFrag& b0, BFrag& b1,
                                                     const char* next_a, const char* next_b0, const char* next_b1,
                                                     bool refill, int ruA, int ruB, int cq) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int fm = 0; fm < 4; ++fm) {
#pragma unroll
            for (int fn = 0; fn < 2; ++fn) {
                acc[I0 + fm][fn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[fm][kk], b0[fn][kk], acc[I0 + fm][fn], 0, 0, 0);
                acc[I0 + fm][2 + fn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[fm][kk], b1[fn][kk], acc[I0 + fm][2 + fn], 0, 0, 0);
            }
            if (refill) a[fm][kk] = unit_frag(next_a, ruA + fm * 16, kk * 4 + cq);
        }
        if (refill) {
#pragma unroll
            for (int fn = 0; fn < 2; ++fn) {
                b0[fn][kk] = unit_frag(next_b0, ruB + fn * 16, kk * 4 + cq);
                b1[fn][kk] = unit_frag(next_b1, ruB + fn * 16, kk * 4 + cq);
            }
        }
    }
}

struct Pipe {
    // constants
    const char* dbbase;
    const char* qbase;
    char* smem;
    char* gstage;
    const uint32_t* gmax_group;
    size_t ldA, ldB;               // row pitches (bytes) of the scanned DB copy and of the query block
    unsigned offA[2], offB[2];     // per-lane byte offsets (zero-extended: SGPR-base addressing form)
    unsigned a1_off, b1_off;        // uniform
    int ruA, ruB, cq;
    int wave, lane, wm, wn;
    int tile_begin, nt, KS, S;
    int kp, trig, gshift, refresh_every;
    int rot;                        // K-loop rotation of this query block (see kernel comment)
    bool no_mma, no_dma, no_filter; // timing experiments (SQE_DBG)
    // (entry, ks) of K steps s, s+1, s+2
    int e0, ks0, e1, ks1, e2, ks2;
    int refresh_pending, refresh_ctr;
    // incremental source pointers of K step s+2
    const char* tile2; const char* a2; const char* b2; long long tile_bytes; int ksl2;

    __device__ __forceinline__ int tile_of(int e) const { return e < nt ? tile_begin + e : tile_begin; }
    // K step ks of this workgroup covers k slice (ks + rot) mod KS: workgroups walk K in rotated
    // order (the dot product does not depend on the order).
    __device__ __forceinline__ int kslice(int ks) const { const int k = ks + rot; return k >= KS ? k - KS : k; }
    __device__ __forceinline__ const char* a_src(int e, int ks) const {
        return dbbase + (size_t)tile_of(e) * SCAN_BM * ldA + (size_t)kslice(ks) * SCAN_ROW_BYTES;
    }
    __device__ __forceinline__ const char* b_src(int ks) const { return qbase + (size_t)kslice(ks) * SCAN_ROW_BYTES; }
    __device__ __forceinline__ char* slot_of(int s, int u) const { return smem + (((s & 1) << 2) + u) * UNIT_BYTES; }
    // a2 / b2: source of K step s+2, maintained incrementally (a handful of scalar adds per K
    // step instead of 64-bit multiplies per DMA issue)
    __device__ __forceinline__ void advance() {
        e0 = e1; ks0 = ks1;
        e1 = e2; ks1 = ks2;
        if (++ks2 == KS) {
            ks2 = 0;
            ++e2;
            tile2 += (e2 == nt) ? -(long long)(nt - 1) * tile_bytes : tile_bytes;   // last entry = first tile again
        }
        if (++ksl2 == KS) ksl2 = 0;
        a2 = tile2 + (size_t)ksl2 * SCAN_ROW_BYTES;
        b2 = qbase + (size_t)ksl2 * SCAN_ROW_BYTES;
    }
};

// One K step.  On entry a = A^0(s) and bX = B^0(s) are in registers; on exit a = A^0(s+1)
// and bY = B^0(s+1) are (the caller swaps bX / bY for the next K step).
__device__ __forceinline__ void kstep(Pipe& P, const Filter& f, f32x4 (&acc)[8][4], AFrag& a,
                                      BFrag& bX, BFrag& bY, int s) {
    char* u0 = P.slot_of(s, 0);
    char* u1 = P.slot_of(s, 1);
    char* u2 = P.slot_of(s, 2);
    char* u3 = P.slot_of(s, 3);
    const bool more2 = s + 2 < P.S && !P.no_dma;
    const bool entry_start = P.ks0 == 0 && s > 0 && !P.no_filter;

    // ================= slot 0: A^0 x B^0
    if (!P.no_mma) read_b(bY, u1, P.ruB, P.cq);
    if (P.refresh_pending >= 0) {            // fetched during an earlier K step, retired by its slot-2 wait
        if (P.wave == (P.refresh_ctr & 7)) refresh_apply(f, P.gstage, P.refresh_pending, P.gshift, fresh_lane());
        P.refresh_pending = -1;
    }
    // Bound refresh schedule.  Entry 1 (the first filtered tile): all slices back to back, but
    // only from K step KS/4 on, when every chunk has published its boot maxima (fetching earlier
    // would read an empty table and leave those queries without a threshold for a whole tile).
    // Then every slice once per tile while the bound still moves fast, one slice per tile later.
    const bool want_refresh = P.e0 == 1 ? (P.ks0 >= P.KS / 4 && P.refresh_ctr < NSLICE8)
                                        : (P.e0 <= 32 ? (P.ks0 % P.refresh_every) == 0 : P.ks0 == 0);
    if (P.gshift >= 0 && P.e0 > 0 && want_refresh && !P.no_filter) {
        P.refresh_pending = P.refresh_ctr % NSLICE8;
        ++P.refresh_ctr;
        refresh_issue(P.gmax_group, f.gstride, P.refresh_pending, P.gstage, P.wave, fresh_lane());
    }
    if (more2) issue_unit(P.a2, P.offA[0], P.offA[1], u0, P.wave);
    if (P.ks0 == 0) {
        if (s > 0 && !P.no_filter) {
            // filter of the entry finished by the previous K step
            const int64_t row0 = (int64_t)P.tile_of(P.e0 - 1) * SCAN_BM;
            const int fl = fresh_lane();
            if (P.e0 - 1 == 0) {
                filter_boot<8, 4>(acc, f, row0, P.wm * 128, P.wn * 64, fl);
            } else {
                // appended keys are NOT drained here: a vmcnt(0) would wait for every DMA piece in
                // flight.  They only have to be visible if a list is compacted, which slot 1 checks.
                filter_tile<8, 4>(acc, f, row0, P.wm * 128, P.wn * 64, fl);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!P.no_mma) mfma_quad<0, 0>(acc, a, bX);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SQE_BARRIER();

    // ================= slot 1: A^0 x B^1
    if (entry_start) {
        // every wave's filter stores for the previous entry are published (drained before the barrier)
        if (P.e0 - 1 == 0) {
            publish_cmax(f, P.wave * 32, 32, fresh_lane());
        } else {
            // flags[w] != 0: a list owned by wave w reached the compaction trigger.  Every wave reads
            // the same eight words after the barrier, so the branch (and its barrier) is uniform.
            const int fl = fresh_lane();
            const int any_flag = __builtin_amdgcn_readfirstlane(__any(f.flags[fl & 7] != 0));
            if (any_flag) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // my appended keys are in memory
                SQE_BARRIER();
                if (__builtin_amdgcn_readfirstlane(f.flags[P.wave]) != 0)
                    compact_owned(f, P.wave * 32, 32, P.trig, P.kp, fl);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                SQE_BARRIER();                                                 // sweeps done before flags are cleared
                if (fl == 0) f.flags[P.wave] = 0;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (more2) issue_unit(P.b2 + P.b1_off, P.offB[0], P.offB[1], u1, P.wave);
    __builtin_amdgcn_sched_barrier(0);
    if (!P.no_mma) mfma_quad_refill<0, 2>(acc, a, bY, u2, true, P.ruA, P.cq);          // a <- A^1(s)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SQE_BARRIER();

    // ================= slot 2: A^1 x B^1; K step s+1 retired
    if (!P.no_mma) read_b(bX, u3, P.ruB, P.cq);
    if (more2) issue_unit(P.a2 + P.a1_off, P.offA[0], P.offA[1], u2, P.wave);
    __builtin_amdgcn_sched_barrier(0);
    if (!P.no_mma) mfma_quad<4, 2>(acc, a, bY);
    if (more2) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    SQE_BARRIER();

    // ================= slot 3: A^1 x B^0; fragments of K step s+1
    const bool more1 = s + 1 < P.S;
    if (more1 && !P.no_mma) read_b(bY, P.slot_of(s + 1, 3), P.ruB, P.cq);
    if (more2) issue_unit(P.b2, P.offB[0], P.offB[1], u3, P.wave);
    __builtin_amdgcn_sched_barrier(0);
    if (!P.no_mma) mfma_quad_refill<4, 0>(acc, a, bX, P.slot_of(s + 1, 0), more1, P.ruA, P.cq);   // a <- A^0(s+1)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SQE_BARRIER();

    P.advance();
}

// Two-slot K step (2 barriers instead of 4).  On entry a = A^0(s), bX = B^0(s), bY = B^1(s); on exit
// the same for K step s+1.  Slot A: 32 MFMAs A^0 x {B^0, B^1}, a <- A^1(s); DMA of A^0, B^1, B^0 of
// K step s+2 (their slots were last read in slot B of K step s-1); counted wait (the three units
// just issued stay in flight, K step s+1 is retired).  Slot B: 32 MFMAs A^1 x {B^0, B^1} while
// every fragment is replaced by K step s+1's; DMA of A^1(s+2).
__device__ __forceinline__ void kstep2(Pipe& P, const Filter& f, f32x4 (&acc)[8][4], AFrag& a, BFrag& bX, BFrag& bY, int s) {
    char* u0 = P.slot_of(s, 0);
    char* u1 = P.slot_of(s, 1);
    char* u2 = P.slot_of(s, 2);
    char* u3 = P.slot_of(s, 3);
    const bool more2 = s + 2 < P.S && !P.no_dma;
    const bool more1 = s + 1 < P.S;
    const bool entry_start = P.ks0 == 0 && s > 0 && !P.no_filter;

    // ================= slot A
    if (P.refresh_pending >= 0) {
        if (P.wave == (P.refresh_ctr & 7)) refresh_apply(f, P.gstage, P.refresh_pending, P.gshift, fresh_lane());
        P.refresh_pending = -1;
    }
    const bool want_refresh = P.e0 == 1 ? (P.ks0 >= P.KS / 4 && P.refresh_ctr < NSLICE8)
                                        : (P.e0 <= 32 ? (P.ks0 % P.refresh_every) == 0 : P.ks0 == 0);
    if (P.gshift >= 0 && P.e0 > 0 && want_refresh && !P.no_filter) {
        P.refresh_pending = P.refresh_ctr % NSLICE8;
        ++P.refresh_ctr;
        refresh_issue(P.gmax_group, f.gstride, P.refresh_pending, P.gstage, P.wave, fresh_lane());
    }
    if (more2) {
        issue_unit(P.a2, P.offA[0], P.offA[1], u0, P.wave);
        issue_unit(P.b2 + P.b1_off, P.offB[0], P.offB[1], u1, P.wave);
        issue_unit(P.b2, P.offB[0], P.offB[1], u3, P.wave);
    }
    if (P.ks0 == 0) {
        if (s > 0 && !P.no_filter) {
            const int64_t row0 = (int64_t)P.tile_of(P.e0 - 1) * SCAN_BM;
            const int fl = fresh_lane();
            if (P.e0 - 1 == 0) filter_boot<8, 4>(acc, f, row0, P.wm * 128, P.wn * 64, fl);
            else filter_tile<8, 4>(acc, f, row0, P.wm * 128, P.wn * 64, fl);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!P.no_mma) mfma_half_refill_a<0>(acc, a, bX, bY, u2, true, P.ruA, P.cq);        // a <- A^1(s)
    if (more2) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    SQE_BARRIER();

    // ================= slot B
    if (entry_start) {
        if (P.e0 - 1 == 0) {
            publish_cmax(f, P.wave * 32, 32, fresh_lane());
        } else {
            const int fl = fresh_lane();
            const int any_flag = __builtin_amdgcn_readfirstlane(__any(f.flags[fl & 7] != 0));
            if (any_flag) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                SQE_BARRIER();
                if (__builtin_amdgcn_readfirstlane(f.flags[P.wave]) != 0)
                    compact_owned(f, P.wave * 32, 32, P.trig, P.kp, fl);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                SQE_BARRIER();
                if (fl == 0) f.flags[P.wave] = 0;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (more2) issue_unit(P.a2 + P.a1_off, P.offA[0], P.offA[1], u2, P.wave);
    __builtin_amdgcn_sched_barrier(0);
    if (!P.no_mma)
        mfma_half_refill_all<4>(acc, a, bX, bY, P.slot_of(s + 1, 0), P.slot_of(s + 1, 3), P.slot_of(s + 1, 1), more1,
                                P.ruA, P.ruB, P.cq);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SQE_BARRIER();

    P.advance();
}

template <int SLOTS>
__global__ __launch_bounds__(SCAN_THREADS) void scan_bf16_p8_kernel(ScanKernelArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    Pipe P;
    P.lane = tid & 63;
    P.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    P.wm = P.wave >> 2;
    P.wn = P.wave & 3;
    P.smem = smem;
    P.gstage = smem + OFF_F + FL8::OFF_GSTAGE;

    int logical = blockIdx.x;
    const int G = gridDim.x;
    if ((G & 7) == 0) logical = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    // integer division runs on the VALU: pull the (uniform) results back into SGPRs so that
    // everything derived from them (tile range, base pointers) stays scalar
    const int chunk = __builtin_amdgcn_readfirstlane(logical / p.qblocks);
    const int qb = __builtin_amdgcn_readfirstlane(logical % p.qblocks);
    const int q0 = qb * BN8;

    int tile_end;
    chunk_tile_range(p.n_tiles, p.n_chunks, chunk, P.tile_begin, tile_end);
    P.nt = tile_end - P.tile_begin;
    P.KS = p.K / SCAN_BK;
    // entries: 0 = first tile (BOOT), 1..nt-1 = the other tiles, nt = the first tile again
    const int n_entries = P.nt > 0 ? P.nt + 1 : 0;
    P.S = n_entries * P.KS;
    P.ldA = (size_t)p.db_pitch;
    P.ldB = (size_t)p.q_pitch;
    P.kp = p.kp; P.trig = p.trig; P.gshift = p.gshift;
    P.refresh_every = P.KS >= NSLICE8 ? P.KS / NSLICE8 : 1;
    P.rot = (logical * p.krot) % P.KS;
    P.no_mma = (p.dbg & 1) != 0; P.no_dma = (p.dbg & 2) != 0; P.no_filter = (p.dbg & 4) != 0;
    if (p.dbg & 8) P.gshift = -1;

    Filter f;
    f.cand_base = p.cand + ((size_t)chunk * p.b_pad + q0) * CAND_CAP;
    f.gstride = p.ngroups * GMAX_COLS * 64;
    bound_rows(p, chunk, q0, P.gmax_group, f.gmax_mine);
    f.thr_key = reinterpret_cast<uint64_t*>(smem + OFF_F + FL8::OFF_THR_KEY);
    f.thr_s = reinterpret_cast<float*>(smem + OFF_F + FL8::OFF_THR_S);
    f.cnt = reinterpret_cast<int*>(smem + OFF_F + FL8::OFF_CNT);
    f.cmax = reinterpret_cast<uint32_t*>(smem + OFF_F + FL8::OFF_CMAX);
    f.flags = reinterpret_cast<int*>(smem + OFF_F + FL8::OFF_FLAGS);
    f.n_rows = p.n_rows;
    f.q_live = min(BN8, p.B - q0);
    f.trig = p.trig;
    f.per_wave = 32;
    f.dbg_no_slow = (p.dbg & 16) != 0;
    f.dbg_counters = (p.dbg & 32) ? p.dbg_counters : nullptr;
    f.collect_keys = nullptr; f.collect_cnt = nullptr;
    for (int i = tid; i < BN8; i += SCAN_THREADS) {
        const bool live = (q0 + i) < p.B;
        f.thr_key[i] = live ? 0ull : ~0ull;
        f.thr_s[i] = live ? -INFINITY : INFINITY;
        f.cnt[i] = 0;
        f.cmax[i] = 0u;
    }
    if (tid < 16) f.flags[tid] = 0;

    // ---- per-lane source offsets of this wave's two DMA pieces per unit (h = 0 form)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ru = (P.wave + 8 * t) * 8 + (P.lane >> 3);      // unit row 0..127
        const int c = (P.lane & 7) ^ ((ru >> 1) & 7);             // source chunk (swizzle on the source)
        P.offA[t] = (unsigned)(((ru >> 6) * 128 + (ru & 63)) * P.ldA) + c * 16;
        P.offB[t] = (unsigned)(((ru >> 5) * 64 + (ru & 31)) * P.ldB) + c * 16;
    }
    P.a1_off = (unsigned)(64 * P.ldA);     // A^1 = A^0 + 64 rows
    P.b1_off = (unsigned)(32 * P.ldB);     // B^1 = B^0 + 32 queries
    P.qbase = reinterpret_cast<const char*>(p.q) + (size_t)q0 * P.ldB;
    P.dbbase = reinterpret_cast<const char*>(p.db);
    P.ruA = P.wm * 64 + (P.lane & 15);
    P.ruB = P.wn * 32 + (P.lane & 15);
    P.cq = P.lane >> 4;

    P.e0 = 0; P.ks0 = 0;
    P.e1 = 0; P.ks1 = 1;
    if (P.ks1 == P.KS) { P.ks1 = 0; ++P.e1; }
    P.e2 = P.e1; P.ks2 = P.ks1 + 1;
    if (P.ks2 == P.KS) { P.ks2 = 0; ++P.e2; }
    P.refresh_pending = -1;
    P.refresh_ctr = 0;
    P.tile_bytes = (long long)SCAN_BM * (long long)P.ldA;
    P.tile2 = P.dbbase + (size_t)P.tile_of(P.e2) * SCAN_BM * P.ldA;
    P.ksl2 = P.kslice(P.ks2);
    P.a2 = P.tile2 + (size_t)P.ksl2 * SCAN_ROW_BYTES;
    P.b2 = P.qbase + (size_t)P.ksl2 * SCAN_ROW_BYTES;

    f32x4 acc[8][4];
    AFrag a;
    BFrag bX, bY;

    // ---- prologue: all units of K steps 0 and 1 (S >= 2 whenever S > 0: boot + rescan entries)
    if (P.S > 0) {
        const char* a0 = P.a_src(0, 0);
        const char* b0 = P.b_src(0);
        issue_unit(a0, P.offA[0], P.offA[1], P.slot_of(0, 0), P.wave);
        issue_unit(b0 + P.b1_off, P.offB[0], P.offB[1], P.slot_of(0, 1), P.wave);
        issue_unit(a0 + P.a1_off, P.offA[0], P.offA[1], P.slot_of(0, 2), P.wave);
        issue_unit(b0, P.offB[0], P.offB[1], P.slot_of(0, 3), P.wave);
        const char* a1 = P.a_src(P.e1, P.ks1);
        const char* b1 = P.b_src(P.ks1);
        issue_unit(a1, P.offA[0], P.offA[1], P.slot_of(1, 0), P.wave);
        issue_unit(b1 + P.b1_off, P.offB[0], P.offB[1], P.slot_of(1, 1), P.wave);
        issue_unit(a1 + P.a1_off, P.offA[0], P.offA[1], P.slot_of(1, 2), P.wave);
        issue_unit(b1, P.offB[0], P.offB[1], P.slot_of(1, 3), P.wave);
    }
    __syncthreads();                       // vmcnt(0) + barrier: prologue landed, state initialised
    if (P.S > 0) {
        read_a(a, P.slot_of(0, 0), P.ruA, P.cq);
        read_b(bX, P.slot_of(0, 3), P.ruB, P.cq);
        if constexpr (SLOTS == 2) read_b(bY, P.slot_of(0, 1), P.ruB, P.cq);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    SQE_BARRIER();                         // every wave holds its first fragments before slot 0 reuses u0

    if constexpr (SLOTS == 2) {
        for (int s = 0; s < P.S; ++s) kstep2(P, f, acc, a, bX, bY, s);
    } else {
        for (int s = 0; s < P.S; s += 2) {
            kstep(P, f, acc, a, bX, bY, s);
            if (s + 1 < P.S) kstep(P, f, acc, a, bY, bX, s + 1);
        }
    }

    // ---- tail: filter of the last entry (the rescan of the first tile)
    if (P.S > 0 && !P.no_filter) {
        if (filter_tile<8, 4>(acc, f, (int64_t)P.tile_begin * SCAN_BM, P.wm * 128, P.wn * 64, P.lane))
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    compact_owned(f, P.wave * 32, 32, p.kp + 1, p.kp, P.lane);
    __syncthreads();
    for (int i = tid; i < BN8; i += SCAN_THREADS)
        p.cand_cnt[(size_t)chunk * p.b_pad + q0 + i] = f.cnt[i];
}

}  // namespace

int launch_scan_bf16_p8(const ScanPlan& plan, const ScanArgs& a, hipStream_t stream) {
    if (plan.bn != BN8) return fail(SQE_ERR_INVALID, "scan p8: query block must be 256");
    ScanKernelArgs k = make_kernel_args(plan, a);
    static const int slots = [] { const char* e = getenv("SQE_P8_SLOTS"); return e && e[0] == '4' ? 4 : 2; }();
    auto kern = slots == 2 ? scan_bf16_p8_kernel<2> : scan_bf16_p8_kernel<4>;
    SQE_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(plan.n_chunks * plan.qblocks), dim3(SCAN_THREADS), LDS_BYTES, stream, k);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

}  // namespace sqe
