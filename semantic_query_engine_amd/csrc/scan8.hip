// scan8.hip -- S2, pipelined form of the bf16 scan for query blocks of 256 (gfx950).
//
// Same contract as scan.hip (256-row DB tile x 256 queries per persistent workgroup, fused
// top-k filter, candidate lists) with a deeper software pipeline:
//
//   * every 64-wide K step is split into four 16-KiB UNITS
//         u0 = A^0  rows    {wm*128 +      0..63}   (wm = 0,1)      read in phase 0
//         u1 = B^1  queries {wn*64  + 32 + 0..31}   (wn = 0..3)     read in phase 1
//         u2 = A^1  rows    {wm*128 + 64 + 0..63}                   read in phase 2
//         u3 = B^0  queries {wn*64  +      0..31}                   read in phases 0 and 3
//     so each unit has ONE last reader phase; 8 unit slots of LDS (128 KiB) form a ring
//     (slot = 4 * (kstep & 1) + u);
//   * phase p of K step s computes one 64x32 quadrant of every wave's 128x64 output
//     (16 x v_mfma_f32_16x16x32_bf16) and issues the global_load_lds of ONE unit, 7 units
//     ahead of the unit it frees:  p0 -> (s+1,u3), p1 -> (s+2,u0), p2 -> (s+2,u1),
//     p3 -> (s+2,u2).  A unit is overwritten one phase after its last reader phase, whose
//     reads were retired (lgkmcnt(0)) before that phase's first barrier;
//   * ONE counted wait per K step: s_waitcnt vmcnt(6) in phase 3 leaves the three youngest
//     units in flight and retires everything K step s+1 needs; it sits before the phase's
//     first barrier, so after the second barrier every wave's portion has landed;
//   * the two wave rows (wm = 0 / 1, one wave of each per SIMD) run staggered by one
//     barrier: while one group issues LDS reads + DMA and waits, the other group's MFMA
//     cluster owns the matrix pipe.
//
// Filter placement: the filter of tile entry e runs at the top of entry e+1's first phase (its
// VALU work overlaps the other group's MFMAs); what needs every wave's contribution (publishing
// the boot maxima, list compaction) runs two phases later, when both groups' filter stores are
// published (each storing wave drains its stores before its next barrier).  The global-bound
// rows are fetched by LDS-DMA in phase 0 of an entry's first K step (older than the six
// youngest DMA pieces at the phase-3 wait, hence retired by it) and folded one K step later.
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
__device__ __forceinline__ void issue_unit(const char* gbase, int off0, int off1, char* slot, int wave) {
    glds16(gbase + off0, slot + wave * 1024);
    glds16(gbase + off1, slot + (wave + 8) * 1024);
}

__device__ __forceinline__ bf16x8 unit_frag(const char* slot, int ru, int c) {
    return *reinterpret_cast<const bf16x8*>(slot + ru * SCAN_ROW_BYTES + ((c ^ ((ru >> 1) & 7)) << 4));
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_bf16_p8_kernel(ScanKernelArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* gstage = smem + OFF_F + FL8::OFF_GSTAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2;      // wave row (group): waves 0-3 / 4-7, one of each per SIMD
    const int wn = wave & 3;

    int logical = blockIdx.x;
    const int G = gridDim.x;
    if ((G & 7) == 0) logical = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int chunk = logical / p.qblocks;
    const int qb = logical % p.qblocks;
    const int q0 = qb * BN8;

    const int tile_begin = chunk * p.tiles_per_chunk;
    const int tile_end = min(p.n_tiles, tile_begin + p.tiles_per_chunk);
    const int nt = tile_end - tile_begin;
    const int KS = p.K / SCAN_BK;
    // entries: 0 = first tile (BOOT), 1..nt-1 = the other tiles, nt = the first tile again
    const int n_entries = nt > 0 ? nt + 1 : 0;
    const int S = n_entries * KS;                    // K steps of this workgroup
    const size_t ld = (size_t)p.K * 2;
    auto tile_of = [&](int e) { return e < nt ? tile_begin + e : tile_begin; };

    Filter f;
    f.cand_base = p.cand + ((size_t)chunk * p.b_pad + q0) * CAND_CAP;
    f.gstride = p.ngroups * GMAX_COLS;
    const uint32_t* gmax_group = p.gmax + ((size_t)q0 * p.ngroups + (chunk % p.ngroups)) * GMAX_COLS;
    f.gmax_mine = const_cast<uint32_t*>(gmax_group) + chunk / p.ngroups;
    f.thr_key = reinterpret_cast<uint64_t*>(smem + OFF_F + FL8::OFF_THR_KEY);
    f.thr_s = reinterpret_cast<float*>(smem + OFF_F + FL8::OFF_THR_S);
    f.cnt = reinterpret_cast<int*>(smem + OFF_F + FL8::OFF_CNT);
    f.cmax = reinterpret_cast<uint32_t*>(smem + OFF_F + FL8::OFF_CMAX);
    f.flags = reinterpret_cast<int*>(smem + OFF_F + FL8::OFF_FLAGS);
    f.n_rows = p.n_rows;
    f.q_live = min(BN8, p.B - q0);
    f.trig = p.trig;
    f.per_wave = 32;
    for (int i = tid; i < BN8; i += SCAN_THREADS) {
        const bool live = (q0 + i) < p.B;
        f.thr_key[i] = live ? 0ull : ~0ull;
        f.thr_s[i] = live ? -INFINITY : INFINITY;
        f.cnt[i] = 0;
        f.cmax[i] = 0u;
    }
    if (tid < 16) f.flags[tid] = 0;

    // ---- per-lane source offsets of this wave's two DMA pieces per unit (h = 0 form)
    int offA[2], offB[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ru = (wave + 8 * t) * 8 + (lane >> 3);        // unit row 0..127
        const int c = (lane & 7) ^ ((ru >> 1) & 7);             // source chunk (swizzle on the source)
        offA[t] = (int)(((ru >> 6) * 128 + (ru & 63)) * ld) + c * 16;
        offB[t] = (int)(((ru >> 5) * 64 + (ru & 31)) * ld) + c * 16;
    }
    const int a1_off = (int)(64 * ld);     // A^1 = A^0 + 64 rows
    const int b1_off = (int)(32 * ld);     // B^1 = B^0 + 32 queries

    const char* qbase = reinterpret_cast<const char*>(p.q) + (size_t)q0 * ld;
    const char* dbbase = reinterpret_cast<const char*>(p.db);
    auto a_src = [&](int e, int ks) { return dbbase + (size_t)tile_of(e) * SCAN_BM * ld + (size_t)ks * SCAN_ROW_BYTES; };
    auto b_src = [&](int ks) { return qbase + (size_t)ks * SCAN_ROW_BYTES; };
    auto slot_of = [&](int s, int u) { return smem + (((s & 1) << 2) + u) * UNIT_BYTES; };

    // ---- fragment read addressing (unit rows of this wave)
    const int ruA = wm * 64 + (lane & 15);     // + fm * 16
    const int ruB = wn * 32 + (lane & 15);     // + fn * 16
    const int cq = lane >> 4;                  // + kk * 4

    f32x4 acc[8][4];
    bf16x8 a[4][2], b[2][2];

    // (entry, ks) of K steps s, s+1, s+2
    int e0 = 0, ks0 = 0;
    int e1 = 0, ks1 = 1;
    if (ks1 == KS) { ks1 = 0; ++e1; }
    int e2 = e1, ks2 = ks1 + 1;
    if (ks2 == KS) { ks2 = 0; ++e2; }

    // ---- prologue: K step 0 (all four units) and K step 1 (u0..u2)
    if (S > 0) {
        const char* a0 = a_src(0, 0);
        const char* b0 = b_src(0);
        issue_unit(a0, offA[0], offA[1], slot_of(0, 0), wave);
        issue_unit(b0 + b1_off, offB[0], offB[1], slot_of(0, 1), wave);
        issue_unit(a0 + a1_off, offA[0], offA[1], slot_of(0, 2), wave);
        issue_unit(b0, offB[0], offB[1], slot_of(0, 3), wave);
        const char* a1 = a_src(e1, ks1);      // S >= 2 whenever S > 0 (boot + rescan entries)
        const char* b1p = b_src(ks1);
        issue_unit(a1, offA[0], offA[1], slot_of(1, 0), wave);
        issue_unit(b1p + b1_off, offB[0], offB[1], slot_of(1, 1), wave);
        issue_unit(a1 + a1_off, offA[0], offA[1], slot_of(1, 2), wave);
    }
    __syncthreads();                       // vmcnt(0) + barrier: prologue landed, state initialised
    if (wm == 1) SQE_BARRIER();            // stagger: group 1 runs one barrier behind group 0

    int refresh_pending = -1;
    int refresh_ctr = 0;
    for (int s = 0; s < S; ++s) {
        const char* u0 = slot_of(s, 0);
        const char* u1 = slot_of(s, 1);
        const char* u2 = slot_of(s, 2);
        const char* u3 = slot_of(s, 3);
        const bool entry_start = ks0 == 0 && s > 0;

        // ================= phase 0: quadrant (A^0, B^0); loads (s+1, u3)
#pragma unroll
        for (int fn = 0; fn < 2; ++fn)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) b[fn][kk] = unit_frag(u3, ruB + fn * 16, kk * 4 + cq);
#pragma unroll
        for (int fm = 0; fm < 4; ++fm)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) a[fm][kk] = unit_frag(u0, ruA + fm * 16, kk * 4 + cq);
        if (refresh_pending >= 0) {          // fetched during the previous K step, retired by its phase-3 wait
            refresh_apply(f, gstage, refresh_pending, p.gshift, wave, lane);
            refresh_pending = -1;
        }
        if (entry_start && p.gshift >= 0) {  // one slice of the global bound per tile entry
            refresh_pending = refresh_ctr % NSLICE8;
            ++refresh_ctr;
            refresh_issue(gmax_group, f.gstride, refresh_pending, gstage, wave, lane);
        }
        if (s + 1 < S) issue_unit(b_src(ks1), offB[0], offB[1], const_cast<char*>(slot_of(s + 1, 3)), wave);
        if (ks0 == 0) {
            if (s > 0) {
                // filter of the entry finished by the previous K step
                const int64_t row0 = (int64_t)tile_of(e0 - 1) * SCAN_BM;
                if (e0 - 1 == 0) {
                    filter_boot<8, 4>(acc, f, row0, wm * 128, wn * 64, lane);
                } else if (filter_tile<8, 4>(acc, f, row0, wm * 128, wn * 64, lane)) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // publish appended keys
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SQE_BARRIER();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int fm = 0; fm < 4; ++fm)
#pragma unroll
                for (int fn = 0; fn < 2; ++fn)
                    acc[fm][fn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[fm][kk], b[fn][kk], acc[fm][fn], 0, 0, 0);
        SQE_BARRIER();

        // ================= phase 1: quadrant (A^0, B^1); u0 is dead -> (s+2, u0)
#pragma unroll
        for (int fn = 0; fn < 2; ++fn)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) b[fn][kk] = unit_frag(u1, ruB + fn * 16, kk * 4 + cq);
        if (s + 2 < S) issue_unit(a_src(e2, ks2), offA[0], offA[1], const_cast<char*>(u0), wave);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SQE_BARRIER();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int fm = 0; fm < 4; ++fm)
#pragma unroll
                for (int fn = 0; fn < 2; ++fn)
                    acc[fm][2 + fn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[fm][kk], b[fn][kk], acc[fm][2 + fn], 0, 0, 0);
        SQE_BARRIER();

        // ================= phase 2: quadrant (A^1, B^1); u1 is dead -> (s+2, u1)
        if (entry_start) {
            // both groups' filter stores for the previous entry are published by now.  Done before
            // the A^1 fragment reads so the sweep's registers do not stack on top of them; the flag
            // word has a wave-uniform address (no per-lane address kept alive across the loop).
            if (e0 - 1 == 0) {
                publish_cmax(f, wave * 32, 32, lane);
            } else if (__builtin_amdgcn_readfirstlane(f.flags[wave]) != 0) {
                if (lane == 0) f.flags[wave] = 0;
                compact_owned(f, wave * 32, 32, p.trig, p.kp, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int fm = 0; fm < 4; ++fm)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) a[fm][kk] = unit_frag(u2, ruA + fm * 16, kk * 4 + cq);
        if (s + 2 < S) issue_unit(b_src(ks2) + b1_off, offB[0], offB[1], const_cast<char*>(u1), wave);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SQE_BARRIER();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int fm = 0; fm < 4; ++fm)
#pragma unroll
                for (int fn = 0; fn < 2; ++fn)
                    acc[4 + fm][2 + fn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[fm][kk], b[fn][kk], acc[4 + fm][2 + fn], 0, 0, 0);
        SQE_BARRIER();

        // ================= phase 3: quadrant (A^1, B^0); u2 is dead -> (s+2, u2); K step s+1 retired
#pragma unroll
        for (int fn = 0; fn < 2; ++fn)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) b[fn][kk] = unit_frag(u3, ruB + fn * 16, kk * 4 + cq);
        if (s + 2 < S) {
            issue_unit(a_src(e2, ks2) + a1_off, offA[0], offA[1], const_cast<char*>(u2), wave);
            asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        SQE_BARRIER();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int fm = 0; fm < 4; ++fm)
#pragma unroll
                for (int fn = 0; fn < 2; ++fn)
                    acc[4 + fm][fn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[fm][kk], b[fn][kk], acc[4 + fm][fn], 0, 0, 0);
        SQE_BARRIER();

        // advance the (entry, ks) windows
        e0 = e1; ks0 = ks1;
        e1 = e2; ks1 = ks2;
        if (++ks2 == KS) { ks2 = 0; ++e2; }
    }

    // ---- tail: filter of the last entry (the rescan of the first tile, or nothing)
    if (S > 0) {
        if (filter_tile<8, 4>(acc, f, (int64_t)tile_begin * SCAN_BM, wm * 128, wn * 64, lane))
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (wm == 0) SQE_BARRIER();            // undo the stagger
    __syncthreads();
    compact_owned(f, wave * 32, 32, p.kp + 1, p.kp, lane);
    __syncthreads();
    for (int i = tid; i < BN8; i += SCAN_THREADS)
        p.cand_cnt[(size_t)chunk * p.b_pad + q0 + i] = f.cnt[i];
}

}  // namespace

int launch_scan_bf16_p8(const ScanPlan& plan, const ScanArgs& a, hipStream_t stream) {
    if (plan.bn != BN8) return fail(SQE_ERR_INVALID, "scan p8: query block must be 256");
    ScanKernelArgs k = make_kernel_args(plan, a);
    static bool attr_set = false;
    if (!attr_set) {
        SQE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(scan_bf16_p8_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL(scan_bf16_p8_kernel, dim3(plan.n_chunks * plan.qblocks), dim3(SCAN_THREADS), LDS_BYTES,
                       stream, k);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

}  // namespace sqe
