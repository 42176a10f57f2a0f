// scan_common.h -- device code shared by the scan kernels: kernel argument block, the
// candidate-list machinery (threshold filter, append, wave-level compaction).
#pragma once

#include "kernels.h"

namespace sqe {

constexpr int SCAN_THREADS = 512;
constexpr int SCAN_NWAVES = 8;
constexpr int SCAN_ROW_BYTES = SCAN_BK * 2;   // 128 B per tile row per 64-wide bf16 K step

struct ScanKernelArgs {
    const bf16_t* db;
    const bf16_t* q;
    int64_t n_rows;
    int K;
    int B;
    int b_pad;
    int n_tiles;
    int tiles_per_chunk;
    int n_chunks;
    int qblocks;
    int kp;
    int trig;            // compaction trigger (kp <= trig <= CAND_CAP - SCAN_BM)
    uint64_t* cand;
    int* cand_cnt;
};

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

// Wave-level compaction of one candidate list to its best `kp` keys; updates the
// threshold of that query.  Caller guarantees n > kp and that no other wave touches
// this list concurrently.
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
        *thr_key_slot = T;
        *thr_s_slot = key_score(T);
    }
}

// Compaction sweep over the `per_wave` queries a wave owns: any list at or above `limit`
// entries is cut back to its best kp.  Must be called wave-uniformly.
__device__ __forceinline__ void compact_owned(uint64_t* cand_base, int first_q, int per_wave, int limit,
                                              int kp, int lane, int* cnt, float* thr_s, uint64_t* thr_key) {
    const int myq = first_q + lane;
    const bool need = lane < per_wave && cnt[myq] >= limit;
    uint64_t mask = __ballot(need);
    while (mask) {
        const int bq = first_q + (int)__builtin_ctzll(mask);
        mask &= mask - 1;
        compact_list(cand_base + (size_t)bq * CAND_CAP, cnt[bq], kp, lane, &cnt[bq], &thr_s[bq], &thr_key[bq]);
    }
}

}  // namespace sqe
