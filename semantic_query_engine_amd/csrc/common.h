// common.h -- shared device/host helpers for libsqe (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>
#include <set>
#include <string>
#include <utility>

#include "host_errors.h"
#include "sqe.h"

namespace sqe {

// hipFuncSetAttribute acts on the CURRENT device's copy of a kernel: a process that opens contexts on two
// GPUs needs it once per (kernel, device), not once per process.
inline hipError_t ensure_dynamic_lds(const void* kern, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({kern, dev})) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.insert({kern, dev});
    return e;
}


// Experiment / schedule knobs (SQE_DBG, SQE_SCAN, SQE_KROT, ...) exist only in -DSQE_DEBUG_KNOBS builds
// (make KNOBS=1 -> libsqe_knobs.so, used by tools/ and the schedule-variant tests).  The shipped library
// reads NO environment variable: nothing outside the ABI can change what a search returns.
inline const char* knob_env(const char* name) {
#ifdef SQE_DEBUG_KNOBS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// ---------------------------------------------------------------- error plumbing (set_error, fail: host_errors.h)

#define SQE_HIP(expr)                                                                   \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            return ::sqe::fail(_e == hipErrorOutOfMemory ? SQE_ERR_OOM : SQE_ERR_HIP,   \
                               std::string(#expr) + ": " + hipGetErrorString(_e));      \
        }                                                                               \
    } while (0)

#define SQE_TRY(expr)              \
    do {                           \
        int _rc = (expr);          \
        if (_rc != SQE_OK) return _rc; \
    } while (0)

// ---------------------------------------------------------------- types
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef uint16_t bf16_t;  // raw bf16 bits in memory

constexpr int WAVE = 64;

// ---------------------------------------------------------------- device helpers
// fp32 -> bf16 bits, round to nearest even; NaN stays NaN (quiet).
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// two fp32 -> packed bf16 (lo in bits 0-15), round to nearest even, in ONE instruction (v_cvt_pk_bf16_f32, new
// on gfx950); NaN -> quiet NaN.  The encoder's epilogues use it; the index keeps f32_to_bf16 above, whose
// rounding the exactness certificate's residual is measured against.
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}

// Order-preserving map float -> uint32 (larger float <=> larger uint); NaNs are
// never fed to it (they fail the `s >= thr` filter first).
__device__ __forceinline__ uint32_t f32_orderable(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_from_orderable(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}

// Candidate key: (orderable(score) << 32) | (0xFFFFFFFF - row).  Larger key = better:
// higher score first, then LOWER row id.  Keys are unique because rows are.
__device__ __forceinline__ uint64_t make_key(float score, uint32_t row) {
    return ((uint64_t)f32_orderable(score) << 32) | (uint64_t)(0xFFFFFFFFu - row);
}
__device__ __forceinline__ uint32_t key_row(uint64_t key) { return 0xFFFFFFFFu - (uint32_t)key; }
__device__ __forceinline__ float key_score(uint64_t key) { return f32_from_orderable((uint32_t)(key >> 32)); }

// global -> LDS copy of 16 B per lane (LDS destination = wave-uniform base + lane * 16) as inline asm.
// Through the builtin, hipcc tracks every LDS-DMA piece as a pending write to LDS and puts
// s_waitcnt vmcnt(0) in front of the next LDS read it sees, which drains the ring; kernels that keep
// stages in flight across LDS reads issue their pieces here and count their own waits.
__device__ __forceinline__ void lds_dma16(const char* src, char* lds_wave_base) {
    const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_wave_base;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory");
}
__device__ __forceinline__ void lds_dma16_sc1(const char* src, char* lds_wave_base) {
    const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_wave_base;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc1" ::"v"(src), "s"(dst) : "memory");
}

// Lane id recomputed on the spot (2 VALU ops).  Used in rarely-run blocks of the pipelined
// kernels so that no per-lane address has to stay alive across the main loop: a spilled one
// is reloaded with a scratch load, whose compiler-inserted vmcnt(0) drains the DMA queue.
__device__ __forceinline__ int fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Radix-select step shared by the select kernels: hist[256] is complete (caller synchronised) and every thread
// of the block (>= 256 threads) calls.  Finds the bin that holds the `remaining`-th largest entry counting down
// from bin 255: bin (or -1 when the histogram holds fewer than `remaining` entries) and what is left to find
// inside it.  One thread per bin and a suffix scan (shuffles inside a wave, four wave totals through LDS)
// instead of one thread walking down from bin 255.
__device__ __forceinline__ void hist_locate(const int* hist, int remaining, int& bin, int& rem) {
    __shared__ int s_wave_total[4];
    __shared__ int s_found[2];
    const int tid = threadIdx.x;
    if (tid == 0) s_found[0] = -1;
    int h = 0, incl = 0;
    if (tid < 256) {
        const int ln = tid & 63;
        h = hist[tid];
        incl = h;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_down(incl, off, 64);
            if (ln + off < 64) incl += v;
        }
        if (ln == 0) s_wave_total[tid >> 6] = incl;
    }
    __syncthreads();
    if (tid < 256) {
        int above = 0;
        for (int w = (tid >> 6) + 1; w < 4; ++w) above += s_wave_total[w];
        const int excl = above + incl - h;          // entries in the bins above this one
        if (excl < remaining && excl + h >= remaining) {
            s_found[0] = tid;
            s_found[1] = remaining - excl;
        }
    }
    __syncthreads();
    bin = s_found[0];
    rem = bin < 0 ? remaining : s_found[1];
    __syncthreads();                                // the next call resets s_found
}

}  // namespace sqe
