// ivf.hip -- IVF-flat on top of the flat index (BASELINE config 5: nlist = 4096, nprobe = 32).
//
//   train   spherical k-means (Lloyd): assignment = flat cosine top-1 of the sample against the
//           current centroids (the bf16 MFMA scan + fp32 rescore of the flat path, S7 "assign"),
//           update = segmented sum by fp32 atomics + renormalise (S7 "update")
//   add     rows are stored by the base flat index (fp32 master + bf16 copy); each row is assigned
//           to its best centroid; the inverted lists (row ids grouped by list) are rebuilt lazily
//   search  S5 coarse quantise = flat top-nprobe over the centroids; S6 list scan = for every list
//           the queries probing it x the list's rows, exact fp32 cosines from the master (rows are
//           gathered by id, 4 KiB each), written to a per-(query, probe) score strip; a per-query
//           radix select over its nprobe strips gives the exact top-k of the probed rows.
//
// The list scan is HBM-bound: with B = 1024 and nprobe = 32 nearly every list is probed, so one
// batch reads the fp32 master about once (41 GB at N = 10M).  Scores are exact fp32, so no rescore.
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "internal.h"

namespace sqe {

namespace {

struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
    ~Buf() { if (p) (void)hipFree(p); }
    int ensure(size_t need) {
        if (need <= bytes) return SQE_OK;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        hipError_t e = hipMalloc(&p, need);
        if (e != hipSuccess) return fail(SQE_ERR_OOM, std::string("ivf hipMalloc: ") + hipGetErrorString(e));
        bytes = need;
        return SQE_OK;
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

constexpr int IVF_QG = 8;               // queries per pass of the list scan
constexpr int IVF_QLDS = 8192;          // floats of LDS for the query group

// ---------------------------------------------------------------- small kernels
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx,
                                                          int n, int dim, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const float4* s = reinterpret_cast<const float4*>(x + (size_t)idx[r] * dim);
    float4* d = reinterpret_cast<float4*>(out + (size_t)r * dim);
    for (int v = lane; v < (dim >> 2); v += 64) d[v] = s[v];
}

// sums[assign[r]] += x[r] (fp32 atomics, 256 contiguous bytes per wave-instruction); counts[assign[r]]++
__global__ __launch_bounds__(256) void kmeans_accum_kernel(const float* __restrict__ x, const int64_t* __restrict__ assign,
                                                           int64_t n, int dim, float* __restrict__ sums,
                                                           int* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const int64_t a = assign[r];
    if (a < 0) return;
    const float* s = x + (size_t)r * dim;
    float* d = sums + (size_t)a * dim;
    for (int i = lane; i < dim; i += 64) atomicAdd(d + i, s[i]);
    if (lane == 0) atomicAdd(counts + a, 1);
}

// centroid = sums / ||sums|| where the list is non-empty; empty lists keep their old centroid
__global__ __launch_bounds__(256) void kmeans_finish_kernel(float* __restrict__ centroids, const float* __restrict__ sums,
                                                            const int* __restrict__ counts, int nlist, int dim) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= nlist || counts[c] == 0) return;
    const float* s = sums + (size_t)c * dim;
    float ss = 0.f;
    for (int i = lane; i < dim; i += 64) ss += s[i] * s[i];
    const float den = sqrtf(wave_sum(ss)) + 1e-9f;
    for (int i = lane; i < dim; i += 64) centroids[(size_t)c * dim + i] = s[i] / den;
}

__global__ void ivf_store_assign_kernel(const int64_t* __restrict__ ids, int64_t n, int* __restrict__ assign,
                                        int* __restrict__ counts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int a = (int)ids[i];
    assign[i] = a;
    if (counts && a >= 0) atomicAdd(counts + a, 1);
}

// assign[rows[i]] = lists[i] (rows overwritten by sqe_index_update)
__global__ void ivf_scatter_assign_kernel(const int* __restrict__ lists, const int64_t* __restrict__ rows, int64_t n,
                                          int* __restrict__ assign) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) assign[rows[i]] = lists[i];
}

__global__ void ivf_count_kernel(const int* __restrict__ assign, int64_t n, int* __restrict__ counts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && assign[i] >= 0) atomicAdd(counts + assign[i], 1);
}

// order[p] = the row at position p of the list-ordered sequence; dpos[p] = its row in the int8 copy, where every list starts
// on a 256-row tile (tile_off: tiles in front of each list)
__global__ void ivf_scatter_kernel(const int* __restrict__ assign, int64_t n, const int64_t* __restrict__ offsets,
                                   const int64_t* __restrict__ tile_off, int* __restrict__ cursor, int* __restrict__ order,
                                   int* __restrict__ dpos) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int a = assign[i];
    if (a < 0) return;
    const int c = atomicAdd(cursor + a, 1);
    order[offsets[a] + c] = (int)i;
    dpos[offsets[a] + c] = (int)(tile_off[a] * 256 + c);
}

// top-nprobe lists of a query from its dense coarse scores [nlist] (score desc, list id asc); one workgroup
// per query, scores staged in LDS, byte-wise radix select on (orderable score, ~id) keys
// The GEMM scores come from bf16 operands: the best nprobe + 8 by those scores are re-scored in fp32 (raw query
// row against the normalised fp32 centroid; dividing by the query norm does not change the order) and the
// best nprobe of them returned, so the list order is the exact fp32 one.
constexpr int PSEL_THREADS = 1024;      // (r04; 256 before: the re-score of the nprobe + 8 best centroids is a chain of row fetches per wave)
__global__ __launch_bounds__(PSEL_THREADS) void ivf_probe_select_kernel(const float* __restrict__ scores, int nlist, int nprobe,
                                                               const float* __restrict__ rows, const float* __restrict__ cent, int dim,
                                                               int64_t* __restrict__ probes, float* __restrict__ probes_cos) {
    extern __shared__ __attribute__((aligned(16))) float ssc[];      // [nlist]
    __shared__ int hist[256];
    __shared__ int scratch[4];
    __shared__ uint64_t top[MAX_KP];
    const int q = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < nlist; i += PSEL_THREADS) ssc[i] = scores[(size_t)q * nlist + i];
    __syncthreads();
    auto key_of = [&](int i) { return make_key(ssc[i] + 0.0f, (uint32_t)i); };
    const int nsel = min(min(nprobe + 8, MAX_KP), nlist);      // candidates kept for the fp32 re-score
    uint64_t prefix = 0;
    int remaining = nsel;
    const bool all = nlist <= nsel;
    for (int byte = 7; byte >= 0 && !all; --byte) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const int shift = byte * 8;
        for (int i = tid; i < nlist; i += PSEL_THREADS) {
            const uint64_t key = key_of(i);
            if (byte == 7 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(int)((key >> shift) & 0xff)], 1);
        }
        __syncthreads();
        {
            int hb, hr;
            hist_locate(hist, remaining, hb, hr);
            if (tid == 0) { scratch[0] = hb < 0 ? 0 : hb; scratch[1] = hr; scratch[3] = hb < 0 ? 0 : hist[hb]; }
        }
        __syncthreads();
        prefix |= ((uint64_t)scratch[0] << shift);
        remaining = scratch[1];
        const bool whole_bin = scratch[3] == remaining;       // keys are unique: every key of the located bin is wanted -> keys >= prefix are the nsel best
        __syncthreads();
        if (whole_bin) break;                                  // (r04c: three or four passes of the eight on scores that differ)
    }
    const uint64_t T = all ? 0ull : prefix;
    if (tid == 0) scratch[2] = 0;
    __syncthreads();
    for (int i = tid; i < nlist; i += PSEL_THREADS) {
        const uint64_t key = key_of(i);
        if (key >= T) {
            const int slot = atomicAdd(&scratch[2], 1);
            if (slot < MAX_KP) top[slot] = key;
        }
    }
    __syncthreads();
    const int m = min(scratch[2], nsel);
    {
        const int lane = tid & 63, wave = tid >> 6;
        const float4* qv = reinterpret_cast<const float4*>(rows + (size_t)q * dim);
        const int nvec = dim >> 2;
        float qq = 0.f;
        for (int v4 = lane; v4 < nvec; v4 += 64) {
            const float4 b = qv[v4];
            qq = fmaf(b.x, b.x, qq); qq = fmaf(b.y, b.y, qq); qq = fmaf(b.z, b.z, qq); qq = fmaf(b.w, b.w, qq);
        }
        const float inv = 1.0f / (sqrtf(wave_sum(qq)) + 1e-9f);
        for (int e = wave; e < m; e += PSEL_THREADS / 64) {
            const uint32_t id = key_row(top[e]);
            const float4* cv = reinterpret_cast<const float4*>(cent + (size_t)id * dim);
            float s = 0.f;
            for (int v4 = lane; v4 < nvec; v4 += 64) {
                const float4 a = cv[v4], b = qv[v4];
                s = fmaf(a.x, b.x, s); s = fmaf(a.y, b.y, s); s = fmaf(a.z, b.z, s); s = fmaf(a.w, b.w, s);
            }
            s = wave_sum(s) * inv + 0.0f;
            if (lane == 0) top[e] = make_key(s, id);
        }
    }
    __syncthreads();
    const int mk = min(m, nprobe);
    for (int i = tid; i < m; i += PSEL_THREADS) {
        const uint64_t ki = top[i];
        int rank = 0;
        for (int j = 0; j < m; ++j) rank += top[j] > ki ? 1 : 0;
        if (rank < nprobe) {
            probes[(size_t)q * nprobe + rank] = (int64_t)key_row(ki);
            probes_cos[(size_t)q * nprobe + rank] = key_score(ki);
        }
    }
    for (int i = mk + tid; i < nprobe; i += PSEL_THREADS) {
        probes[(size_t)q * nprobe + i] = -1;
        probes_cos[(size_t)q * nprobe + i] = -INFINITY;
    }
}

// (query, probe) pairs bucketed by list
__global__ void ivf_bucket_kernel(const int64_t* __restrict__ probes, int B, int nprobe, int* __restrict__ lcount,
                                  int* __restrict__ lq, int cap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * nprobe) return;
    const int64_t L = probes[i];
    if (L < 0) return;
    const int slot = atomicAdd(lcount + L, 1);
    if (slot < cap) lq[(size_t)L * cap + slot] = i;
}

// S6: one workgroup per list; exact fp32 cosines of the list's rows against the queries probing it
__global__ __launch_bounds__(256) void ivf_list_scan_kernel(const float* __restrict__ master, const float* __restrict__ qn,
                                                            const int* __restrict__ order,
                                                            const int64_t* __restrict__ offsets,
                                                            const int* __restrict__ lcount, const int* __restrict__ lq,
                                                            int cap, int nprobe, int K, int max_len,
                                                            float* __restrict__ pair_scores) {
    __shared__ __attribute__((aligned(16))) float sq[IVF_QLDS];
    __shared__ int spair[IVF_QG];
    const int L = blockIdx.x;
    const int m = min(lcount[L], cap);
    const int64_t off = offsets[L];
    const int len = (int)(offsets[L + 1] - off);
    if (m == 0 || len == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qg = min(IVF_QG, IVF_QLDS / K);
    const int nvec = K >> 2;
    for (int g0 = 0; g0 < m; g0 += qg) {
        const int gq = min(qg, m - g0);
        __syncthreads();
        if (tid < gq) spair[tid] = lq[(size_t)L * cap + g0 + tid];
        __syncthreads();
        for (int i = tid; i < gq * K; i += 256) {
            const int j = i / K, d = i - j * K;
            sq[i] = qn[(size_t)(spair[j] / nprobe) * K + d];
        }
        __syncthreads();
        if (K == 1024) {
            // fast path for the reference dimension: two rows per wave in flight, all eight 1-KiB
            // row loads issued before the first use, query fragments read once for both rows
            for (int i = wave * 2; i < len; i += 8) {
                const bool two = i + 1 < len;
                const float4* r0 = reinterpret_cast<const float4*>(master + (size_t)order[off + i] * 1024);
                const float4* r1 = reinterpret_cast<const float4*>(master + (size_t)order[off + (two ? i + 1 : i)] * 1024);
                float4 a0[4], a1[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) { a0[t] = r0[lane + 64 * t]; a1[t] = r1[lane + 64 * t]; }
                float acc0[IVF_QG], acc1[IVF_QG];
#pragma unroll
                for (int j = 0; j < IVF_QG; ++j) {
                    acc0[j] = 0.f; acc1[j] = 0.f;
                    if (j < gq) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const float4 b = *reinterpret_cast<const float4*>(&sq[j * 1024 + (lane + 64 * t) * 4]);
                            acc0[j] = fmaf(a0[t].x, b.x, acc0[j]); acc0[j] = fmaf(a0[t].y, b.y, acc0[j]);
                            acc0[j] = fmaf(a0[t].z, b.z, acc0[j]); acc0[j] = fmaf(a0[t].w, b.w, acc0[j]);
                            acc1[j] = fmaf(a1[t].x, b.x, acc1[j]); acc1[j] = fmaf(a1[t].y, b.y, acc1[j]);
                            acc1[j] = fmaf(a1[t].z, b.z, acc1[j]); acc1[j] = fmaf(a1[t].w, b.w, acc1[j]);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < IVF_QG; ++j) {
                    if (j < gq) {
                        const float s0 = wave_sum(acc0[j]) + 0.0f, s1 = wave_sum(acc1[j]) + 0.0f;
                        if (lane == 0) {
                            pair_scores[(size_t)spair[j] * max_len + i] = s0;
                            if (two) pair_scores[(size_t)spair[j] * max_len + i + 1] = s1;
                        }
                    }
                }
            }
            continue;
        }
        for (int i = wave; i < len; i += 4) {
            const float4* rv = reinterpret_cast<const float4*>(master + (size_t)order[off + i] * K);
            float acc[IVF_QG];
#pragma unroll
            for (int j = 0; j < IVF_QG; ++j) acc[j] = 0.f;
            for (int v = lane; v < nvec; v += 64) {
                const float4 a = rv[v];
#pragma unroll
                for (int j = 0; j < IVF_QG; ++j) {
                    if (j < gq) {
                        const float4 b = *reinterpret_cast<const float4*>(&sq[j * K + v * 4]);
                        acc[j] = fmaf(a.x, b.x, acc[j]); acc[j] = fmaf(a.y, b.y, acc[j]);
                        acc[j] = fmaf(a.z, b.z, acc[j]); acc[j] = fmaf(a.w, b.w, acc[j]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < IVF_QG; ++j) {
                if (j < gq) {
                    const float s = wave_sum(acc[j]) + 0.0f;
                    if (lane == 0) pair_scores[(size_t)spair[j] * max_len + i] = s;
                }
            }
        }
    }
}

// S6 (MFMA form): one workgroup per list.  Rows of the list are gathered from the bf16 scan copy by
// their ids (the LDS-DMA takes a per-lane source address, so a gather costs nothing extra), 256 rows x
// up to 64 of the queries that probe the list per tile, v_mfma_f32_16x16x32_bf16, 3-stage LDS ring
// (two K steps in flight).  Scores go to the same per-(query, probe) strips as the fp32 kernel above;
// ivf_select_kernel re-scores the best of them in fp32.  HBM-bound: every probed row is read once per
// group of 64 queries, at 2 bytes per element.
constexpr int LS_ROWS = 256, LS_Q = 64, LS_NST = 3, LS_SEG = 4096;
constexpr int LS_STAGE = (LS_ROWS + LS_Q) * 128;
constexpr int LS_LDS = LS_NST * LS_STAGE + LS_SEG * 4 + LS_Q * 4;

// Stores through inline asm: a store hipcc knows about would make it guard the ring's invisible DMA pieces with
// vmcnt(0).  hipcc pads no hazard whose producer or consumer sits inside an asm string, so the two software wait-state
// rules that touch this store are met INSIDE it ("asm stores"):
//   (a) XDL (MFMA) write of a VGPR -> VMEM read of it as store data: 11 wait states behind an 8-pass MFMA
//       (v_mfma_f32_16x16x32_bf16) -- mfma_results_to_vmem() below, once per epilogue, tied to the accumulators;
//   (b) a VMEM store of more than 8 bytes reads its data registers up to two wait states after it issued: nothing
//       may write them before -- the s_nop 1 that ends the string (a nop in FRONT of the store, r02's encoder attempt,
//       covers nothing: the next v_cvt_pk_bf16_f32 overwrote the registers of the store just issued).
__device__ __forceinline__ void global_store_f4_asm(float* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// 12 wait states between the MFMAs that produced `acc` and whatever follows (the "+v" operands keep those MFMAs above)
template <int N, int M>
__device__ __forceinline__ void mfma_results_to_vmem(f32x4 (&acc)[N][M]) {
    static_assert(N == 2 && M == 4, "the list-scan tile: 2 x 4 fragments per wave");
    asm volatile("s_nop 11"
                 : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]), "+v"(acc[1][1]),
                   "+v"(acc[1][2]), "+v"(acc[1][3]));
}

__global__ __launch_bounds__(512) void ivf_list_scan_mfma_kernel(const bf16_t* __restrict__ scan, int pitch,
                                                                 const bf16_t* __restrict__ qb, int qpitch,
                                                                 const int* __restrict__ order,
                                                                 const int64_t* __restrict__ offsets,
                                                                 const int* __restrict__ lcount, const int* __restrict__ lq,
                                                                 int cap, int nprobe, int K, int max_len,
                                                                 float* __restrict__ pair_scores) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* sorder = reinterpret_cast<int*>(smem + LS_NST * LS_STAGE);
    int* spair = sorder + LS_SEG;
    const int L = blockIdx.x;
    const int m = min(lcount[L], cap);
    const int64_t off = offsets[L];
    const int len_all = (int)(offsets[L + 1] - off);
    if (m == 0 || len_all == 0) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KS = K / 64;
    const char* scan_b = reinterpret_cast<const char*>(scan);
    const char* qb_b = reinterpret_cast<const char*>(qb);
    const int chunk_lo = lane & 7, r_in_piece = lane >> 3;

    for (int seg0 = 0; seg0 < len_all; seg0 += LS_SEG) {
        const int len = min(LS_SEG, len_all - seg0);
        __syncthreads();
        for (int i = tid; i < len; i += 512) sorder[i] = order[off + seg0 + i];
        const int n_tiles = (len + LS_ROWS - 1) / LS_ROWS;
        for (int g0 = 0; g0 < m; g0 += LS_Q) {
            const int gq = min(LS_Q, m - g0);
            __syncthreads();
            if (tid < LS_Q) spair[tid] = lq[(size_t)L * cap + g0 + min(tid, gq - 1)];
            __syncthreads();
            // this lane's query row of the one Q piece its wave issues per stage
            const int qj = wave * 8 + r_in_piece;
            const size_t qoff = (size_t)(spair[qj] / nprobe) * qpitch + ((chunk_lo ^ ((qj >> 1) & 7)) << 4);

            const int total = n_tiles * KS;
            int i_tile = -1, i_ks = KS - 1;                 // issue cursor
            size_t aoff[4];
            auto issue = [&](int s) {
                if (++i_ks == KS) {                         // the cursor enters a new tile: its rows' addresses
                    i_ks = 0;
                    ++i_tile;
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int r = (wave + 8 * it) * 8 + r_in_piece;
                        const int rr = min(i_tile * LS_ROWS + r, len - 1);          // rows past the end repeat the last one
                        aoff[it] = (size_t)sorder[rr] * pitch + ((chunk_lo ^ ((r >> 1) & 7)) << 4);
                    }
                }
                char* buf = smem + (s % LS_NST) * LS_STAGE;
#pragma unroll
                for (int it = 0; it < 4; ++it) lds_dma16(scan_b + aoff[it] + (size_t)i_ks * 128, buf + (wave + 8 * it) * 1024);
                lds_dma16(qb_b + qoff + (size_t)i_ks * 128, buf + LS_ROWS * 128 + wave * 1024);
            };
            for (int s = 0; s < LS_NST - 1 && s < total; ++s) issue(s);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);

            f32x4 acc[2][4];
            int ks = 0, tile = 0;
            for (int s = 0; s < total; ++s) {
                const bool more = s + LS_NST - 1 < total;
                if (more) issue(s + LS_NST - 1);
                if (ks == 0) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                const char* tA = smem + (s % LS_NST) * LS_STAGE;
                const char* tB = tA + LS_ROWS * 128;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    bf16x8 a[2], b[4];
                    const int c = kk * 4 + (lane >> 4);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int r = wave * 32 + i * 16 + (lane & 15);
                        a[i] = *reinterpret_cast<const bf16x8*>(tA + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = j * 16 + (lane & 15);
                        b[j] = *reinterpret_cast<const bf16x8*>(tB + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                }
                if (++ks == KS) {
                    // scores of this tile: a lane holds 4 consecutive rows of one query per fragment
                    mfma_results_to_vmem(acc);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int col = j * 16 + (lane & 15);
                        if (col < gq) {
                            float* strip = pair_scores + (size_t)spair[col] * max_len + seg0 + tile * LS_ROWS;
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                const int r = wave * 32 + i * 16 + (lane >> 4) * 4;
                                if (tile * LS_ROWS + r < len)      // len and max_len are padded reads, not padded strips:
                                    global_store_f4_asm(strip + r, acc[i][j]);
                            }
                        }
                    }
                    ks = 0;
                    ++tile;
                }
                if (more) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// int8 twin of the list scan (r03): the same tile, ring and LDS image -- a K step is 128 int8 elements = 128 B per row --
// over an int8 copy with per-row scales (quant.hip) kept in LIST order: every list is a run of whole 256-row tiles in the flat scan's
// tiled layout (64-B K slices, 16 KiB apart), so a K step of a tile is 32 contiguous KiB -- a stream, where the bf16 kernel gathers rows
// by id, 128 B at a time -- v_mfma_i32_16x16x64_i8, half the bytes per probed row.  The
// strips receive ESTIMATED cosines (acc x row scale x query scale); ivf_select_kernel re-scores the best of them in fp32
// exactly as before.  IVF answers are approximate by nature (no certificate): the int8 estimate (error ~2e-3 at most on
// Gaussian-like rows) only decides which kp = max(32, 4k) rows are re-scored.
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
constexpr int LS_LDS_I8 = LS_NST * LS_STAGE + LS_SEG * 4 + LS_SEG * 4 + LS_Q * 4 + LS_Q * 4;
static_assert(LS_LDS_I8 <= 160 * 1024, "LDS budget of the int8 list scan");

__global__ __launch_bounds__(512) void ivf_list_scan_i8_kernel(const int8_t* __restrict__ scan, int64_t tile_stride, const uint32_t* __restrict__ sxi,
                                                               const int8_t* __restrict__ qb, int qpitch, const uint32_t* __restrict__ sqi, float unit2,
                                                                 const int64_t* __restrict__ tile_off,
                                                                 const int64_t* __restrict__ offsets,
                                                                 const int* __restrict__ lcount, const int* __restrict__ lq,
                                                                 int cap, int nprobe, int K, int max_len,
                                                                 float* __restrict__ pair_scores,
                                                                 const int64_t* __restrict__ probes, int n_pairs, int split) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* sorder = reinterpret_cast<int*>(smem + LS_NST * LS_STAGE);
    float* sscale = reinterpret_cast<float*>(sorder + LS_SEG);     // [LS_SEG] row scales (sxi) of the segment's rows
    int* spair = reinterpret_cast<int*>(sscale + LS_SEG);
    float* sqscale = reinterpret_cast<float*>(spair + LS_Q);       // [LS_Q] unit^2 * sqi of the group's queries
    // Two grids.  List mode (n_pairs == 0): one workgroup per list, every query that probes it.  Pair mode (a handful of
    // queries: fewer probed lists than CUs, and a CU takes in ~25 GB/s): one workgroup per (query, probe) pair and SEGMENT of its
    // list -- `split` workgroups share a list's tiles, so one query's 32 lists are read by a few hundred CUs instead of 32.
    const bool pair_mode = n_pairs > 0;
    int L, m, seg = 0, pair0 = 0;
    if (pair_mode) {
        pair0 = (int)blockIdx.x / split;
        seg = (int)blockIdx.x % split;
        const int64_t l64 = probes[pair0];
        if (l64 < 0) return;
        L = (int)l64;
        m = 1;
    } else {
        L = blockIdx.x;
        m = min(lcount[L], cap);
    }
    const int64_t off = offsets[L];
    const int len_all = (int)(offsets[L + 1] - off);
    if (m == 0 || len_all == 0) return;
    int row_lo = 0, row_hi = len_all;
    if (pair_mode && split > 1) {
        const int nt_all = (len_all + LS_ROWS - 1) / LS_ROWS, per = (nt_all + split - 1) / split;
        row_lo = min(len_all, seg * per * LS_ROWS);
        row_hi = min(len_all, (seg + 1) * per * LS_ROWS);
        if (row_lo >= row_hi) return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KS = K / 128;                        // 128 int8 elements = 128 B per row per K step: the bf16 kernel's LDS image
    const char* scan_b = reinterpret_cast<const char*>(scan);
    const char* qb_b = reinterpret_cast<const char*>(qb);
    const int chunk_lo = lane & 7, r_in_piece = lane >> 3;

    for (int seg0 = row_lo; seg0 < row_hi; seg0 += LS_SEG) {
        const int len = min(LS_SEG, row_hi - seg0);
        __syncthreads();
        const int64_t tile0 = tile_off[L] + seg0 / LS_ROWS;                        // first tile of the segment in the copy
        for (int i = tid; i < len; i += 512) sscale[i] = (float)sxi[tile0 * LS_ROWS + i];
        const int n_tiles = (len + LS_ROWS - 1) / LS_ROWS;
        for (int g0 = 0; g0 < m; g0 += LS_Q) {
            const int gq = min(LS_Q, m - g0);
            __syncthreads();
            if (tid < LS_Q) {
                const int pr = pair_mode ? pair0 : lq[(size_t)L * cap + g0 + min(tid, gq - 1)];
                spair[tid] = pr;
                sqscale[tid] = unit2 * (float)sqi[pr / nprobe];
            }
            __syncthreads();
            // this lane's query row of the one Q piece its wave issues per stage
            const int qj = wave * 8 + r_in_piece;
            const size_t qoff = (size_t)(spair[qj] / nprobe) * qpitch + ((chunk_lo ^ ((qj >> 1) & 7)) << 4);

            const int total = n_tiles * KS;
            int i_tile = -1, i_ks = KS - 1;                 // issue cursor
            size_t aoff[4];
            auto issue = [&](int s) {
                if (++i_ks == KS) {                         // the cursor enters a new tile: its rows' addresses
                    i_ks = 0;
                    ++i_tile;
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int r = (wave + 8 * it) * 8 + r_in_piece;
                        const int rr = min(i_tile * LS_ROWS + r, len - 1) - i_tile * LS_ROWS;   // rows past the end repeat the last one (same tile)
                        // the lane's 16 bytes of a K step: chunk c of the row's 128 B = bytes (c & 3) * 16 of its 64-B slice 2 ks + (c >> 2)
                        const int c = chunk_lo ^ ((r >> 1) & 7);
                        aoff[it] = (size_t)(tile0 + i_tile) * tile_stride + (size_t)(c >> 2) * 16384 + rr * 64 + (c & 3) * 16;
                    }
                }
                char* buf = smem + (s % LS_NST) * LS_STAGE;
#pragma unroll
                for (int it = 0; it < 4; ++it) lds_dma16(scan_b + aoff[it] + (size_t)i_ks * 32768, buf + (wave + 8 * it) * 1024);
                lds_dma16(qb_b + qoff + (size_t)i_ks * 128, buf + LS_ROWS * 128 + wave * 1024);
            };
            for (int s = 0; s < LS_NST - 1 && s < total; ++s) issue(s);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);

            i32x4_t acc[2][4];
            int ks = 0, tile = 0;
            for (int s = 0; s < total; ++s) {
                const bool more = s + LS_NST - 1 < total;
                if (more) issue(s + LS_NST - 1);
                if (ks == 0) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = i32x4_t{0, 0, 0, 0};
                }
                const char* tA = smem + (s % LS_NST) * LS_STAGE;
                const char* tB = tA + LS_ROWS * 128;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    i32x4_t a[2], b[4];
                    const int c = kk * 4 + (lane >> 4);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int r = wave * 32 + i * 16 + (lane & 15);
                        a[i] = *reinterpret_cast<const i32x4_t*>(tA + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = j * 16 + (lane & 15);
                        b[j] = *reinterpret_cast<const i32x4_t*>(tB + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[j], acc[i][j], 0, 0, 0);
                }
                if (++ks == KS) {
                    // scores of this tile: a lane holds 4 consecutive rows of one query per fragment
                    // estimated cosines: acc * (sxi[row] unit) * (sqi[query] unit); the conversions are VALU instructions
                    // hipcc sees (it pads the MFMA -> VALU distance), their results feed the asm store
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int col = j * 16 + (lane & 15);
                        if (col < gq) {
                            float* strip = pair_scores + (size_t)spair[col] * max_len + seg0 + tile * LS_ROWS;
                            const float qs = sqscale[col];
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                const int r = wave * 32 + i * 16 + (lane >> 4) * 4;
                                if (tile * LS_ROWS + r < len) {     // len and max_len are padded reads, not padded strips:
                                    const f32x4 rs = *reinterpret_cast<const f32x4*>(sscale + tile * LS_ROWS + r);
                                    f32x4 v;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] = (float)acc[i][j][e] * rs[e] * qs;
                                    global_store_f4_asm(strip + r, v);
                                }
                            }
                        }
                    }
                    ks = 0;
                    ++tile;
                }
                if (more) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// r04: STREAMING form of the int8 list scan -- the rows never touch LDS.  The tiled int8 copy keeps the 64-byte K slice h of
// tile row r at h * 16 KiB + r * 64, so the A operand of v_mfma_i32_16x16x64_i8 for 16 consecutive rows and one slice -- lane l:
// row l & 15, bytes (l >> 4) * 16 .. + 15 -- is ONE contiguous KiB: a wave loads it with a single global_load_dwordx4 straight into
// the operand registers.  No staging ring, no DMA pieces, no workgroup barrier in the loop: every wave streams its own 64 rows of
// each tile through a register ring of ST_RING fragments (16 KiB in flight per wave; workgroups of four waves, two or more per CU:
// one loads its queries while the other streams; hipcc counts the vmcnt waits -- the loads are plain loads), and only the few queries
// that probe the list (8 on average at batch 1024, nprobe 32 of 4096) sit in LDS, loaded once per unit.  The staged kernel above kept 2 x 40 KiB in flight per CU behind one
// barrier per K step and re-sent the query block with every step: 0.50 of HBM at batch 1024 (profiles/r03_configs/cfg_ivf.json).
// A workgroup takes one UNIT: up to ST_UNIT_TILES consecutive tiles of one list (table built with the lists), so long lists
// spread over several CUs and the grid is a few thousand even units.  Strips receive estimated cosines as before.
constexpr int ST_Q = 32;                  // queries of a list resident in LDS per pass (more: the unit's rows are streamed again)
constexpr int ST_RING = 16;               // 1-KiB row fragments in flight per wave
constexpr int ST_FR = 4;                  // fragments (16 rows each) of a wave per K slice: 4 waves x 64 rows = a 256-row tile
constexpr int ST_SL = ST_RING / ST_FR;    // K slices the ring holds
constexpr int ST_UNIT_TILES = 5;
constexpr int ST_THREADS = 256;
constexpr int ST_PATCH_PITCH = 68;        // floats per query row of a wave's transpose patch: 64 rows + 4 (16-byte rows of different queries on different banks)
constexpr int ST_PATCH_ROWS = 36;         // query rows of patch per wave: four tiles of <= 8 (padded: 9) queries, one tile of 32

// issue cursor of a wave's stream: the next K slice (its ST_FR fragments)
struct StCursor {
    const char* tile;                     // base of the cursor's tile (+ the wave's rows)
    int h;                                // its K slice
    // r04c: a list ends inside its last tile (2,441 rows on average = 9.5 tiles: 5 % of the copy is padding).  The wave's fragments of
    // that tile that lie wholly behind the list's end are read from `dummy` -- a KiB every unit keeps hot in L2 (the head of the query
    // block) -- instead of from HBM: their scores are dropped by the row < len tests anyway.  A select on the address, no branch.
    const char* last;                     // the list's last tile if it is partial and belongs to this unit (else null)
    int nvalid;                           // fragments of this wave in it that hold rows of the list
    const char* dummy;
};
__device__ __forceinline__ void st_issue_slice(StCursor& c, int HS, int64_t tile_stride, unsigned aoff, i32x4_t* r) {
    const char* p = c.tile + (size_t)c.h * 16384 + aoff;
    const bool pad = c.tile == c.last;
#pragma unroll
    for (int f = 0; f < ST_FR; ++f) {
        const char* pf = (pad && f >= c.nvalid) ? c.dummy + aoff : p + f * 1024;
#ifdef SQE_ST_NO_NT              // (timing build: default cache policy)
        r[f] = *reinterpret_cast<const i32x4_t*>(pf);
#else
        r[f] = __builtin_nontemporal_load(reinterpret_cast<const i32x4_t*>(pf));
#endif
    }
    if (++c.h == HS) { c.h = 0; c.tile += tile_stride; }
}
// ST_SL K slices: the fragments of the ring against the query fragments of slices hb .. hb + ST_SL - 1 (bq: this lane's query
// row in LDS + hb * 64).  ISSUE: a slice's registers are refilled, as soon as they have been read, with the slice ST_SL ahead.
// The refill is unconditional inside this body and the LAST group of a unit (ISSUE = false) is code of its own behind the loop: with
// a branch around the loads hipcc has to assume the path without them at every join and waits for vmcnt(0) in front of each use --
// the ring drained at every step (first version of this kernel); this way it counts vmcnt(12) in the steady state.
template <bool ISSUE>
__device__ __forceinline__ void st_group(i32x4_t (&ring)[ST_RING], i32x4_t (&acc)[ST_FR][2], const char* bq, int qrow, int bcq, int bsw, StCursor& c,
                                         int HS, int64_t tile_stride, unsigned aoff) {
#pragma unroll
    for (int e = 0; e < ST_SL; ++e) {
        const int boff = ((e * 4 + bcq) ^ bsw) << 4;
        const i32x4_t b0 = *reinterpret_cast<const i32x4_t*>(bq + boff);
        const i32x4_t b1 = *reinterpret_cast<const i32x4_t*>(bq + 16 * qrow + boff);
#pragma unroll
        for (int f = 0; f < ST_FR; ++f) {
            acc[f][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[e * ST_FR + f], b0, acc[f][0], 0, 0, 0);
            acc[f][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[e * ST_FR + f], b1, acc[f][1], 0, 0, 0);
        }
        if (ISSUE) st_issue_slice(c, HS, tile_stride, aoff, &ring[e * ST_FR]);
    }
}

// COLLECT (r04b): the scores are not written to strips.  A sample pass (this kernel in strip mode over the FIRST tile of every list,
// ~10 % of the rows) and ivf_threshold_kernel have given every query a threshold that a few hundred of its ~78 k probed rows reach;
// the pass over the other tiles keeps only the (score, row) keys at or above it: they wait in an LDS buffer of the workgroup and go to
// the query's list (one global atomic per key) behind the unit's last load.  ivf_select_list_kernel ranks the list.  The 320 MB of
// strips per batch of 1024 and the pass that re-read them are gone (what is left of them: the sample's 32 MB).
struct StCollect {
    const float* thr;          // [B] per-query threshold on the estimated cosine
    int* cnt;                  // [B] keys appended to a query's list (may exceed list_cap: overflow -> the fallback)
    uint64_t* list;            // [B, list_cap]
    int list_cap;
    const int* order;          // row id of every list position
};
constexpr int ST_CBUF = 1024;  // keys a workgroup can hold back per unit and pass (~40 expected)
template <bool COLLECT>
__global__ __launch_bounds__(ST_THREADS) void ivf_list_stream_i8_kernel(const int8_t* __restrict__ scan, int64_t tile_stride, const uint32_t* __restrict__ sxi,
                                                                        const int8_t* __restrict__ qb, int qpitch, const uint32_t* __restrict__ sqi, float unit2,
                                                                        const int4* __restrict__ units, const int64_t* __restrict__ tile_off,
                                                                        const int64_t* __restrict__ offsets, const int* __restrict__ lcount,
                                                                        const int* __restrict__ lq, int cap, int nprobe, int K, int max_len,
                                                                        float* __restrict__ pair_scores, StCollect col, const int* __restrict__ gate,
                                                                        int* __restrict__ queue, int n_units,
                                                                        const int64_t* __restrict__ pair_probes, int pair_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_next;
    if (gate && *gate == 0) return;                            // (the strip-mode fallback of a collect search: only if some query asked for it)
    // queue != null: PERSISTENT workgroups (two per CU) take units from a counter -- ~12 k workgroups of ~100 us each otherwise, and the
    // stream read at 5.5 TB/s where the flat small-batch kernel, persistent, reads at 6.5
    for (int unit = blockIdx.x;;) {
    if (queue) {
        __syncthreads();                                       // (the previous unit is done with LDS)
        if (threadIdx.x == 0) s_next = atomicAdd(queue, 1);
        __syncthreads();
        unit = s_next;
    }
    if (unit >= n_units) return;
    const int qrow = K + 128;                                  // LDS pitch of a query row: rows r and r + 1 start 32 banks apart
    float* sscale = reinterpret_cast<float*>(smem + ST_Q * qrow);          // [ST_UNIT_TILES * 256] row scales of the unit
    int* spair = reinterpret_cast<int*>(sscale + ST_UNIT_TILES * LS_ROWS);
    float* sqscale = reinterpret_cast<float*>(spair + ST_Q);
    // per wave: ST_PATCH_ROWS query rows of [64 rows + 4] floats: the transposed scores of finished tiles, waiting for their stores
    float* spatch = sqscale + ST_Q + (threadIdx.x >> 6) * (ST_PATCH_ROWS * ST_PATCH_PITCH);
    // COLLECT: the patch region holds the workgroup's key buffer instead: [ST_CBUF] keys, [ST_CBUF] query columns, a counter, thresholds
    uint64_t* cbuf_key = reinterpret_cast<uint64_t*>(sqscale + ST_Q);
    unsigned char* cbuf_col = reinterpret_cast<unsigned char*>(cbuf_key + ST_CBUF);
    int* cbuf_n = reinterpret_cast<int*>(cbuf_col + ST_CBUF);
    float* sthr = reinterpret_cast<float*>(cbuf_n + 4);        // [ST_Q]
    // pair_probes != null (a handful of queries, no queue): the grid is (query, probe) pair x tile of its list -- a few hundred workgroups
    // that all have work.  The unit table of single tiles it replaces is ~40 k workgroups for 10 M rows, of which one query uses 320:
    // handing the others their table entry and list count took most of the 72 us this launch lasted (r04c).
    int4 u;
    int m;
    if (pair_probes) {
        const int pair = unit / pair_tiles;
        const int64_t Lp = pair_probes[pair];
        if (Lp < 0) return;
        u = make_int4((int)Lp, unit - pair * pair_tiles, 1, pair);
        if (u.y >= (int)(tile_off[Lp + 1] - tile_off[Lp])) return;
        m = 1;
    } else {
        u = units[unit];
        m = min(lcount[u.x], cap);
    }
    const int L = u.x, ntiles = u.z;
    if (m == 0) {                                              // nobody probes this list
        if (!queue) return;
        continue;
    }
    const int64_t loff = offsets[L];
    const int len_all = (int)(offsets[L + 1] - loff);
    const int row0 = u.y * LS_ROWS;                            // first row of the unit inside its list
    const int64_t gt0 = tile_off[L] + u.y;                     // ... and its first tile in the copy
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HS = K >> 6;                                     // 64-byte K slices per row (a multiple of ST_SL: the launcher checks)
    const int GPT = HS / ST_SL;                                // groups per tile
    // this lane's 16 bytes of a fragment: row (lane & 15) of a 16-row block, bytes (lane >> 4) * 16 of its slice
    const char* abase = reinterpret_cast<const char*>(scan) + gt0 * tile_stride + wave * (ST_FR * 1024);
    // (StCursor: the wave's fragments behind the end of the list, when the list's partial last tile is this unit's last)
    const int list_tiles = (len_all + LS_ROWS - 1) / LS_ROWS;
    const int pad_nvalid = min(ST_FR, max(0, (len_all - (list_tiles - 1) * LS_ROWS - wave * (ST_FR * 16) + 15) >> 4));
    const char* pad_tile = (u.y + ntiles == list_tiles && pad_nvalid < ST_FR) ? abase + (int64_t)(ntiles - 1) * tile_stride : nullptr;
#ifdef SQE_ST_NATURAL            // (timing build, results wrong: lane-contiguous addresses -- what a fragment-major copy would read)
    const unsigned aoff = (unsigned)(lane * 16);
#else
    const unsigned aoff = (unsigned)((lane & 15) * 64 + (lane >> 4) * 16);
#endif
    // ... and of a query fragment in LDS: row j * 16 + (lane & 15), chunk (h * 4 + (lane >> 4)) ^ ((row >> 1) & 7)
    const int br = lane & 15, bsw = (br >> 1) & 7, bcq = lane >> 4;
    const char* bbase = smem + br * qrow;
    for (int i = tid; i < ntiles * LS_ROWS; i += ST_THREADS) sscale[i] = (float)sxi[gt0 * LS_ROWS + i];

    for (int g0 = 0; g0 < m; g0 += ST_Q) {
        const int gq = min(ST_Q, m - g0);
        __syncthreads();                                       // the previous pass is done with the query block
        if (tid < ST_Q) {
            const int pr = pair_probes ? u.w : lq[(size_t)L * cap + g0 + min(tid, gq - 1)];
            spair[tid] = pr;
            sqscale[tid] = unit2 * (float)sqi[pr / nprobe];
            if (COLLECT) sthr[tid] = tid < gq ? col.thr[pr / nprobe] : INFINITY;
        }
        if (COLLECT && tid == 0) *cbuf_n = 0;
        __syncthreads();
        {
            const int cpr = K >> 4;                            // 16-byte chunks per query row
            for (int c = tid; c < gq * cpr; c += ST_THREADS) {   // (rows >= gq keep stale bytes: their columns are never stored or compared)
                const int r = c / cpr, cc = c - r * cpr;
                const i32x4_t v = *reinterpret_cast<const i32x4_t*>(qb + (size_t)(spair[r] / nprobe) * qpitch + cc * 16);
                *reinterpret_cast<i32x4_t*>(smem + r * qrow + ((cc ^ ((r >> 1) & 7)) << 4)) = v;
            }
        }
        __syncthreads();

        // ---- the stream: group g = slices (g % GPT) * ST_SL .. of tile g / GPT; the ring holds group g while group g + 1 is on its way
        i32x4_t ring[ST_RING];
        StCursor cur{abase, 0, pad_tile, pad_nvalid, reinterpret_cast<const char*>(qb)};
#pragma unroll
        for (int e = 0; e < ST_SL; ++e) st_issue_slice(cur, HS, tile_stride, aoff, &ring[e * ST_FR]);
        i32x4_t acc[ST_FR][2];
#pragma unroll
        for (int i = 0; i < ST_FR; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = i32x4_t{0, 0, 0, 0};
        // Scores of a finished tile.  A lane holds 4 consecutive rows of ONE query per fragment: the wave transposes its 64 rows x gq
        // queries through a private LDS patch, so that 16 lanes hold the 256 contiguous bytes of one query's 64 rows (four queries per
        // store instruction).  WHEN the stores leave matters more than their shape: vector-memory operations retire in issue order, so a
        // row fragment loaded behind a store cannot be consumed before that store has been acknowledged -- a stall of one store latency
        // per tile (measured: 2.03 ms with the stores behind every tile, in 64-byte segments or 256-byte runs alike, 1.61 ms without
        // stores; profiles/r04_configs/ivf_stream_ablation.log).  The patches of up to four tiles therefore wait in LDS (as many as fit:
        // 36 query rows of 64 scores per wave) and leave together -- for the usual handful of probing queries once per unit, behind
        // its last load.
        const int gq_pad = (gq + 3) & ~3;
        const int max_slots = ST_PATCH_ROWS / gq_pad;          // tiles whose patches fit (gq <= 32: at least one)
        int nslot = 0, slot_t0 = 0;
        auto flush = [&]() {
            const int r4 = (lane & 15) * 4;                    // this lane's 4 rows of the wave's 64
            for (int sl = 0; sl < nslot; ++sl) {
                const int wrow = row0 + (slot_t0 + sl) * LS_ROWS + wave * (ST_FR * 16) + r4;   // row inside the list
                const float* patch = spatch + sl * gq_pad * ST_PATCH_PITCH;
                for (int c0 = 0; c0 < gq; c0 += 4) {
                    const int col = c0 + (lane >> 4);
                    if (col < gq && wrow < len_all) {          // (strips are padded to 4 floats: max_len)
                        const f32x4 v = *reinterpret_cast<const f32x4*>(patch + col * ST_PATCH_PITCH + r4);
#ifdef SQE_ST_NO_STORE           // (timing build, results wrong: no strip stores unless a score is NaN)
                        if (v[0] != v[0])
#endif
                        *reinterpret_cast<f32x4*>(pair_scores + (size_t)spair[col] * max_len + wrow) = v;
                    }
                }
            }
            nslot = 0;
        };
        auto write_tile = [&](int t, bool last) {
            if (COLLECT) {
                // keys at or above the query's threshold into the workgroup's buffer (LDS atomics: nothing here enters the vector-memory queue)
                const int trow = row0 + t * LS_ROWS + wave * (ST_FR * 16);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int c = j * 16 + (lane & 15);
                    const float qs = sqscale[c], th = sthr[c];
#pragma unroll
                    for (int i = 0; i < ST_FR; ++i) {
                        const int r = i * 16 + (lane >> 4) * 4;
                        const f32x4 rs = *reinterpret_cast<const f32x4*>(sscale + t * LS_ROWS + wave * (ST_FR * 16) + r);
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (float)acc[i][j][e] * rs[e] * qs;
                        const float m4 = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                        if (__any(m4 >= th)) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (v[e] >= th && trow + r + e < len_all) {
                                    const int at = atomicAdd(cbuf_n, 1);
                                    if (at < ST_CBUF) {
                                        cbuf_key[at] = make_key(v[e], (uint32_t)(trow + r + e));     // (row = position inside the list, for now)
                                        cbuf_col[at] = (unsigned char)c;
                                    } else {
                                        // the buffer is full (a crowd: every row of the unit passes for some query): straight to the list -- the
                                        // atomic's return drains this wave's ring, once in a long while
                                        const int q = spair[c] / nprobe;
                                        const int slot = atomicAdd(col.cnt + q, 1);
                                        if (slot < col.list_cap)
                                            col.list[(size_t)q * col.list_cap + slot] = make_key(v[e], (uint32_t)col.order[loff + trow + r + e]);
                                    }
                                }
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < ST_FR; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = i32x4_t{0, 0, 0, 0};
                return;
            }
            if (nslot == 0) slot_t0 = t;
            float* patch = spatch + nslot * gq_pad * ST_PATCH_PITCH;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = j * 16 + (lane & 15);
                if (col < gq_pad) {
                    const float qs = sqscale[col];
#pragma unroll
                    for (int i = 0; i < ST_FR; ++i) {
                        const int r = i * 16 + (lane >> 4) * 4;    // row inside the wave's 64
                        const f32x4 rs = *reinterpret_cast<const f32x4*>(sscale + t * LS_ROWS + wave * (ST_FR * 16) + r);
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (float)acc[i][j][e] * rs[e] * qs;
                        *reinterpret_cast<f32x4*>(patch + col * ST_PATCH_PITCH + r) = v;
                    }
                }
            }
            // (wave-private patches: the LDS operations of a wave execute in order, no barrier)
            if (++nslot == max_slots || last) flush();
#pragma unroll
            for (int i = 0; i < ST_FR; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = i32x4_t{0, 0, 0, 0};
        };
        const int n_grp = ntiles * GPT;
        int gt = 0, t = 0;                                     // group inside its tile, tile
        for (int g = 0; g + 1 < n_grp; ++g) {
            st_group<true>(ring, acc, bbase + gt * ST_SL * 64, qrow, bcq, bsw, cur, HS, tile_stride, aoff);
            if (++gt == GPT) {
                write_tile(t, false);
                gt = 0;
                ++t;
            }
        }
        st_group<false>(ring, acc, bbase + gt * ST_SL * 64, qrow, bcq, bsw, cur, HS, tile_stride, aoff);
        write_tile(t, true);
        if (COLLECT) {
            __syncthreads();                                   // every wave's keys are in the buffer; no load is in flight any more
            const int n = *cbuf_n;
            const int64_t off = loff;
            for (int i = tid; i < min(n, ST_CBUF); i += ST_THREADS) {
                const uint64_t key = cbuf_key[i];
                const int q = spair[cbuf_col[i]] / nprobe;
                const int slot = atomicAdd(col.cnt + q, 1);
                if (slot < col.list_cap) col.list[(size_t)q * col.list_cap + slot] = make_key(key_score(key), (uint32_t)col.order[off + key_row(key)]);
            }
        }
    }
    if (!queue) return;
    }
}

// Threshold of a query for the collect pass, from the sample strips (the first min(256, len) scores of each of its probed lists): the
// r-th largest sample score, r placed so that ~8 kp rows of the whole probed set are expected at or above it (r = 8 kp x sample
// fraction, plus three standard deviations of that count and 2), and the sample's own entries at or above it start the query's list.
// A probed set small enough to fit the list whole gets -inf.  One workgroup of 1,024 threads per query (r04c; 256 before: the sample
// came in as 32 dependent round trips, one strip after the other -- 36 us for a kernel that moves 32 KiB): 32 threads per strip, the
// eight loads of a thread issued together.
constexpr int IVF_LIST_CAP = 8192;      // keys of a query's list: a whole cluster of near-ties (2,441 rows in SURVEY 8(d)'s set) must fit
constexpr int THR_THREADS = 1024;
__global__ __launch_bounds__(THR_THREADS) void ivf_threshold_kernel(const int64_t* __restrict__ probes, const int64_t* __restrict__ offsets,
                                                            const int* __restrict__ order, const float* __restrict__ pair_scores,
                                                            int nprobe, int max_len, int kp, float* __restrict__ thr_out,
                                                            int* __restrict__ cnt, uint64_t* __restrict__ list) {
    __shared__ uint32_t samp[MAX_KP * 32];                   // up to 256 scores of each of up to 32 probed lists... (nprobe <= 32 here)
    __shared__ int s_len[32], s_pos[33];
    __shared__ int64_t s_off[32];
    __shared__ int hist[256];
    __shared__ int scratch[4];
    const int q = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) { scratch[0] = 0; scratch[2] = 0; }
    if (tid < nprobe) {
        const int64_t L = probes[(size_t)q * nprobe + tid];
        const int64_t off = L >= 0 ? offsets[L] : 0;
        s_off[tid] = off;
        s_len[tid] = L >= 0 ? (int)(offsets[L + 1] - off) : 0;
    }
    __syncthreads();
    if (tid == 0) {
        int pos = 0, total = 0;
        for (int p = 0; p < nprobe; ++p) { s_pos[p] = pos; pos += min(s_len[p], LS_ROWS); total += s_len[p]; }
        s_pos[nprobe] = pos;
        scratch[3] = total;
    }
    __syncthreads();
    const int S = s_pos[nprobe], total = scratch[3];
    {
        const int p = tid >> 5, sub = tid & 31;                // (nprobe <= 32: the host takes this mode only then)
        if (p < nprobe) {
            const int n = s_pos[p + 1] - s_pos[p];
            const float* strip = pair_scores + ((size_t)q * nprobe + p) * max_len;
            float v[LS_ROWS / 32];
#pragma unroll
            for (int j = 0; j < LS_ROWS / 32; ++j) v[j] = sub + 32 * j < n ? strip[sub + 32 * j] : 0.f;
#pragma unroll
            for (int j = 0; j < LS_ROWS / 32; ++j)
                if (sub + 32 * j < n) samp[s_pos[p] + sub + 32 * j] = v[j] == v[j] ? f32_orderable(v[j] + 0.0f) : 0u;      // (NaN rows never rank)
        }
    }
    __syncthreads();
    uint32_t thr = 0;                                          // orderable -inf: everything
    if (total > IVF_LIST_CAP / 4 && S > 0) {
        const float t = 8.0f * (float)kp * (float)S / (float)total;
        const int want = min(S, (int)(t + 3.0f * sqrtf(t) + 2.0f));
        uint32_t pre = 0;
        int rem = want;
        for (int byte = 3; byte >= 0; --byte) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const int shift = byte * 8;
            for (int i = tid; i < S; i += THR_THREADS) {
                const uint32_t v32 = samp[i];
                if (byte == 3 || (v32 >> (shift + 8)) == (pre >> (shift + 8))) atomicAdd(&hist[(v32 >> shift) & 0xff], 1);
            }
            __syncthreads();
            {
                int hb, hr;
                hist_locate(hist, rem, hb, hr);
                if (tid == 0) { scratch[1] = hb < 0 ? 0 : hb; scratch[0] = hr; }
            }
            __syncthreads();
            pre |= ((uint32_t)scratch[1] << shift);
            rem = scratch[0];
            __syncthreads();
        }
        thr = pre;
    }
    // the sample's own entries at or above the threshold open the list
    {
        const int p = tid >> 5, sub = tid & 31;
        if (p < nprobe) {
            const int n = s_pos[p + 1] - s_pos[p];
            for (int i = sub; i < n; i += 32) {
                const uint32_t v32 = samp[s_pos[p] + i];
                if (v32 != 0u && v32 >= thr) {
                    const int slot = atomicAdd(&scratch[2], 1);
                    if (slot < IVF_LIST_CAP) list[(size_t)q * IVF_LIST_CAP + slot] = make_key(f32_from_orderable(v32), (uint32_t)order[s_off[p] + i]);
                }
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        cnt[q] = scratch[2];
        cnt[gridDim.x + q] = total;                            // (ivf_select_list_kernel: a list shorter than min(kp, total) cannot answer)
        thr_out[q] = thr == 0u ? -INFINITY : f32_from_orderable(thr);
    }
}

// per query: the kp best scan scores over the strips of its probed lists, re-scored in fp32 against the
// master, then the top-k by (fp32 cosine desc, row id asc)
// 1,024 threads per query (r04; 256 before): the strip passes and the re-score are chains of dependent loads per wave -- one query
// alone took 83 us here, of a 196 us search (profiles/r04_configs/ivf_stream_ablation.log) -- sixteen waves shorten every chain fourfold.
constexpr int SEL_THREADS = 1024, SEL_WAVES = SEL_THREADS / 64;
// fp32 re-score of the kept rows top[0 .. m) against the master: out[e] = (exact cosine, row).  One wave per row; every thread of the
// block calls (SEL_WAVES waves); out may be top.
__device__ __forceinline__ void rescore_top(const uint64_t* top, uint64_t* out, int m, const float* __restrict__ master,
                                            const float* __restrict__ qrow, int K, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const float4* qv = reinterpret_cast<const float4*>(qrow);
    const int nvec = K >> 2;
    if (K == 1024 && m <= 3 * SEL_WAVES) {
        // the usual shape (kp = 40 rows of 4 KiB, sixteen waves): a wave's up to three rows are random rows of the master, each a full
        // memory round trip (and a TLB miss) -- all their loads are issued before the first sum, one round trip instead of three
        float4 a[3][4], b[4];
        uint32_t rows[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int e = wave + r * SEL_WAVES;
            rows[r] = e < m ? key_row(top[e]) : 0u;
            const float4* rv = reinterpret_cast<const float4*>(master + (size_t)rows[r] * K);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[r][i] = e < m ? rv[lane + 64 * i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = qv[lane + 64 * i];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int e = wave + r * SEL_WAVES;
            float sc = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {          // (the order of the general loop below: same bits)
                sc = fmaf(a[r][i].x, b[i].x, sc); sc = fmaf(a[r][i].y, b[i].y, sc);
                sc = fmaf(a[r][i].z, b[i].z, sc); sc = fmaf(a[r][i].w, b[i].w, sc);
            }
            sc = wave_sum(sc) + 0.0f;
            if (lane == 0 && e < m) out[e] = make_key(sc, rows[r]);
        }
        return;
    }
    for (int e = wave; e < m; e += SEL_WAVES) {
        const uint32_t row = key_row(top[e]);
        const float4* rv = reinterpret_cast<const float4*>(master + (size_t)row * K);
        float sc = 0.f;
        for (int v4 = lane; v4 < nvec; v4 += 64) {
            const float4 a = rv[v4], b = qv[v4];
            sc = fmaf(a.x, b.x, sc); sc = fmaf(a.y, b.y, sc); sc = fmaf(a.z, b.z, sc); sc = fmaf(a.w, b.w, sc);
        }
        sc = wave_sum(sc) + 0.0f;
        if (lane == 0) out[e] = make_key(sc, row);
    }
}

__global__ __launch_bounds__(SEL_THREADS) void ivf_select_kernel(const int64_t* __restrict__ probes, const int64_t* __restrict__ offsets,
                                                         const int* __restrict__ order, const float* __restrict__ pair_scores,
                                                         int nprobe, int max_len, int k, int kp, int64_t id_base,
                                                         const float* __restrict__ master, const float* __restrict__ qn, int K,
                                                         float* __restrict__ cos_out, int64_t* __restrict__ id_out, const int* __restrict__ gate) {
    __shared__ int hist[256];
    __shared__ int scratch[4];
    __shared__ uint64_t top[MAX_KP];
    if (gate && *gate == 0) return;                  // (the strip-mode fallback of a collect search: only if some query asked for it)
    const int q = blockIdx.x, tid = threadIdx.x;
    auto key_at = [&](int p, int i, int64_t off) {
        return make_key(pair_scores[((size_t)q * nprobe + p) * max_len + i], (uint32_t)order[off + i]);
    };
    // the query's probed lists into LDS once (r04: every pass below used to re-read probes / offsets list by list, a chain of
    // dependent loads per list; batch 1024: 476 us for the kernel)
    __shared__ int s_len[MAX_KP];
    __shared__ int64_t s_off[MAX_KP];
    __shared__ int s_total;
    if (tid == 0) s_total = 0;
    __syncthreads();
    for (int p = tid; p < nprobe; p += SEL_THREADS) {
        const int64_t L = probes[(size_t)q * nprobe + p];
        const int64_t off = L >= 0 ? offsets[L] : 0;
        const int len = L >= 0 ? (int)(offsets[L + 1] - off) : 0;
        s_off[p] = off;
        s_len[p] = len;
        if (len) atomicAdd(&s_total, len);
    }
    __syncthreads();
    const int total = s_total;                       // total number of probed rows
    const int wave4 = tid >> 6, lane64 = tid & 63;          // (wave4: one of SEL_WAVES)
    // ---- fast path: a threshold from a strided sample, ONE pass over the strips that collects every key
    // at or above it, exact top-kp among the few collected.  (The general path below walks the strips
    // eight times; it remains the fallback when the sample misjudges the tail.)  One wave per strip in both passes.
    constexpr int SAMPLE_CAP = 8704, COLLECT_CAP = 1024;
    __shared__ uint32_t sample[SAMPLE_CAP];
    __shared__ uint64_t coll[COLLECT_CAP];
    bool done = false;
    if (total > COLLECT_CAP) {
        const int stride = (total + 8191) / 8192;
        if (tid == 0) { scratch[0] = 0; scratch[2] = 0; }
        __syncthreads();
        for (int p = wave4; p < nprobe; p += SEL_WAVES) {
            const int len = s_len[p];
            const float* strip = pair_scores + ((size_t)q * nprobe + p) * max_len;
            // (r04c: four loads of a lane in flight; one at a time the strip came in as a chain of dependent round trips)
            for (int i0 = lane64 * stride + (p % stride); i0 < len; i0 += 4 * 64 * stride) {
                float sc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * 64 * stride;
                    sc[u] = i < len ? strip[i] : __builtin_nanf("");
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (sc[u] != sc[u]) continue;
                    const int slot = atomicAdd(&scratch[0], 1);
                    if (slot < SAMPLE_CAP) sample[slot] = f32_orderable(sc[u] + 0.0f);
                }
            }
        }
        __syncthreads();
        const int ns = min(scratch[0], SAMPLE_CAP);
        // rank in the sample whose value is, with margin, below the kp-th best of the full set
        const float t = (float)kp / (float)stride;
        const int want = (int)(t + 4.0f * sqrtf(t) + 6.0f);
        uint32_t thr = 0;
        if (ns > want) {
            uint32_t pre = 0;
            int rem = want;
            for (int byte = 3; byte >= 0; --byte) {
                if (tid < 256) hist[tid] = 0;
                __syncthreads();
                const int shift = byte * 8;
                for (int i = tid; i < ns; i += SEL_THREADS) {
                    const uint32_t v32 = sample[i];
                    if (byte == 3 || (v32 >> (shift + 8)) == (pre >> (shift + 8))) atomicAdd(&hist[(v32 >> shift) & 0xff], 1);
                }
                __syncthreads();
                {
                    int hb, hr;
                    hist_locate(hist, rem, hb, hr);
                    if (tid == 0) { scratch[1] = hb < 0 ? 0 : hb; scratch[3] = hr; }
                }
                __syncthreads();
                pre |= ((uint32_t)scratch[1] << shift);
                rem = scratch[3];
                __syncthreads();
            }
            thr = pre;
        }
        for (int p = wave4; p < nprobe; p += SEL_WAVES) {
            const int len = s_len[p];
            const int64_t off = s_off[p];
            const float* strip = pair_scores + ((size_t)q * nprobe + p) * max_len;   // (16-byte aligned: max_len is a multiple of 4)
            for (int i0 = lane64 * 4; i0 < len; i0 += 4 * 256) {         // (four 16-byte loads of a lane in flight)
                f32x4 v4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * 256;
                    const f32x4 nan4 = {__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
                    v4[u] = i < len ? *reinterpret_cast<const f32x4*>(strip + i) : nan4;           // (strips are padded to 4 floats)
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * 256;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float sc = v4[u][e];
                        if (i + e < len && sc == sc && f32_orderable(sc + 0.0f) >= thr) {
                            const int slot = atomicAdd(&scratch[2], 1);
                            if (slot < COLLECT_CAP) coll[slot] = make_key(sc, (uint32_t)(off + i + e));      // (position in the lists, for now)
                        }
                    }
                }
            }
        }
        __syncthreads();
        const int nc = scratch[2];
        // positions -> row ids, all collected keys at once (r04c: looked up where a key was found, every hit of a wave was a dependent
        // round trip of its own)
        for (int i = tid; i < min(nc, COLLECT_CAP); i += SEL_THREADS) {
            const uint64_t kpos = coll[i];
            coll[i] = make_key(key_score(kpos), (uint32_t)order[key_row(kpos)]);
        }
        __syncthreads();
        if (nc >= kp && nc <= COLLECT_CAP) {
            for (int i = tid; i < nc; i += SEL_THREADS) {
                const uint64_t ki = coll[i];
                int rank = 0;
                for (int j = 0; j < nc; ++j) rank += coll[j] > ki ? 1 : 0;
                if (rank < kp) top[rank] = ki;
            }
            if (tid == 0) scratch[2] = kp;
            done = true;
        }
        __syncthreads();
    }

    uint64_t prefix = 0;
    int remaining = kp;
    const bool all = total <= kp;
    for (int byte = 7; byte >= 0 && !all && !done; --byte) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const int shift = byte * 8;
        for (int p = 0; p < nprobe; ++p) {
            const int64_t L = probes[(size_t)q * nprobe + p];
            if (L < 0) continue;
            const int64_t off = offsets[L];
            const int len = (int)(offsets[L + 1] - off);
            for (int i = tid; i < len; i += SEL_THREADS) {
                const uint64_t key = key_at(p, i, off);
                const float sc = key_score(key);
                if (sc != sc) continue;                                   // NaN rows never rank
                if (byte == 7 || (key >> (shift + 8)) == (prefix >> (shift + 8)))
                    atomicAdd(&hist[(int)((key >> shift) & 0xff)], 1);
            }
        }
        __syncthreads();
        {
            int hb, hr;
            hist_locate(hist, remaining, hb, hr);
            if (tid == 0) { scratch[0] = hb < 0 ? 0 : hb; scratch[1] = hr; }
        }
        __syncthreads();
        prefix |= ((uint64_t)scratch[0] << shift);
        remaining = scratch[1];
        __syncthreads();
    }
    const uint64_t T = all ? 0ull : prefix;
    if (tid == 0 && !done) scratch[2] = 0;
    __syncthreads();
    for (int p = 0; p < nprobe && !done; ++p) {
        const int64_t L = probes[(size_t)q * nprobe + p];
        if (L < 0) continue;
        const int64_t off = offsets[L];
        const int len = (int)(offsets[L + 1] - off);
        for (int i = tid; i < len; i += SEL_THREADS) {
            const uint64_t key = key_at(p, i, off);
            const float sc = key_score(key);
            if (sc == sc && key >= T) {
                const int slot = atomicAdd(&scratch[2], 1);
                if (slot < MAX_KP) top[slot] = key;
            }
        }
    }
    __syncthreads();
    const int m = min(scratch[2], kp);
    // fp32 re-score of the kept rows (one wave per row), then rank by the exact cosines
    rescore_top(top, top, m, master, qn + (size_t)q * K, K, tid);
    __syncthreads();
    const int mk = min(m, k);
    for (int i = tid; i < m; i += SEL_THREADS) {
        const uint64_t ki = top[i];
        int rank = 0;
        for (int j = 0; j < m; ++j) rank += top[j] > ki ? 1 : 0;
        if (rank < k) {
            cos_out[(size_t)q * k + rank] = key_score(ki);
            id_out[(size_t)q * k + rank] = (int64_t)key_row(ki) + id_base;
        }
    }
    for (int i = mk + tid; i < k; i += SEL_THREADS) {
        cos_out[(size_t)q * k + i] = -INFINITY;
        id_out[(size_t)q * k + i] = -1;
    }
}

// Collect mode: the query's list holds every (estimated score, row) at or above its threshold -- a few hundred to a few thousand keys.
// The kp best by estimate (radix select in LDS) are re-scored in fp32 against the master and ranked, as ivf_select_kernel does for the strips.  A
// list that overflowed, or holds fewer than kp keys although more rows were probed, cannot answer: the query raises the fallback flag
// and the strip-mode pass (gated launches behind this kernel) answers the whole batch.
__global__ __launch_bounds__(SEL_THREADS) void ivf_select_list_kernel(const int* __restrict__ cnt, const int* __restrict__ totals,
                                                                      const uint64_t* __restrict__ list, int k, int kp, int64_t id_base,
                                                                      const float* __restrict__ master, const float* __restrict__ qn, int K,
                                                                      float* __restrict__ cos_out, int64_t* __restrict__ id_out, int* __restrict__ fallback) {
    __shared__ uint64_t keys[IVF_LIST_CAP];
    __shared__ uint64_t top[MAX_KP];
    __shared__ int hist[256];
    __shared__ int scratch[4];
    const int q = blockIdx.x, tid = threadIdx.x;
    const int n = cnt[q], total = totals[q];                  // (total: rows in the query's probed lists, from ivf_threshold_kernel)
    if (n > IVF_LIST_CAP || n < min(kp, total)) {
        if (tid == 0) atomicExch(fallback, 1);
        return;
    }
    for (int i = tid; i < n; i += SEL_THREADS) keys[i] = list[(size_t)q * IVF_LIST_CAP + i];
    if (tid == 0) scratch[2] = 0;
    __syncthreads();
    const int m = min(n, kp);
    // The kp best keys by byte-wise radix select (r04c; rank counting before: n^2 / 1,024 LDS reads per thread -- timing builds without
    // it: 49 -> 8 us for one query, 154 -> 47 us at batch 1024, profiles/r04_configs/ivf_select_list_ablation.log).  Keys are unique
    // (score | ~row), so the pass after which the located bin holds exactly the keys still wanted ends the search: three or four
    // passes of the eight on scores that differ.
    uint64_t T = 0;
    if (n > kp) {
        uint64_t prefix = 0;
        int remaining = kp;
        for (int byte = 7; byte >= 0; --byte) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const int shift = byte * 8;
            for (int i = tid; i < n; i += SEL_THREADS) {
                const uint64_t key = keys[i];
                if (byte == 7 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(int)((key >> shift) & 0xff)], 1);
            }
            __syncthreads();
            {
                int hb, hr;
                hist_locate(hist, remaining, hb, hr);
                if (tid == 0) { scratch[0] = hb < 0 ? 0 : hb; scratch[1] = hr; scratch[3] = hb < 0 ? 0 : hist[hb]; }
            }
            __syncthreads();
            prefix |= ((uint64_t)scratch[0] << shift);
            remaining = scratch[1];
            const bool whole_bin = scratch[3] == remaining;   // every key of the bin is wanted: keys >= prefix (lower bytes 0) are the kp best
            __syncthreads();
            if (whole_bin) break;
        }
        T = prefix;
    }
    for (int i = tid; i < n; i += SEL_THREADS) {
        const uint64_t key = keys[i];
        if (key >= T) {
            const int slot = atomicAdd(&scratch[2], 1);
            if (slot < MAX_KP) top[slot] = key;
        }
    }
    __syncthreads();
    rescore_top(top, keys, m, master, qn + (size_t)q * K, K, tid);          // (keys[] is free: the kp best are in top[])
    __syncthreads();
    const int mk = min(m, k);
    for (int i = tid; i < m; i += SEL_THREADS) {
        const uint64_t ki = keys[i];
        int rank = 0;
        for (int j = 0; j < m; ++j) rank += keys[j] > ki ? 1 : 0;
        if (rank < k) {
            cos_out[(size_t)q * k + rank] = key_score(ki);
            id_out[(size_t)q * k + rank] = (int64_t)key_row(ki) + id_base;
        }
    }
    for (int i = mk + tid; i < k; i += SEL_THREADS) {
        cos_out[(size_t)q * k + i] = -INFINITY;
        id_out[(size_t)q * k + i] = -1;
    }
}

uint64_t splitmix(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

}  // namespace

struct IvfState {
    sqe_index* coarse = nullptr;       // flat index over the centroids
    bool trained = false;
    bool lists_dirty = true;
    bool cent_dirty = true;            // dense bf16 copy of the centroids (coarse GEMM) is stale
    int64_t n_assigned = 0;            // rows of the base index that have an assignment
    int max_len = 0;
    Buf centroids, assign, order, offsets, counts, cursor;   // device
    // int8 list scan (r03): int8 copy of the base rows IN LIST ORDER at i8_pitch bytes + per-row scales, (re)built lazily
    Buf i8rows, i8sxi, q8, q8sqi, tile_off, dpos;
    int64_t i8_cap = 0, i8_done = 0, total_tiles = 0;    // i8_cap: tiles allocated
    int64_t i8_tile_stride = 0;
    bool use_i8 = true;              // knobs build: SQE_IVF_I8=0 keeps the bf16 list scan (A/B)
    Buf units4, units1;              // work units of the streaming list scan: (list, first tile in the list, tiles, 0), <= ST_UNIT_TILES tiles / 1 tile each
    int n_units4 = 0, n_units1 = 0;
    Buf unitsS, unitsR;              // collect mode: the first tile of every list (the sample) / the other tiles in runs of <= 4
    int n_unitsS = 0, n_unitsR = 0;
    Buf cthr, ccnt, clist;           // collect mode, per search: thresholds [B], list lengths [B], lists [B, IVF_LIST_CAP]
    Buf qn, qb, qd, cent_bf16, cscores, probes_cos, probes_ids, lcount, lq, pair_scores, tmp_ids, tmp_cos, sums;
    std::vector<int64_t> h_offsets;
};

// Top-kk lists of b rows (raw fp32, normalised here) against the centroids.  A few thousand centroids are too
// few rows for the streaming scan (its chunks could not even fill the global-bound table), so when nlist
// allows it the whole [b, nlist] score matrix is one small GEMM (dense bf16 copies of both sides, fp32 out,
// the encoder's ring GEMM) followed by a per-row radix select; otherwise the flat index over the centroids.
static int ivf_coarse_topk(sqe_index* base, IvfState* st, const float* rows_dev, int64_t b, int kk, int64_t* ids_out,
                           float* cos_out, hipStream_t s) {
    sqe_ctx* ctx = base->ctx;
    const int nlist = base->nlist, dim = base->dim;
    const bool dense = nlist % 128 == 0 && (size_t)nlist * 4 <= 64 * 1024 && kk + 8 <= MAX_KP;
    if (!dense) return index_search_impl(st->coarse, rows_dev, (int)b, kk, 0, cos_out, ids_out, s);
    if (st->cent_dirty) {
        SQE_TRY(st->cent_bf16.ensure((size_t)nlist * dim * 2));
        SQE_TRY(launch_normalize_rows(st->coarse->master, nlist, dim, dim, nullptr, st->cent_bf16.as<bf16_t>(), dim, nullptr, nullptr, s));
        st->cent_dirty = false;
    }
    SQE_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(ivf_probe_select_kernel), 64 * 1024));
    const int64_t step = 16384;
    const int64_t cap = std::min(step, (b + 127) / 128 * 128);
    SQE_TRY(st->qd.ensure((size_t)cap * dim * 2));
    SQE_TRY(st->cscores.ensure((size_t)cap * nlist * 4));
    for (int64_t off = 0; off < b; off += step) {
        const int m = (int)std::min(step, b - off);
        const int t_pad = (m + 127) / 128 * 128;
        // (rows m .. t_pad of qd keep whatever they held: the GEMM computes their scores and stores none of them)
        SQE_TRY(launch_normalize_rows(rows_dev + (size_t)off * dim, m, dim, dim, nullptr, st->qd.as<bf16_t>(), dim, nullptr, nullptr, s));
        SQE_TRY(launch_scores_gemm(st->cent_bf16.as<bf16_t>(), st->qd.as<bf16_t>(), st->cscores.as<float>(), nlist, dim, m, t_pad,
                                   ctx->cu_count, s));
        hipLaunchKernelGGL(ivf_probe_select_kernel, dim3(m), dim3(PSEL_THREADS), (size_t)nlist * 4, s, st->cscores.as<float>(), nlist, kk,
                           rows_dev + (size_t)off * dim, st->coarse->master, dim, ids_out + off * kk, cos_out + off * kk);
        SQE_HIP(hipGetLastError());
    }
    return SQE_OK;
}

static int ivf_assign_rows(sqe_index* base, IvfState* st, const float* rows_dev, int64_t n, int* assign_out,
                           int64_t* ids64_out /* optional [n] */, hipStream_t s) {
    // cosine top-1 of `rows_dev` against the centroids, in batches
    const int dim = base->dim;
    const int64_t batch = 65536;
    SQE_TRY(st->tmp_ids.ensure((size_t)std::min(batch, n) * 8));
    SQE_TRY(st->tmp_cos.ensure((size_t)std::min(batch, n) * 4));
    for (int64_t off = 0; off < n; off += batch) {
        const int b = (int)std::min(batch, n - off);
        SQE_TRY(ivf_coarse_topk(base, st, rows_dev + (size_t)off * dim, b, 1, st->tmp_ids.as<int64_t>(), st->tmp_cos.as<float>(), s));
        if (assign_out)
            hipLaunchKernelGGL(ivf_store_assign_kernel, dim3((b + 255) / 256), dim3(256), 0, s, st->tmp_ids.as<int64_t>(),
                               (int64_t)b, assign_out + off, (int*)nullptr);
        if (ids64_out)
            SQE_HIP(hipMemcpyAsync(ids64_out + off, st->tmp_ids.p, (size_t)b * 8, hipMemcpyDeviceToDevice, s));
        SQE_HIP(hipGetLastError());
    }
    return SQE_OK;
}

int ivf_create(sqe_index* base, IvfState** out) {
    IvfState* st = new (std::nothrow) IvfState;   // (struct defined above)
    if (!st) return fail(SQE_ERR_OOM, "ivf: host allocation failed");
    int rc = index_create_impl(base->ctx, base->dim, SQE_INDEX_FLAT, 0, true, &st->coarse);
    if (rc != SQE_OK) { delete st; return rc; }
    *out = st;
    return SQE_OK;
}

void ivf_destroy(IvfState* st) {
    if (!st) return;
    if (st->coarse) sqe_index_destroy(st->coarse);
    delete st;
}

int ivf_rows_added(sqe_index* base, IvfState* st, hipStream_t s) {
    // assign rows [n_assigned, rows) once the centroids exist
    if (!st->trained) return SQE_OK;
    const int64_t n = base->n.load();
    if (n <= st->n_assigned) return SQE_OK;
    if ((size_t)n * 4 > st->assign.bytes) {
        // grows geometrically: an index fed 64 rows at a time must not copy the whole array on every add
        Buf grown;
        SQE_TRY(grown.ensure((size_t)std::max<int64_t>(n, (int64_t)(st->assign.bytes / 4) * 3 / 2) * 4));
        if (st->n_assigned > 0)
            SQE_HIP(hipMemcpyAsync(grown.p, st->assign.p, (size_t)st->n_assigned * 4, hipMemcpyDeviceToDevice, s));
        SQE_HIP(hipStreamSynchronize(s));
        std::swap(grown.p, st->assign.p);
        std::swap(grown.bytes, st->assign.bytes);
    }
    SQE_TRY(ivf_assign_rows(base, st, base->master + (size_t)st->n_assigned * base->dim, n - st->n_assigned,
                            st->assign.as<int>() + st->n_assigned, nullptr, s));
    st->n_assigned = n;
    st->lists_dirty = true;
    return SQE_OK;
}

// sqe_index_update overwrote `n` stored rows (ids on the device): only those are re-assigned
int ivf_rows_updated(sqe_index* base, IvfState* st, const int64_t* rows_dev, int64_t n, hipStream_t s) {
    if (!st->trained || n <= 0) return SQE_OK;
    SQE_TRY(ivf_rows_added(base, st, s));                 // rows appended since the last search get their list first
    const int dim = base->dim;
    Buf rows, lists;
    SQE_TRY(rows.ensure((size_t)n * dim * 4));
    SQE_TRY(lists.ensure((size_t)n * 4));
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, base->master, rows_dev, (int)n, dim,
                       rows.as<float>());
    SQE_HIP(hipGetLastError());
    SQE_TRY(ivf_assign_rows(base, st, rows.as<float>(), n, lists.as<int>(), nullptr, s));
    hipLaunchKernelGGL(ivf_scatter_assign_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, lists.as<int>(), rows_dev, n,
                       st->assign.as<int>());
    SQE_HIP(hipGetLastError());
    SQE_HIP(hipStreamSynchronize(s));                     // the temporaries die here
    st->lists_dirty = true;
    st->i8_done = 0;                                      // the int8 copy is rebuilt from the master by the next search
    return SQE_OK;
}

static int ivf_build_lists(sqe_index* base, IvfState* st, hipStream_t s) {
    const int nlist = base->nlist;
    const int64_t n = st->n_assigned;
    SQE_TRY(st->counts.ensure((size_t)nlist * 4));
    SQE_TRY(st->cursor.ensure((size_t)nlist * 4));
    SQE_TRY(st->offsets.ensure((size_t)(nlist + 1) * 8));
    SQE_TRY(st->order.ensure((size_t)std::max<int64_t>(n, 1) * 4));
    SQE_HIP(hipMemsetAsync(st->counts.p, 0, (size_t)nlist * 4, s));
    SQE_HIP(hipMemsetAsync(st->cursor.p, 0, (size_t)nlist * 4, s));
    if (n > 0) hipLaunchKernelGGL(ivf_count_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, st->assign.as<int>(), n, st->counts.as<int>());
    std::vector<int> h_counts(nlist);
    SQE_HIP(hipMemcpyAsync(h_counts.data(), st->counts.p, (size_t)nlist * 4, hipMemcpyDeviceToHost, s));
    SQE_HIP(hipStreamSynchronize(s));
    st->h_offsets.assign(nlist + 1, 0);
    st->max_len = 0;
    for (int i = 0; i < nlist; ++i) {
        st->h_offsets[i + 1] = st->h_offsets[i] + h_counts[i];
        st->max_len = std::max(st->max_len, h_counts[i]);
    }
    SQE_HIP(hipMemcpyAsync(st->offsets.p, st->h_offsets.data(), (size_t)(nlist + 1) * 8, hipMemcpyHostToDevice, s));
    std::vector<int64_t> h_tile_off(nlist + 1, 0);
    for (int i = 0; i < nlist; ++i) h_tile_off[i + 1] = h_tile_off[i] + (h_counts[i] + 255) / 256;
    st->total_tiles = h_tile_off[nlist];
    if (st->total_tiles * 256 > 0x7fffffffLL) return fail(SQE_ERR_INVALID, "ivf: too many rows for one index shard");
    SQE_TRY(st->tile_off.ensure((size_t)(nlist + 1) * 8));
    SQE_TRY(st->dpos.ensure((size_t)std::max<int64_t>(n, 1) * 4));
    SQE_HIP(hipMemcpyAsync(st->tile_off.p, h_tile_off.data(), (size_t)(nlist + 1) * 8, hipMemcpyHostToDevice, s));
    if (n > 0) hipLaunchKernelGGL(ivf_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, st->assign.as<int>(), n,
                                  st->offsets.as<int64_t>(), st->tile_off.as<int64_t>(), st->cursor.as<int>(), st->order.as<int>(),
                                  st->dpos.as<int>());
    SQE_HIP(hipGetLastError());
    {
        // work units of the streaming list scan (ivf_list_stream_i8_kernel): runs of <= ST_UNIT_TILES tiles of one list, and single tiles
        // (the grid of a search with a handful of queries, where few lists are probed and every CU should get some of them)
        std::vector<int4> u4, u1, uS, uR;
        for (int i = 0; i < nlist; ++i) {
            const int nt = (h_counts[i] + 255) / 256;
            // (tiles [lo, nt) of a list in nearly equal runs of <= ST_UNIT_TILES: a unit pays its query load and the fill and drain of
            // its register ring once, so 9 tiles are 5 + 4, not 4 + 4 + 1)
            auto split = [&](std::vector<int4>& out, int lo) {
                const int rem = nt - lo;
                if (rem <= 0) return;
                const int nu = (rem + ST_UNIT_TILES - 1) / ST_UNIT_TILES;
                for (int u = 0, t0 = lo; u < nu; ++u) {
                    const int len = rem / nu + (u < rem % nu ? 1 : 0);
                    out.push_back(make_int4(i, t0, len, 0));
                    t0 += len;
                }
            };
            split(u4, 0);
            for (int t0 = 0; t0 < nt; ++t0) u1.push_back(make_int4(i, t0, 1, 0));
            if (nt > 0) uS.push_back(make_int4(i, 0, 1, 0));
            split(uR, 1);
        }
        // the unit queue hands units out in table order: longest first, so that the launch ends on its shortest units
        auto longer = [](const int4& a, const int4& b) { return a.z > b.z; };
        std::stable_sort(u4.begin(), u4.end(), longer);
        std::stable_sort(uR.begin(), uR.end(), longer);
        st->n_units4 = (int)u4.size(); st->n_units1 = (int)u1.size(); st->n_unitsS = (int)uS.size(); st->n_unitsR = (int)uR.size();
        SQE_TRY(st->units4.ensure(std::max<size_t>(1, u4.size()) * sizeof(int4)));
        SQE_TRY(st->units1.ensure(std::max<size_t>(1, u1.size()) * sizeof(int4)));
        SQE_TRY(st->unitsS.ensure(std::max<size_t>(1, uS.size()) * sizeof(int4)));
        SQE_TRY(st->unitsR.ensure(std::max<size_t>(1, uR.size()) * sizeof(int4)));
        if (!u4.empty()) SQE_HIP(hipMemcpyAsync(st->units4.p, u4.data(), u4.size() * sizeof(int4), hipMemcpyHostToDevice, s));
        if (!u1.empty()) SQE_HIP(hipMemcpyAsync(st->units1.p, u1.data(), u1.size() * sizeof(int4), hipMemcpyHostToDevice, s));
        if (!uS.empty()) SQE_HIP(hipMemcpyAsync(st->unitsS.p, uS.data(), uS.size() * sizeof(int4), hipMemcpyHostToDevice, s));
        if (!uR.empty()) SQE_HIP(hipMemcpyAsync(st->unitsR.p, uR.data(), uR.size() * sizeof(int4), hipMemcpyHostToDevice, s));
        SQE_HIP(hipStreamSynchronize(s));                 // (host vectors)
    }
    SQE_HIP(hipStreamSynchronize(s));
    st->lists_dirty = false;
    st->i8_done = 0;                                      // the int8 copy follows the list order
    return SQE_OK;
}

int ivf_train(sqe_index* base, IvfState* st, const float* x_dev, int64_t n, int iters, uint64_t seed, hipStream_t s) {
    const int nlist = base->nlist, dim = base->dim;
    if (n < nlist) return fail(SQE_ERR_INVALID, "sqe_index_train: need at least nlist training rows");
    if (iters < 1) iters = 1;
    // normalised copy of the sample
    Buf xs, pick, assign64;
    SQE_TRY(xs.ensure((size_t)n * dim * 4));
    SQE_TRY(launch_normalize_rows(x_dev, n, dim, dim, xs.as<float>(), nullptr, dim, nullptr, nullptr, s));
    // initial centroids: nlist distinct rows of the sample (seeded partial Fisher-Yates)
    std::vector<int64_t> perm(n);
    for (int64_t i = 0; i < n; ++i) perm[i] = i;
    uint64_t rs = seed ^ 0x5eed5eedULL;
    for (int i = 0; i < nlist; ++i) std::swap(perm[i], perm[i + (int64_t)(splitmix(rs) % (uint64_t)(n - i))]);
    SQE_TRY(pick.ensure((size_t)nlist * 8));
    SQE_HIP(hipMemcpyAsync(pick.p, perm.data(), (size_t)nlist * 8, hipMemcpyHostToDevice, s));
    SQE_TRY(st->centroids.ensure((size_t)nlist * dim * 4));
    hipLaunchKernelGGL(gather_rows_kernel, dim3((nlist + 3) / 4), dim3(256), 0, s, xs.as<float>(), pick.as<int64_t>(), nlist, dim,
                       st->centroids.as<float>());
    SQE_HIP(hipGetLastError());
    SQE_HIP(hipStreamSynchronize(s));      // perm is a host buffer
    SQE_TRY(assign64.ensure((size_t)n * 8));
    SQE_TRY(st->sums.ensure((size_t)nlist * dim * 4));
    SQE_TRY(st->counts.ensure((size_t)nlist * 4));
    for (int it = 0; it < iters; ++it) {
        st->coarse->n.store(0);
        SQE_TRY(index_add_impl(st->coarse, st->centroids.as<float>(), nlist, dim, false, s));
        st->cent_dirty = true;
        SQE_TRY(ivf_assign_rows(base, st, xs.as<float>(), n, nullptr, assign64.as<int64_t>(), s));
        SQE_HIP(hipMemsetAsync(st->sums.p, 0, (size_t)nlist * dim * 4, s));
        SQE_HIP(hipMemsetAsync(st->counts.p, 0, (size_t)nlist * 4, s));
        hipLaunchKernelGGL(kmeans_accum_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, xs.as<float>(), assign64.as<int64_t>(), n,
                           dim, st->sums.as<float>(), st->counts.as<int>());
        hipLaunchKernelGGL(kmeans_finish_kernel, dim3((nlist + 3) / 4), dim3(256), 0, s, st->centroids.as<float>(), st->sums.as<float>(),
                           st->counts.as<int>(), nlist, dim);
        SQE_HIP(hipGetLastError());
    }
    st->coarse->n.store(0);
    SQE_TRY(index_add_impl(st->coarse, st->centroids.as<float>(), nlist, dim, false, s));
    SQE_HIP(hipStreamSynchronize(s));
    st->trained = true;
    st->cent_dirty = true;
    st->n_assigned = 0;                    // (re)assign everything stored so far
    st->lists_dirty = true;
    return ivf_rows_added(base, st, s);
}

static bool ivf_i8_off() {
    static const bool v = [] { const char* e = knob_env("SQE_IVF_I8"); return e && e[0] == '0'; }();   // knobs build only
    return v;
}

int ivf_search(sqe_index* base, IvfState* st, const float* q_dev, int B, int k, int nprobe, float* cos_out, int64_t* id_out,
               hipStream_t s) {
    if (!st->trained) return fail(SQE_ERR_STATE, "sqe_index_search: IVF index is not trained (sqe_index_train)");
    const int nlist = base->nlist, dim = base->dim;
    if (nprobe <= 0) nprobe = 32;
    nprobe = std::min(std::min(nprobe, nlist), MAX_KP);
    SQE_TRY(ivf_rows_added(base, st, s));
    if (st->lists_dirty) SQE_TRY(ivf_build_lists(base, st, s));
    const int max_len = (std::max(st->max_len, 1) + 3) / 4 * 4;        // strips are written 4 floats at a time
    // The score strips are [queries, nprobe, longest list] floats.  One over-long list (duplicate-heavy or
    // tightly clustered data) must not turn that into terabytes: the batch is cut into sub-batches whose
    // strips fit a fixed budget, each a complete search of its queries.
    constexpr size_t STRIP_BUDGET = 6ull << 30;
    const size_t per_query = (size_t)nprobe * max_len * 4;
    const int sub = (int)std::max<size_t>(1, std::min<size_t>((size_t)B, STRIP_BUDGET / per_query));
    if (sub < B) {
        for (int off = 0; off < B; off += sub) {
            const int m = std::min(sub, B - off);
            SQE_TRY(ivf_search(base, st, q_dev + (size_t)off * dim, m, k, nprobe, cos_out + (size_t)off * k, id_out + (size_t)off * k, s));
        }
        return SQE_OK;
    }
    const int pitch = base->pitch;
    const int kp = std::min(MAX_KP, std::max(32, 4 * k));
    SQE_TRY(st->qn.ensure((size_t)B * dim * 4));
    SQE_TRY(st->qb.ensure((size_t)(B + LS_Q) * pitch));
    SQE_TRY(st->probes_cos.ensure((size_t)B * nprobe * 4));
    SQE_TRY(st->probes_ids.ensure((size_t)B * nprobe * 8));
    SQE_TRY(st->lcount.ensure((size_t)(nlist + 4) * 4));      // (+ 4 ints behind the list counts: the collect mode's flag and unit counters)
    SQE_TRY(st->lq.ensure((size_t)nlist * B * 4));
    SQE_TRY(st->pair_scores.ensure((size_t)B * per_query));
    SQE_TRY(launch_normalize_rows(q_dev, B, dim, dim, st->qn.as<float>(), st->qb.as<bf16_t>(), pitch / 2, nullptr, nullptr, s));
    // S5: coarse quantise
    SQE_TRY(ivf_coarse_topk(base, st, q_dev, B, nprobe, st->probes_ids.as<int64_t>(), st->probes_cos.as<float>(), s));
    // S6: list scan
    static const bool fp32_lists = [] { const char* e = knob_env("SQE_IVF_FP32"); return e && e[0] == '1'; }();   // knobs build only
    static const bool staged = [] { const char* e = knob_env("SQE_IVF_STAGED"); return e && e[0] == '1'; }();   // knobs build: the r03 kernel, for A/B
    static const bool unit_table = [] { const char* e = knob_env("SQE_IVF_FEW_TABLE"); return e && e[0] == '1'; }();   // knobs build: the single-tile unit table for a handful of queries, for A/B
    const size_t st_lds = (size_t)ST_Q * (dim + 128) + ST_UNIT_TILES * LS_ROWS * 4 + ST_Q * 8 + (ST_THREADS / 64) * ST_PATCH_ROWS * ST_PATCH_PITCH * 4;
    const bool i8_lists = !fp32_lists && st->use_i8 && dim >= 256 && dim % 128 == 0 && !ivf_i8_off();
    const bool streaming = i8_lists && !staged && dim % (64 * ST_SL) == 0 && st_lds <= 80 * 1024 && st->n_units4 > 0;      // (two workgroups per CU)
    // a handful of queries through the streaming scan: one workgroup per (query, probe) pair and tile -- no per-list query buckets needed
    // (... unless one list is so long that a grid of pairs x tiles-of-the-longest-list would be mostly empty: the unit table then)
    const bool pair_grid = streaming && B * nprobe <= 512 && !unit_table &&
                           (int64_t)B * nprobe * ((max_len + LS_ROWS - 1) / LS_ROWS) <= std::max<int64_t>(8192, 2 * (int64_t)st->n_units1);
    if (!pair_grid) {
        SQE_HIP(hipMemsetAsync(st->lcount.p, 0, (size_t)(nlist + 4) * 4, s));
        hipLaunchKernelGGL(ivf_bucket_kernel, dim3((B * nprobe + 255) / 256), dim3(256), 0, s, st->probes_ids.as<int64_t>(), B, nprobe,
                           st->lcount.as<int>(), st->lq.as<int>(), B);
    }
    if (fp32_lists) {
        hipLaunchKernelGGL(ivf_list_scan_kernel, dim3(nlist), dim3(256), 0, s, base->master, st->qn.as<float>(), st->order.as<int>(),
                           st->offsets.as<int64_t>(), st->lcount.as<int>(), st->lq.as<int>(), B, nprobe, dim, max_len,
                           st->pair_scores.as<float>());
    } else if (i8_lists) {
        // ---- int8 list scan: half the bytes per probed row, every list a run of whole 256-row tiles in the flat scan's tiled
        // layout (a K step of a tile is 32 contiguous KiB).  The copy is in list order, so it is rebuilt from the master whenever
        // the lists were (rows added, rows overwritten): ivf_build_lists / ivf_rows_updated reset i8_done.  ~12 ms per 10 M rows,
        // on the first search after the change.
        const int64_t n = st->n_assigned;
        const int p8 = dim + 128;                          // (pitch of the quantised QUERY rows)
        const int64_t tile_stride = (int64_t)(dim / 64) * 16384 + 2048;
        if (st->i8_cap < st->total_tiles || st->i8_tile_stride != tile_stride) {
            const int64_t cap = std::max<int64_t>(st->total_tiles, (base->cap + 255) / 256 + nlist);
            SQE_TRY(st->i8rows.ensure((size_t)cap * tile_stride));
            SQE_TRY(st->i8sxi.ensure((size_t)cap * 256 * 4));
            st->i8_cap = cap;
            st->i8_tile_stride = tile_stride;
            st->i8_done = 0;
        }
        if (st->i8_done != n) {
            SQE_TRY(launch_quantize_gather_i8(base->master, st->order.as<int>(), st->dpos.as<int>(), n, dim, st->i8rows.as<int8_t>(), tile_stride,
                                              st->i8sxi.as<uint32_t>(), s));
            st->i8_done = n;
        }
        SQE_TRY(st->q8.ensure((size_t)(B + LS_Q) * p8));
        SQE_TRY(st->q8sqi.ensure((size_t)(B + LS_Q) * 4));
        SQE_TRY(launch_quantize_queries_i8(st->qn.as<float>(), B, dim, st->q8.as<int8_t>(), p8, st->q8sqi.as<uint32_t>(), nullptr, s));
        const float unit = i8_scale_unit(dim);
        if (streaming) {
            // streaming form: one workgroup per unit of <= 4 tiles (single tiles when only a handful of lists are probed)
            const bool few = B * nprobe <= 512;
            static const bool strips_only = [] { const char* e = knob_env("SQE_IVF_STRIPS"); return e && e[0] == '1'; }();   // knobs build: r04a's form, for A/B
            auto strips = ivf_list_stream_i8_kernel<false>;
            auto collect = ivf_list_stream_i8_kernel<true>;
            SQE_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(strips), (int)st_lds));
            SQE_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(collect), (int)st_lds));
            static const bool no_queue = [] { const char* e = knob_env("SQE_IVF_QUEUE"); return e && e[0] == '0'; }();   // knobs build: one workgroup per unit, for A/B
            const int persistent = 2 * base->ctx->cu_count;                  // two workgroups per CU
            auto launch_strips = [&](const Buf& units, int n_units, const int* gate, int* queue, const int64_t* pair_probes = nullptr, int pair_tiles = 0) {
                if (no_queue || n_units <= persistent) queue = nullptr;
                hipLaunchKernelGGL(strips, dim3(queue ? persistent : n_units), dim3(ST_THREADS), st_lds, s, st->i8rows.as<int8_t>(), tile_stride,
                                   st->i8sxi.as<uint32_t>(), st->q8.as<int8_t>(), p8, st->q8sqi.as<uint32_t>(), unit * unit, units.as<int4>(),
                                   st->tile_off.as<int64_t>(), st->offsets.as<int64_t>(), st->lcount.as<int>(), st->lq.as<int>(), B, nprobe, dim, max_len,
                                   st->pair_scores.as<float>(), StCollect{}, gate, queue, n_units, pair_probes, pair_tiles);
            };
            if (!few && !strips_only && nprobe <= 32 && st->n_unitsR > 0) {
                // ---- collect mode (r04b): sample pass over the first tile of every list -> per-query thresholds -> the other tiles keep only
                // the keys at or above them -> the lists are ranked.  Any query whose list cannot answer sets the flag, and the two gated
                // launches at the end redo the batch through the strips (they return at once otherwise).
                SQE_TRY(st->cthr.ensure((size_t)B * 4));
                SQE_TRY(st->ccnt.ensure((size_t)B * 8));      // (list lengths [B], rows in the probed lists [B])
                SQE_TRY(st->clist.ensure((size_t)B * IVF_LIST_CAP * 8));
                int* cflag = st->lcount.as<int>() + nlist;      // [0] fallback flag, [1..3] unit counters: zeroed with the list counts above
                launch_strips(st->unitsS, st->n_unitsS, nullptr, cflag + 1);
                hipLaunchKernelGGL(ivf_threshold_kernel, dim3(B), dim3(THR_THREADS), 0, s, st->probes_ids.as<int64_t>(), st->offsets.as<int64_t>(),
                                   st->order.as<int>(), st->pair_scores.as<float>(), nprobe, max_len, kp, st->cthr.as<float>(), st->ccnt.as<int>(),
                                   st->clist.as<uint64_t>());
                StCollect col{st->cthr.as<float>(), st->ccnt.as<int>(), st->clist.as<uint64_t>(), IVF_LIST_CAP, st->order.as<int>()};
                int* cqueue = (no_queue || st->n_unitsR <= persistent) ? nullptr : cflag + 2;
                hipLaunchKernelGGL(collect, dim3(cqueue ? persistent : st->n_unitsR), dim3(ST_THREADS), st_lds, s, st->i8rows.as<int8_t>(), tile_stride,
                                   st->i8sxi.as<uint32_t>(), st->q8.as<int8_t>(), p8, st->q8sqi.as<uint32_t>(), unit * unit, st->unitsR.as<int4>(),
                                   st->tile_off.as<int64_t>(), st->offsets.as<int64_t>(), st->lcount.as<int>(), st->lq.as<int>(), B, nprobe, dim, max_len,
                                   st->pair_scores.as<float>(), col, (const int*)nullptr, cqueue, st->n_unitsR, (const int64_t*)nullptr, 0);
                hipLaunchKernelGGL(ivf_select_list_kernel, dim3(B), dim3(SEL_THREADS), 0, s, st->ccnt.as<int>(), st->ccnt.as<int>() + B,
                                   st->clist.as<uint64_t>(), k, kp, base->id_base, base->master, st->qn.as<float>(), dim,
                                   cos_out, id_out, cflag);
                launch_strips(st->units4, st->n_units4, cflag, cflag + 3);
                hipLaunchKernelGGL(ivf_select_kernel, dim3(B), dim3(SEL_THREADS), 0, s, st->probes_ids.as<int64_t>(), st->offsets.as<int64_t>(),
                                   st->order.as<int>(), st->pair_scores.as<float>(), nprobe, max_len, k, kp, base->id_base,
                                   base->master, st->qn.as<float>(), dim, cos_out, id_out, cflag);
                SQE_HIP(hipGetLastError());
                return SQE_OK;
            }
            if (few) {
                // a handful of queries: one workgroup per (query, probe) pair and tile of its list (the lists of different queries are read
                // separately; at most 512 pairs)
                const int pair_tiles = (max_len + LS_ROWS - 1) / LS_ROWS;
                if (!pair_grid) launch_strips(st->units1, st->n_units1, nullptr, nullptr);
                else launch_strips(st->units1, B * nprobe * pair_tiles, nullptr, nullptr, st->probes_ids.as<int64_t>(), pair_tiles);
            } else {
                launch_strips(st->units4, st->n_units4, nullptr, nullptr);
            }
        } else {
        SQE_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(ivf_list_scan_i8_kernel), LS_LDS_I8));
        // a handful of queries: pair mode (the kernel's comment), up to 16 workgroups per probed list
        const int n_pairs = B * nprobe <= 512 ? B * nprobe : 0;
        const int split = n_pairs ? std::max(1, std::min(16, 512 / n_pairs)) : 1;
        hipLaunchKernelGGL(ivf_list_scan_i8_kernel, dim3(n_pairs ? n_pairs * split : nlist), dim3(512), LS_LDS_I8, s, st->i8rows.as<int8_t>(),
                           tile_stride, st->i8sxi.as<uint32_t>(), st->q8.as<int8_t>(), p8, st->q8sqi.as<uint32_t>(), unit * unit,
                           st->tile_off.as<int64_t>(), st->offsets.as<int64_t>(), st->lcount.as<int>(), st->lq.as<int>(), B, nprobe, dim, max_len,
                           st->pair_scores.as<float>(), st->probes_ids.as<int64_t>(), n_pairs, split);
        }
    } else {
        SQE_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(ivf_list_scan_mfma_kernel), LS_LDS));
        hipLaunchKernelGGL(ivf_list_scan_mfma_kernel, dim3(nlist), dim3(512), LS_LDS, s, base->scan, pitch, st->qb.as<bf16_t>(),
                           pitch, st->order.as<int>(), st->offsets.as<int64_t>(), st->lcount.as<int>(), st->lq.as<int>(), B, nprobe,
                           dim, max_len, st->pair_scores.as<float>());
    }
    hipLaunchKernelGGL(ivf_select_kernel, dim3(B), dim3(SEL_THREADS), 0, s, st->probes_ids.as<int64_t>(), st->offsets.as<int64_t>(),
                       st->order.as<int>(), st->pair_scores.as<float>(), nprobe, max_len, k, kp, base->id_base,
                       base->master, st->qn.as<float>(), dim, cos_out, id_out, (const int*)nullptr);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

void ivf_invalidate(IvfState* st) { st->n_assigned = 0; st->lists_dirty = true; }
sqe_index* ivf_coarse(IvfState* st) { return st->coarse; }

bool ivf_trained(IvfState* st) { return st->trained; }

// sqe_index_load: centroids (normalised, as exported) and the per-row list assignment of a saved index
int ivf_restore(sqe_index* base, IvfState* st, const float* centroids_dev, const int32_t* assign_dev, int64_t n, hipStream_t s) {
    const int nlist = base->nlist;
    st->coarse->n.store(0);
    SQE_TRY(index_add_impl(st->coarse, centroids_dev, nlist, base->dim, true, s));
    SQE_TRY(st->assign.ensure((size_t)std::max<int64_t>(n, 1) * 4));
    if (n > 0) SQE_HIP(hipMemcpyAsync(st->assign.p, assign_dev, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    SQE_HIP(hipStreamSynchronize(s));
    st->trained = true;
    st->cent_dirty = true;
    st->n_assigned = n;
    st->lists_dirty = true;
    return SQE_OK;
}

// introspection for tests / tools: centroids and assignments as stored
int ivf_export(sqe_index* base, IvfState* st, float* centroids_host, int32_t* assign_host, hipStream_t s) {
    if (!st->trained) return fail(SQE_ERR_STATE, "ivf export: not trained");
    SQE_TRY(ivf_rows_added(base, st, s));
    if (centroids_host)
        SQE_HIP(hipMemcpyAsync(centroids_host, st->coarse->master, (size_t)base->nlist * base->dim * 4,
                               hipMemcpyDeviceToHost, s));
    if (assign_host && st->n_assigned > 0)
        SQE_HIP(hipMemcpyAsync(assign_host, st->assign.p, (size_t)st->n_assigned * 4, hipMemcpyDeviceToHost, s));
    SQE_HIP(hipStreamSynchronize(s));
    return SQE_OK;
}

}  // namespace sqe
