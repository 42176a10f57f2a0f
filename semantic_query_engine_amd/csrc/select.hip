// select.hip -- S3 + S4: per query, merge the per-chunk candidate lists of the scan, keep
// the best kp keys, re-score them in fp32 against the fp32 master rows, and emit the final
// top-k ordered by (fp32 cosine desc, row id asc).  Also the [P,B,k] merge used after the
// multi-GPU all-gather.
//
// Tiny kernels (B workgroups, a few KB each): latency-bound, not roofline-relevant.  The
// rescore reads kp * dim * 4 bytes of the master per query (128 KiB at kp=32, dim=1024).
#include "kernels.h"

namespace sqe {

namespace {

// Two shapes of the same kernel: small batches (many DB chunks per query, most CUs idle) give a query
// 16 waves -- the fp32 re-score runs 16 rows at a time -- and 128 KiB of LDS for its keys; large batches
// (64 chunks) get 4 waves and 32 KiB so that several queries share a CU.

struct SelectKernelArgs {
    const uint64_t* cand;
    const int* cand_cnt;
    int n_chunks, b_pad, kp;
    const float* master;
    const float* qn;
    int K, B, k;
    float* cos_out;
    int64_t* id_out;
    int64_t id_base;
    const float* q_resid;
    const uint32_t* db_resid_max;
    int* unc_count;
    float* collect_thr;
    const uint32_t* gmax;
    int gshift;
};

// MSB-first byte-wise radix select of the `kth` largest of n keys.  `get(e)` returns key e (0 = no key).
// Returns 0 when there are fewer than kth keys (everything qualifies).
template <int SEL_THREADS, typename Get>
__device__ uint64_t block_select_kth(Get get, int n, int kth, int* hist, int* scratch) {
    static_assert(SEL_THREADS >= 256, "hist_locate: one thread per histogram bin");
    const int tid = threadIdx.x;
    uint64_t prefix = 0;      // determined high bytes
    int remaining = kth;
    for (int byte = 7; byte >= 0; --byte) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const int shift = byte * 8;
        for (int e = tid; e < n; e += SEL_THREADS) {
            const uint64_t key = get(e);
            if (key == 0ull) continue;
            const bool match = (byte == 7) || ((key >> (shift + 8)) == (prefix >> (shift + 8)));
            if (match) atomicAdd(&hist[(int)((key >> shift) & 0xff)], 1);
        }
        __syncthreads();
        int bin, rem;
        hist_locate(hist, remaining, bin, rem);      // ends with a barrier: hist may be cleared again
        if (bin < 0) return 0ull;                    // fewer than `remaining` keys in total
        prefix |= ((uint64_t)bin << shift);
        remaining = rem;
    }
    return prefix;
}

template <int SEL_THREADS, int SEL_KEYS_CAP>
__global__ __launch_bounds__(SEL_THREADS) void select_rescore_kernel(SelectKernelArgs p) {
    extern __shared__ __attribute__((aligned(16))) uint64_t keys[];      // [SEL_KEYS_CAP]
    __shared__ int hist[256];
    __shared__ int scratch[4];
    __shared__ uint32_t sel_row[MAX_KP];
    __shared__ float sel_score[MAX_KP];
    __shared__ float kth_score;
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    constexpr int NW = SEL_THREADS / 64;

    // ---- gather the valid keys of this query's per-chunk lists into LDS: one wave per chunk, positions
    // by ballot (one LDS atomic per wave and 64 slots)
    if (tid == 0) { scratch[2] = 0; scratch[3] = 0; }
    __syncthreads();
    for (int c = wave; c < p.n_chunks; c += NW) {
        const int cnt = min(p.cand_cnt[(size_t)c * p.b_pad + q], p.kp);
        const uint64_t* list = p.cand + ((size_t)c * p.b_pad + q) * CAND_CAP;
        for (int s0 = 0; s0 < cnt; s0 += 64) {
            const bool have = s0 + lane < cnt;
            const uint64_t key = have ? list[s0 + lane] : 0ull;
            const uint64_t m = __ballot(have);
            int base = 0;
            if (lane == 0) base = atomicAdd(&scratch[3], __popcll(m));
            base = __shfl(base, 0, 64);
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            if (have && pos < SEL_KEYS_CAP) keys[pos] = key;
        }
    }
    __syncthreads();
    const int n_keys = scratch[3];
    const bool in_lds = n_keys <= SEL_KEYS_CAP;
    const int total_slots = p.n_chunks * p.kp;
    auto global_key = [&](int e) -> uint64_t {
        const int chunk = e / p.kp, slot = e - chunk * p.kp;
        return slot < p.cand_cnt[(size_t)chunk * p.b_pad + q] ? p.cand[((size_t)chunk * p.b_pad + q) * CAND_CAP + slot] : 0ull;
    };
    auto lds_key = [&](int e) -> uint64_t { return keys[e]; };

    const uint64_t T = in_lds ? block_select_kth<SEL_THREADS>(lds_key, n_keys, p.kp, hist, scratch)
                              : block_select_kth<SEL_THREADS>(global_key, total_slots, p.kp, hist, scratch);
    __syncthreads();
    const int n_scan = in_lds ? n_keys : total_slots;
    for (int e = tid; e < n_scan; e += SEL_THREADS) {
        const uint64_t key = in_lds ? keys[e] : global_key(e);
        if (key != 0ull && key >= T) {
            const int pos = atomicAdd(&scratch[2], 1);
            if (pos < MAX_KP) sel_row[pos] = key_row(key);
        }
    }
    __syncthreads();
    const int m = min(scratch[2], p.kp);
    if (tid == 0) kth_score = -INFINITY;

    // fp32 re-score: one wave per candidate, float4 per lane per step
    const float4* qv = reinterpret_cast<const float4*>(p.qn + (size_t)q * p.K);
    const int nvec = p.K >> 2;
    // (four rows per step: their loads are in flight together, each row keeps its own fmaf chain, so the
    // sums are the ones a row-at-a-time loop produces)
    constexpr int RU = 4;
    for (int i0 = wave * RU; i0 < m; i0 += NW * RU) {
        const float4* rv[RU];
        float s[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            rv[u] = reinterpret_cast<const float4*>(p.master + (size_t)sel_row[min(i0 + u, m - 1)] * p.K);
            s[u] = 0.f;
        }
        for (int v = lane; v < nvec; v += 64) {
            const float4 b = qv[v];
            float4 a[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) a[u] = rv[u][v];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                s[u] = fmaf(a[u].x, b.x, s[u]); s[u] = fmaf(a[u].y, b.y, s[u]);
                s[u] = fmaf(a[u].z, b.z, s[u]); s[u] = fmaf(a[u].w, b.w, s[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const float t = wave_sum(s[u]);
            if (lane == 0 && i0 + u < m) sel_score[i0 + u] = t + 0.0f;
        }
    }
    __syncthreads();

    // final order: (score desc, row asc); NaN scores rank last
    float* cos_out = p.cos_out + (size_t)q * p.k;
    int64_t* id_out = p.id_out + (size_t)q * p.k;
    for (int i = tid; i < m; i += SEL_THREADS) {
        const float si = sel_score[i];
        const uint32_t ri = sel_row[i];
        int rank = 0;
        for (int j = 0; j < m; ++j) {
            const float sj = sel_score[j];
            const uint32_t rj = sel_row[j];
            const bool better = (si != si) ? (sj == sj || rj < ri)
                                           : (sj > si || (sj == si && rj < ri));
            rank += (j != i && better) ? 1 : 0;
        }
        if (rank < p.k) {
            cos_out[rank] = si;
            id_out[rank] = (int64_t)ri + p.id_base;
        }
        if (rank == p.k - 1) kth_score = si;
    }
    for (int i = m + tid; i < p.k; i += SEL_THREADS) {
        cos_out[i] = -INFINITY;
        id_out[i] = -1;
    }
    // ---- exactness certificate.  A row that is not a candidate was
    //   (a) dropped under the scan's k-row bound: excluded rigorously there (scan_common.h: refresh_apply);
    //   (b) dropped under the kp-row bound: its scan score is below that bound's FINAL value, which the table
    //       the scan left behind still gives (columns and bound only rise);
    //   (c) cut from a chunk's list or from the union here: its key is below T, the kp-th best key of the union
    //       (T == 0: the union has fewer than kp keys and nothing was cut).
    // A scan score differs from the true fp32 cosine by at most eps (kernels.h: scan_eps), so no row of (b) or
    // (c) can reach max(score(T), bound) + eps.
    if (p.unc_count) {
        float gfin = -INFINITY;
        if (p.gmax && p.gshift >= 0 && wave == 0) {
            uint32_t v = p.gmax[((size_t)(q >> 6) * GMAX_COLS + lane) * 64 + (q & 63)];      // column `lane`
            if (p.gshift >= 1) v = max(v, (uint32_t)__shfl_xor((int)v, 1, 64));
            if (p.gshift >= 2) v = max(v, (uint32_t)__shfl_xor((int)v, 2, 64));
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) v = min(v, (uint32_t)__shfl_xor((int)v, d, 64));
            if (v != 0u) gfin = f32_from_orderable(v);
        }
        __syncthreads();
        if (tid == 0) {
            float thr = INFINITY;                    // certified: nothing to collect
            const float unseen = fmaxf(T != 0ull ? key_score(T) : -INFINITY, gfin);
            if (unseen > -INFINITY) {
                const float eps = scan_eps(p.q_resid[q], __uint_as_float(*p.db_resid_max), p.K);
                const bool certified = m >= p.k && kth_score > unseen + eps;
                if (!certified) {
                    atomicAdd(p.unc_count, 1);
                    thr = (m >= p.k ? kth_score : -1.0f) - eps;
                }
            }
            p.collect_thr[q] = thr;
        }
    }
}

// ---- merge of P partial top-k lists per query (after the all-gather of a sharded index)
__global__ __launch_bounds__(64) void merge_topk_kernel(const char* __restrict__ cos_parts,
                                                        const char* __restrict__ id_parts,
                                                        int64_t cos_stride, int64_t id_stride,
                                                        int P, int B, int k,
                                                        float* __restrict__ cos_out,
                                                        int64_t* __restrict__ id_out,
                                                        int64_t id_mul, int64_t id_part_add, int64_t id_add) {
    const int q = blockIdx.x;
    const int lane = threadIdx.x;
    const int total = P * k;
    auto score = [&](int e) {
        const int part = e / k, j = e - part * k;
        return reinterpret_cast<const float*>(cos_parts + part * cos_stride)[(size_t)q * k + j];
    };
    auto ident = [&](int e) {
        const int part = e / k, j = e - part * k;
        const int64_t raw = reinterpret_cast<const int64_t*>(id_parts + part * id_stride)[(size_t)q * k + j];
        return raw < 0 ? raw : raw * id_mul + part * id_part_add + id_add;      // shard-local -> global row id
    };
    // rank by counting: entry e beats f when cos higher, or equal cos and lower id
    int valid = 0;
    for (int e = lane; e < total; e += 64) {
        const float se = score(e);
        const int64_t ie = ident(e);
        if (ie < 0) continue;
        ++valid;
        int rank = 0;
        for (int f = 0; f < total; ++f) {
            const float sf = score(f);
            const int64_t idf = ident(f);
            if (f == e || idf < 0) continue;
            const bool better = (se != se) ? (sf == sf || idf < ie) : (sf > se || (sf == se && idf < ie));
            rank += better ? 1 : 0;
        }
        if (rank < k) {
            cos_out[(size_t)q * k + rank] = se;
            id_out[(size_t)q * k + rank] = ie;
        }
    }
    for (int off = 32; off > 0; off >>= 1) valid += __shfl_xor(valid, off, 64);
    for (int i = valid + lane; i < k; i += 64) {
        cos_out[(size_t)q * k + i] = -INFINITY;
        id_out[(size_t)q * k + i] = -1;
    }
}

}  // namespace

int launch_select_rescore(const SelectArgs& a, hipStream_t stream) {
    if (a.B <= 0) return SQE_OK;
    if (a.k < 1 || a.k > MAX_KP || a.kp < a.k || a.kp > MAX_KP)
        return fail(SQE_ERR_INVALID, "select: need 1 <= k <= kp <= 256");
    SelectKernelArgs k;
    k.cand = a.cand; k.cand_cnt = a.cand_cnt; k.n_chunks = a.n_chunks; k.b_pad = a.b_pad; k.kp = a.kp;
    k.master = a.master; k.qn = a.qn; k.K = a.K; k.B = a.B; k.k = a.k;
    k.cos_out = a.cos_out; k.id_out = a.id_out; k.id_base = a.id_base;
    k.q_resid = a.q_resid; k.db_resid_max = a.db_resid_max; k.unc_count = a.unc_count; k.collect_thr = a.collect_thr;
    k.gmax = a.gmax; k.gshift = a.gmax ? a.gshift : -1;
    if ((int64_t)a.n_chunks * a.kp > 4096) {              // few query blocks, many chunks: B <= 256
        constexpr int LDS = 16384 * 8;
        SQE_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(select_rescore_kernel<1024, 16384>), LDS));
        hipLaunchKernelGGL((select_rescore_kernel<1024, 16384>), dim3(a.B), dim3(1024), LDS, stream, k);
    } else {
        hipLaunchKernelGGL((select_rescore_kernel<256, 4096>), dim3(a.B), dim3(256), 4096 * 8, stream, k);
    }
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

int launch_merge_topk(const float* cos_parts, const int64_t* id_parts, int64_t part_stride_bytes,
                      int P, int B, int k, float* cos_out, int64_t* id_out, int64_t id_mul, int64_t id_part_add, int64_t id_add,
                      hipStream_t stream) {
    if (B <= 0) return SQE_OK;
    if (P < 1 || k < 1) return fail(SQE_ERR_INVALID, "merge: P and k must be >= 1");
    const int64_t cs = part_stride_bytes ? part_stride_bytes : (int64_t)B * k * 4;
    const int64_t is = part_stride_bytes ? part_stride_bytes : (int64_t)B * k * 8;
    hipLaunchKernelGGL(merge_topk_kernel, dim3(B), dim3(64), 0, stream,
                       reinterpret_cast<const char*>(cos_parts), reinterpret_cast<const char*>(id_parts),
                       cs, is, P, B, k, cos_out, id_out, id_mul, id_part_add, id_add);
    SQE_HIP(hipGetLastError());
    return SQE_OK;
}

}  // namespace sqe
