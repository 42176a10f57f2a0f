// kernels.h -- host-callable launchers of the HIP kernels (gfx950).
#pragma once

#include "common.h"

namespace sqe {

// ------------------------------------------------------------------ normalise (S1)
// out = x / (||x||_2 + 1e-9) per row, fp32 (main.py:315-316, 353-354); optionally also a
// bf16 copy (the scanned copy).  Either output may be null.  dim % 4 == 0.
// `bf16_pitch` = elements between consecutive bf16 output rows (>= dim; the scanned copy pads
// its rows so that consecutive rows do not map to the same memory channel).
// `resid_rows` (per row) / `resid_max` (atomic max, float bits) receive || x_hat - bf16(x_hat) ||_2;
// either may be null.
// `x_stride` = floats between consecutive INPUT rows (>= dim: a strided view, e.g. every P-th row of a block).
int launch_normalize_rows(const float* x, int64_t n, int dim, int64_t x_stride, float* out_f32, bf16_t* out_bf16,
                          int bf16_pitch, float* resid_rows, uint32_t* resid_max, hipStream_t stream);
// Rows that are already normalised (read back from a saved index): out_f32 = x bit for bit, bf16 copy
// and residual rebuilt (sqe_index_load).
int launch_restore_rows(const float* x, int64_t n, int dim, int64_t x_stride, float* out_f32, bf16_t* out_bf16, int bf16_pitch,
                        uint32_t* resid_max, hipStream_t stream);
// Same, rows scattered to out row ids `rows[i]` (sqe_index_update).
int launch_normalize_rows_scatter(const float* x, const int64_t* rows, int64_t n, int dim,
                                  float* out_f32, bf16_t* out_bf16, int bf16_pitch, uint32_t* resid_max,
                                  hipStream_t stream);

// ------------------------------------------------------------------ flat scan (S2)
constexpr int SCAN_BM = 256;        // DB rows per tile
constexpr int SCAN_BK = 64;         // K elements per pipeline stage
constexpr int CAND_CAP = 512;       // candidate slots per (chunk, query): kp kept + a compaction window + one tile of appends
constexpr int MAX_KP = 256;         // max candidates kept per (chunk, query)
constexpr int GMAX_COLS = 64;       // chunk maxima per (query, group) row of the global-bound table

struct ScanPlan {
    int bn;               // queries per workgroup tile (256 / 64 / 16)
    int qblocks;          // ceil(B / bn)
    int b_pad;            // qblocks * bn
    int n_tiles;          // ceil(n_rows / SCAN_BM)
    int n_chunks;         // DB chunks (persistent workgroups per query block)
    int tiles_per_chunk;  // ceil(n_tiles / n_chunks): the longest chunk (tiles are dealt out evenly)
    int kp;               // candidates kept per (chunk, query)
    int ngroups;          // rows of the global-bound table per query slice (1: chunk c folds its maxima into column c % 64)
    int gshift;           // log2 group size of the global bound (64 >> gshift >= kp); -1 = off
    int gshift_k;         // log2 group size of the k-row bound (64 >> gshift_k >= k), used with the 2 eps slack; -1 = off
    int k_rows;           // k when the k-row bound can take the k-th largest of the 16 group maxima (k <= 16), else 0
};
// k = 0: no k-row bound (the caller has no error bound for the scan scores)
ScanPlan make_scan_plan(int64_t n_rows, int B, int kp, int cu_count, int k = 0);

// Upper bound of |bf16 scan score - fp32 cosine| for a query with bf16 residual norm dq over rows whose residual
// norms are at most dx:  |<q_b, x_b> - <q, x>| <= ||q_b|| ||x - x_b|| + ||q - q_b|| ||x||, plus the rounding of the
// two fp32 accumulations that are compared (the scan's MFMA chain and the re-score's FMA chain): each is within
// K * 2^-24 * sum |a_i b_i| <= K * 2^-24 of the exact dot product of its unit-norm operands, together K * 2^-23.
// The term is floored at 2e-4 (its r01/r02 value, which covers K <= 1677): at dim 1024 nothing changes, and the
// proof holds for every dim sqe_index_create admits (r02 kept the constant while admitting dim 8192, where the worst
// case is 9.8e-4).  ONE definition: the scan's k-row bound and the certificate must agree on it.
__host__ __device__ inline float scan_eps(float dq, float dx, int K) {
    const float acc = fmaxf(2.0e-4f, (float)K * 1.1920929e-7f * 1.05f);
    return (1.0f + dq) * dx * 1.000001f + dq * 1.000001f + acc;
}

struct ScanArgs {
    const bf16_t* db;     // [round_up(n_rows, 256)] rows of K bf16 at `db_pitch` bytes, zero rows past n_rows
    const bf16_t* q;      // [b_pad] rows of K bf16 at `q_pitch` bytes, zero rows past B
    int db_pitch, q_pitch;
    int64_t n_rows;
    int K;
    int B;
    uint64_t* cand;       // [n_chunks, b_pad, CAND_CAP] candidate keys
    int* cand_cnt;        // [n_chunks, b_pad]
    uint32_t* gmax;       // [b_pad, ngroups, GMAX_COLS], zeroed before the launch
    unsigned long long* dbg_counters;   // null unless SQE_DBG has bit 32
    // k-row bound (plan.gshift_k >= 0): error-bound inputs, as SelectArgs; null = bound off
    const float* q_resid;
    const uint32_t* db_resid_max;
    // collect pass (launch_scan_collect): see ExactArgs
    const float* collect_thr;
    uint64_t* collect_keys;
    int* collect_cnt;
    const int* unc_count;
    int collect_lo, collect_hi;   // collect pass: this launch runs when collect_lo <= *unc_count <= collect_hi
    int tile_step = 1;            // > 1: scan DB tiles 0, step, 2 step, ... only (plan.n_tiles counts the scanned tiles); row ids stay global
};
int launch_scan_bf16(const ScanPlan& plan, const ScanArgs& args, hipStream_t stream);
// ping-pong form for 256-query blocks (scan_pp.hip, the default): the two waves of a SIMD alternate between
// a compute phase and a memory phase.  (Knobs build only: SQE_SCAN=v0 picks the two-stage form of scan.hip.)
int launch_scan_bf16_pp(const ScanPlan& plan, const ScanArgs& args, hipStream_t stream);
// second pass for uncertified queries: same scan, fixed thresholds args.collect_thr, every row at or
// above its query's threshold goes to args.collect_keys; exits at once when *args.unc_count == 0
int launch_scan_collect(const ScanPlan& plan, const ScanArgs& args, hipStream_t stream);

// ------------------------------------------------------------------ int8 first-pass scan (quant.hip, scan_i8.hip, select_i8.hip)
// Scale unit of the int8 copies of `dim`-d unit vectors: row scale = sxi * unit (quant.hip).
float i8_scale_unit(int dim);
// int8 copy of the 256-row TILES that hold master rows [first_row, first_row + n) (rows != null: the tiles of the n listed rows)
// into the TILED layout (tile t at t * tile_stride; K slice h of row r at h * 16 KiB + (r % 256) * 64), ONE scale per tile
// (written to sxi[row] of each of its rows), atomic max of the rounding residual || x_hat - sxi unit x8 || (float bits).
// Rows >= n_rows of a tile are written as zero vectors.
int launch_quantize_rows_i8(const float* master, const int64_t* rows, int64_t first_row, int64_t n, int64_t n_rows, int dim, int8_t* out,
                            int64_t tile_stride, uint32_t* sxi, uint32_t* resid_max, hipStream_t stream);
// int8 copy of normalised query rows, row-major at q_pitch bytes; sqi[q], resid_rows[q]
int launch_quantize_queries_i8(const float* qn, int B, int dim, int8_t* out, int q_pitch, uint32_t* sqi, float* resid_rows, hipStream_t stream);
int launch_quantize_gather_i8(const float* x, const int* gather, const int* scatter, int64_t n, int dim, int8_t* out, int64_t tile_stride,
                              uint32_t* sxi, hipStream_t stream);
// collect thresholds from the sample pass: tau[q] = cos_s[q][m - 1] (true cosines, best first)
int launch_i8_thresholds(const float* cos_s, int m, const uint32_t* sqi, int dim, int B, int b_pad, int* thr_int, float* thr_eff,
                         hipStream_t stream);
struct I8ScanArgs {
    const int8_t* db8; int64_t tile_stride; const uint32_t* sxi;
    const int8_t* q8; int q_pitch; const int* thr_int;
    int64_t n_rows; int K, B, b_pad, n_tiles, n_chunks, qblocks;
    int bn;                              // queries per workgroup tile: 256 (ping-pong kernel), 128 or 64 (staged kernels, HBM-bound)
    uint64_t* cand; int* cand_cnt;      // the bf16 scan's candidate lists: [n_chunks, b_pad, CAND_CAP], [n_chunks, b_pad]
    uint64_t* ovf; int* ovf_cnt;        // overflow pool: [b_pad, I8_OVF_CAP] keys that found their (chunk, query) list full, [b_pad] (zeroed by the caller)
    unsigned long long* stamps = nullptr;   // knobs build: 256 x u64, zeroed (scan_i8.hip: tile_end)
};
// A (chunk, query) list holds CAND_CAP keys; rows that belong together often sit together (the chunks of one document, a
// cluster appended in one call), so one query can collect thousands of keys from ONE chunk: what does not fit its list goes to
// the query's pool (one global atomic per key, rare) and the query stays certifiable.
constexpr int I8_OVF_CAP = 4096;
int launch_scan_i8(const I8ScanArgs& args, hipStream_t stream);
// the same scan built with a five-stage row ring (scan_i8_deep.hip): one 256-query block per chunk, where every tile comes from HBM
int launch_scan_i8_deep(const I8ScanArgs& args, hipStream_t stream);
// Threshold pass in int8: every step-th (whole) tile against query blocks of 256; out[chunk][query][16] = the two best
// (scaled score, row) of each of the 8 row lanes (scan_i8.hip: sample_i8_pp_kernel).  b_pad is a multiple of 256.
struct I8SampleArgs {
    const int8_t* db8; int64_t tile_stride; const uint32_t* sxi;
    const int8_t* q8; int q_pitch;
    int K, b_pad, n_tiles_s, step, n_chunks;
    void* out;                           // int2 [n_chunks][b_pad][16]
};
int launch_sample_i8(const I8SampleArgs& args, hipStream_t stream);
// Per query: the m-th largest of its n_chunks x 16 sample scores -> thr_int / thr_eff; the k best sample rows re-scored in
// fp32 -> sample_cos / sample_ids [B][m] (k entries valid, best first): lower bounds of the k-th cosine for select_i8.
struct I8SampleSelectArgs {
    const void* cand; int n_chunks, b_pad_s;          // the sample kernel's output and its query padding
    int m, k, B, b_pad, K;                            // b_pad: padding of thr_int / thr_eff (the collect scan's)
    const uint32_t* sqi; const float* master; const float* qn;
    int* thr_int; float* thr_eff; float* sample_cos; int64_t* sample_ids;
    // anchor of the threshold on a true cosine (select_i8.hip: i8_sample_select_kernel): the int8 residuals behind eps, the
    // margin as a fraction of eps, the sampling step (expected keys = sample values above the threshold x step) and the key budget
    const float* q_resid8; const uint32_t* db_resid8_max; float margin; int step, key_budget;
};
int launch_i8_sample_select(const I8SampleSelectArgs& args, hipStream_t stream);
// Per query: gather the collected keys, fp32 re-score in two stages (the best 64 by int8 score give t = k-th true cosine so
// far, then every collected row whose int8 score can still reach t), exact top-k, certificate thr_eff + eps < k-th cosine.
// collect_thr[q] = +inf if certified, else (k-th cosine so far) - bf16 eps: the input of the bf16 collect pass (exact.hip).
struct I8SelectArgs {
    const uint64_t* cand; const int* cand_cnt; int n_chunks, b_pad;
    const float* master; const float* qn; int K, B, k;
    const bf16_t* scan16; int pitch16;                        // the bf16 scan copy (rows of K bf16 at pitch16 bytes): middle stage
    const uint32_t* sxi; const uint32_t* sqi;                 // row / query scales
    const float* q_resid8; const uint32_t* db_resid8_max;     // int8 residuals (eps of the certificate)
    const float* q_resid16; const uint32_t* db_resid16_max;   // bf16 residuals (threshold of the fallback)
    const float* thr_eff;                                     // [b_pad] estimated-score bound of an uncollected row
    const float* sample_cos; const int64_t* sample_ids; int sample_m;   // threshold pass: top-m true cosines / rows of the sample, best first
    float* cos_out; int64_t* id_out; int64_t id_base;
    int* unc_count; float* collect_thr;
    unsigned long long* stats;                                 // null or [4]: keys gathered, rows re-scored, overflows, uncertified
    const uint64_t* ovf; const int* ovf_cnt;                   // the scan's overflow pool (I8ScanArgs)
};
int launch_select_i8(const I8SelectArgs& args, hipStream_t stream);

// ------------------------------------------------------------------ select + rescore (S3+S4)
struct SelectArgs {
    const uint64_t* cand;
    const int* cand_cnt;
    int n_chunks, b_pad, kp;
    const float* master;   // [n_rows, K] fp32 normalised rows
    const float* qn;       // [B, K] fp32 normalised queries
    int K, B, k;
    float* cos_out;        // [B, k]
    int64_t* id_out;       // [B, k]
    int64_t id_base;       // added to local row ids
    // exactness certificate (all null = off): a query whose k-th re-scored cosine does not beat the
    // best score any unseen row could have is queued for the exact fp32 rescan (exact.hip)
    const float* q_resid;          // [B]  || q_hat - bf16(q_hat) ||
    const uint32_t* db_resid_max;  // max over rows of || x_hat - bf16(x_hat) || (float bits)
    int* unc_count;                // number of uncertified queries
    float* collect_thr;            // [b_pad] per query: +inf if certified, else (k-th true cosine) - eps
    // the scan's global-bound table as the scan left it and the plan's gshift (null / -1: the scan ran without
    // that bound): rows the scan dropped below the kp-row bound are bounded by the table's final value
    const uint32_t* gmax;
    int gshift;
};
int launch_select_rescore(const SelectArgs& args, hipStream_t stream);

// ------------------------------------------------------------------ certified fallback
// For a query whose certificate failed, select.hip leaves collect_thr[q] = t - eps (t = lower bound
// of the true k-th cosine, eps = bf16 error bound): every row that can be in the exact top-k has a
// bf16 scan score >= that.  launch_scan_collect gathers those rows, launch_collect_rescore re-scores
// them in fp32 and writes the exact top-k over the uncertified result.
constexpr int EXACT_CAP = 4096;    // keys collected per uncertified query
struct ExactArgs {
    const float* master;   // [n_rows, K]
    const float* qn;       // [B, K] normalised queries
    int K, B, k;
    const float* collect_thr;   // [B]
    const int* unc_ids;    // [count] compact index -> query (launch_compact_uncertified)
    const int* unc_count;  // device: number of uncertified queries
    uint64_t* keys;        // [count, EXACT_CAP], by compact index
    const int* key_cnt;    // [count]
    float* cos_out;
    int64_t* id_out;
    int64_t id_base;
};
int launch_collect_rescore(const ExactArgs& args, hipStream_t stream);
// uncertified queries (collect_thr != +inf) -> dense batch: ids, thresholds (padded with +inf to thr_cap),
// bf16 rows; *unc_count = their number.  Query order is kept.
int launch_compact_uncertified(const float* collect_thr, int B, const bf16_t* qb, int pitch_bytes, int row_bytes, int* unc_ids,
                               float* thr_out, int thr_cap, bf16_t* qb_out, int* unc_count, hipStream_t stream);

// out[t][n] = <X[t], W[n]> fp32 from bf16 rows of K elements (dense, K * 2 bytes apart); N % 128 == 0,
// K % 64 == 0, X holds t_pad % 128 == 0 rows.  The encoder's small-batch GEMM (encoder.hip); IVF coarse scores.
int launch_scores_gemm(const bf16_t* W, const bf16_t* X, float* out, int N, int K, int T, int t_pad, int cu_count, hipStream_t stream);

// merge of [P,B,k] partial results (multi-GPU all-gather output).  A valid id of part p is mapped to
// id * id_mul + p * id_part_add + id_add before it is compared and written: (1, 0, 0) for parts that already
// hold global ids (torch.distributed row shards), (P, 1, base) for the round-robin shards of a device group.
int launch_merge_topk(const float* cos_parts, const int64_t* id_parts, int64_t part_stride_bytes,
                      int P, int B, int k, float* cos_out, int64_t* id_out, int64_t id_mul, int64_t id_part_add, int64_t id_add,
                      hipStream_t stream);

// ------------------------------------------------------------------ cache scan (S8)
// sims[i] = cosine(mat[slot(i)], q) with the zero-norm rule; slot(i) = order ? order[i] : i.
// best = first strict max over i from (-1.0, -1).
int launch_cosine_scan(const float* mat, const int32_t* order, int m, int dim, const float* q,
                       float* sims, float* best_sim, int32_t* best_idx, hipStream_t stream);

}  // namespace sqe
