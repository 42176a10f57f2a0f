/* sqe.h -- C ABI of libsqe.so, the MI355X (gfx950) embedding + cosine k-NN + cosine-cache
 * engine that replaces the Ollama-embed / OpenSearch-HNSW / Redis-scan leg of the
 * reference's /ask pipeline (/root/reference/app/main.py:56-180, 250-373).
 *
 * The reference has no FFI of its own: its hot path is three network clients called by
 * plain Python names.  Each entry point below states which reference call it stands
 * behind; INTEGRATION.md shows the ctypes stub a maintainer adds on the reference side.
 *
 * Conventions
 *   - return 0 (SQE_OK) or a negative error code; sqe_last_error() gives thread-local text
 *   - the caller allocates every output buffer; the library owns all device memory
 *   - host inputs are copied before return and never retained
 *   - "_device" variants take device pointers, enqueue on the context stream and do not
 *     synchronise (sqe_synchronize does); they exist so a caller that already holds its
 *     data in HBM (bench.py, the encoder -> search hand-off, torch.distributed shards)
 *     pays no PCIe copy
 *   - every entry point is thread-safe: the reference calls add_embeddings from a
 *     thread-pool thread while search runs on the event-loop thread (main.py:454-455, 499).
 *     There is no context-wide lock: every index, cache and encoder has its own mutex and its
 *     own stream (host entry points run there and synchronise it before returning), so an add
 *     on one index never blocks a search on another, the cache scan or the encoder
 *   - multi-GPU, two forms: ONE host process driving several devices (sqe_create with
 *     n_dev > 1: what the reference's single uvicorn process needs, main.py:738-739), or one
 *     process per GPU over torch.distributed with single-device contexts (bench.py, sharded.py)
 *   - the library reads no environment variable
 */
#ifndef SQE_H
#define SQE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SQE_VERSION 100 /* 0.1.0 */

typedef struct sqe_ctx sqe_ctx;
typedef struct sqe_index sqe_index;
typedef struct sqe_cache sqe_cache;
typedef struct sqe_encoder sqe_encoder;
typedef struct sqe_tokenizer sqe_tokenizer;

enum {
    SQE_OK = 0,
    SQE_ERR_INVALID = -1,     /* bad argument */
    SQE_ERR_HIP = -2,         /* a HIP runtime call failed */
    SQE_ERR_OOM = -3,         /* device or host allocation failed */
    SQE_ERR_STATE = -4,       /* object not in a state that allows the call */
    SQE_ERR_UNSUPPORTED = -5, /* valid request this build does not implement */
    SQE_ERR_IO = -6           /* file could not be opened, read or written, or is not a saved index */
};

enum { SQE_INDEX_FLAT = 0, SQE_INDEX_IVF_FLAT = 1 };

/* How the stored rows are scanned.  Both return the exact fp32 top-k: cosines are fp32 re-scores from the fp32
 * master copy, and a per-query certificate (option "certify") proves that no row outside the re-scored set can
 * reach the k-th cosine -- the rounding of every scanned copy is measured and bounded -- with a bf16 collect pass
 * for the queries where the proof fails.
 *   BF16_RESCORE: bf16 MFMA scan with a fused top-k filter (every batch size, every index size; what IVF indexes and
 *                 every case outside the int8 conditions below run);
 *   INT8_RESCORE (default of FLAT indexes): int8 MFMA first pass at twice the bf16 rate over a per-row-scaled int8 copy (+1 byte per element):
 *                 a 2 % row sample fixes per-query thresholds, the int8 scan collects every row above them, the
 *                 collected rows are re-scored in fp32.  Used on FLAT indexes of >= "i8_min_rows" rows (default 1 M),
 *                 dim >= 256 and a multiple of 128, k <= "i8_sample_m" (20), rows that quantise within "i8_max_resid";
 *                 everything else runs the bf16 scan.  Suited to rows whose
 *                 elements are of similar magnitude (unit Gaussian-like embeddings): the int8 error bound is ~8 x the
 *                 bf16 one, and data on which it is too wide simply takes the bf16 pass.
 * (r02's header also listed an fp32 scan mode that was never built; it is gone.) */
enum { SQE_SCAN_BF16_RESCORE = 0, SQE_SCAN_INT8_RESCORE = 2 };

/* ---- library / context ---------------------------------------------------------- */
int sqe_version(void);
const char* sqe_last_error(void);

/* n_dev == 1: the context drives HIP device device_ids[0].
 * n_dev > 1: single process, many devices (the reference is one uvicorn process, main.py:738-739).  The
 * context leads one member context per device; flat indexes created on it are sharded row-wise (global row g
 * on shard g % n_dev), a search sends the query batch to every device, runs the per-shard top-k there and
 * exchanges the packed [B,k] results in ONE step -- RCCL all-gather over xGMI (ncclCommInitAll, one
 * communicator per device) or, where RCCL is unavailable, peer copies to device_ids[0] -- before the merge
 * on device_ids[0].  Device pointers of "_device" entry points are memory of device_ids[0].  Caches and
 * encoders live on device_ids[0] (replicas only).  IVF indexes shard the same way: sqe_index_train
 * runs k-means once on device_ids[0] and replicates the centroids, every device keeps the lists of the rows it owns,
 * all devices probe the same lists and the per-device top-k go through the same exchange + merge. */
int sqe_create(const int* device_ids, int n_dev, sqe_ctx** out);
/* General form: `exchange` picks the exchange step, and device ids may repeat (several logical shards on
 * one device -- how a one-GPU box rehearses the sharded path; those use the copy exchange).  A group of ONE
 * shard with SQE_EXCHANGE_RCCL runs the in-library RCCL leg on a single device. */
enum { SQE_EXCHANGE_AUTO = 0, SQE_EXCHANGE_RCCL = 1, SQE_EXCHANGE_COPY = 2 };
int sqe_create_sharded(const int* device_ids, int n_shards, int exchange, sqe_ctx** out);
/* Shards of the context (1 for a single-device context), the exchange in use and the device of each shard. */
int sqe_group_info(sqe_ctx* ctx, int* n_shards, int* exchange, int* device_ids, int cap);
void sqe_destroy(sqe_ctx* ctx);
int sqe_synchronize(sqe_ctx* ctx);
/* hipStream_t the "_device" entry points enqueue on (for event timing by the caller). */
void* sqe_stream(sqe_ctx* ctx);
/* Make the context enqueue on a caller-owned hipStream_t (e.g. torch's current stream, so
 * collectives and library kernels order without host synchronisation); NULL restores the
 * context's own stream.  The caller keeps the stream alive while it is set. */
int sqe_set_stream(sqe_ctx* ctx, void* hip_stream);
int sqe_device_info(sqe_ctx* ctx, char* name, int name_cap, int* cu_count, int64_t* hbm_bytes);

/* ---- vector index: stands behind OpenSearchIndexer (main.py:291-373) ------------- */
/* Index mapping of main.py:262-282: cosine similarity over `dim`-d vectors.  `kind`
 * selects exact brute force (FLAT) or IVF-flat with `nlist` lists. dim % 64 == 0. */
int sqe_index_create(sqe_ctx* ctx, int dim, int kind, int nlist, sqe_index** out);
void sqe_index_destroy(sqe_index* idx);
int sqe_index_reserve(sqe_index* idx, int64_t rows);

/* add_embeddings (main.py:309-338): L2-normalises each row as x / (||x|| + 1e-9) in fp32
 * (main.py:315-316) and appends; rows get ids count .. count+n-1.  x is [n, dim] row-major. */
int sqe_index_add(sqe_index* idx, const float* x_host, int64_t n);
int sqe_index_add_device(sqe_index* idx, const float* x_dev, int64_t n);
/* Re-indexing an existing `_id` overwrites the document (OpenSearch "index" op,
 * main.py:321-325): replace the given rows in place. */
int sqe_index_update(sqe_index* idx, const int64_t* rows_host, const float* x_host, int64_t n);
/* has_any_data (main.py:300-307) is count > 0. */
int sqe_index_count(const sqe_index* idx, int64_t* out);
/* Normalised fp32 rows as stored (`_source.embedding` of a hit, main.py:327-331). */
int sqe_index_get_rows(sqe_index* idx, const int64_t* rows_host, int64_t n, float* out_host);

/* Options: "scan_mode" (SQE_SCAN_*; int8 tuning: "i8_min_rows", "i8_sample_step" = sample every n-th tile,
 * "i8_sample_m" = the threshold is the m-th best score of the sample, "i8_anchor_margin" (default 0.25): the threshold never
 * lies above (best true cosine of the sample) - eps * (1 + margin), which makes the int8 certificate hold by construction on rows
 * that crowd together, "i8_key_budget" (default 6144; 0 = no limit): where the sample predicts that the anchored threshold would collect more
 * keys than this, the m-th sample score alone stands, "i8_sample_int8" = 1 (default): the sample is scanned
 * in int8 too, 0: by the bf16 kernels with an fp32 re-score, "i8_max_resid" = the largest int8 rounding residual
 * of a stored row, default 0.02, beyond which the index answers with the bf16 scan), "rescore_k" (candidates kept by the bf16 scan,
 * 0 = automatic), "nprobe" default for IVF, "id_base" (added to every returned row id:
 * the first global row of this shard in a row-sharded index), "certify" (default 1: prove
 * per query that no row outside the re-scored candidates can reach the k-th cosine -- the
 * bf16 rounding of every vector is bounded -- and re-scan the fp32 master for the queries
 * where that proof fails; 0 = skip both). */
int sqe_index_set_option(sqe_index* idx, const char* key, double value);

/* search (main.py:347-373): q is [B, dim] row-major raw query embeddings; each is
 * normalised as q / (||q|| + 1e-9) (main.py:353-354) and its cosine top-k returned best
 * first, ties to the lowest id.  cos_out [B,k] fp32 cosines, id_out [B,k] int64 row ids,
 * padded with (-inf, -1) when fewer than k rows qualify.  The reference searches row 0
 * only (main.py:355); B > 1 is the batched form of the same call.  nprobe is ignored for
 * FLAT (0 = index default for IVF).  1 <= k <= 256. */
int sqe_index_search(sqe_index* idx, const float* q_host, int B, int k, int nprobe,
                     float* cos_out_host, int64_t* id_out_host);
int sqe_index_search_device(sqe_index* idx, const float* q_dev, int B, int k, int nprobe,
                            float* cos_out_dev, int64_t* id_out_dev);

/* IVF only: k-means (spherical, Lloyd) on a sample, then (re)assignment of stored rows. */
int sqe_index_train(sqe_index* idx, const float* x_host, int64_t n, int iters, uint64_t seed);
int sqe_index_train_device(sqe_index* idx, const float* x_dev, int64_t n, int iters, uint64_t seed);
/* IVF introspection: normalised centroids [nlist, dim] and the list of every stored row [count]
 * (either pointer may be NULL). */
int sqe_index_ivf_export(sqe_index* idx, float* centroids_host, int32_t* assign_host);

/* INT8_RESCORE introspection (tests/test_i8_gpu.py: the bit-exact parity check of the int8 kernels' integer arithmetic -- the
 * set of keys one collect launch appended against {(acc * s, row) : acc * s >= thr} recomputed from the same int8 operands in
 * NumPy).  sqe_index_i8_last describes the LAST search this index answered with the int8 first pass (SQE_ERR_STATE if
 * there was none); sqe_index_i8_read copies one of the device buffers that search read or wrote to the host, `bytes` bytes
 * from byte offset `offset`.  The buffers are overwritten by the next search.  Single-device FLAT indexes only.
 *   SQE_I8_ROWS        int8 copy of the rows, TILED: tile t (tile_rows rows) at t * tile_stride bytes; inside a tile the 64-element
 *                      slice h of row r at h * tile_rows * 64 + r * 64
 *   SQE_I8_ROW_SCALES  uint32 [tiles * tile_rows]: the integer scale of each row (one value per tile)
 *   SQE_I8_QUERIES     int8 [b_pad][q_pitch]: the quantised queries, row-major (rows >= B are zero)
 *   SQE_I8_THRESHOLDS  int32 [b_pad]: collect threshold of each query on acc * scale
 *   SQE_I8_LIST_COUNTS int32 [n_chunks][b_pad]: keys APPENDED to each (chunk, query) list (the first list_cap of them are in the
 *                      list, the rest went to the query's overflow pool)
 *   SQE_I8_LISTS       uint64 [n_chunks][b_pad][list_cap]: keys, (score ^ 0x80000000) << 32 | (0xFFFFFFFF - row), unordered
 *   SQE_I8_POOL_COUNTS int32 [b_pad]: keys that found their (chunk, query) list full and went to the query's overflow pool
 *   SQE_I8_POOLS       uint64 [b_pad][pool_cap]: those keys
 *   SQE_I8_SAMPLE_BEST int32 [sample_chunks][sample_b_pad][8 row lanes][2][2]: threshold pass, the two best (score, row) of each
 *                      lane stream (row lane l of a sampled tile: rows (l >> 2) * 128 + (l & 3) * 4 + 16 i + j, i < 8, j < 4) */
typedef struct sqe_i8_launch_t {
    int64_t rows;          /* rows scanned */
    int64_t tile_stride;   /* bytes between tiles of the int8 copy */
    int32_t dim, B, b_pad, k;
    int32_t tile_rows;     /* 256 */
    int32_t q_pitch;       /* bytes between quantised query rows */
    int32_t query_block;   /* queries per workgroup: 256 (ping-pong kernel), 128 or 64 (staged kernels) */
    int32_t n_chunks;      /* row chunks of the collect launch: chunk c holds tiles [c * (T / n) + min(c, T % n), ...), T tiles dealt out evenly */
    int32_t list_cap;      /* slots per (chunk, query) list */
    int32_t sample_int8;   /* 1: the threshold pass ran in int8 (SQE_I8_SAMPLE_BEST is valid) */
    int32_t sample_step;   /* the threshold pass scanned tiles 0, step, 2 step, ... (whole tiles only) */
    int32_t sample_tiles, sample_chunks, sample_b_pad, sample_m;
    int32_t uncertified;   /* queries that went on to the bf16 collect pass (it reuses the list buffers: they are valid only when 0) */
    int32_t pool_cap;      /* slots of a query's overflow pool */
} sqe_i8_launch_t;
enum { SQE_I8_ROWS = 0, SQE_I8_ROW_SCALES = 1, SQE_I8_QUERIES = 2, SQE_I8_THRESHOLDS = 3, SQE_I8_LIST_COUNTS = 4, SQE_I8_LISTS = 5,
       SQE_I8_SAMPLE_BEST = 6, SQE_I8_POOL_COUNTS = 7, SQE_I8_POOLS = 8 };
int sqe_index_i8_last(sqe_index* idx, sqe_i8_launch_t* out);
int sqe_index_i8_read(sqe_index* idx, int what, int64_t offset, void* out_host, int64_t bytes);

/* Persistence.  The reference keeps its vectors in the OpenSearch index across restarts and skips the
 * rebuild when `has_any_data()` is true (main.py:300-307, :422-424); here the index lives in HBM, so it
 * is written to / read from a local file: the normalised fp32 rows (plus IVF centroids and list
 * assignments) exactly as stored -- a loaded index returns bit-identical results.  The scanned bf16
 * copy and the IVF lists are rebuilt on load.  `sqe_index_load` creates the index. */
int sqe_index_save(sqe_index* idx, const char* path);
int sqe_index_load(sqe_ctx* ctx, const char* path, sqe_index** out);

/* Merge of per-shard results after the all-gather of a row-sharded index: part p holds
 * cos [B,k] fp32 at cos_parts_dev + p * part_stride_bytes and global ids [B,k] int64
 * (-1 padded) at id_parts_dev + p * part_stride_bytes (part_stride_bytes = 0: dense
 * [P,B,k] arrays).  out is [B,k], best first, ties to the lowest id.  Device pointers,
 * context stream. */
int sqe_merge_topk_device(sqe_ctx* ctx, const float* cos_parts_dev, const int64_t* id_parts_dev,
                          int64_t part_stride_bytes, int P, int B, int k,
                          float* cos_out_dev, int64_t* id_out_dev);

/* ---- semantic cache scan: stands behind the loop of lfu_cache_get (main.py:73-87) -- */
/* One-shot form: mat is [m, dim] raw (un-normalised) cached embeddings in list order
 * (index 0 = newest), q is [dim] raw.  cosine = dot / (||a|| * ||b||) in fp32 with the
 * zero-norm rule (main.py:59-64); returns the FIRST strict maximum starting from
 * (-1.0, -1) exactly as main.py:74-87 does (NaN never wins). */
int sqe_cosine_best(sqe_ctx* ctx, const float* mat_host, int m, int dim, const float* q_host,
                    float* best_sim, int32_t* best_idx);
/* All m cosines (the values cosine_similarity returns, main.py:59-64). */
int sqe_cosine_all(sqe_ctx* ctx, const float* mat_host, int m, int dim, const float* q_host,
                   float* sims_out_host);

/* Resident form: the cache matrix lives in HBM; the host keeps the LFU bookkeeping
 * (freq counters, JSON payloads) and tells the library which slot holds which list
 * position.  `order_host[i]` = slot of list position i (position 0 = newest). */
int sqe_cache_create(sqe_ctx* ctx, int capacity, int dim, sqe_cache** out);
void sqe_cache_destroy(sqe_cache* c);
int sqe_cache_set_slot(sqe_cache* c, int slot, const float* vec_host);
int sqe_cache_best(sqe_cache* c, const int32_t* order_host, int m, const float* q_host,
                   float* best_sim, int32_t* best_pos);

/* ---- encoder: stands behind ollama_embed_text (main.py:134-145) ------------------ */
typedef struct sqe_bert_cfg {
    int32_t vocab_size;  /* 30522 */
    int32_t hidden;      /* 1024 */
    int32_t layers;      /* 24 */
    int32_t heads;       /* 16 */
    int32_t inter;       /* 4096 */
    int32_t max_pos;     /* 512 */
    int32_t type_vocab;  /* 2 */
    float ln_eps;        /* 1e-12 */
} sqe_bert_cfg;

/* Tensors are fp32 host arrays named as in the HF BertModel state dict
 * ("embeddings.word_embeddings.weight", "encoder.layer.0.attention.self.query.weight"...);
 * they are converted to bf16 (matrices, embeddings) / fp32 (biases, LayerNorm) on load. */
int sqe_encoder_create(sqe_ctx* ctx, const sqe_bert_cfg* cfg, sqe_encoder** out);
void sqe_encoder_destroy(sqe_encoder* enc);
int sqe_encoder_load_tensor(sqe_encoder* enc, const char* name, const float* data_host,
                            const int64_t* shape, int ndim);
int sqe_encoder_finalize(sqe_encoder* enc);
/* Host-only BERT WordPiece (clean, CJK spacing, NFD + accent strip, lower-case, punctuation
 * split, greedy longest match with "##" continuations, [CLS] ... [SEP], truncation to max_len
 * ids): the tokenizer that ran inside Ollama for main.py:139-142.  Needs no GPU and no context.
 * vocab_utf8: one token per line, line number = id (a local vocab.txt; nothing is fetched). */
int sqe_tokenizer_create(const char* vocab_utf8, int64_t vocab_bytes, sqe_tokenizer** out);
void sqe_tokenizer_destroy(sqe_tokenizer* tok);
int sqe_tokenize(const sqe_tokenizer* tok, const char* text_utf8, int64_t text_bytes, int max_len,
                 int32_t* ids_out, int* len_out);
/* n texts -> ids_out [n, max_len] ([PAD] = 0 filled), lens_out [n]; multi-threaded for n >= 64. */
int sqe_tokenize_batch(const sqe_tokenizer* tok, const char* const* texts, const int64_t* text_bytes, int n,
                       int max_len, int32_t* ids_out, int32_t* lens_out);
/* ids [B,S] int32 row-major (anything past lens[b] is ignored), lens [B];
 * out [B, hidden] fp32 = final-layer CLS row (no normalisation, as Ollama's
 * /api/embeddings output feeds main.py:315-316 and :59-64 un-normalised). */
int sqe_encode(sqe_encoder* enc, const int32_t* ids_host, const int32_t* lens_host, int B, int S,
               float* out_host);
int sqe_encode_device(sqe_encoder* enc, const int32_t* ids_dev, const int32_t* lens_dev, int B, int S,
                      float* out_dev);

/* ---- stats ----------------------------------------------------------------------- */
/* When profiling is on, every stage is bracketed by hipEvents on the context stream;
 * sqe_stats reads the accumulated totals (it synchronises the stream). */
typedef struct sqe_stats_t {
    double scan_ms;        /* bf16 scan kernels: the main launches plus the collect-pass scans of uncertified queries */
    double prep_ms;        /* query normalise + cast */
    double select_ms;      /* candidate merge + fp32 rescore + final top-k */
    double add_ms;         /* normalise + cast of added rows */
    double encode_ms;      /* encoder forward */
    double cache_ms;       /* cache scan */
    int64_t scan_calls;    /* main scan launches (the collect pass is not counted) */
    int64_t search_calls;
    int64_t scan_rows;     /* rows scanned by the last search */
    int64_t scan_flops;    /* 2 * rows * dim * B of the last search */
    int64_t scan_bytes;    /* algorithmic bytes of the last search */
    int64_t uncertified;   /* queries of the last search that needed the exact fp32 rescan */
    double sample_ms;      /* int8 mode: threshold pass (the int8 sample scan + order statistic, or with "i8_sample_int8" = 0 the bf16
                            * scan + fp32 re-score of the row sample); a stage of its own, not part of scan_ms */
    int64_t i8_collected;  /* int8 mode, last search: keys the collect scan appended (all queries) */
    int64_t i8_rescored;   /*   rows re-scored in fp32 */
    int64_t i8_overflows;  /*   queries whose lists or buffers overflowed (they took the bf16 pass) */
} sqe_stats_t;
int sqe_set_profiling(sqe_ctx* ctx, int on);
int sqe_stats(sqe_ctx* ctx, sqe_stats_t* out);
int sqe_stats_reset(sqe_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* SQE_H */
