// internal.h -- the objects behind the opaque handles of include/sqe.h (shared by api.hip, group.hip, ivf.hip,
// encoder.hip) and the locking / stream discipline every entry point follows.
//
// Threading model (the reference calls add_embeddings from a pool thread while search runs on the
// event loop, main.py:454-455 vs :499; ctypes drops the GIL):
//   * every object that owns device state (index, cache, encoder) has its OWN mutex and its OWN stream; there
//     is no context-wide lock, so an add on one index never blocks a search on another, the cache scan or the
//     encoder, and their kernels overlap on the device;
//   * host entry points (host pointers in, host pointers out) enqueue on the object's stream and synchronise
//     it before returning; "_device" entry points enqueue on the CONTEXT stream (sqe_stream / sqe_set_stream)
//     and do not synchronise -- that is the stream a caller orders its own work against;
//   * the operations of one object are serialised ACROSS streams by an event: an operation that runs on
//     another stream than the object's previous one first waits for that one's event (OpScope);
//   * a caller-owned stream installed with sqe_set_stream is used by every entry point of the context.
#pragma once

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "kernels.h"

namespace sqe {

// ---------------------------------------------------------------- device buffer helper
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    // grows (never shrinks); contents are NOT preserved
    int ensure(size_t need) {
        if (need <= bytes) return SQE_OK;
        release();
        hipError_t e = hipMalloc(&p, need);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(SQE_ERR_OOM, std::string("hipMalloc(") + std::to_string(need) + "): " + hipGetErrorString(e));
        }
        bytes = need;
        return SQE_OK;
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// ---------------------------------------------------------------- profiling
enum Stage { ST_SCAN = 0, ST_PREP, ST_SELECT, ST_ADD, ST_ENCODE, ST_CACHE, ST_COLLECT, ST_SAMPLE, ST_COUNT };

// hipEvent pairs around the stages, on whatever stream the stage ran on; totals are read by sqe_stats.
struct Profiler {
    std::atomic<bool> on{false};
    std::mutex mu;
    struct Pending { int stage; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    double ms[ST_COUNT] = {0};
    int64_t calls[ST_COUNT] = {0};

    hipEvent_t get() {                       // caller holds mu
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    void drain_locked() {
        for (auto& pd : pending) {
            float t = 0.f;
            if (hipEventSynchronize(pd.b) == hipSuccess && hipEventElapsedTime(&t, pd.a, pd.b) == hipSuccess) {
                ms[pd.stage] += t;
                calls[pd.stage]++;
            }
            pool.push_back(pd.a);
            pool.push_back(pd.b);
        }
        pending.clear();
    }
    void drain() {
        std::lock_guard<std::mutex> lk(mu);
        drain_locked();
    }
    ~Profiler() {
        for (auto& pd : pending) { (void)hipEventDestroy(pd.a); (void)hipEventDestroy(pd.b); }
        for (auto e : pool) (void)hipEventDestroy(e);
    }
};

struct StageTimer {
    Profiler& pf; hipStream_t s; int stage; hipEvent_t a = nullptr;
    StageTimer(Profiler& p, hipStream_t st, int stg) : pf(p), s(st), stage(stg) {
        if (pf.on.load(std::memory_order_relaxed)) {
            std::lock_guard<std::mutex> lk(pf.mu);
            if (pf.pending.size() >= 2048) pf.drain_locked();
            a = pf.get();
            (void)hipEventRecord(a, s);
        }
    }
    ~StageTimer() {
        if (a) {
            std::lock_guard<std::mutex> lk(pf.mu);
            hipEvent_t b = pf.get();
            (void)hipEventRecord(b, s);
            pf.pending.push_back({stage, a, b});
        }
    }
};

// ---------------------------------------------------------------- per-object operation order
struct OpOrder {
    std::mutex mu;                 // one operation of the object at a time (host side)
    hipStream_t own = nullptr;     // the object's stream (host entry points)
    hipEvent_t ev = nullptr;       // recorded after every operation
    hipStream_t last = nullptr;    // stream of the previous operation
    bool armed = false;            // ev has been recorded at least once
    int init() {                   // current device = the object's device
        SQE_HIP(hipStreamCreateWithFlags(&own, hipStreamNonBlocking));
        SQE_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        return SQE_OK;
    }
    // wait (host) for the object's last operation, whatever stream it ran on -- never touches a stream handle,
    // so a caller-owned stream that has died since is no hazard
    void quiesce() {
        if (armed) (void)hipEventSynchronize(ev);
    }
    void destroy() {
        quiesce();
        if (ev) (void)hipEventDestroy(ev);
        if (own) (void)hipStreamDestroy(own);
        ev = nullptr; own = nullptr; armed = false;
    }
};

struct IvfState;
struct Group;        // group.hip: the member contexts / shards of a multi-device context
struct GroupIndex;

}  // namespace sqe

// ================================================================ objects
struct sqe_ctx {
    int device = 0;
    std::atomic<hipStream_t> stream{nullptr};   // stream of the "_device" entry points (own_stream unless sqe_set_stream)
    hipStream_t own_stream = nullptr;           // created by sqe_create
    std::atomic<bool> foreign{false};           // a caller-owned stream is installed: every entry point uses it
    std::mutex mu;                              // stream swaps and the one-shot cosine scan's buffer
    sqe::OpOrder host;                          // context-level host operations (one-shot cosine scan)
    int cu_count = 256;
    int64_t hbm_bytes = 0;
    std::string name;
    sqe::Profiler prof;
    std::atomic<int64_t> last_scan_rows{0}, last_scan_flops{0}, last_scan_bytes{0}, search_calls{0};
    sqe::DevBuf unc_last;    // 16 B owned by the context: uncertified-query count of the last certified search
                             //   (copied on the search's stream; never a pointer into an index's buffers)
    std::atomic<bool> unc_valid{false};
    sqe::DevBuf i8_last;     // 32 B: keys collected / rows re-scored / overflows / uncertified of the last int8 search
    std::atomic<bool> i8_valid{false};
    sqe::DevBuf cache_tmp;   // one-shot cosine scan: matrix + q + sims + best
    sqe::Group* group = nullptr;                // n_dev > 1: this context leads a device group (group.hip)
};

struct sqe_index {
    sqe_ctx* ctx = nullptr;
    sqe::OpOrder ord;
    int dim = 0;
    int kind = SQE_INDEX_FLAT;
    int nlist = 0;
    std::atomic<int64_t> n{0};
    int64_t cap = 0;               // rows allocated (multiple of 256)
    float* master = nullptr;       // [cap, dim] fp32 normalised
    sqe::bf16_t* scan = nullptr;   // [cap] rows of dim bf16 at `pitch` bytes, zero past n
    int pitch = 0;                 // bytes between rows of the scanned copy and of the bf16 query block
    int scan_mode = SQE_SCAN_BF16_RESCORE;
    int rescore_k = 0;             // 0 = automatic
    int nprobe = 0;
    int64_t id_base = 0;           // added to returned ids (row-sharded index)
    sqe::DevBuf qn;                // [B, dim] fp32 normalised queries
    sqe::DevBuf qb;                // [b_pad, dim] bf16 queries
    sqe::DevBuf cand;              // [n_chunks, b_pad, CAND_CAP] u64
    sqe::DevBuf cand_cnt;          // [n_chunks, b_pad] int
    sqe::DevBuf gmax;              // [b_pad, ngroups, 64] u32 chunk maxima (global bound table)
    sqe::DevBuf dbg;               // 8 x u64 debug counters (knobs build, SQE_DBG bit 32)
    sqe::DevBuf resid_max;         // u32 float bits: max over rows of || x_hat - bf16(x_hat) ||
    sqe::DevBuf q_resid;           // [B] the same per query
    sqe::DevBuf unc;               // int count (16 B) | float collect_thr[b_pad]
    sqe::DevBuf fb_keys, fb_cnt;   // exact-rescan collection buffers (by compact index)
    sqe::DevBuf unc_ids, thr_c, qb_c;   // uncertified queries compacted into a dense batch: ids, thresholds, bf16 rows
    sqe::DevBuf stage_in, stage_out;    // H2D / D2H staging of the host entry points
    int certify = 1;               // run the exactness certificate + fp32 rescan fallback
    // ---- int8 first pass (scan_mode == SQE_SCAN_INT8_RESCORE; quant.hip, scan_i8.hip, select_i8.hip)
    sqe::DevBuf i8db;              // tiled int8 copy: tile t (256 rows) at t * i8_tile_stride
    sqe::DevBuf i8sxi;             // [i8_cap_tiles * 256] u32 row scales
    sqe::DevBuf i8resid_max;       // u32 float bits: max over rows of || x_hat - sxi unit x8 ||
    int64_t i8_cap_tiles = 0, i8_tile_stride = 0;
    int64_t i8_rows = 0;           // rows [0, i8_rows) of the int8 copy are current (filled lazily by the first search after an add)
    sqe::DevBuf q8, q8sqi, q8resid, i8thr_int, i8thr_eff, i8cos_s, i8ids_s, i8stats, i8samp;   // per search
    int64_t i8_min_rows = 1000000; // below this many rows (or batches <= 128, dim < 256, k > 32) the bf16 scan answers
    int i8_sample_step = 100;      // the threshold pass scans every i8_sample_step-th tile with the bf16 kernels ...
    int i8_sample_m = 20;          // ... and the collect threshold of a query is its m-th best true cosine there (~step x m = 2,000
                                   //   rows collected per query; r03 sweep, profiles/r03_search/i8_sample_sweep.log: a proof starts
                                   //   to fail when fewer than ~380 rows of a query are collected, i.e. when >= m sample rows beat
                                   //   the rank-380 score -- Poisson(3.8) >= 20: 1e-8 per query; any failure costs a 3 ms bf16 pass)
    float i8_dx = 0.f;             // host copy of the int8 residual maximum (refreshed when rows were quantised)
    bool i8_dx_stale = true;
    bool i8_oom_logged = false;    // the int8 copy did not fit: scan_mode fell back to BF16_RESCORE (api.hip: ensure_i8_copy)
    int i8_sample_int8 = 1;        // threshold pass: 1 = int8 sample scan + order statistic (r03c), 0 = bf16 scan + fp32 re-score of the sample
    double i8_max_resid = 0.02;    // rows that quantise worse than this (one element 40 x the others: 0.05 at dim 1024) would
                                   //   make every certificate fail: the index then answers with the bf16 scan
    sqe::DevBuf i8ovf, i8ovf_cnt;  // overflow pool of the collect scan: [b_pad, I8_OVF_CAP] keys, [b_pad] counts (kernels.h)
    double i8_anchor_margin = 0.25;// the collect threshold never lies above (best true cosine of the sample) - eps (1 + margin): select_i8.hip
    int i8_key_budget = 6144;      // where the sample predicts that the anchored threshold collects more keys than this, it is not used (0 = no limit)
    sqe_i8_launch_t i8_launch{};   // the last int8 search (sqe_index_i8_last); rows == 0: none yet
    sqe::IvfState* ivf = nullptr;  // kind == SQE_INDEX_IVF_FLAT
    bool internal = false;         // sub-index of another object (IVF coarse quantiser): runs under its owner's lock and stream
    sqe::GroupIndex* group = nullptr;   // index of a multi-device context: one shard per member device (group.hip)
};

struct sqe_cache {
    sqe_ctx* ctx = nullptr;
    sqe::OpOrder ord;
    int capacity = 0, dim = 0;
    sqe::DevBuf mat;     // [capacity, dim] raw fp32
    sqe::DevBuf work;    // q [dim] | sims [capacity] | best_sim | best_idx | order [capacity]
};

namespace sqe {

inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// Scope of one operation on an object: device set, object locked, stream chosen and ordered after the
// object's previous operation; the destructor records the object's event on that stream.
struct OpScope {
    OpOrder& o;
    hipStream_t s;
    bool locked;
    OpScope(sqe_ctx* c, OpOrder& ord, bool host_call, bool lock = true) : o(ord), locked(lock) {
        (void)hipSetDevice(c->device);
        if (locked) o.mu.lock();
        s = (host_call && !c->foreign.load()) ? o.own : c->stream.load();
        if (o.armed && o.last != s) (void)hipStreamWaitEvent(s, o.ev, 0);
    }
    ~OpScope() {
        (void)hipEventRecord(o.ev, s);
        o.last = s;
        o.armed = true;
        if (locked) o.mu.unlock();
    }
    OpScope(const OpScope&) = delete;
    OpScope& operator=(const OpScope&) = delete;
};

// ---- internal forms of the index operations: no locking, explicit stream (api.hip)
int index_create_impl(sqe_ctx* ctx, int dim, int kind, int nlist, bool internal, sqe_index** out);
int index_grow(sqe_index* idx, int64_t need_rows, hipStream_t s);
// rows are [n] x dim floats, `x_stride` floats apart (>= dim; a strided view of a row-major block)
int index_add_impl(sqe_index* idx, const float* x_dev, int64_t n, int64_t x_stride, bool restore, hipStream_t s);
int index_update_impl(sqe_index* idx, const int64_t* rows_dev, const float* x_dev, int64_t n, hipStream_t s);
int index_search_impl(sqe_index* idx, const float* q_dev, int B, int k, int nprobe, float* cos_out_dev, int64_t* id_out_dev,
                      hipStream_t s, int pass_index = 0);

// ---- IVF layer (ivf.hip); every call runs under the base index's lock, on stream s
int ivf_create(sqe_index* base, IvfState** out);
void ivf_destroy(IvfState* st);
int ivf_rows_added(sqe_index* base, IvfState* st, hipStream_t s);
int ivf_rows_updated(sqe_index* base, IvfState* st, const int64_t* rows_dev, int64_t n, hipStream_t s);
int ivf_train(sqe_index* base, IvfState* st, const float* x_dev, int64_t n, int iters, uint64_t seed, hipStream_t s);
int ivf_search(sqe_index* base, IvfState* st, const float* q_dev, int B, int k, int nprobe, float* cos_out, int64_t* id_out,
               hipStream_t s);
int ivf_export(sqe_index* base, IvfState* st, float* centroids_host, int32_t* assign_host, hipStream_t s);
void ivf_invalidate(IvfState* st);
sqe_index* ivf_coarse(IvfState* st);
bool ivf_trained(IvfState* st);
int ivf_restore(sqe_index* base, IvfState* st, const float* centroids_dev, const int32_t* assign_dev, int64_t n, hipStream_t s);

// ---- device groups (group.hip): n_dev > 1 contexts, one shard per member device
int group_create(sqe_ctx* leader, const int* device_ids, int n, int exchange);
void group_destroy(sqe_ctx* leader);
int group_index_create(sqe_ctx* leader, int dim, int kind, int nlist, sqe_index** out);
void group_index_destroy(sqe_index* idx);
int group_index_reserve(sqe_index* idx, int64_t rows);
int group_index_add(sqe_index* idx, const float* x, int64_t n, bool x_on_device, bool restore);
int group_index_update(sqe_index* idx, const int64_t* rows_host, const float* x_host, int64_t n);
int group_index_count(const sqe_index* idx, int64_t* out);
int group_index_get_rows(sqe_index* idx, const int64_t* rows_host, int64_t n, float* out_host);
int group_index_set_option(sqe_index* idx, const char* key, double value);
int group_index_search(sqe_index* idx, const float* q, int B, int k, int nprobe, float* cos_out, int64_t* id_out, bool on_device);
int group_index_save_rows(sqe_index* idx, FILE* f, void* pinned, size_t pinned_bytes);
int group_index_train(sqe_index* idx, const float* x, int64_t n, int iters, uint64_t seed, bool x_on_device);
int group_index_ivf_export(sqe_index* idx, float* centroids_host, int32_t* assign_host);
bool group_index_ivf_trained(sqe_index* idx);
int group_index_ivf_restore(sqe_index* idx, const float* centroids_host, const int32_t* assign_host, int64_t n);
int group_describe(sqe_ctx* leader, int* n_shards, int* exchange, int* device_ids, int cap);
int group_member_count(const sqe_ctx* leader);          // shards of the context (1 without a group)
sqe_ctx* group_member(sqe_ctx* leader, int p);          // member context p (0 = the leader itself)

}  // namespace sqe
