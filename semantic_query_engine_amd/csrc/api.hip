// api.hip -- the C ABI of include/sqe.h: context, flat vector index, search pipeline,
// cache scan and stats.  No C++ types or exceptions cross this boundary.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <new>
#include <vector>

#include "kernels.h"

namespace sqe {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

// ---------------------------------------------------------------- device buffer helper
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
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
enum Stage { ST_SCAN = 0, ST_PREP, ST_SELECT, ST_ADD, ST_ENCODE, ST_CACHE, ST_COUNT };

struct Profiler {
    bool on = false;
    struct Pending { int stage; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    double ms[ST_COUNT] = {0};
    int64_t calls[ST_COUNT] = {0};

    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    void drain(hipStream_t s) {
        if (pending.empty()) return;
        (void)hipStreamSynchronize(s);
        for (auto& pd : pending) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, pd.a, pd.b) == hipSuccess) { ms[pd.stage] += t; calls[pd.stage]++; }
            pool.push_back(pd.a);
            pool.push_back(pd.b);
        }
        pending.clear();
    }
    ~Profiler() {
        for (auto& pd : pending) { (void)hipEventDestroy(pd.a); (void)hipEventDestroy(pd.b); }
        for (auto e : pool) (void)hipEventDestroy(e);
    }
};

struct StageTimer {
    Profiler& pf; hipStream_t s; int stage; hipEvent_t a = nullptr;
    StageTimer(Profiler& p, hipStream_t st, int stg) : pf(p), s(st), stage(stg) {
        if (pf.on) {
            if (pf.pending.size() >= 2048) pf.drain(s);
            a = pf.get();
            (void)hipEventRecord(a, s);
        }
    }
    ~StageTimer() {
        if (pf.on && a) {
            hipEvent_t b = pf.get();
            (void)hipEventRecord(b, s);
            pf.pending.push_back({stage, a, b});
        }
    }
};

}  // namespace sqe

using namespace sqe;

// ================================================================ objects
namespace sqe { struct IvfState; }

struct sqe_ctx {
    int device = 0;
    hipStream_t stream = nullptr;       // stream in use
    hipStream_t own_stream = nullptr;   // created by sqe_create
    std::recursive_mutex mu;
    int cu_count = 256;
    int64_t hbm_bytes = 0;
    std::string name;
    Profiler prof;
    int64_t last_scan_rows = 0, last_scan_flops = 0, last_scan_bytes = 0, search_calls = 0;
    const int* last_unc_count = nullptr;   // device counter of the last certified search
    DevBuf stage_in;    // generic H2D staging
    DevBuf stage_out;   // generic D2H staging
    DevBuf cache_tmp;   // one-shot cosine scan: matrix + q + sims + best
};

struct sqe_index {
    sqe_ctx* ctx = nullptr;
    int dim = 0;
    int kind = SQE_INDEX_FLAT;
    int nlist = 0;
    int64_t n = 0;
    int64_t cap = 0;               // rows allocated (multiple of 256)
    float* master = nullptr;       // [cap, dim] fp32 normalised
    bf16_t* scan = nullptr;        // [cap] rows of dim bf16 at `pitch` bytes, zero past n
    int pitch = 0;                 // bytes between rows of the scanned copy and of the bf16 query block
    int scan_mode = SQE_SCAN_BF16_RESCORE;
    int rescore_k = 0;             // 0 = automatic
    int nprobe = 0;
    int64_t id_base = 0;           // added to returned ids (row-sharded index)
    DevBuf qn;                     // [B, dim] fp32 normalised queries
    DevBuf qb;                     // [b_pad, dim] bf16 queries
    DevBuf cand;                   // [n_chunks, b_pad, CAND_CAP] u64
    DevBuf cand_cnt;               // [n_chunks, b_pad] int
    DevBuf gmax;                   // [b_pad, ngroups, 64] u32 chunk maxima (global bound table)
    DevBuf dbg;                    // 8 x u64 debug counters (SQE_DBG bit 32)
    DevBuf resid_max;              // u32 float bits: max over rows of || x_hat - bf16(x_hat) ||
    DevBuf q_resid;                // [B] the same per query
    DevBuf unc;                    // int count (16 B) | float collect_thr[b_pad]
    DevBuf fb_keys, fb_cnt;        // exact-rescan collection buffers (by compact index)
    DevBuf unc_ids, thr_c, qb_c;   // uncertified queries compacted into a dense batch: ids, thresholds, bf16 rows
    int certify = 1;               // run the exactness certificate + fp32 rescan fallback
    sqe::IvfState* ivf = nullptr;  // kind == SQE_INDEX_IVF_FLAT
};

struct sqe_cache {
    sqe_ctx* ctx = nullptr;
    int capacity = 0, dim = 0;
    DevBuf mat;     // [capacity, dim] raw fp32
    DevBuf work;    // q [dim] | sims [capacity] | best_sim | best_idx | order [capacity]
};

namespace {

struct DeviceGuard {
    explicit DeviceGuard(sqe_ctx* c) { (void)hipSetDevice(c->device); }
};

#define SQE_ENTER(ctxptr)                                              \
    if (!(ctxptr)) return fail(SQE_ERR_INVALID, "null handle");        \
    std::lock_guard<std::recursive_mutex> _lk((ctxptr)->mu);           \
    DeviceGuard _dg(ctxptr)

int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

int index_grow(sqe_index* idx, int64_t need_rows) {
    if (need_rows <= idx->cap) return SQE_OK;
    sqe_ctx* c = idx->ctx;
    int64_t new_cap = std::max<int64_t>(need_rows, idx->cap + idx->cap / 2);
    new_cap = round_up(std::max<int64_t>(new_cap, 1024), SCAN_BM);
    const size_t row_f = (size_t)idx->dim * 4, row_b = (size_t)idx->pitch;
    float* nm = nullptr;
    bf16_t* ns = nullptr;
    hipError_t e = hipMalloc((void**)&nm, (size_t)new_cap * row_f);
    if (e != hipSuccess) return fail(SQE_ERR_OOM, std::string("index master alloc: ") + hipGetErrorString(e));
    e = hipMalloc((void**)&ns, (size_t)new_cap * row_b);
    if (e != hipSuccess) { (void)hipFree(nm); return fail(SQE_ERR_OOM, std::string("index scan-copy alloc: ") + hipGetErrorString(e)); }
    // rows past n of the scanned copy must read as zero (tile padding)
    SQE_HIP(hipMemsetAsync(ns, 0, (size_t)new_cap * row_b, c->stream));
    if (idx->n > 0) {
        SQE_HIP(hipMemcpyAsync(nm, idx->master, (size_t)idx->n * row_f, hipMemcpyDeviceToDevice, c->stream));
        SQE_HIP(hipMemcpyAsync(ns, idx->scan, (size_t)idx->n * row_b, hipMemcpyDeviceToDevice, c->stream));
    }
    SQE_HIP(hipStreamSynchronize(c->stream));
    if (idx->master) (void)hipFree(idx->master);
    if (idx->scan) (void)hipFree(idx->scan);
    idx->master = nm;
    idx->scan = ns;
    idx->cap = new_cap;
    return SQE_OK;
}

int auto_kp(const sqe_index* idx, int k) {
    if (idx->rescore_k > 0) return std::min(MAX_KP, std::max(idx->rescore_k, k));
    // The certificate needs the kp-th scan score to sit more than eps (~2.5e-3 at dim 1024) below the
    // k-th true cosine.  On 10M random rows the 10th -> 32nd gap is only ~3 sigma above eps (a few
    // queries per 1024 would need the fp32 rescan); 10th -> 64th makes that a 1e-5 event.
    // kp <= 64 keeps the cross-chunk bound of the scan filter (64 table columns), which halves the scan
    // time, so with the certificate guarding exactness kp stays 64 up to k = 32; beyond that the candidate
    // set has to grow with k and the scan runs on per-chunk thresholds only.
    // 128 candidates (compaction window 128) up to k = 128.  Above that the 512-slot lists leave a window of
    // only 256 - kp entries, so kp = k exactly (largest window; the certificate then fails and the collect
    // pass supplies the answer) -- far outside the reference's k = 3.
    if (idx->certify) return k <= 32 ? 64 : k <= 128 ? 128 : k;
    return std::min(MAX_KP, std::max(32, 4 * k));
}

}  // namespace

// IVF layer (ivf.hip)
namespace sqe {
struct IvfState;
int ivf_create(sqe_index* base, IvfState** out);
void ivf_destroy(IvfState* st);
int ivf_rows_added(sqe_index* base, IvfState* st);
int ivf_train(sqe_index* base, IvfState* st, const float* x_dev, int64_t n, int iters, uint64_t seed);
int ivf_search(sqe_index* base, IvfState* st, const float* q_dev, int B, int k, int nprobe, float* cos_out, int64_t* id_out);
int ivf_export(sqe_index* base, IvfState* st, float* centroids_host, int32_t* assign_host);
void ivf_invalidate(IvfState* st);
sqe_index* ivf_coarse(IvfState* st);
bool ivf_trained(IvfState* st);
int ivf_restore(sqe_index* base, IvfState* st, const float* centroids_dev, const int32_t* assign_dev, int64_t n);
}  // namespace sqe

// accessors used by encoder.hip / ivf.hip (sqe_ctx and sqe_index are defined in this file only)
namespace sqe {
float* index_master(sqe_index* idx) { return idx->master; }
const bf16_t* index_scan(sqe_index* idx) { return idx->scan; }
int index_pitch(sqe_index* idx) { return idx->pitch; }
int64_t index_rows(sqe_index* idx) { return idx->n; }
int index_dim(sqe_index* idx) { return idx->dim; }
int index_nlist(sqe_index* idx) { return idx->nlist; }
int64_t index_id_base(sqe_index* idx) { return idx->id_base; }
sqe_ctx* index_ctx(sqe_index* idx) { return idx->ctx; }
void index_clear(sqe_index* idx) { idx->n = 0; }
// append rows that are already normalised (saved index): master bit for bit, bf16 copy + residual rebuilt
int index_add_restored(sqe_index* idx, const float* x_dev, int64_t n) {
    if (n <= 0) return SQE_OK;
    SQE_TRY(index_grow(idx, idx->n + n));
    if (!idx->resid_max.p) {
        SQE_TRY(idx->resid_max.ensure(16));
        SQE_HIP(hipMemsetAsync(idx->resid_max.p, 0, 16, idx->ctx->stream));
    }
    SQE_TRY(launch_restore_rows(x_dev, n, idx->dim, idx->master + (size_t)idx->n * idx->dim,
                                idx->scan + (size_t)idx->n * (idx->pitch / 2), idx->pitch / 2,
                                idx->resid_max.as<uint32_t>(), idx->ctx->stream));
    idx->n += n;
    return SQE_OK;
}
hipStream_t ctx_stream(sqe_ctx* ctx) { return ctx->stream; }
int ctx_cu_count(sqe_ctx* ctx) { return ctx->cu_count; }
int ctx_device(sqe_ctx* ctx) { return ctx->device; }
void ctx_lock(sqe_ctx* ctx) { ctx->mu.lock(); }
void ctx_unlock(sqe_ctx* ctx) { ctx->mu.unlock(); }
}  // namespace sqe

// ================================================================ library / context
extern "C" {

int sqe_version(void) { return SQE_VERSION; }
const char* sqe_last_error(void) { return g_last_error.c_str(); }

int sqe_create(const int* device_ids, int n_dev, sqe_ctx** out) {
    if (!out) return fail(SQE_ERR_INVALID, "sqe_create: out is null");
    *out = nullptr;
    if (n_dev != 1 || !device_ids)
        return fail(SQE_ERR_INVALID, "sqe_create: one context drives one device (one process per GPU)");
    int count = 0;
    SQE_HIP(hipGetDeviceCount(&count));
    if (device_ids[0] < 0 || device_ids[0] >= count)
        return fail(SQE_ERR_INVALID, "sqe_create: no such HIP device");
    std::unique_ptr<sqe_ctx> c(new (std::nothrow) sqe_ctx);
    if (!c) return fail(SQE_ERR_OOM, "sqe_create: host allocation failed");
    c->device = device_ids[0];
    SQE_HIP(hipSetDevice(c->device));
    hipDeviceProp_t prop;
    SQE_HIP(hipGetDeviceProperties(&prop, c->device));
    c->cu_count = prop.multiProcessorCount;
    c->hbm_bytes = (int64_t)prop.totalGlobalMem;
    c->name = prop.name[0] ? prop.name : prop.gcnArchName;
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(SQE_ERR_UNSUPPORTED, std::string("libsqe is built for gfx950 only, device is ") + prop.gcnArchName);
    SQE_HIP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    *out = c.release();
    return SQE_OK;
}

void sqe_destroy(sqe_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    ctx->prof.drain(ctx->stream);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int sqe_synchronize(sqe_ctx* ctx) {
    SQE_ENTER(ctx);
    SQE_HIP(hipStreamSynchronize(ctx->stream));
    return SQE_OK;
}

void* sqe_stream(sqe_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int sqe_set_stream(sqe_ctx* ctx, void* hip_stream) {
    SQE_ENTER(ctx);
    ctx->prof.drain(ctx->stream);
    SQE_HIP(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return SQE_OK;
}

int sqe_device_info(sqe_ctx* ctx, char* name, int name_cap, int* cu_count, int64_t* hbm_bytes) {
    SQE_ENTER(ctx);
    if (name && name_cap > 0) {
        strncpy(name, ctx->name.c_str(), (size_t)name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (cu_count) *cu_count = ctx->cu_count;
    if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
    return SQE_OK;
}

// ================================================================ index
int sqe_index_create(sqe_ctx* ctx, int dim, int kind, int nlist, sqe_index** out) {
    SQE_ENTER(ctx);
    if (!out) return fail(SQE_ERR_INVALID, "sqe_index_create: out is null");
    *out = nullptr;
    if (dim <= 0 || dim % SCAN_BK != 0 || dim > 8192)
        return fail(SQE_ERR_INVALID, "sqe_index_create: dim must be a positive multiple of 64 (<= 8192)");
    if (kind != SQE_INDEX_FLAT && kind != SQE_INDEX_IVF_FLAT) return fail(SQE_ERR_INVALID, "sqe_index_create: unknown index kind");
    if (kind == SQE_INDEX_IVF_FLAT && (nlist < 1 || nlist > (1 << 20)))
        return fail(SQE_ERR_INVALID, "sqe_index_create: IVF needs 1 <= nlist <= 2^20");
    sqe_index* idx = new (std::nothrow) sqe_index;
    if (!idx) return fail(SQE_ERR_OOM, "sqe_index_create: host allocation failed");
    idx->ctx = ctx;
    idx->dim = dim;
    idx->kind = kind;
    idx->nlist = nlist;
    {
        // rows of the scanned copy are padded by one 128-B line by default: with a 2^n pitch every
        // row of a K slice would sit in the same memory channel
        const char* e = knob_env("SQE_ROW_PAD");
        const int pad = e ? atoi(e) : 128;
        idx->pitch = dim * 2 + (pad >= 0 && pad % 8 == 0 ? pad : 128);
    }
    if (kind == SQE_INDEX_IVF_FLAT) {
        int rc = ivf_create(idx, &idx->ivf);
        if (rc == SQE_OK) rc = sqe_index_set_option(ivf_coarse(idx->ivf), "certify", 0.0);   // probes are approximate by nature
        if (rc != SQE_OK) { ivf_destroy(idx->ivf); delete idx; return rc; }
    }
    *out = idx;
    return SQE_OK;
}

void sqe_index_destroy(sqe_index* idx) {
    if (!idx) return;
    if (idx->ivf) { ivf_destroy(idx->ivf); idx->ivf = nullptr; }
    {
        std::lock_guard<std::recursive_mutex> lk(idx->ctx->mu);
        (void)hipSetDevice(idx->ctx->device);
        (void)hipStreamSynchronize(idx->ctx->stream);
        if (idx->master) (void)hipFree(idx->master);
        if (idx->scan) (void)hipFree(idx->scan);
        idx->qn.release(); idx->qb.release(); idx->cand.release(); idx->cand_cnt.release(); idx->gmax.release(); idx->dbg.release(); idx->resid_max.release(); idx->q_resid.release(); idx->unc.release();
        idx->fb_keys.release(); idx->fb_cnt.release(); idx->unc_ids.release(); idx->thr_c.release(); idx->qb_c.release();
    }
    delete idx;
}

int sqe_index_reserve(sqe_index* idx, int64_t rows) {
    if (!idx) return fail(SQE_ERR_INVALID, "null index");
    SQE_ENTER(idx->ctx);
    if (rows < 0) return fail(SQE_ERR_INVALID, "sqe_index_reserve: rows < 0");
    return index_grow(idx, rows);
}

int sqe_index_add_device(sqe_index* idx, const float* x_dev, int64_t n) {
    if (!idx) return fail(SQE_ERR_INVALID, "null index");
    SQE_ENTER(idx->ctx);
    if (n < 0 || (n > 0 && !x_dev)) return fail(SQE_ERR_INVALID, "sqe_index_add: bad arguments");
    if (n == 0) return SQE_OK;
    if (idx->n + n > 0xFFFFFFF0LL) return fail(SQE_ERR_INVALID, "sqe_index_add: more than 2^32 rows per shard");
    SQE_TRY(index_grow(idx, idx->n + n));
    if (!idx->resid_max.p) {
        SQE_TRY(idx->resid_max.ensure(16));
        SQE_HIP(hipMemsetAsync(idx->resid_max.p, 0, 16, idx->ctx->stream));
    }
    {
        StageTimer t(idx->ctx->prof, idx->ctx->stream, ST_ADD);
        SQE_TRY(launch_normalize_rows(x_dev, n, idx->dim, idx->master + (size_t)idx->n * idx->dim,
                                      idx->scan + (size_t)idx->n * (idx->pitch / 2), idx->pitch / 2, nullptr,
                                      idx->resid_max.as<uint32_t>(), idx->ctx->stream));
    }
    idx->n += n;
    if (idx->ivf) SQE_TRY(ivf_rows_added(idx, idx->ivf));
    return SQE_OK;
}

int sqe_index_add(sqe_index* idx, const float* x_host, int64_t n) {
    if (!idx) return fail(SQE_ERR_INVALID, "null index");
    SQE_ENTER(idx->ctx);
    if (n < 0 || (n > 0 && !x_host)) return fail(SQE_ERR_INVALID, "sqe_index_add: bad arguments");
    if (n == 0) return SQE_OK;
    sqe_ctx* c = idx->ctx;
    SQE_TRY(index_grow(idx, idx->n + n));
    const int64_t rows_per_step = std::max<int64_t>(1, (64ll << 20) / ((int64_t)idx->dim * 4));
    SQE_TRY(c->stage_in.ensure((size_t)std::min(rows_per_step, n) * idx->dim * 4));
    for (int64_t off = 0; off < n; off += rows_per_step) {
        const int64_t m = std::min(rows_per_step, n - off);
        SQE_HIP(hipMemcpyAsync(c->stage_in.p, x_host + (size_t)off * idx->dim, (size_t)m * idx->dim * 4,
                               hipMemcpyHostToDevice, c->stream));
        SQE_TRY(sqe_index_add_device(idx, c->stage_in.as<float>(), m));
    }
    SQE_HIP(hipStreamSynchronize(c->stream));   // x_host is not retained past return
    return SQE_OK;
}

int sqe_index_update(sqe_index* idx, const int64_t* rows_host, const float* x_host, int64_t n) {
    if (!idx) return fail(SQE_ERR_INVALID, "null index");
    SQE_ENTER(idx->ctx);
    if (n < 0 || (n > 0 && (!rows_host || !x_host))) return fail(SQE_ERR_INVALID, "sqe_index_update: bad arguments");
    if (n == 0) return SQE_OK;
    for (int64_t i = 0; i < n; ++i)
        if (rows_host[i] < 0 || rows_host[i] >= idx->n) return fail(SQE_ERR_INVALID, "sqe_index_update: row out of range");
    sqe_ctx* c = idx->ctx;
    const size_t xb = (size_t)n * idx->dim * 4, rb = (size_t)n * 8;
    SQE_TRY(c->stage_in.ensure(xb + rb));
    SQE_HIP(hipMemcpyAsync(c->stage_in.p, x_host, xb, hipMemcpyHostToDevice, c->stream));
    SQE_HIP(hipMemcpyAsync((char*)c->stage_in.p + xb, rows_host, rb, hipMemcpyHostToDevice, c->stream));
    SQE_TRY(launch_normalize_rows_scatter(c->stage_in.as<float>(), (const int64_t*)((char*)c->stage_in.p + xb), n,
                                          idx->dim, idx->master, idx->scan, idx->pitch / 2, idx->resid_max.as<uint32_t>(), c->stream));
    SQE_HIP(hipStreamSynchronize(c->stream));
    if (idx->ivf) ivf_invalidate(idx->ivf);       // overwritten rows are re-assigned at the next search
    return SQE_OK;
}

int sqe_index_count(const sqe_index* idx, int64_t* out) {
    if (!idx || !out) return fail(SQE_ERR_INVALID, "sqe_index_count: null argument");
    std::lock_guard<std::recursive_mutex> lk(idx->ctx->mu);
    *out = idx->n;
    return SQE_OK;
}

int sqe_index_get_rows(sqe_index* idx, const int64_t* rows_host, int64_t n, float* out_host) {
    if (!idx) return fail(SQE_ERR_INVALID, "null index");
    SQE_ENTER(idx->ctx);
    if (n < 0 || (n > 0 && (!rows_host || !out_host))) return fail(SQE_ERR_INVALID, "sqe_index_get_rows: bad arguments");
    for (int64_t i = 0; i < n; ++i) {
        if (rows_host[i] < 0 || rows_host[i] >= idx->n) return fail(SQE_ERR_INVALID, "sqe_index_get_rows: row out of range");
        SQE_HIP(hipMemcpyAsync(out_host + (size_t)i * idx->dim, idx->master + (size_t)rows_host[i] * idx->dim,
                               (size_t)idx->dim * 4, hipMemcpyDeviceToHost, idx->ctx->stream));
    }
    SQE_HIP(hipStreamSynchronize(idx->ctx->stream));
    return SQE_OK;
}

int sqe_index_set_option(sqe_index* idx, const char* key, double value) {
    if (!idx || !key) return fail(SQE_ERR_INVALID, "sqe_index_set_option: null argument");
    SQE_ENTER(idx->ctx);
    const std::string k(key);
    if (k == "scan_mode") {
        if ((int)value != SQE_SCAN_BF16_RESCORE)
            return fail(SQE_ERR_UNSUPPORTED, "scan_mode: only SQE_SCAN_BF16_RESCORE is implemented in this build");
        idx->scan_mode = (int)value;
    } else if (k == "rescore_k") {
        if (value < 0 || value > MAX_KP) return fail(SQE_ERR_INVALID, "rescore_k must be in [0, 256]");
        idx->rescore_k = (int)value;
    } else if (k == "nprobe") {
        idx->nprobe = (int)value;
    } else if (k == "certify") {
        idx->certify = value != 0.0;
    } else if (k == "id_base") {
        if (value < 0) return fail(SQE_ERR_INVALID, "id_base must be >= 0");
        idx->id_base = (int64_t)value;
    } else {
        return fail(SQE_ERR_INVALID, "unknown option: " + k);
    }
    return SQE_OK;
}

int sqe_index_search_device(sqe_index* idx, const float* q_dev, int B, int k, int nprobe,
                            float* cos_out_dev, int64_t* id_out_dev) {
    (void)nprobe;
    if (!idx) return fail(SQE_ERR_INVALID, "null index");
    SQE_ENTER(idx->ctx);
    if (B < 0 || k < 1 || k > MAX_KP) return fail(SQE_ERR_INVALID, "sqe_index_search: need B >= 0 and 1 <= k <= 256");
    if (B == 0) return SQE_OK;
    if (!q_dev || !cos_out_dev || !id_out_dev) return fail(SQE_ERR_INVALID, "sqe_index_search: null buffer");
    if (idx->ivf) {
        StageTimer t(idx->ctx->prof, idx->ctx->stream, ST_SCAN);
        SQE_TRY(ivf_search(idx, idx->ivf, q_dev, B, k, nprobe > 0 ? nprobe : idx->nprobe, cos_out_dev, id_out_dev));
        idx->ctx->search_calls++;
        return SQE_OK;
    }
    sqe_ctx* c = idx->ctx;
    const int K = idx->dim;
    // More than four 256-query blocks would leave fewer than 64 DB chunks (one workgroup per CU), too few to
    // fill a row of the global-bound table: the filter would lose its cross-chunk threshold.  Larger batches
    // run as passes of 1024 queries, each at the full-batch rate.
    constexpr int MAX_PASS = 1024;
    if (B > MAX_PASS) {
        for (int off = 0; off < B; off += MAX_PASS) {
            const int m = std::min(MAX_PASS, B - off);
            SQE_TRY(sqe_index_search_device(idx, q_dev + (size_t)off * K, m, k, nprobe, cos_out_dev + (size_t)off * k,
                                            id_out_dev + (size_t)off * k));
        }
        return SQE_OK;
    }
    const int kp = auto_kp(idx, k);
    const ScanPlan plan = make_scan_plan(idx->n, B, kp, c->cu_count);

    SQE_TRY(idx->qn.ensure((size_t)B * K * 4));
    SQE_TRY(idx->qb.ensure((size_t)plan.b_pad * idx->pitch));
    SQE_TRY(idx->cand.ensure((size_t)plan.n_chunks * plan.b_pad * CAND_CAP * 8));
    SQE_TRY(idx->cand_cnt.ensure((size_t)plan.n_chunks * plan.b_pad * 4));
    const size_t gmax_bytes = (size_t)plan.b_pad * plan.ngroups * GMAX_COLS * 4;
    SQE_TRY(idx->gmax.ensure(gmax_bytes));
    const bool certify = idx->certify && idx->n > 0;
    SQE_TRY(idx->q_resid.ensure((size_t)B * 4));
    if (certify) {
        SQE_TRY(idx->unc.ensure(16 + (size_t)plan.b_pad * 4));
        SQE_TRY(idx->fb_keys.ensure((size_t)B * EXACT_CAP * 8));
        SQE_TRY(idx->fb_cnt.ensure((size_t)B * 4));
        SQE_TRY(idx->unc_ids.ensure((size_t)B * 4));
        SQE_TRY(idx->thr_c.ensure((size_t)(plan.b_pad + 256) * 4));
        if ((size_t)(plan.b_pad + 256) * idx->pitch > idx->qb_c.bytes) {
            SQE_TRY(idx->qb_c.ensure((size_t)(plan.b_pad + 256) * idx->pitch));
            SQE_HIP(hipMemsetAsync(idx->qb_c.p, 0, idx->qb_c.bytes, c->stream));     // rows past the count read as zero
        }
    }
    {
        StageTimer t(c->prof, c->stream, ST_PREP);
        if (plan.b_pad > B)
            SQE_HIP(hipMemsetAsync(idx->qb.as<char>() + (size_t)B * idx->pitch, 0, (size_t)(plan.b_pad - B) * idx->pitch, c->stream));
        SQE_TRY(launch_normalize_rows(q_dev, B, K, idx->qn.as<float>(), idx->qb.as<bf16_t>(), idx->pitch / 2,
                                      idx->q_resid.as<float>(), nullptr, c->stream));
        if (certify) {
            SQE_HIP(hipMemsetAsync(idx->unc.p, 0, 16, c->stream));
            SQE_HIP(hipMemsetAsync(idx->fb_cnt.p, 0, (size_t)B * 4, c->stream));
        }
        SQE_HIP(hipMemsetAsync(idx->gmax.p, 0, gmax_bytes, c->stream));
    }
    if (idx->n > 0) {
        StageTimer t(c->prof, c->stream, ST_SCAN);
        ScanArgs a;
        a.db = idx->scan; a.q = idx->qb.as<bf16_t>(); a.n_rows = idx->n; a.K = K; a.B = B;
        a.db_pitch = idx->pitch; a.q_pitch = idx->pitch;
        a.cand = idx->cand.as<uint64_t>(); a.cand_cnt = idx->cand_cnt.as<int>(); a.gmax = idx->gmax.as<uint32_t>();
        a.dbg_counters = nullptr;
        a.collect_thr = nullptr; a.collect_keys = nullptr; a.collect_cnt = nullptr; a.unc_count = nullptr;
        {
            static const bool want = [] { const char* e = knob_env("SQE_DBG"); return e && (atoi(e) & 32); }();
            if (want) {
                SQE_TRY(idx->dbg.ensure(64));
                SQE_HIP(hipMemsetAsync(idx->dbg.p, 0, 64, c->stream));
                a.dbg_counters = idx->dbg.as<unsigned long long>();
            }
        }
        SQE_TRY(launch_scan_bf16(plan, a, c->stream));
    } else {
        SQE_HIP(hipMemsetAsync(idx->cand_cnt.p, 0, (size_t)plan.n_chunks * plan.b_pad * 4, c->stream));
    }
    {
        StageTimer t(c->prof, c->stream, ST_SELECT);
        SelectArgs s;
        s.cand = idx->cand.as<uint64_t>(); s.cand_cnt = idx->cand_cnt.as<int>();
        s.n_chunks = plan.n_chunks; s.b_pad = plan.b_pad; s.kp = kp;
        s.master = idx->master; s.qn = idx->qn.as<float>(); s.K = K; s.B = B; s.k = k;
        s.cos_out = cos_out_dev; s.id_out = id_out_dev; s.id_base = idx->id_base;
        int* unc_count = certify ? idx->unc.as<int>() : nullptr;
        float* collect_thr = certify ? reinterpret_cast<float*>(idx->unc.as<int>() + 4) : nullptr;
        s.q_resid = certify ? idx->q_resid.as<float>() : nullptr;
        s.db_resid_max = certify ? idx->resid_max.as<uint32_t>() : nullptr;
        s.unc_count = unc_count; s.collect_thr = collect_thr;
        SQE_TRY(launch_select_rescore(s, c->stream));
        if (certify) {
            // second pass for the queries whose certificate failed.  They are compacted into a dense batch on
            // the device (no host round trip): the collect scan then costs what a batch of that size costs.
            // Both scan shapes are enqueued; each returns at once unless the count is in its range
            // (1..64: the HBM-bound 64-query kernel, more: 256-query blocks), and at once when it is 0.
            const int thr_cap = plan.b_pad + 256;
            SQE_TRY(launch_compact_uncertified(collect_thr, B, idx->qb.as<bf16_t>(), idx->pitch, K * 2, idx->unc_ids.as<int>(),
                                               idx->thr_c.as<float>(), thr_cap, idx->qb_c.as<bf16_t>(), unc_count, c->stream));
            ScanArgs a;
            a.db = idx->scan; a.q = idx->qb_c.as<bf16_t>(); a.n_rows = idx->n; a.K = K; a.B = B;
            a.db_pitch = idx->pitch; a.q_pitch = idx->pitch;
            a.cand = idx->cand.as<uint64_t>(); a.cand_cnt = idx->cand_cnt.as<int>(); a.gmax = idx->gmax.as<uint32_t>();
            a.dbg_counters = nullptr;
            a.collect_thr = idx->thr_c.as<float>(); a.collect_keys = idx->fb_keys.as<uint64_t>(); a.collect_cnt = idx->fb_cnt.as<int>();
            a.unc_count = unc_count;
            // one plan per range of counts, each sized like a search of that batch (all CUs busy in every case)
            const int bounds[5] = {0, 64, 256, 512, 1 << 30};
            for (int r = 0; r < 4 && bounds[r] < B; ++r) {
                const int hi = std::min(bounds[r + 1], B);
                const ScanPlan cp = make_scan_plan(idx->n, hi, kp, c->cu_count);
                a.collect_lo = bounds[r] + 1;
                a.collect_hi = r == 3 ? (1 << 30) : bounds[r + 1];
                SQE_TRY(launch_scan_collect(cp, a, c->stream));
            }
            // ... and re-score them in fp32
            ExactArgs e;
            e.master = idx->master; e.qn = idx->qn.as<float>(); e.K = K; e.B = B; e.k = k;
            e.collect_thr = collect_thr; e.keys = idx->fb_keys.as<uint64_t>(); e.key_cnt = idx->fb_cnt.as<int>();
            e.unc_ids = idx->unc_ids.as<int>(); e.unc_count = unc_count;
            e.cos_out = cos_out_dev; e.id_out = id_out_dev; e.id_base = idx->id_base;
            SQE_TRY(launch_collect_rescore(e, c->stream));
            c->last_unc_count = unc_count;
        }
    }
    if (idx->dbg.p) {
        unsigned long long h[8];
        SQE_HIP(hipMemcpyAsync(h, idx->dbg.p, 64, hipMemcpyDeviceToHost, c->stream));
        SQE_HIP(hipStreamSynchronize(c->stream));
        fprintf(stderr, "[sqe dbg] appends=%llu slow_path_entries=%llu compactions=%llu\n", h[0], h[1], h[2]);
    }
    c->search_calls++;
    c->last_scan_rows = idx->n;
    c->last_scan_flops = 2 * idx->n * (int64_t)K * B;
    c->last_scan_bytes = idx->n * (int64_t)K * 2 + (int64_t)B * K * 4 + (int64_t)B * k * 12;
    return SQE_OK;
}

int sqe_index_search(sqe_index* idx, const float* q_host, int B, int k, int nprobe,
                     float* cos_out_host, int64_t* id_out_host) {
    if (!idx) return fail(SQE_ERR_INVALID, "null index");
    SQE_ENTER(idx->ctx);
    if (B < 0 || k < 1 || k > MAX_KP) return fail(SQE_ERR_INVALID, "sqe_index_search: need B >= 0 and 1 <= k <= 256");
    if (B == 0) return SQE_OK;
    if (!q_host || !cos_out_host || !id_out_host) return fail(SQE_ERR_INVALID, "sqe_index_search: null buffer");
    sqe_ctx* c = idx->ctx;
    const size_t qbytes = (size_t)B * idx->dim * 4, cb = (size_t)B * k * 4, ib = (size_t)B * k * 8;
    SQE_TRY(c->stage_in.ensure(qbytes));
    SQE_TRY(c->stage_out.ensure(round_up((int64_t)cb, 16) + ib));
    float* cos_dev = c->stage_out.as<float>();
    int64_t* id_dev = reinterpret_cast<int64_t*>(c->stage_out.as<char>() + round_up((int64_t)cb, 16));
    SQE_HIP(hipMemcpyAsync(c->stage_in.p, q_host, qbytes, hipMemcpyHostToDevice, c->stream));
    SQE_TRY(sqe_index_search_device(idx, c->stage_in.as<float>(), B, k, nprobe, cos_dev, id_dev));
    SQE_HIP(hipMemcpyAsync(cos_out_host, cos_dev, cb, hipMemcpyDeviceToHost, c->stream));
    SQE_HIP(hipMemcpyAsync(id_out_host, id_dev, ib, hipMemcpyDeviceToHost, c->stream));
    SQE_HIP(hipStreamSynchronize(c->stream));
    return SQE_OK;
}

int sqe_index_train_device(sqe_index* idx, const float* x_dev, int64_t n, int iters, uint64_t seed) {
    if (!idx) return fail(SQE_ERR_INVALID, "null index");
    SQE_ENTER(idx->ctx);
    if (!idx->ivf) return fail(SQE_ERR_STATE, "sqe_index_train: not an IVF index");
    if (!x_dev || n <= 0) return fail(SQE_ERR_INVALID, "sqe_index_train: empty training set");
    return ivf_train(idx, idx->ivf, x_dev, n, iters, seed);
}

int sqe_index_train(sqe_index* idx, const float* x_host, int64_t n, int iters, uint64_t seed) {
    if (!idx) return fail(SQE_ERR_INVALID, "null index");
    SQE_ENTER(idx->ctx);
    if (!idx->ivf) return fail(SQE_ERR_STATE, "sqe_index_train: not an IVF index");
    if (!x_host || n <= 0) return fail(SQE_ERR_INVALID, "sqe_index_train: empty training set");
    DevBuf tmp;
    SQE_TRY(tmp.ensure((size_t)n * idx->dim * 4));
    SQE_HIP(hipMemcpyAsync(tmp.p, x_host, (size_t)n * idx->dim * 4, hipMemcpyHostToDevice, idx->ctx->stream));
    SQE_TRY(ivf_train(idx, idx->ivf, tmp.as<float>(), n, iters, seed));
    SQE_HIP(hipStreamSynchronize(idx->ctx->stream));
    return SQE_OK;
}

int sqe_index_ivf_export(sqe_index* idx, float* centroids_host, int32_t* assign_host) {
    if (!idx) return fail(SQE_ERR_INVALID, "null index");
    SQE_ENTER(idx->ctx);
    if (!idx->ivf) return fail(SQE_ERR_STATE, "sqe_index_ivf_export: not an IVF index");
    return ivf_export(idx, idx->ivf, centroids_host, assign_host);
}

// ---------------------------------------------------------------- persistence
// File: 64-byte header | master rows [n, dim] fp32 | (IVF, trained) centroids [nlist, dim] fp32 | assign [n] int32.
// The bf16 scan copy, residuals and IVF lists are derived data and are rebuilt on load.
namespace {
struct SaveHeader {
    char magic[8];          // "SQEIDX01"
    uint32_t version, dim, kind, nlist;
    int64_t n, id_base;
    uint32_t flags;         // bit 0: IVF centroids + assignments follow
    uint32_t certify;
    uint8_t pad[16];
};
static_assert(sizeof(SaveHeader) == 64, "header layout");
constexpr size_t IO_CHUNK = 64u << 20;

struct PinnedBuf {
    void* p = nullptr;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    int alloc(size_t bytes) {
        hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
        if (e != hipSuccess) return fail(SQE_ERR_OOM, std::string("pinned staging: ") + hipGetErrorString(e));
        return SQE_OK;
    }
};
struct FileCloser {
    FILE* f;
    ~FileCloser() { if (f) fclose(f); }
};

int write_device_range(FILE* f, const void* dev, size_t bytes, void* pinned, hipStream_t s) {
    for (size_t off = 0; off < bytes; off += IO_CHUNK) {
        const size_t m = std::min(IO_CHUNK, bytes - off);
        SQE_HIP(hipMemcpyAsync(pinned, (const char*)dev + off, m, hipMemcpyDeviceToHost, s));
        SQE_HIP(hipStreamSynchronize(s));
        if (fwrite(pinned, 1, m, f) != m) return fail(SQE_ERR_IO, "sqe_index_save: short write");
    }
    return SQE_OK;
}
}  // namespace

int sqe_index_save(sqe_index* idx, const char* path) {
    if (!idx || !path) return fail(SQE_ERR_INVALID, "sqe_index_save: null argument");
    SQE_ENTER(idx->ctx);
    sqe_ctx* c = idx->ctx;
    const bool ivf = idx->ivf && ivf_trained(idx->ivf);
    if (ivf) SQE_TRY(ivf_rows_added(idx, idx->ivf));
    FileCloser fc{fopen(path, "wb")};
    if (!fc.f) return fail(SQE_ERR_IO, std::string("sqe_index_save: cannot open ") + path);
    SaveHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "SQEIDX01", 8);
    h.version = 1; h.dim = (uint32_t)idx->dim; h.kind = (uint32_t)idx->kind; h.nlist = (uint32_t)idx->nlist;
    h.n = idx->n; h.id_base = idx->id_base; h.flags = ivf ? 1u : 0u; h.certify = (uint32_t)idx->certify;
    if (fwrite(&h, 1, sizeof(h), fc.f) != sizeof(h)) return fail(SQE_ERR_IO, "sqe_index_save: short write");
    PinnedBuf pin;
    SQE_TRY(pin.alloc(IO_CHUNK));
    SQE_HIP(hipStreamSynchronize(c->stream));
    SQE_TRY(write_device_range(fc.f, idx->master, (size_t)idx->n * idx->dim * 4, pin.p, c->stream));
    if (ivf) {
        std::vector<float> cent((size_t)idx->nlist * idx->dim);
        std::vector<int32_t> assign((size_t)std::max<int64_t>(idx->n, 1));
        SQE_TRY(ivf_export(idx, idx->ivf, cent.data(), assign.data()));
        if (fwrite(cent.data(), 4, cent.size(), fc.f) != cent.size()) return fail(SQE_ERR_IO, "sqe_index_save: short write");
        if (idx->n > 0 && fwrite(assign.data(), 4, (size_t)idx->n, fc.f) != (size_t)idx->n)
            return fail(SQE_ERR_IO, "sqe_index_save: short write");
    }
    if (fflush(fc.f) != 0) return fail(SQE_ERR_IO, "sqe_index_save: flush failed");
    return SQE_OK;
}

int sqe_index_load(sqe_ctx* ctx, const char* path, sqe_index** out) {
    if (!ctx || !path || !out) return fail(SQE_ERR_INVALID, "sqe_index_load: null argument");
    SQE_ENTER(ctx);
    FileCloser fc{fopen(path, "rb")};
    if (!fc.f) return fail(SQE_ERR_IO, std::string("sqe_index_load: cannot open ") + path);
    SaveHeader h;
    if (fread(&h, 1, sizeof(h), fc.f) != sizeof(h) || memcmp(h.magic, "SQEIDX01", 8) != 0 || h.version != 1)
        return fail(SQE_ERR_IO, "sqe_index_load: not a saved index (bad header)");
    if (h.n < 0 || h.dim == 0 || h.dim % 64 != 0) return fail(SQE_ERR_IO, "sqe_index_load: corrupt header");
    sqe_index* idx = nullptr;
    SQE_TRY(sqe_index_create(ctx, (int)h.dim, (int)h.kind, (int)h.nlist, &idx));
    struct Guard {
        sqe_index* i;
        ~Guard() { if (i) sqe_index_destroy(i); }
    } guard{idx};
    idx->id_base = h.id_base;
    idx->certify = (int)h.certify;
    SQE_TRY(index_grow(idx, h.n));
    PinnedBuf pin;
    SQE_TRY(pin.alloc(IO_CHUNK));
    const size_t row_bytes = (size_t)h.dim * 4;
    const int64_t rows_per_step = std::max<int64_t>(1, (int64_t)(IO_CHUNK / row_bytes));
    SQE_TRY(ctx->stage_in.ensure((size_t)std::min<int64_t>(rows_per_step, std::max<int64_t>(h.n, 1)) * row_bytes));
    for (int64_t off = 0; off < h.n; off += rows_per_step) {
        const int64_t m = std::min(rows_per_step, h.n - off);
        if (fread(pin.p, row_bytes, (size_t)m, fc.f) != (size_t)m) return fail(SQE_ERR_IO, "sqe_index_load: file is truncated");
        SQE_HIP(hipMemcpyAsync(ctx->stage_in.p, pin.p, (size_t)m * row_bytes, hipMemcpyHostToDevice, ctx->stream));
        SQE_TRY(index_add_restored(idx, ctx->stage_in.as<float>(), m));
        SQE_HIP(hipStreamSynchronize(ctx->stream));    // the pinned buffer is reused
    }
    if (h.flags & 1u) {
        if (!idx->ivf) return fail(SQE_ERR_IO, "sqe_index_load: IVF section in a flat index file");
        const size_t cb = (size_t)h.nlist * row_bytes, ab = (size_t)h.n * 4;
        std::vector<char> host(cb + ab);
        if (fread(host.data(), 1, cb + ab, fc.f) != cb + ab) return fail(SQE_ERR_IO, "sqe_index_load: file is truncated");
        DevBuf tmp;
        SQE_TRY(tmp.ensure(cb + std::max<size_t>(ab, 4)));
        SQE_HIP(hipMemcpyAsync(tmp.p, host.data(), cb + ab, hipMemcpyHostToDevice, ctx->stream));
        SQE_TRY(ivf_restore(idx, idx->ivf, tmp.as<float>(), (const int32_t*)((char*)tmp.p + cb), h.n));
        SQE_HIP(hipStreamSynchronize(ctx->stream));
    }
    guard.i = nullptr;
    *out = idx;
    return SQE_OK;
}

int sqe_merge_topk_device(sqe_ctx* ctx, const float* cos_parts_dev, const int64_t* id_parts_dev,
                          int64_t part_stride_bytes, int P, int B, int k,
                          float* cos_out_dev, int64_t* id_out_dev) {
    SQE_ENTER(ctx);
    if (!cos_parts_dev || !id_parts_dev || !cos_out_dev || !id_out_dev)
        return fail(SQE_ERR_INVALID, "sqe_merge_topk: null buffer");
    if (part_stride_bytes < 0 || part_stride_bytes % 8 != 0) return fail(SQE_ERR_INVALID, "sqe_merge_topk: bad part stride");
    return launch_merge_topk(cos_parts_dev, id_parts_dev, part_stride_bytes, P, B, k, cos_out_dev, id_out_dev, ctx->stream);
}

// ================================================================ cache scan
static int cosine_scan_host(sqe_ctx* ctx, const float* mat_host, int m, int dim, const float* q_host,
                            float* sims_out_host, float* best_sim, int32_t* best_idx) {
    if (m < 0 || dim <= 0 || dim % 4 != 0) return fail(SQE_ERR_INVALID, "cosine scan: bad m/dim");
    if (!q_host || (m > 0 && !mat_host)) return fail(SQE_ERR_INVALID, "cosine scan: null buffer");
    const size_t mb = (size_t)m * dim * 4, qb = (size_t)dim * 4, sb = round_up((int64_t)m * 4 + 4, 16);
    SQE_TRY(ctx->cache_tmp.ensure(mb + qb + sb + 16));
    char* base = ctx->cache_tmp.as<char>();
    float* d_mat = (float*)base;
    float* d_q = (float*)(base + mb);
    float* d_sims = (float*)(base + mb + qb);
    float* d_best = (float*)(base + mb + qb + sb);
    int32_t* d_idx = (int32_t*)(base + mb + qb + sb + 4);
    if (m > 0) SQE_HIP(hipMemcpyAsync(d_mat, mat_host, mb, hipMemcpyHostToDevice, ctx->stream));
    SQE_HIP(hipMemcpyAsync(d_q, q_host, qb, hipMemcpyHostToDevice, ctx->stream));
    {
        StageTimer t(ctx->prof, ctx->stream, ST_CACHE);
        SQE_TRY(launch_cosine_scan(d_mat, nullptr, m, dim, d_q, d_sims, d_best, d_idx, ctx->stream));
    }
    if (sims_out_host && m > 0)
        SQE_HIP(hipMemcpyAsync(sims_out_host, d_sims, (size_t)m * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (best_sim) SQE_HIP(hipMemcpyAsync(best_sim, d_best, 4, hipMemcpyDeviceToHost, ctx->stream));
    if (best_idx) SQE_HIP(hipMemcpyAsync(best_idx, d_idx, 4, hipMemcpyDeviceToHost, ctx->stream));
    SQE_HIP(hipStreamSynchronize(ctx->stream));
    return SQE_OK;
}

int sqe_cosine_best(sqe_ctx* ctx, const float* mat_host, int m, int dim, const float* q_host,
                    float* best_sim, int32_t* best_idx) {
    SQE_ENTER(ctx);
    if (!best_sim || !best_idx) return fail(SQE_ERR_INVALID, "sqe_cosine_best: null output");
    return cosine_scan_host(ctx, mat_host, m, dim, q_host, nullptr, best_sim, best_idx);
}

int sqe_cosine_all(sqe_ctx* ctx, const float* mat_host, int m, int dim, const float* q_host,
                   float* sims_out_host) {
    SQE_ENTER(ctx);
    if (m > 0 && !sims_out_host) return fail(SQE_ERR_INVALID, "sqe_cosine_all: null output");
    float bs; int32_t bi;
    return cosine_scan_host(ctx, mat_host, m, dim, q_host, sims_out_host, &bs, &bi);
}

int sqe_cache_create(sqe_ctx* ctx, int capacity, int dim, sqe_cache** out) {
    SQE_ENTER(ctx);
    if (!out) return fail(SQE_ERR_INVALID, "sqe_cache_create: out is null");
    *out = nullptr;
    if (capacity <= 0 || dim <= 0 || dim % 4 != 0) return fail(SQE_ERR_INVALID, "sqe_cache_create: bad capacity/dim");
    std::unique_ptr<sqe_cache> c(new (std::nothrow) sqe_cache);
    if (!c) return fail(SQE_ERR_OOM, "sqe_cache_create: host allocation failed");
    c->ctx = ctx; c->capacity = capacity; c->dim = dim;
    SQE_TRY(c->mat.ensure((size_t)capacity * dim * 4));
    SQE_TRY(c->work.ensure((size_t)dim * 4 + (size_t)capacity * 8 + 64));
    SQE_HIP(hipMemsetAsync(c->mat.p, 0, (size_t)capacity * dim * 4, ctx->stream));
    *out = c.release();
    return SQE_OK;
}

void sqe_cache_destroy(sqe_cache* c) {
    if (!c) return;
    {
        std::lock_guard<std::recursive_mutex> lk(c->ctx->mu);
        (void)hipSetDevice(c->ctx->device);
        (void)hipStreamSynchronize(c->ctx->stream);
        c->mat.release(); c->work.release();
    }
    delete c;
}

int sqe_cache_set_slot(sqe_cache* c, int slot, const float* vec_host) {
    if (!c) return fail(SQE_ERR_INVALID, "null cache");
    SQE_ENTER(c->ctx);
    if (slot < 0 || slot >= c->capacity || !vec_host) return fail(SQE_ERR_INVALID, "sqe_cache_set_slot: bad slot");
    SQE_HIP(hipMemcpyAsync(c->mat.as<float>() + (size_t)slot * c->dim, vec_host, (size_t)c->dim * 4,
                           hipMemcpyHostToDevice, c->ctx->stream));
    SQE_HIP(hipStreamSynchronize(c->ctx->stream));
    return SQE_OK;
}

int sqe_cache_best(sqe_cache* c, const int32_t* order_host, int m, const float* q_host,
                   float* best_sim, int32_t* best_pos) {
    if (!c) return fail(SQE_ERR_INVALID, "null cache");
    SQE_ENTER(c->ctx);
    if (m < 0 || m > c->capacity || !q_host || !best_sim || !best_pos || (m > 0 && !order_host))
        return fail(SQE_ERR_INVALID, "sqe_cache_best: bad arguments");
    for (int i = 0; i < m; ++i)
        if (order_host[i] < 0 || order_host[i] >= c->capacity) return fail(SQE_ERR_INVALID, "sqe_cache_best: slot out of range");
    sqe_ctx* ctx = c->ctx;
    char* base = c->work.as<char>();
    float* d_q = (float*)base;
    float* d_sims = (float*)(base + (size_t)c->dim * 4);
    int32_t* d_order = (int32_t*)(base + (size_t)c->dim * 4 + (size_t)c->capacity * 4);
    float* d_best = (float*)(base + (size_t)c->dim * 4 + (size_t)c->capacity * 8);
    int32_t* d_idx = (int32_t*)(d_best + 1);
    SQE_HIP(hipMemcpyAsync(d_q, q_host, (size_t)c->dim * 4, hipMemcpyHostToDevice, ctx->stream));
    if (m > 0) SQE_HIP(hipMemcpyAsync(d_order, order_host, (size_t)m * 4, hipMemcpyHostToDevice, ctx->stream));
    {
        StageTimer t(ctx->prof, ctx->stream, ST_CACHE);
        SQE_TRY(launch_cosine_scan(c->mat.as<float>(), d_order, m, c->dim, d_q, d_sims, d_best, d_idx, ctx->stream));
    }
    SQE_HIP(hipMemcpyAsync(best_sim, d_best, 4, hipMemcpyDeviceToHost, ctx->stream));
    SQE_HIP(hipMemcpyAsync(best_pos, d_idx, 4, hipMemcpyDeviceToHost, ctx->stream));
    SQE_HIP(hipStreamSynchronize(ctx->stream));
    return SQE_OK;
}

// ================================================================ stats
int sqe_set_profiling(sqe_ctx* ctx, int on) {
    SQE_ENTER(ctx);
    ctx->prof.drain(ctx->stream);
    ctx->prof.on = on != 0;
    return SQE_OK;
}

int sqe_stats(sqe_ctx* ctx, sqe_stats_t* out) {
    SQE_ENTER(ctx);
    if (!out) return fail(SQE_ERR_INVALID, "sqe_stats: out is null");
    ctx->prof.drain(ctx->stream);
    memset(out, 0, sizeof(*out));
    out->scan_ms = ctx->prof.ms[ST_SCAN];
    out->prep_ms = ctx->prof.ms[ST_PREP];
    out->select_ms = ctx->prof.ms[ST_SELECT];
    out->add_ms = ctx->prof.ms[ST_ADD];
    out->encode_ms = ctx->prof.ms[ST_ENCODE];
    out->cache_ms = ctx->prof.ms[ST_CACHE];
    out->scan_calls = ctx->prof.calls[ST_SCAN];
    out->search_calls = ctx->search_calls;
    out->scan_rows = ctx->last_scan_rows;
    out->scan_flops = ctx->last_scan_flops;
    out->scan_bytes = ctx->last_scan_bytes;
    if (ctx->last_unc_count) {
        int v = 0;
        if (hipMemcpy(&v, ctx->last_unc_count, 4, hipMemcpyDeviceToHost) == hipSuccess) out->uncertified = v;
    }
    return SQE_OK;
}

int sqe_stats_reset(sqe_ctx* ctx) {
    SQE_ENTER(ctx);
    ctx->prof.drain(ctx->stream);
    for (int i = 0; i < ST_COUNT; ++i) { ctx->prof.ms[i] = 0; ctx->prof.calls[i] = 0; }
    ctx->search_calls = 0;
    return SQE_OK;
}

}  // extern "C"
