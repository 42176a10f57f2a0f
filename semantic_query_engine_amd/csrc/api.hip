// api.hip -- the C ABI of include/sqe.h: context, flat vector index, search pipeline,
// cache scan and stats.  No C++ types or exceptions cross this boundary.  Locking and stream
// discipline: internal.h.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <new>
#include <vector>

#include "internal.h"

namespace sqe {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

}  // namespace sqe

using namespace sqe;

namespace {

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

namespace sqe {

// Rows allocated for the index (never shrinks).  Runs on the operation's stream, which OpScope has ordered
// after every earlier operation of this index on any stream: once the copy on `s` has finished nothing
// can still read the old buffers.
int index_grow(sqe_index* idx, int64_t need_rows, hipStream_t s) {
    if (need_rows <= idx->cap) return SQE_OK;
    const int64_t n = idx->n.load();
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
    SQE_HIP(hipMemsetAsync(ns, 0, (size_t)new_cap * row_b, s));
    if (n > 0) {
        SQE_HIP(hipMemcpyAsync(nm, idx->master, (size_t)n * row_f, hipMemcpyDeviceToDevice, s));
        SQE_HIP(hipMemcpyAsync(ns, idx->scan, (size_t)n * row_b, hipMemcpyDeviceToDevice, s));
    }
    SQE_HIP(hipStreamSynchronize(s));
    if (idx->master) (void)hipFree(idx->master);
    if (idx->scan) (void)hipFree(idx->scan);
    idx->master = nm;
    idx->scan = ns;
    idx->cap = new_cap;
    return SQE_OK;
}

// append rows (normalise, or pass through bit for bit when `restore`: rows read back from a saved index)
int index_add_impl(sqe_index* idx, const float* x_dev, int64_t n, int64_t x_stride, bool restore, hipStream_t s) {
    if (n <= 0) return SQE_OK;
    const int64_t have = idx->n.load();
    if (have + n > 0xFFFFFFF0LL) return fail(SQE_ERR_INVALID, "sqe_index_add: more than 2^32 rows per shard");
    SQE_TRY(index_grow(idx, have + n, s));
    if (!idx->resid_max.p) {
        SQE_TRY(idx->resid_max.ensure(16));
        SQE_HIP(hipMemsetAsync(idx->resid_max.p, 0, 16, s));
    }
    {
        StageTimer t(idx->ctx->prof, s, ST_ADD);
        float* mdst = idx->master + (size_t)have * idx->dim;
        bf16_t* sdst = idx->scan + (size_t)have * (idx->pitch / 2);
        if (restore) SQE_TRY(launch_restore_rows(x_dev, n, idx->dim, x_stride, mdst, sdst, idx->pitch / 2, idx->resid_max.as<uint32_t>(), s));
        else SQE_TRY(launch_normalize_rows(x_dev, n, idx->dim, x_stride, mdst, sdst, idx->pitch / 2, nullptr, idx->resid_max.as<uint32_t>(), s));
    }
    idx->n.store(have + n);
    if (idx->ivf) SQE_TRY(ivf_rows_added(idx, idx->ivf, s));
    return SQE_OK;
}

int index_update_impl(sqe_index* idx, const int64_t* rows_dev, const float* x_dev, int64_t n, hipStream_t s) {
    if (n <= 0) return SQE_OK;
    SQE_TRY(launch_normalize_rows_scatter(x_dev, rows_dev, n, idx->dim, idx->master, idx->scan, idx->pitch / 2,
                                          idx->resid_max.as<uint32_t>(), s));
    if (idx->ivf) SQE_TRY(ivf_rows_updated(idx, idx->ivf, rows_dev, n, s));   // only the overwritten rows are re-assigned
    // the int8 copy follows (rows past i8_rows are quantised again by the next search; the residual maximum only grows)
    if (idx->i8db.p && idx->i8_cap_tiles == idx->cap / SCAN_BM)
        SQE_TRY(launch_quantize_rows_i8(idx->master, rows_dev, 0, n, idx->n.load(), idx->dim, idx->i8db.as<int8_t>(), idx->i8_tile_stride,
                                        idx->i8sxi.as<uint32_t>(), idx->i8resid_max.as<uint32_t>(), s));
    else idx->i8_rows = 0;
    idx->i8_dx_stale = true;      // the index grew since the copy was made: the next search rebuilds it from the master, whole
    return SQE_OK;
}

int index_create_impl(sqe_ctx* ctx, int dim, int kind, int nlist, bool internal, sqe_index** out) {
    *out = nullptr;
    if (dim <= 0 || dim % SCAN_BK != 0 || dim > 8192)
        return fail(SQE_ERR_INVALID, "sqe_index_create: dim must be a positive multiple of 64 (<= 8192)");
    if (kind != SQE_INDEX_FLAT && kind != SQE_INDEX_IVF_FLAT) return fail(SQE_ERR_INVALID, "sqe_index_create: unknown index kind");
    if (kind == SQE_INDEX_IVF_FLAT && (nlist < 1 || nlist > (1 << 20)))
        return fail(SQE_ERR_INVALID, "sqe_index_create: IVF needs 1 <= nlist <= 2^20");
    SQE_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<sqe_index> idx(new (std::nothrow) sqe_index);
    if (!idx) return fail(SQE_ERR_OOM, "sqe_index_create: host allocation failed");
    idx->ctx = ctx;
    idx->dim = dim;
    idx->kind = kind;
    idx->nlist = nlist;
    idx->internal = internal;
    // FLAT indexes take the int8 first pass wherever it applies (>= i8_min_rows rows, dim >= 256 and a multiple of 128,
    // k <= i8_sample_m, rows that quantise within i8_max_resid); everything else -- small indexes, IVF, the coarse quantiser's
    // own index -- runs the bf16 scan.  Both are exact (sqe.h: SQE_SCAN_*).
    idx->scan_mode = (kind == SQE_INDEX_FLAT && !internal) ? SQE_SCAN_INT8_RESCORE : SQE_SCAN_BF16_RESCORE;
    if (!internal) SQE_TRY(idx->ord.init());
    {
        // rows of the scanned copy are padded by one 128-B line by default: with a 2^n pitch every
        // row of a K slice would sit in the same memory channel
        const char* e = knob_env("SQE_ROW_PAD");
        const int pad = e ? atoi(e) : 128;
        idx->pitch = dim * 2 + (pad >= 0 && pad % 8 == 0 ? pad : 128);
    }
    if (kind == SQE_INDEX_IVF_FLAT) {
        int rc = ivf_create(idx.get(), &idx->ivf);
        if (rc == SQE_OK) ivf_coarse(idx->ivf)->certify = 0;   // probes are approximate by nature
        if (rc != SQE_OK) { ivf_destroy(idx->ivf); idx->ord.destroy(); return rc; }
    }
    *out = idx.release();
    return SQE_OK;
}

}  // namespace sqe

// ================================================================ library / context
extern "C" {

int sqe_version(void) { return SQE_VERSION; }
const char* sqe_last_error(void) { return g_last_error.c_str(); }

static int ctx_init_device(sqe_ctx* c, int device) {
    int count = 0;
    SQE_HIP(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return fail(SQE_ERR_INVALID, "sqe_create: no such HIP device");
    c->device = device;
    SQE_HIP(hipSetDevice(c->device));
    hipDeviceProp_t prop;
    SQE_HIP(hipGetDeviceProperties(&prop, c->device));
    c->cu_count = prop.multiProcessorCount;
    c->hbm_bytes = (int64_t)prop.totalGlobalMem;
    c->name = prop.name[0] ? prop.name : prop.gcnArchName;
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(SQE_ERR_UNSUPPORTED, std::string("libsqe is built for gfx950 only, device is ") + prop.gcnArchName);
    SQE_HIP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream.store(c->own_stream);
    SQE_TRY(c->host.init());
    SQE_TRY(c->unc_last.ensure(16));
    SQE_HIP(hipMemsetAsync(c->unc_last.p, 0, 16, c->own_stream));
    SQE_TRY(c->i8_last.ensure(32));
    SQE_HIP(hipMemsetAsync(c->i8_last.p, 0, 32, c->own_stream));
    return SQE_OK;
}

int sqe_create_sharded(const int* device_ids, int n_shards, int exchange, sqe_ctx** out) {
    if (!out) return fail(SQE_ERR_INVALID, "sqe_create: out is null");
    *out = nullptr;
    if (n_shards < 1 || n_shards > 64 || !device_ids) return fail(SQE_ERR_INVALID, "sqe_create: need 1 <= n_dev <= 64 device ids");
    if (exchange < SQE_EXCHANGE_AUTO || exchange > SQE_EXCHANGE_COPY) return fail(SQE_ERR_INVALID, "sqe_create_sharded: unknown exchange mode");
    std::unique_ptr<sqe_ctx> c(new (std::nothrow) sqe_ctx);
    if (!c) return fail(SQE_ERR_OOM, "sqe_create: host allocation failed");
    int rc = ctx_init_device(c.get(), device_ids[0]);
    if (rc == SQE_OK) rc = group_create(c.get(), device_ids, n_shards, exchange);
    if (rc != SQE_OK) { sqe_destroy(c.release()); return rc; }
    *out = c.release();
    return SQE_OK;
}

int sqe_create(const int* device_ids, int n_dev, sqe_ctx** out) {
    if (!out) return fail(SQE_ERR_INVALID, "sqe_create: out is null");
    *out = nullptr;
    if (n_dev < 1 || !device_ids) return fail(SQE_ERR_INVALID, "sqe_create: need at least one device id");
    if (n_dev > 1) return sqe_create_sharded(device_ids, n_dev, SQE_EXCHANGE_AUTO, out);
    std::unique_ptr<sqe_ctx> c(new (std::nothrow) sqe_ctx);
    if (!c) return fail(SQE_ERR_OOM, "sqe_create: host allocation failed");
    int rc = ctx_init_device(c.get(), device_ids[0]);
    if (rc != SQE_OK) { sqe_destroy(c.release()); return rc; }
    *out = c.release();
    return SQE_OK;
}

void sqe_destroy(sqe_ctx* ctx) {
    if (!ctx) return;
    if (ctx->group) group_destroy(ctx);
    (void)hipSetDevice(ctx->device);
    // Wait for everything on the device rather than for ctx->stream: an installed caller-owned stream
    // (sqe_set_stream) may already be gone when the context is torn down.
    (void)hipDeviceSynchronize();
    ctx->prof.drain();
    ctx->host.destroy();
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int sqe_synchronize(sqe_ctx* ctx) {
    if (!ctx) return fail(SQE_ERR_INVALID, "null handle");
    SQE_HIP(hipSetDevice(ctx->device));
    SQE_HIP(hipStreamSynchronize(ctx->stream.load()));
    return SQE_OK;
}

void* sqe_stream(sqe_ctx* ctx) { return ctx ? (void*)ctx->stream.load() : nullptr; }

int sqe_set_stream(sqe_ctx* ctx, void* hip_stream) {
    if (!ctx) return fail(SQE_ERR_INVALID, "null handle");
    std::lock_guard<std::mutex> lk(ctx->mu);
    SQE_HIP(hipSetDevice(ctx->device));
    ctx->prof.drain();
    // the OLD stream is only synchronised when it is ours: a caller-owned one may be dead already; work
    // still queued there is ordered by the per-object events (OpScope) in any case
    if (!ctx->foreign.load()) SQE_HIP(hipStreamSynchronize(ctx->own_stream));
    ctx->stream.store(hip_stream ? (hipStream_t)hip_stream : ctx->own_stream);
    ctx->foreign.store(hip_stream != nullptr);
    return SQE_OK;
}

int sqe_device_info(sqe_ctx* ctx, char* name, int name_cap, int* cu_count, int64_t* hbm_bytes) {
    if (!ctx) return fail(SQE_ERR_INVALID, "null handle");
    if (name && name_cap > 0) {
        strncpy(name, ctx->name.c_str(), (size_t)name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (cu_count) *cu_count = ctx->cu_count;
    if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
    return SQE_OK;
}

int sqe_group_info(sqe_ctx* ctx, int* n_shards, int* exchange, int* device_ids, int cap) {
    if (!ctx) return fail(SQE_ERR_INVALID, "null handle");
    if (!ctx->group) {
        if (n_shards) *n_shards = 1;
        if (exchange) *exchange = SQE_EXCHANGE_COPY;
        if (device_ids && cap > 0) device_ids[0] = ctx->device;
        return SQE_OK;
    }
    return group_describe(ctx, n_shards, exchange, device_ids, cap);
}

// ================================================================ index
int sqe_index_create(sqe_ctx* ctx, int dim, int kind, int nlist, sqe_index** out) {
    if (!ctx) return fail(SQE_ERR_INVALID, "null handle");
    if (!out) return fail(SQE_ERR_INVALID, "sqe_index_create: out is null");
    if (ctx->group) return group_index_create(ctx, dim, kind, nlist, out);
    return index_create_impl(ctx, dim, kind, nlist, false, out);
}

void sqe_index_destroy(sqe_index* idx) {
    if (!idx) return;
    if (idx->group) { group_index_destroy(idx); return; }
    (void)hipSetDevice(idx->ctx->device);
    {
        std::lock_guard<std::mutex> lk(idx->ord.mu);
        idx->ord.quiesce();                       // the last operation on this index, on whatever stream it ran
    }
    if (idx->ivf) { ivf_destroy(idx->ivf); idx->ivf = nullptr; }
    if (idx->master) (void)hipFree(idx->master);
    if (idx->scan) (void)hipFree(idx->scan);
    idx->ord.destroy();
    delete idx;                                   // DevBufs release in the destructor
}

int sqe_index_reserve(sqe_index* idx, int64_t rows) {
    if (!idx) return fail(SQE_ERR_INVALID, "null index");
    if (rows < 0) return fail(SQE_ERR_INVALID, "sqe_index_reserve: rows < 0");
    if (idx->group) return group_index_reserve(idx, rows);
    OpScope op(idx->ctx, idx->ord, true);
    return index_grow(idx, rows, op.s);
}

int sqe_index_add_device(sqe_index* idx, const float* x_dev, int64_t n) {
    if (!idx) return fail(SQE_ERR_INVALID, "null index");
    if (n < 0 || (n > 0 && !x_dev)) return fail(SQE_ERR_INVALID, "sqe_index_add: bad arguments");
    if (n == 0) return SQE_OK;
    if (idx->group) return group_index_add(idx, x_dev, n, true, false);
    OpScope op(idx->ctx, idx->ord, false);
    return index_add_impl(idx, x_dev, n, idx->dim, false, op.s);
}

int sqe_index_add(sqe_index* idx, const float* x_host, int64_t n) {
    if (!idx) return fail(SQE_ERR_INVALID, "null index");
    if (n < 0 || (n > 0 && !x_host)) return fail(SQE_ERR_INVALID, "sqe_index_add: bad arguments");
    if (n == 0) return SQE_OK;
    if (idx->group) return group_index_add(idx, x_host, n, false, false);
    OpScope op(idx->ctx, idx->ord, true);
    SQE_TRY(index_grow(idx, idx->n.load() + n, op.s));
    const int64_t rows_per_step = std::max<int64_t>(1, (64ll << 20) / ((int64_t)idx->dim * 4));
    SQE_TRY(idx->stage_in.ensure((size_t)std::min(rows_per_step, n) * idx->dim * 4));
    for (int64_t off = 0; off < n; off += rows_per_step) {
        const int64_t m = std::min(rows_per_step, n - off);
        SQE_HIP(hipMemcpyAsync(idx->stage_in.p, x_host + (size_t)off * idx->dim, (size_t)m * idx->dim * 4,
                               hipMemcpyHostToDevice, op.s));
        SQE_TRY(index_add_impl(idx, idx->stage_in.as<float>(), m, idx->dim, false, op.s));
    }
    SQE_HIP(hipStreamSynchronize(op.s));   // x_host is not retained past return
    return SQE_OK;
}

int sqe_index_update(sqe_index* idx, const int64_t* rows_host, const float* x_host, int64_t n) {
    if (!idx) return fail(SQE_ERR_INVALID, "null index");
    if (n < 0 || (n > 0 && (!rows_host || !x_host))) return fail(SQE_ERR_INVALID, "sqe_index_update: bad arguments");
    if (n == 0) return SQE_OK;
    if (idx->group) return group_index_update(idx, rows_host, x_host, n);
    OpScope op(idx->ctx, idx->ord, true);
    const int64_t have = idx->n.load();
    for (int64_t i = 0; i < n; ++i)
        if (rows_host[i] < 0 || rows_host[i] >= have) return fail(SQE_ERR_INVALID, "sqe_index_update: row out of range");
    const size_t xb = (size_t)n * idx->dim * 4, rb = (size_t)n * 8;
    SQE_TRY(idx->stage_in.ensure(xb + rb));
    SQE_HIP(hipMemcpyAsync(idx->stage_in.p, x_host, xb, hipMemcpyHostToDevice, op.s));
    SQE_HIP(hipMemcpyAsync((char*)idx->stage_in.p + xb, rows_host, rb, hipMemcpyHostToDevice, op.s));
    SQE_TRY(index_update_impl(idx, (const int64_t*)((char*)idx->stage_in.p + xb), idx->stage_in.as<float>(), n, op.s));
    SQE_HIP(hipStreamSynchronize(op.s));
    return SQE_OK;
}

int sqe_index_count(const sqe_index* idx, int64_t* out) {
    if (!idx || !out) return fail(SQE_ERR_INVALID, "sqe_index_count: null argument");
    if (idx->group) return group_index_count(idx, out);
    *out = idx->n.load();
    return SQE_OK;
}

int sqe_index_get_rows(sqe_index* idx, const int64_t* rows_host, int64_t n, float* out_host) {
    if (!idx) return fail(SQE_ERR_INVALID, "null index");
    if (n < 0 || (n > 0 && (!rows_host || !out_host))) return fail(SQE_ERR_INVALID, "sqe_index_get_rows: bad arguments");
    if (idx->group) return group_index_get_rows(idx, rows_host, n, out_host);
    OpScope op(idx->ctx, idx->ord, true);
    const int64_t have = idx->n.load();
    for (int64_t i = 0; i < n; ++i) {
        if (rows_host[i] < 0 || rows_host[i] >= have) return fail(SQE_ERR_INVALID, "sqe_index_get_rows: row out of range");
        SQE_HIP(hipMemcpyAsync(out_host + (size_t)i * idx->dim, idx->master + (size_t)rows_host[i] * idx->dim,
                               (size_t)idx->dim * 4, hipMemcpyDeviceToHost, op.s));
    }
    SQE_HIP(hipStreamSynchronize(op.s));
    return SQE_OK;
}

int sqe_index_set_option(sqe_index* idx, const char* key, double value) {
    if (!idx || !key) return fail(SQE_ERR_INVALID, "sqe_index_set_option: null argument");
    if (idx->group) return group_index_set_option(idx, key, value);
    std::lock_guard<std::mutex> lk(idx->ord.mu);
    const std::string k(key);
    if (k == "scan_mode") {
        if ((int)value != SQE_SCAN_BF16_RESCORE && (int)value != SQE_SCAN_INT8_RESCORE) return fail(SQE_ERR_INVALID, "scan_mode: unknown mode");
        if ((int)value == SQE_SCAN_INT8_RESCORE && idx->kind != SQE_INDEX_FLAT)
            return fail(SQE_ERR_INVALID, "scan_mode: the int8 first pass is for FLAT indexes");
        idx->scan_mode = (int)value;
    } else if (k == "i8_min_rows") {
        if (value < 0) return fail(SQE_ERR_INVALID, "i8_min_rows must be >= 0");
        idx->i8_min_rows = (int64_t)value;
    } else if (k == "i8_sample_step") {
        if (value < 1 || value > 4096) return fail(SQE_ERR_INVALID, "i8_sample_step must be in [1, 4096]");
        idx->i8_sample_step = (int)value;
    } else if (k == "i8_sample_int8") {
        idx->i8_sample_int8 = value != 0 ? 1 : 0;
    } else if (k == "i8_max_resid") {
        if (!(value > 0)) return fail(SQE_ERR_INVALID, "i8_max_resid must be > 0");
        idx->i8_max_resid = value;
    } else if (k == "i8_anchor_margin") {
        if (!(value >= 0) || value > 4) return fail(SQE_ERR_INVALID, "i8_anchor_margin must be in [0, 4]");
        idx->i8_anchor_margin = value;
    } else if (k == "i8_key_budget") {
        if (value < 0 || value > 1e9) return fail(SQE_ERR_INVALID, "i8_key_budget must be >= 0");
        idx->i8_key_budget = (int)value;
    } else if (k == "i8_sample_m") {
        if (value < 1 || value > 64) return fail(SQE_ERR_INVALID, "i8_sample_m must be in [1, 64]");
        idx->i8_sample_m = (int)value;
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

}  // extern "C"

namespace sqe {

// The search pipeline on stream s (caller holds the index lock): query normalise -> bf16 scan with the fused
// top-k filter -> select + fp32 rescore + certificate -> collect pass for uncertified queries.
// The int8 copy of the stored rows (scan_mode INT8) is derived data, filled lazily: the first search after an add
// quantises rows [i8_rows, n) from the fp32 master (one streaming pass, ~5 KiB per row).  Tiles are independent, so a
// grown index keeps what it has.
static int ensure_i8_copy(sqe_index* idx, hipStream_t s) {
    const int K = idx->dim;
    const int64_t n_rows = idx->n.load();
    const int64_t cap_tiles = idx->cap / SCAN_BM;
    const int64_t stride = (int64_t)(K / 64) * 16384 + 2048;     // + 2 KiB: chunk streams do not start at the same address modulo 256 KiB
    if (idx->i8_cap_tiles != cap_tiles || idx->i8_tile_stride != stride) {
        DevBuf nd, ns;
        // The copy is derived data (+1 byte per element): an index that fits without it must keep answering.  SQE_ERR_OOM here
        // sends THIS search and every later one to the bf16 scan, which needs no extra memory (index_search_impl).
        if (nd.ensure((size_t)cap_tiles * stride) != SQE_OK || ns.ensure((size_t)cap_tiles * SCAN_BM * 4) != SQE_OK) {
            (void)hipGetLastError();
            return SQE_ERR_OOM;
        }
        SQE_HIP(hipMemsetAsync(nd.p, 0, nd.bytes, s));            // rows past n read as zero vectors
        SQE_HIP(hipMemsetAsync(ns.p, 0, ns.bytes, s));
        if (idx->i8_rows > 0 && idx->i8_tile_stride == stride) {
            const int64_t tiles = (idx->i8_rows + SCAN_BM - 1) / SCAN_BM;
            SQE_HIP(hipMemcpyAsync(nd.p, idx->i8db.p, (size_t)tiles * stride, hipMemcpyDeviceToDevice, s));
            SQE_HIP(hipMemcpyAsync(ns.p, idx->i8sxi.p, (size_t)tiles * SCAN_BM * 4, hipMemcpyDeviceToDevice, s));
        } else {
            idx->i8_rows = 0;
        }
        SQE_HIP(hipStreamSynchronize(s));                         // the old buffers die here
        std::swap(nd.p, idx->i8db.p); std::swap(nd.bytes, idx->i8db.bytes);
        std::swap(ns.p, idx->i8sxi.p); std::swap(ns.bytes, idx->i8sxi.bytes);
        idx->i8_cap_tiles = cap_tiles;
        idx->i8_tile_stride = stride;
    }
    if (!idx->i8resid_max.p) {
        SQE_TRY(idx->i8resid_max.ensure(16));
        SQE_HIP(hipMemsetAsync(idx->i8resid_max.p, 0, 16, s));
    }
    if (idx->i8_rows < n_rows) {
        StageTimer t(idx->ctx->prof, s, ST_ADD);
        SQE_TRY(launch_quantize_rows_i8(idx->master, nullptr, idx->i8_rows, n_rows - idx->i8_rows, n_rows, K, idx->i8db.as<int8_t>(), stride,
                                        idx->i8sxi.as<uint32_t>(), idx->i8resid_max.as<uint32_t>(), s));
        idx->i8_rows = n_rows;
        idx->i8_dx_stale = true;
    }
    if (idx->i8_dx_stale) {
        // one 4-byte read-back per batch of newly quantised (or overwritten) rows: the host decides from it whether the
        // int8 bound is worth using at all (index_search_impl)
        uint32_t bits = 0;
        SQE_HIP(hipMemcpyAsync(&bits, idx->i8resid_max.p, 4, hipMemcpyDeviceToHost, s));
        SQE_HIP(hipStreamSynchronize(s));
        memcpy(&idx->i8_dx, &bits, 4);
        idx->i8_dx_stale = false;
    }
    return SQE_OK;
}

__global__ void add_count_kernel(int* acc, const int* v) { *acc += *v; }

// Second pass for the queries whose certificate failed (bf16 or int8 first pass alike): they are compacted into a
// dense batch on the device, a bf16 collect scan gathers every row whose scan score can still reach the query's k-th
// cosine (collect_thr[q] = that cosine - bf16 eps, +inf for certified queries) and the gathered rows are re-scored in fp32.
static int run_collect_fallback(sqe_index* idx, int B, int k, int kp, int b_pad, int* unc_count, float* collect_thr,
                                float* cos_out_dev, int64_t* id_out_dev, int pass_index, hipStream_t s) {
    sqe_ctx* c = idx->ctx;
    const int K = idx->dim;
    const int64_t n_rows = idx->n.load();
    {
        StageTimer t(c->prof, s, ST_SELECT);
        SQE_TRY(launch_compact_uncertified(collect_thr, B, idx->qb.as<bf16_t>(), idx->pitch, K * 2, idx->unc_ids.as<int>(),
                                           idx->thr_c.as<float>(), b_pad + 256, idx->qb_c.as<bf16_t>(), unc_count, s));
    }
        {
            // The collect scans are scans: they are booked under scan_ms (scan_calls counts the main launches
            // only).  One launch per range of counts is enqueued, each planned like a search of that batch size
            // and returning at once unless the count is in its range (and at once when it is 0).
            StageTimer t(c->prof, s, ST_COLLECT);
            ScanArgs a;
            a.db = idx->scan; a.q = idx->qb_c.as<bf16_t>(); a.n_rows = n_rows; a.K = K; a.B = B;
            a.db_pitch = idx->pitch; a.q_pitch = idx->pitch;
            a.cand = idx->cand.as<uint64_t>(); a.cand_cnt = idx->cand_cnt.as<int>(); a.gmax = idx->gmax.as<uint32_t>();
            a.dbg_counters = nullptr;
            a.q_resid = nullptr; a.db_resid_max = nullptr;
            a.collect_thr = idx->thr_c.as<float>(); a.collect_keys = idx->fb_keys.as<uint64_t>(); a.collect_cnt = idx->fb_cnt.as<int>();
            a.unc_count = unc_count;
            const int bounds[5] = {0, 64, 256, 512, 1 << 30};
            for (int r = 0; r < 4 && bounds[r] < B; ++r) {
                const int hi = std::min(bounds[r + 1], B);
                const ScanPlan cp = make_scan_plan(n_rows, hi, kp, c->cu_count);
                a.collect_lo = bounds[r] + 1;
                a.collect_hi = r == 3 ? (1 << 30) : bounds[r + 1];
                SQE_TRY(launch_scan_collect(cp, a, s));
            }
        }
        {
            // ... and re-score them in fp32
            StageTimer t(c->prof, s, ST_SELECT);
            ExactArgs e;
            e.master = idx->master; e.qn = idx->qn.as<float>(); e.K = K; e.B = B; e.k = k;
            e.collect_thr = collect_thr; e.keys = idx->fb_keys.as<uint64_t>(); e.key_cnt = idx->fb_cnt.as<int>();
            e.unc_ids = idx->unc_ids.as<int>(); e.unc_count = unc_count;
            e.cos_out = cos_out_dev; e.id_out = id_out_dev; e.id_base = idx->id_base;
            SQE_TRY(launch_collect_rescore(e, s));
        }
        // the count goes to a buffer the CONTEXT owns (sqe_stats reads it long after this index may be gone); the
        // passes of a batch above MAX_PASS add up (r02: the last pass's count overwrote the others)
        if (pass_index == 0) SQE_HIP(hipMemcpyAsync(c->unc_last.p, unc_count, 4, hipMemcpyDeviceToDevice, s));
        else hipLaunchKernelGGL(add_count_kernel, dim3(1), dim3(1), 0, s, c->unc_last.as<int>(), unc_count);
        c->unc_valid.store(true);
    return SQE_OK;
}


int index_search_impl(sqe_index* idx, const float* q_dev, int B, int k, int nprobe, float* cos_out_dev, int64_t* id_out_dev,
                      hipStream_t s, int pass_index) {
    sqe_ctx* c = idx->ctx;
    if (idx->ivf) {
        StageTimer t(c->prof, s, ST_SCAN);
        SQE_TRY(ivf_search(idx, idx->ivf, q_dev, B, k, nprobe > 0 ? nprobe : idx->nprobe, cos_out_dev, id_out_dev, s));
        c->search_calls++;
        return SQE_OK;
    }
    const int K = idx->dim;
    // More than four 256-query blocks would leave fewer than 64 DB chunks (one workgroup per CU), too few to
    // fill a row of the global-bound table: the filter would lose its cross-chunk threshold.  Larger batches
    // run as passes of 1024 queries, each at the full-batch rate.
    constexpr int MAX_PASS = 1024;
    if (B > MAX_PASS) {
        for (int off = 0; off < B; off += MAX_PASS) {
            const int m = std::min(MAX_PASS, B - off);
            SQE_TRY(index_search_impl(idx, q_dev + (size_t)off * K, m, k, nprobe, cos_out_dev + (size_t)off * k,
                                      id_out_dev + (size_t)off * k, s, off / MAX_PASS));
        }
        return SQE_OK;
    }
    const int64_t n_rows = idx->n.load();
    const int kp = auto_kp(idx, k);
    const ScanPlan plan = make_scan_plan(n_rows, B, kp, c->cu_count, k);

    SQE_TRY(idx->qn.ensure((size_t)B * K * 4));
    SQE_TRY(idx->qb.ensure((size_t)plan.b_pad * idx->pitch));
    SQE_TRY(idx->cand.ensure((size_t)plan.n_chunks * plan.b_pad * CAND_CAP * 8));
    SQE_TRY(idx->cand_cnt.ensure((size_t)plan.n_chunks * plan.b_pad * 4));
    const size_t gmax_bytes = (size_t)plan.b_pad * plan.ngroups * GMAX_COLS * 4;
    SQE_TRY(idx->gmax.ensure(gmax_bytes));
    // knobs build, timing experiments whose scan scores are wrong on purpose (SQE_DBG=8192): no certificate, no collect pass
    static const bool no_collect = [] { const char* e = knob_env("SQE_NO_COLLECT"); return e && atoi(e) != 0; }();
    const bool certify = idx->certify && n_rows > 0 && !no_collect;
    SQE_TRY(idx->q_resid.ensure((size_t)B * 4));
    if (certify) {
        SQE_TRY(idx->unc.ensure(16 + (size_t)plan.b_pad * 4));
        SQE_TRY(idx->fb_keys.ensure((size_t)B * EXACT_CAP * 8));
        SQE_TRY(idx->fb_cnt.ensure((size_t)B * 4));
        SQE_TRY(idx->unc_ids.ensure((size_t)B * 4));
        SQE_TRY(idx->thr_c.ensure((size_t)(plan.b_pad + 256) * 4));
        if ((size_t)(plan.b_pad + 256) * idx->pitch > idx->qb_c.bytes) {
            SQE_TRY(idx->qb_c.ensure((size_t)(plan.b_pad + 256) * idx->pitch));
            SQE_HIP(hipMemsetAsync(idx->qb_c.p, 0, idx->qb_c.bytes, s));     // rows past the count read as zero
        }
    }
    {
        StageTimer t(c->prof, s, ST_PREP);
        if (plan.b_pad > B)
            SQE_HIP(hipMemsetAsync(idx->qb.as<char>() + (size_t)B * idx->pitch, 0, (size_t)(plan.b_pad - B) * idx->pitch, s));
        SQE_TRY(launch_normalize_rows(q_dev, B, K, K, idx->qn.as<float>(), idx->qb.as<bf16_t>(), idx->pitch / 2,
                                      idx->q_resid.as<float>(), nullptr, s));
        if (certify) {
            SQE_HIP(hipMemsetAsync(idx->unc.p, 0, 16, s));
            SQE_HIP(hipMemsetAsync(idx->fb_cnt.p, 0, (size_t)B * 4, s));
        }
        SQE_HIP(hipMemsetAsync(idx->gmax.p, 0, gmax_bytes, s));
    }
    int* unc_count = certify ? idx->unc.as<int>() : nullptr;
    float* collect_thr = certify ? reinterpret_cast<float*>(idx->unc.as<int>() + 4) : nullptr;
    // ---- int8 first pass (scan_mode INT8): threshold pass on a row sample (bf16 kernels, every step-th tile) -> fixed
    // per-query collect thresholds -> int8 collect scan over all rows -> staged fp32 re-score + certificate -> the bf16
    // collect pass for what is left.  Small indexes give the sample nothing to estimate from: they stay with the bf16 scan.
    int step8 = idx->i8_sample_step, m8 = idx->i8_sample_m;
    {
        // The bf16 kernels of the threshold pass exchange their bounds between chunks only when the sample has at least 64
        // chunks (scan.hip: make_scan_plan); below that every workgroup keeps a quarter of its rows and the pass takes three
        // times as long (1.25 M rows -- an eighth of the 10 M-row index, one shard of eight: 0.72 ms against 0.22 ms at
        // 2.5 M).  Small indexes therefore sample MORE tiles (at least 128) and take a deeper place of the sample in proportion,
        // which leaves the expected number of collected rows (~ step x m) where the options put it.
        const int64_t tiles = (n_rows + SCAN_BM - 1) / SCAN_BM;
        const int min_tiles = 2 * GMAX_COLS;
        if (tiles / step8 < min_tiles && tiles / min_tiles >= 1 && tiles / min_tiles < step8) {
            // (the place is capped at 64: the step does not go below what keeps step x m)
            const int step_e = (int)std::max<int64_t>(std::max(1, (step8 * m8 + 63) / 64), tiles / min_tiles);
            const int m_e = std::min(64, std::max(m8, (step8 * m8 + step_e - 1) / step_e));
            step8 = step_e;
            m8 = m_e;
        }
    }
    const bool use_i8 = idx->scan_mode == SQE_SCAN_INT8_RESCORE && certify && K >= 256 && K % 128 == 0 && k <= m8 &&
                        n_rows >= idx->i8_min_rows && n_rows >= (int64_t)step8 * SCAN_BM * 4;
    bool i8_ok = use_i8;
    if (use_i8) {
        const int rc8 = ensure_i8_copy(idx, s);
        if (rc8 == SQE_ERR_OOM) {
            if (!idx->i8_oom_logged) fprintf(stderr, "[sqe] no memory for the int8 copy of the rows: this index answers with the bf16 scan\n");
            idx->i8_oom_logged = true;
            idx->scan_mode = SQE_SCAN_BF16_RESCORE;
            i8_ok = false;
        } else {
            SQE_TRY(rc8);
            i8_ok = idx->i8_dx <= (float)idx->i8_max_resid;  // else: the bf16 scan below
        }
    }
    if (i8_ok) {
        const int q8_pitch = K + 128;
        const int n_tiles_s = (plan.n_tiles + step8 - 1) / step8;
        ScanPlan ps = make_scan_plan((int64_t)n_tiles_s * SCAN_BM, B, auto_kp(idx, m8), c->cu_count, m8);
        // the int8 threshold pass runs on query blocks of 256 whatever the batch (1 % of the tiles: padding costs nothing)
        const int b_pad_s = (B + 255) / 256 * 256;
        const int b_pad_q = std::max(plan.b_pad, b_pad_s);
        const int64_t full_tiles = n_rows / SCAN_BM;
        const int n_tiles_i8s = (int)(full_tiles / step8);          // sampled tiles t * step8, whole tiles only
        const bool sample_i8 = idx->i8_sample_int8 != 0 && n_tiles_i8s >= 1;
        SQE_TRY(idx->q8.ensure((size_t)b_pad_q * q8_pitch));
        SQE_TRY(idx->q8sqi.ensure((size_t)b_pad_q * 4));
        SQE_TRY(idx->q8resid.ensure((size_t)plan.b_pad * 4));
        SQE_TRY(idx->i8thr_int.ensure((size_t)plan.b_pad * 4));
        SQE_TRY(idx->i8thr_eff.ensure((size_t)plan.b_pad * 4));
        SQE_TRY(idx->i8cos_s.ensure((size_t)B * m8 * 4));
        SQE_TRY(idx->i8ids_s.ensure((size_t)B * m8 * 8));
        SQE_TRY(idx->i8stats.ensure(64));
        SQE_TRY(idx->i8ovf.ensure((size_t)plan.b_pad * I8_OVF_CAP * 8));
        SQE_TRY(idx->i8ovf_cnt.ensure((size_t)plan.b_pad * 4));
        {
            StageTimer t(c->prof, s, ST_PREP);
            SQE_HIP(hipMemsetAsync(idx->i8ovf_cnt.p, 0, (size_t)plan.b_pad * 4, s));
            if (b_pad_q > B)
                SQE_HIP(hipMemsetAsync(idx->q8.as<char>() + (size_t)B * q8_pitch, 0, (size_t)(b_pad_q - B) * q8_pitch, s));
            SQE_TRY(launch_quantize_queries_i8(idx->qn.as<float>(), B, K, idx->q8.as<int8_t>(), q8_pitch, idx->q8sqi.as<uint32_t>(),
                                               idx->q8resid.as<float>(), s));
            SQE_HIP(hipMemsetAsync(idx->i8stats.p, 0, 64, s));
        }
        int chunks_s_used = 0;
        if (sample_i8) {
            // threshold pass in int8 (r03c): the collect scan's own tile loop over every step-th tile, two best scores per lane,
            // then per query the m-th largest of them (scan_i8.hip: sample_i8_pp_kernel; select_i8.hip: i8_sample_select_kernel)
            StageTimer t(c->prof, s, ST_SAMPLE);
            const int qblocks_s = b_pad_s / 256;
            const int chunks_s = std::max(1, std::min(std::min(c->cu_count / qblocks_s, 256), n_tiles_i8s));
            chunks_s_used = chunks_s;
            SQE_TRY(idx->i8samp.ensure((size_t)chunks_s * b_pad_s * 16 * 8));
            I8SampleArgs sp;
            sp.db8 = idx->i8db.as<int8_t>(); sp.tile_stride = idx->i8_tile_stride; sp.sxi = idx->i8sxi.as<uint32_t>();
            sp.q8 = idx->q8.as<int8_t>(); sp.q_pitch = q8_pitch; sp.K = K; sp.b_pad = b_pad_s; sp.n_tiles_s = n_tiles_i8s; sp.step = step8;
            sp.n_chunks = chunks_s; sp.out = idx->i8samp.p;
            SQE_TRY(launch_sample_i8(sp, s));
            I8SampleSelectArgs ss;
            ss.cand = idx->i8samp.p; ss.n_chunks = chunks_s; ss.b_pad_s = b_pad_s; ss.m = m8; ss.k = k; ss.B = B; ss.b_pad = plan.b_pad; ss.K = K;
            ss.sqi = idx->q8sqi.as<uint32_t>(); ss.master = idx->master; ss.qn = idx->qn.as<float>();
            ss.thr_int = idx->i8thr_int.as<int>(); ss.thr_eff = idx->i8thr_eff.as<float>();
            ss.sample_cos = idx->i8cos_s.as<float>(); ss.sample_ids = idx->i8ids_s.as<int64_t>();
            ss.q_resid8 = idx->q8resid.as<float>(); ss.db_resid8_max = idx->i8resid_max.as<uint32_t>();
            ss.margin = (float)idx->i8_anchor_margin; ss.step = step8; ss.key_budget = idx->i8_key_budget;
            SQE_TRY(launch_i8_sample_select(ss, s));
        } else {
            // threshold pass: the bf16 scan + fp32 re-score of the row sample, top-m true cosines per query
            StageTimer t(c->prof, s, ST_SAMPLE);
            ScanArgs a;
            a.db = idx->scan; a.q = idx->qb.as<bf16_t>(); a.n_rows = n_rows; a.K = K; a.B = B;
            a.db_pitch = idx->pitch; a.q_pitch = idx->pitch;
            a.cand = idx->cand.as<uint64_t>(); a.cand_cnt = idx->cand_cnt.as<int>(); a.gmax = idx->gmax.as<uint32_t>();
            a.dbg_counters = nullptr;
            a.q_resid = idx->q_resid.as<float>(); a.db_resid_max = idx->resid_max.as<uint32_t>();
            a.collect_thr = nullptr; a.collect_keys = nullptr; a.collect_cnt = nullptr; a.unc_count = nullptr;
            a.tile_step = step8;
            SQE_TRY(launch_scan_bf16(ps, a, s));
            SelectArgs sa;
            sa.cand = idx->cand.as<uint64_t>(); sa.cand_cnt = idx->cand_cnt.as<int>();
            sa.n_chunks = ps.n_chunks; sa.b_pad = ps.b_pad; sa.kp = ps.kp;
            sa.master = idx->master; sa.qn = idx->qn.as<float>(); sa.K = K; sa.B = B; sa.k = m8;
            sa.cos_out = idx->i8cos_s.as<float>(); sa.id_out = idx->i8ids_s.as<int64_t>(); sa.id_base = 0;
            sa.q_resid = nullptr; sa.db_resid_max = nullptr; sa.unc_count = nullptr; sa.collect_thr = nullptr;
            sa.gmax = nullptr; sa.gshift = -1;
            SQE_TRY(launch_select_rescore(sa, s));
            SQE_TRY(launch_i8_thresholds(idx->i8cos_s.as<float>(), m8, idx->q8sqi.as<uint32_t>(), K, B, plan.b_pad,
                                         idx->i8thr_int.as<int>(), idx->i8thr_eff.as<float>(), s));
        }
        {
            StageTimer t(c->prof, s, ST_SCAN);
            I8ScanArgs ia;
            ia.db8 = idx->i8db.as<int8_t>(); ia.tile_stride = idx->i8_tile_stride; ia.sxi = idx->i8sxi.as<uint32_t>();
            ia.q8 = idx->q8.as<int8_t>(); ia.q_pitch = q8_pitch; ia.thr_int = idx->i8thr_int.as<int>();
            ia.n_rows = n_rows; ia.K = K; ia.B = B; ia.b_pad = plan.b_pad; ia.n_tiles = plan.n_tiles; ia.n_chunks = plan.n_chunks;
            ia.qblocks = plan.qblocks; ia.bn = plan.bn; ia.cand = idx->cand.as<uint64_t>(); ia.cand_cnt = idx->cand_cnt.as<int>();
            ia.ovf = idx->i8ovf.as<uint64_t>(); ia.ovf_cnt = idx->i8ovf_cnt.as<int>();
            static const bool want_stamps = [] { const char* e = knob_env("SQE_I8_STAMPS"); return e && e[0] == '1'; }();   // knobs build only
            if (want_stamps) {
                SQE_TRY(idx->dbg.ensure(8192));
                SQE_HIP(hipMemsetAsync(idx->dbg.p, 0, 8192, s));
                ia.stamps = idx->dbg.as<unsigned long long>();
            }
            static const int deep_max = [] { const char* e = knob_env("SQE_I8_DEEP_MAX"); return e ? atoi(e) : 1; }();   // knobs build: A/B of the cut
            if (plan.bn == 256 && plan.qblocks <= deep_max) SQE_TRY(launch_scan_i8_deep(ia, s));      // (scan_i8_deep.hip)
            else SQE_TRY(launch_scan_i8(ia, s));
            if (want_stamps) {
                unsigned long long h[256];
                SQE_HIP(hipMemcpyAsync(h, idx->dbg.p, sizeof(h), hipMemcpyDeviceToHost, s));
                SQE_HIP(hipStreamSynchronize(s));
                int wall_khz = 0;
                (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, c->device);
                for (int blk = 0; blk < 2; ++blk) {
                    const unsigned long long* o = h + blk * 128;
                    int prev = -1;
                    for (int sl = 0; sl < 32; ++sl) {
                        if (!o[sl * 3]) continue;
                        if (prev >= 0 && o[sl * 3] > o[prev * 3]) {
                            const double tiles = (double)(o[sl * 3] - o[prev * 3]);
                            const double us = (double)(o[sl * 3 + 1] - o[prev * 3 + 1]) / (wall_khz / 1e3);
                            const double cyc = (double)(o[sl * 3 + 2] - o[prev * 3 + 2]);
                            fprintf(stderr, "[sqe i8 stamps] wg %3d tiles %5llu..%5llu: %7.2f us per tile, %7.0f core cycles per tile, %5.0f MHz\n",
                                    blk ? 100 : 0, o[prev * 3], o[sl * 3], us / tiles, cyc / tiles, us > 0 ? cyc / us : 0.0);
                        }
                        prev = sl;
                    }
                }
                idx->dbg.release();
            }
        }
        {
            StageTimer t(c->prof, s, ST_SELECT);
            I8SelectArgs sa;
            sa.cand = idx->cand.as<uint64_t>(); sa.cand_cnt = idx->cand_cnt.as<int>(); sa.n_chunks = plan.n_chunks; sa.b_pad = plan.b_pad;
            sa.master = idx->master; sa.qn = idx->qn.as<float>(); sa.K = K; sa.B = B; sa.k = k;
            sa.scan16 = idx->scan; sa.pitch16 = idx->pitch;
            sa.sxi = idx->i8sxi.as<uint32_t>(); sa.sqi = idx->q8sqi.as<uint32_t>();
            sa.q_resid8 = idx->q8resid.as<float>(); sa.db_resid8_max = idx->i8resid_max.as<uint32_t>();
            sa.q_resid16 = idx->q_resid.as<float>(); sa.db_resid16_max = idx->resid_max.as<uint32_t>();
            sa.thr_eff = idx->i8thr_eff.as<float>();
            sa.sample_cos = idx->i8cos_s.as<float>(); sa.sample_ids = idx->i8ids_s.as<int64_t>(); sa.sample_m = m8;
            sa.cos_out = cos_out_dev; sa.id_out = id_out_dev; sa.id_base = idx->id_base;
            sa.unc_count = unc_count; sa.collect_thr = collect_thr;
            sa.stats = idx->i8stats.as<unsigned long long>();
            sa.ovf = idx->i8ovf.as<uint64_t>(); sa.ovf_cnt = idx->i8ovf_cnt.as<int>();
            SQE_TRY(launch_select_i8(sa, s));
        }
        SQE_TRY(run_collect_fallback(idx, B, k, kp, plan.b_pad, unc_count, collect_thr, cos_out_dev, id_out_dev, pass_index, s));
        {
            sqe_i8_launch_t& L = idx->i8_launch;           // what sqe_index_i8_last reports (the uncertified count is read there)
            L.rows = n_rows; L.tile_stride = idx->i8_tile_stride; L.dim = K; L.B = B; L.b_pad = plan.b_pad; L.k = k;
            L.tile_rows = SCAN_BM; L.q_pitch = q8_pitch; L.query_block = plan.bn; L.n_chunks = plan.n_chunks; L.list_cap = CAND_CAP; L.pool_cap = I8_OVF_CAP;
            L.sample_int8 = sample_i8 ? 1 : 0; L.sample_step = step8; L.sample_tiles = n_tiles_i8s;
            L.sample_chunks = chunks_s_used;
            L.sample_b_pad = b_pad_s; L.sample_m = m8; L.uncertified = -1;
        }
        SQE_HIP(hipMemcpyAsync(c->i8_last.p, idx->i8stats.p, 32, hipMemcpyDeviceToDevice, s));
        c->i8_valid.store(true);
        c->search_calls++;
        c->last_scan_rows.store(n_rows);
        c->last_scan_flops.store(2 * n_rows * (int64_t)K * B);
        c->last_scan_bytes.store(n_rows * (int64_t)K + (int64_t)B * K * 4 + (int64_t)B * k * 12);   // SURVEY 8(d) with s = 1 byte per element
        return SQE_OK;
    }
    c->i8_valid.store(false);        // this search runs the bf16 first pass
    if (n_rows > 0) {
        StageTimer t(c->prof, s, ST_SCAN);
        ScanArgs a;
        a.db = idx->scan; a.q = idx->qb.as<bf16_t>(); a.n_rows = n_rows; a.K = K; a.B = B;
        a.db_pitch = idx->pitch; a.q_pitch = idx->pitch;
        a.cand = idx->cand.as<uint64_t>(); a.cand_cnt = idx->cand_cnt.as<int>(); a.gmax = idx->gmax.as<uint32_t>();
        a.dbg_counters = nullptr;
        a.q_resid = idx->q_resid.as<float>(); a.db_resid_max = idx->resid_max.as<uint32_t>();   // the k-row bound's eps
        a.collect_thr = nullptr; a.collect_keys = nullptr; a.collect_cnt = nullptr; a.unc_count = nullptr;
        {
            static const bool want = [] { const char* e = knob_env("SQE_DBG"); return e && (atoi(e) & 32); }();   // knobs build only
            if (want) {
                SQE_TRY(idx->dbg.ensure(8192));
                SQE_HIP(hipMemsetAsync(idx->dbg.p, 0, 8192, s));
                a.dbg_counters = idx->dbg.as<unsigned long long>();
            }
        }
        SQE_TRY(launch_scan_bf16(plan, a, s));
    } else {
        SQE_HIP(hipMemsetAsync(idx->cand_cnt.p, 0, (size_t)plan.n_chunks * plan.b_pad * 4, s));
    }
    {
        StageTimer t(c->prof, s, ST_SELECT);
        SelectArgs sa;
        sa.cand = idx->cand.as<uint64_t>(); sa.cand_cnt = idx->cand_cnt.as<int>();
        sa.n_chunks = plan.n_chunks; sa.b_pad = plan.b_pad; sa.kp = kp;
        sa.master = idx->master; sa.qn = idx->qn.as<float>(); sa.K = K; sa.B = B; sa.k = k;
        sa.cos_out = cos_out_dev; sa.id_out = id_out_dev; sa.id_base = idx->id_base;
        sa.q_resid = certify ? idx->q_resid.as<float>() : nullptr;
        sa.db_resid_max = certify ? idx->resid_max.as<uint32_t>() : nullptr;
        sa.unc_count = unc_count; sa.collect_thr = collect_thr;
        sa.gmax = (n_rows > 0 && plan.gshift >= 0) ? idx->gmax.as<uint32_t>() : nullptr; sa.gshift = plan.gshift;
        SQE_TRY(launch_select_rescore(sa, s));
    }
    if (certify) SQE_TRY(run_collect_fallback(idx, B, k, kp, plan.b_pad, unc_count, collect_thr, cos_out_dev, id_out_dev, pass_index, s));
    if (idx->dbg.p) {
        unsigned long long h[1024];
        SQE_HIP(hipMemcpyAsync(h, idx->dbg.p, 8192, hipMemcpyDeviceToHost, s));
        SQE_HIP(hipStreamSynchronize(s));
        int wall_khz = 0;
        (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, c->device);
        fprintf(stderr, "[sqe dbg] appends=%llu slow_path_entries=%llu compactions=%llu block0_core_ticks=%llu wall_ticks=%llu core_mhz=%.0f\n",
                h[0], h[1], h[2], h[4], h[5], h[5] ? (double)h[4] / (double)h[5] * wall_khz / 1e3 : 0.0);
        // ping-pong scan, counters build: core cycles per steady-state phase, per wave of workgroups 0 and 100
        for (int blk = 0; blk < 2; ++blk)
            for (int w = 0; w < 8; ++w) {
                const unsigned long long* o = h + 8 + blk * 64 + w * 8;
                if (!o[0]) continue;
                const double n = (double)o[0];
                fprintf(stderr, "[sqe dbg] wg %d wave %d: phases=%llu  cmp %.0f  barrier after cmp %.0f | mem: dma issue %.0f  lds reads issue %.0f  waitcnt %.0f  barrier after mem %.0f\n",
                        blk ? 100 : 0, w, o[0], o[1] / n, o[2] / n, o[3] / n, o[4] / n, o[5] / n, o[6] / n);
            }
        for (int blk = 0; blk < 2; ++blk) {
            const unsigned long long* o = h + 8 + 128 + blk * 16;
            if (!o[11]) continue;
            const double t = (double)o[11];
            fprintf(stderr, "[sqe dbg] wg %d wave 0, cycles per tile: first cmp %.0f + barrier %.0f, mem %.0f + barrier %.0f | last cmp (fast-path test) %.0f + barrier %.0f, "
                            "mem of next tile %.0f + barrier %.0f, tile_end %.0f | %.2f general middle half-steps per tile at %.0f cycles each\n",
                    blk ? 100 : 0, o[0] / t, o[1] / t, o[2] / t, o[3] / t, o[4] / t, o[5] / t, o[6] / t, o[7] / t, o[8] / t, o[10] / t,
                    o[10] ? (double)o[9] / (double)o[10] : 0.0);
        }
        for (int g = 0; g < 2; ++g) {
            const unsigned long long* o = h + 300 + g * 8;
            if (!o[4]) continue;
            const double n = (double)o[4];
            fprintf(stderr, "[sqe dbg] wg 0 group %d general memory phase x %llu: bookkeeping before %.0f, pieces + reads %.0f, wait %.0f, bookkeeping after %.0f\n",
                    g, o[4], o[0] / n, o[1] / n, o[2] / n, o[3] / n);
        }
        // drift between the workgroups that share a DB chunk (STAMPS build): spread of their arrival at two tiles
        if (h[512] && plan.qblocks > 1 && plan.n_chunks * plan.qblocks <= 256) {
            const int G = plan.n_chunks * plan.qblocks;
            double worst[2] = {0, 0}, mean[2] = {0, 0};
            for (int c = 0; c < plan.n_chunks; ++c)
                for (int t = 0; t < 2; ++t) {
                    unsigned long long lo = ~0ull, hi = 0;
                    for (int qb = 0; qb < plan.qblocks; ++qb) {
                        const int logical = c * plan.qblocks + qb;
                        const int blk = (G & 7) == 0 ? (logical % (G >> 3)) * 8 + logical / (G >> 3) : logical;   // inverse of the kernel's remap
                        const unsigned long long v = h[512 + blk * 2 + t];
                        lo = std::min(lo, v); hi = std::max(hi, v);
                    }
                    const double d = (double)(hi - lo) / (wall_khz / 1e3);      // microseconds
                    worst[t] = std::max(worst[t], d); mean[t] += d / plan.n_chunks;
                }
            fprintf(stderr, "[sqe dbg] spread between the %d workgroups of a chunk when they finish tile 100 / 400: mean %.1f / %.1f us, worst %.1f / %.1f us (one tile = ~30 us)\n",
                    plan.qblocks, mean[0], mean[1], worst[0], worst[1]);
        }
        for (int blk = 0; blk < 2; ++blk)
            for (int w = 0; w < 8; ++w) {
                const unsigned long long* o = h + 8 + 160 + blk * 48 + w * 6;
                if (!o[5]) continue;
                const double n = (double)o[5];
                fprintf(stderr, "[sqe dbg] wg %d wave %d tile_end x %llu: mark read %.0f, own slow path %.0f (flagged in %.0f %%), barrier %.0f, entry_sync %.0f\n",
                        blk ? 100 : 0, w, o[5], o[0] / n, o[1] / n, 100.0 * o[4] / n, o[2] / n, o[3] / n);
            }
    }
    c->search_calls++;
    c->last_scan_rows.store(n_rows);
    c->last_scan_flops.store(2 * n_rows * (int64_t)K * B);
    c->last_scan_bytes.store(n_rows * (int64_t)K * 2 + (int64_t)B * K * 4 + (int64_t)B * k * 12);
    return SQE_OK;
}

}  // namespace sqe

extern "C" {

static int search_args_ok(sqe_index* idx, const void* q, int B, int k, const void* cos, const void* ids) {
    if (!idx) return fail(SQE_ERR_INVALID, "null index");
    if (B < 0 || k < 1 || k > MAX_KP) return fail(SQE_ERR_INVALID, "sqe_index_search: need B >= 0 and 1 <= k <= 256");
    if (B > 0 && (!q || !cos || !ids)) return fail(SQE_ERR_INVALID, "sqe_index_search: null buffer");
    return SQE_OK;
}

int sqe_index_i8_last(sqe_index* idx, sqe_i8_launch_t* out) {
    if (!idx || !out) return fail(SQE_ERR_INVALID, "sqe_index_i8_last: null argument");
    if (idx->group || idx->ivf) return fail(SQE_ERR_UNSUPPORTED, "sqe_index_i8_last: single-device FLAT indexes only");
    OpScope op(idx->ctx, idx->ord, true);
    if (idx->i8_launch.rows == 0) return fail(SQE_ERR_STATE, "sqe_index_i8_last: this index has not answered a search with the int8 first pass");
    int unc = 0;
    SQE_HIP(hipMemcpyAsync(&unc, idx->unc.p, 4, hipMemcpyDeviceToHost, op.s));
    SQE_HIP(hipStreamSynchronize(op.s));
    *out = idx->i8_launch;
    out->uncertified = unc;
    return SQE_OK;
}

int sqe_index_i8_read(sqe_index* idx, int what, int64_t offset, void* out_host, int64_t bytes) {
    if (!idx || (!out_host && bytes > 0)) return fail(SQE_ERR_INVALID, "sqe_index_i8_read: null argument");
    if (idx->group || idx->ivf) return fail(SQE_ERR_UNSUPPORTED, "sqe_index_i8_read: single-device FLAT indexes only");
    OpScope op(idx->ctx, idx->ord, true);
    const sqe_i8_launch_t& L = idx->i8_launch;
    if (L.rows == 0) return fail(SQE_ERR_STATE, "sqe_index_i8_read: this index has not answered a search with the int8 first pass");
    const int64_t tiles = (L.rows + L.tile_rows - 1) / L.tile_rows;
    const void* src = nullptr;
    int64_t size = 0;
    switch (what) {
        case SQE_I8_ROWS: src = idx->i8db.p; size = tiles * L.tile_stride; break;
        case SQE_I8_ROW_SCALES: src = idx->i8sxi.p; size = tiles * L.tile_rows * 4; break;
        case SQE_I8_QUERIES: src = idx->q8.p; size = (int64_t)L.b_pad * L.q_pitch; break;
        case SQE_I8_THRESHOLDS: src = idx->i8thr_int.p; size = (int64_t)L.b_pad * 4; break;
        case SQE_I8_LIST_COUNTS: src = idx->cand_cnt.p; size = (int64_t)L.n_chunks * L.b_pad * 4; break;
        case SQE_I8_LISTS: src = idx->cand.p; size = (int64_t)L.n_chunks * L.b_pad * L.list_cap * 8; break;
        case SQE_I8_POOL_COUNTS: src = idx->i8ovf_cnt.p; size = (int64_t)L.b_pad * 4; break;
        case SQE_I8_POOLS: src = idx->i8ovf.p; size = (int64_t)L.b_pad * L.pool_cap * 8; break;
        case SQE_I8_SAMPLE_BEST:
            if (!L.sample_int8) return fail(SQE_ERR_STATE, "sqe_index_i8_read: the threshold pass of the last search did not run in int8");
            src = idx->i8samp.p; size = (int64_t)L.sample_chunks * L.sample_b_pad * 16 * 8; break;
        default: return fail(SQE_ERR_INVALID, "sqe_index_i8_read: unknown buffer");
    }
    if (offset < 0 || bytes < 0 || offset + bytes > size) return fail(SQE_ERR_INVALID, "sqe_index_i8_read: range outside the buffer");
    if (bytes == 0) return SQE_OK;
    SQE_HIP(hipMemcpyAsync(out_host, static_cast<const char*>(src) + offset, (size_t)bytes, hipMemcpyDeviceToHost, op.s));
    SQE_HIP(hipStreamSynchronize(op.s));
    return SQE_OK;
}

int sqe_index_search_device(sqe_index* idx, const float* q_dev, int B, int k, int nprobe,
                            float* cos_out_dev, int64_t* id_out_dev) {
    SQE_TRY(search_args_ok(idx, q_dev, B, k, cos_out_dev, id_out_dev));
    if (B == 0) return SQE_OK;
    if (idx->group) return group_index_search(idx, q_dev, B, k, nprobe, cos_out_dev, id_out_dev, true);
    OpScope op(idx->ctx, idx->ord, false);
    return index_search_impl(idx, q_dev, B, k, nprobe, cos_out_dev, id_out_dev, op.s);
}

int sqe_index_search(sqe_index* idx, const float* q_host, int B, int k, int nprobe,
                     float* cos_out_host, int64_t* id_out_host) {
    SQE_TRY(search_args_ok(idx, q_host, B, k, cos_out_host, id_out_host));
    if (B == 0) return SQE_OK;
    if (idx->group) return group_index_search(idx, q_host, B, k, nprobe, cos_out_host, id_out_host, false);
    OpScope op(idx->ctx, idx->ord, true);
    const size_t qbytes = (size_t)B * idx->dim * 4, cb = (size_t)B * k * 4, ib = (size_t)B * k * 8;
    SQE_TRY(idx->stage_in.ensure(qbytes));
    SQE_TRY(idx->stage_out.ensure(round_up((int64_t)cb, 16) + ib));
    float* cos_dev = idx->stage_out.as<float>();
    int64_t* id_dev = reinterpret_cast<int64_t*>(idx->stage_out.as<char>() + round_up((int64_t)cb, 16));
    SQE_HIP(hipMemcpyAsync(idx->stage_in.p, q_host, qbytes, hipMemcpyHostToDevice, op.s));
    SQE_TRY(index_search_impl(idx, idx->stage_in.as<float>(), B, k, nprobe, cos_dev, id_dev, op.s));
    SQE_HIP(hipMemcpyAsync(cos_out_host, cos_dev, cb, hipMemcpyDeviceToHost, op.s));
    SQE_HIP(hipMemcpyAsync(id_out_host, id_dev, ib, hipMemcpyDeviceToHost, op.s));
    SQE_HIP(hipStreamSynchronize(op.s));
    return SQE_OK;
}

int sqe_index_train_device(sqe_index* idx, const float* x_dev, int64_t n, int iters, uint64_t seed) {
    if (!idx) return fail(SQE_ERR_INVALID, "null index");
    if (idx->group) {
        if (!x_dev || n <= 0) return fail(SQE_ERR_INVALID, "sqe_index_train: empty training set");
        return group_index_train(idx, x_dev, n, iters, seed, true);
    }
    if (!idx->ivf) return fail(SQE_ERR_STATE, "sqe_index_train: not an IVF index");
    if (!x_dev || n <= 0) return fail(SQE_ERR_INVALID, "sqe_index_train: empty training set");
    OpScope op(idx->ctx, idx->ord, false);
    return ivf_train(idx, idx->ivf, x_dev, n, iters, seed, op.s);
}

int sqe_index_train(sqe_index* idx, const float* x_host, int64_t n, int iters, uint64_t seed) {
    if (!idx) return fail(SQE_ERR_INVALID, "null index");
    if (idx->group) {
        if (!x_host || n <= 0) return fail(SQE_ERR_INVALID, "sqe_index_train: empty training set");
        return group_index_train(idx, x_host, n, iters, seed, false);
    }
    if (!idx->ivf) return fail(SQE_ERR_STATE, "sqe_index_train: not an IVF index");
    if (!x_host || n <= 0) return fail(SQE_ERR_INVALID, "sqe_index_train: empty training set");
    OpScope op(idx->ctx, idx->ord, true);
    DevBuf tmp;
    SQE_TRY(tmp.ensure((size_t)n * idx->dim * 4));
    SQE_HIP(hipMemcpyAsync(tmp.p, x_host, (size_t)n * idx->dim * 4, hipMemcpyHostToDevice, op.s));
    SQE_TRY(ivf_train(idx, idx->ivf, tmp.as<float>(), n, iters, seed, op.s));
    SQE_HIP(hipStreamSynchronize(op.s));
    return SQE_OK;
}

int sqe_index_ivf_export(sqe_index* idx, float* centroids_host, int32_t* assign_host) {
    if (!idx) return fail(SQE_ERR_INVALID, "null index");
    if (idx->group) {
        if (!group_index_ivf_trained(idx)) return fail(SQE_ERR_STATE, "sqe_index_ivf_export: not a trained IVF index");
        return group_index_ivf_export(idx, centroids_host, assign_host);
    }
    if (!idx->ivf) return fail(SQE_ERR_STATE, "sqe_index_ivf_export: not an IVF index");
    OpScope op(idx->ctx, idx->ord, true);
    return ivf_export(idx, idx->ivf, centroids_host, assign_host, op.s);
}

// ---------------------------------------------------------------- persistence
// File: 64-byte header | master rows [n, dim] fp32 | (IVF, trained) centroids [nlist, dim] fp32 | assign [n] int32.
// The bf16 scan copy, residuals and IVF lists are derived data and are rebuilt on load.  Rows are in GLOBAL row
// order whatever the number of devices the index was spread over, so a file written by an 8-device context
// loads on one device and the other way round.
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
    FileCloser fc{fopen(path, "wb")};
    if (!fc.f) return fail(SQE_ERR_IO, std::string("sqe_index_save: cannot open ") + path);
    PinnedBuf pin;
    SQE_TRY(pin.alloc(IO_CHUNK));
    SaveHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "SQEIDX01", 8);
    h.version = 1; h.dim = (uint32_t)idx->dim; h.kind = (uint32_t)idx->kind; h.nlist = (uint32_t)idx->nlist;
    if (idx->group) {
        SQE_TRY(group_index_count(idx, &h.n));
        const bool givf = group_index_ivf_trained(idx);
        h.id_base = idx->id_base; h.flags = givf ? 1u : 0u; h.certify = (uint32_t)idx->certify;
        if (fwrite(&h, 1, sizeof(h), fc.f) != sizeof(h)) return fail(SQE_ERR_IO, "sqe_index_save: short write");
        SQE_TRY(group_index_save_rows(idx, fc.f, pin.p, IO_CHUNK));
        if (givf) {
            // the same IVF section a single-device index writes: centroids, then the list of every row in global order
            std::vector<float> cent((size_t)idx->nlist * idx->dim);
            std::vector<int32_t> assign((size_t)std::max<int64_t>(h.n, 1));
            SQE_TRY(group_index_ivf_export(idx, cent.data(), assign.data()));
            if (fwrite(cent.data(), 4, cent.size(), fc.f) != cent.size()) return fail(SQE_ERR_IO, "sqe_index_save: short write");
            if (h.n > 0 && fwrite(assign.data(), 4, (size_t)h.n, fc.f) != (size_t)h.n)
                return fail(SQE_ERR_IO, "sqe_index_save: short write");
        }
        if (fflush(fc.f) != 0) return fail(SQE_ERR_IO, "sqe_index_save: flush failed");
        return SQE_OK;
    }
    OpScope op(idx->ctx, idx->ord, true);
    const bool ivf = idx->ivf && ivf_trained(idx->ivf);
    if (ivf) SQE_TRY(ivf_rows_added(idx, idx->ivf, op.s));
    const int64_t n = idx->n.load();
    h.n = n; h.id_base = idx->id_base; h.flags = ivf ? 1u : 0u; h.certify = (uint32_t)idx->certify;
    if (fwrite(&h, 1, sizeof(h), fc.f) != sizeof(h)) return fail(SQE_ERR_IO, "sqe_index_save: short write");
    SQE_HIP(hipStreamSynchronize(op.s));
    SQE_TRY(write_device_range(fc.f, idx->master, (size_t)n * idx->dim * 4, pin.p, op.s));
    if (ivf) {
        std::vector<float> cent((size_t)idx->nlist * idx->dim);
        std::vector<int32_t> assign((size_t)std::max<int64_t>(n, 1));
        SQE_TRY(ivf_export(idx, idx->ivf, cent.data(), assign.data(), op.s));
        if (fwrite(cent.data(), 4, cent.size(), fc.f) != cent.size()) return fail(SQE_ERR_IO, "sqe_index_save: short write");
        if (n > 0 && fwrite(assign.data(), 4, (size_t)n, fc.f) != (size_t)n)
            return fail(SQE_ERR_IO, "sqe_index_save: short write");
    }
    if (fflush(fc.f) != 0) return fail(SQE_ERR_IO, "sqe_index_save: flush failed");
    return SQE_OK;
}

int sqe_index_load(sqe_ctx* ctx, const char* path, sqe_index** out) {
    if (!ctx || !path || !out) return fail(SQE_ERR_INVALID, "sqe_index_load: null argument");
    *out = nullptr;
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
    SQE_TRY(sqe_index_set_option(idx, "id_base", (double)h.id_base));
    SQE_TRY(sqe_index_set_option(idx, "certify", (double)h.certify));
    SQE_TRY(sqe_index_reserve(idx, h.n));
    PinnedBuf pin;
    SQE_TRY(pin.alloc(IO_CHUNK));
    const size_t row_bytes = (size_t)h.dim * 4;
    const int64_t rows_per_step = std::max<int64_t>(1, (int64_t)(IO_CHUNK / row_bytes));
    if (idx->group) {
        for (int64_t off = 0; off < h.n; off += rows_per_step) {
            const int64_t m = std::min(rows_per_step, h.n - off);
            if (fread(pin.p, row_bytes, (size_t)m, fc.f) != (size_t)m) return fail(SQE_ERR_IO, "sqe_index_load: file is truncated");
            SQE_TRY(group_index_add(idx, (const float*)pin.p, m, false, true));   // synchronises: the pinned buffer is reused
        }
        if (h.flags & 1u) {
            if (idx->kind != SQE_INDEX_IVF_FLAT) return fail(SQE_ERR_IO, "sqe_index_load: IVF section in a flat index file");
            const size_t cb = (size_t)h.nlist * row_bytes, ab = (size_t)h.n * 4;
            std::vector<char> host(cb + ab);
            if (fread(host.data(), 1, cb + ab, fc.f) != cb + ab) return fail(SQE_ERR_IO, "sqe_index_load: file is truncated");
            SQE_TRY(group_index_ivf_restore(idx, (const float*)host.data(), (const int32_t*)(host.data() + cb), h.n));
        }
        guard.i = nullptr;
        *out = idx;
        return SQE_OK;
    }
    OpScope op(ctx, idx->ord, true);
    SQE_TRY(idx->stage_in.ensure((size_t)std::min<int64_t>(rows_per_step, std::max<int64_t>(h.n, 1)) * row_bytes));
    for (int64_t off = 0; off < h.n; off += rows_per_step) {
        const int64_t m = std::min(rows_per_step, h.n - off);
        if (fread(pin.p, row_bytes, (size_t)m, fc.f) != (size_t)m) return fail(SQE_ERR_IO, "sqe_index_load: file is truncated");
        SQE_HIP(hipMemcpyAsync(idx->stage_in.p, pin.p, (size_t)m * row_bytes, hipMemcpyHostToDevice, op.s));
        SQE_TRY(index_add_impl(idx, idx->stage_in.as<float>(), m, h.dim, true, op.s));
        SQE_HIP(hipStreamSynchronize(op.s));    // the pinned buffer is reused
    }
    if (h.flags & 1u) {
        if (!idx->ivf) return fail(SQE_ERR_IO, "sqe_index_load: IVF section in a flat index file");
        const size_t cb = (size_t)h.nlist * row_bytes, ab = (size_t)h.n * 4;
        std::vector<char> host(cb + ab);
        if (fread(host.data(), 1, cb + ab, fc.f) != cb + ab) return fail(SQE_ERR_IO, "sqe_index_load: file is truncated");
        DevBuf tmp;
        SQE_TRY(tmp.ensure(cb + std::max<size_t>(ab, 4)));
        SQE_HIP(hipMemcpyAsync(tmp.p, host.data(), cb + ab, hipMemcpyHostToDevice, op.s));
        SQE_TRY(ivf_restore(idx, idx->ivf, tmp.as<float>(), (const int32_t*)((char*)tmp.p + cb), h.n, op.s));
        SQE_HIP(hipStreamSynchronize(op.s));
    }
    guard.i = nullptr;
    *out = idx;
    return SQE_OK;
}

int sqe_merge_topk_device(sqe_ctx* ctx, const float* cos_parts_dev, const int64_t* id_parts_dev,
                          int64_t part_stride_bytes, int P, int B, int k,
                          float* cos_out_dev, int64_t* id_out_dev) {
    if (!ctx) return fail(SQE_ERR_INVALID, "null handle");
    if (!cos_parts_dev || !id_parts_dev || !cos_out_dev || !id_out_dev)
        return fail(SQE_ERR_INVALID, "sqe_merge_topk: null buffer");
    if (part_stride_bytes < 0 || part_stride_bytes % 8 != 0) return fail(SQE_ERR_INVALID, "sqe_merge_topk: bad part stride");
    SQE_HIP(hipSetDevice(ctx->device));
    return launch_merge_topk(cos_parts_dev, id_parts_dev, part_stride_bytes, P, B, k, cos_out_dev, id_out_dev, 1, 0, 0,
                             ctx->stream.load());
}

// ================================================================ cache scan
// one-shot form: context-level host operation (its buffer and stream order belong to the context)
static int cosine_scan_host(sqe_ctx* ctx, const float* mat_host, int m, int dim, const float* q_host,
                            float* sims_out_host, float* best_sim, int32_t* best_idx) {
    if (m < 0 || dim <= 0 || dim % 4 != 0) return fail(SQE_ERR_INVALID, "cosine scan: bad m/dim");
    if (!q_host || (m > 0 && !mat_host)) return fail(SQE_ERR_INVALID, "cosine scan: null buffer");
    OpScope op(ctx, ctx->host, true);
    const size_t mb = (size_t)m * dim * 4, qb = (size_t)dim * 4, sb = round_up((int64_t)m * 4 + 4, 16);
    SQE_TRY(ctx->cache_tmp.ensure(mb + qb + sb + 16));
    char* base = ctx->cache_tmp.as<char>();
    float* d_mat = (float*)base;
    float* d_q = (float*)(base + mb);
    float* d_sims = (float*)(base + mb + qb);
    float* d_best = (float*)(base + mb + qb + sb);
    int32_t* d_idx = (int32_t*)(base + mb + qb + sb + 4);
    if (m > 0) SQE_HIP(hipMemcpyAsync(d_mat, mat_host, mb, hipMemcpyHostToDevice, op.s));
    SQE_HIP(hipMemcpyAsync(d_q, q_host, qb, hipMemcpyHostToDevice, op.s));
    {
        StageTimer t(ctx->prof, op.s, ST_CACHE);
        SQE_TRY(launch_cosine_scan(d_mat, nullptr, m, dim, d_q, d_sims, d_best, d_idx, op.s));
    }
    if (sims_out_host && m > 0)
        SQE_HIP(hipMemcpyAsync(sims_out_host, d_sims, (size_t)m * 4, hipMemcpyDeviceToHost, op.s));
    if (best_sim) SQE_HIP(hipMemcpyAsync(best_sim, d_best, 4, hipMemcpyDeviceToHost, op.s));
    if (best_idx) SQE_HIP(hipMemcpyAsync(best_idx, d_idx, 4, hipMemcpyDeviceToHost, op.s));
    SQE_HIP(hipStreamSynchronize(op.s));
    return SQE_OK;
}

int sqe_cosine_best(sqe_ctx* ctx, const float* mat_host, int m, int dim, const float* q_host,
                    float* best_sim, int32_t* best_idx) {
    if (!ctx) return fail(SQE_ERR_INVALID, "null handle");
    if (!best_sim || !best_idx) return fail(SQE_ERR_INVALID, "sqe_cosine_best: null output");
    return cosine_scan_host(ctx, mat_host, m, dim, q_host, nullptr, best_sim, best_idx);
}

int sqe_cosine_all(sqe_ctx* ctx, const float* mat_host, int m, int dim, const float* q_host,
                   float* sims_out_host) {
    if (!ctx) return fail(SQE_ERR_INVALID, "null handle");
    if (m > 0 && !sims_out_host) return fail(SQE_ERR_INVALID, "sqe_cosine_all: null output");
    float bs; int32_t bi;
    return cosine_scan_host(ctx, mat_host, m, dim, q_host, sims_out_host, &bs, &bi);
}

int sqe_cache_create(sqe_ctx* ctx, int capacity, int dim, sqe_cache** out) {
    if (!ctx) return fail(SQE_ERR_INVALID, "null handle");
    if (!out) return fail(SQE_ERR_INVALID, "sqe_cache_create: out is null");
    *out = nullptr;
    if (capacity <= 0 || dim <= 0 || dim % 4 != 0) return fail(SQE_ERR_INVALID, "sqe_cache_create: bad capacity/dim");
    SQE_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<sqe_cache> c(new (std::nothrow) sqe_cache);
    if (!c) return fail(SQE_ERR_OOM, "sqe_cache_create: host allocation failed");
    c->ctx = ctx; c->capacity = capacity; c->dim = dim;
    SQE_TRY(c->ord.init());
    int rc = c->mat.ensure((size_t)capacity * dim * 4);
    if (rc == SQE_OK) rc = c->work.ensure((size_t)dim * 4 + (size_t)capacity * 8 + 64);
    if (rc == SQE_OK) {
        OpScope op(ctx, c->ord, true);
        hipError_t e = hipMemsetAsync(c->mat.p, 0, (size_t)capacity * dim * 4, op.s);
        if (e != hipSuccess) rc = fail(SQE_ERR_HIP, std::string("cache memset: ") + hipGetErrorString(e));
    }
    if (rc != SQE_OK) { c->ord.destroy(); return rc; }
    *out = c.release();
    return SQE_OK;
}

void sqe_cache_destroy(sqe_cache* c) {
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    {
        std::lock_guard<std::mutex> lk(c->ord.mu);
        c->ord.quiesce();
    }
    c->ord.destroy();
    delete c;
}

int sqe_cache_set_slot(sqe_cache* c, int slot, const float* vec_host) {
    if (!c) return fail(SQE_ERR_INVALID, "null cache");
    if (slot < 0 || slot >= c->capacity || !vec_host) return fail(SQE_ERR_INVALID, "sqe_cache_set_slot: bad slot");
    OpScope op(c->ctx, c->ord, true);
    SQE_HIP(hipMemcpyAsync(c->mat.as<float>() + (size_t)slot * c->dim, vec_host, (size_t)c->dim * 4,
                           hipMemcpyHostToDevice, op.s));
    SQE_HIP(hipStreamSynchronize(op.s));
    return SQE_OK;
}

int sqe_cache_best(sqe_cache* c, const int32_t* order_host, int m, const float* q_host,
                   float* best_sim, int32_t* best_pos) {
    if (!c) return fail(SQE_ERR_INVALID, "null cache");
    if (m < 0 || m > c->capacity || !q_host || !best_sim || !best_pos || (m > 0 && !order_host))
        return fail(SQE_ERR_INVALID, "sqe_cache_best: bad arguments");
    for (int i = 0; i < m; ++i)
        if (order_host[i] < 0 || order_host[i] >= c->capacity) return fail(SQE_ERR_INVALID, "sqe_cache_best: slot out of range");
    sqe_ctx* ctx = c->ctx;
    OpScope op(ctx, c->ord, true);
    char* base = c->work.as<char>();
    float* d_q = (float*)base;
    float* d_sims = (float*)(base + (size_t)c->dim * 4);
    int32_t* d_order = (int32_t*)(base + (size_t)c->dim * 4 + (size_t)c->capacity * 4);
    float* d_best = (float*)(base + (size_t)c->dim * 4 + (size_t)c->capacity * 8);
    int32_t* d_idx = (int32_t*)(d_best + 1);
    SQE_HIP(hipMemcpyAsync(d_q, q_host, (size_t)c->dim * 4, hipMemcpyHostToDevice, op.s));
    if (m > 0) SQE_HIP(hipMemcpyAsync(d_order, order_host, (size_t)m * 4, hipMemcpyHostToDevice, op.s));
    {
        StageTimer t(ctx->prof, op.s, ST_CACHE);
        SQE_TRY(launch_cosine_scan(c->mat.as<float>(), d_order, m, c->dim, d_q, d_sims, d_best, d_idx, op.s));
    }
    SQE_HIP(hipMemcpyAsync(best_sim, d_best, 4, hipMemcpyDeviceToHost, op.s));
    SQE_HIP(hipMemcpyAsync(best_pos, d_idx, 4, hipMemcpyDeviceToHost, op.s));
    SQE_HIP(hipStreamSynchronize(op.s));
    return SQE_OK;
}

// ================================================================ stats
int sqe_set_profiling(sqe_ctx* ctx, int on) {
    if (!ctx) return fail(SQE_ERR_INVALID, "null handle");
    ctx->prof.drain();
    ctx->prof.on.store(on != 0);
    return SQE_OK;
}

// one context's share of the statistics, ADDED to *out (timings, calls, uncertified queries of its last search)
static int stats_accumulate(sqe_ctx* ctx, sqe_stats_t* out) {
    SQE_HIP(hipSetDevice(ctx->device));
    ctx->prof.drain();
    {
        std::lock_guard<std::mutex> lk(ctx->prof.mu);
        out->scan_ms += ctx->prof.ms[ST_SCAN] + ctx->prof.ms[ST_COLLECT];   // collect-pass scans are scans
        out->prep_ms += ctx->prof.ms[ST_PREP];
        out->select_ms += ctx->prof.ms[ST_SELECT];
        out->add_ms += ctx->prof.ms[ST_ADD];
        out->encode_ms += ctx->prof.ms[ST_ENCODE];
        out->cache_ms += ctx->prof.ms[ST_CACHE];
        out->sample_ms += ctx->prof.ms[ST_SAMPLE];
        out->scan_calls += ctx->prof.calls[ST_SCAN];
    }
    if (ctx->i8_valid.load()) {
        unsigned long long v[4] = {0, 0, 0, 0};
        SQE_HIP(hipDeviceSynchronize());
        if (hipMemcpy(v, ctx->i8_last.p, 32, hipMemcpyDeviceToHost) == hipSuccess) {
            out->i8_collected += (int64_t)v[0]; out->i8_rescored += (int64_t)v[1]; out->i8_overflows += (int64_t)v[2];
        }
    }
    if (ctx->unc_valid.load()) {
        // the count was written on the stream of that search; a device-wide wait orders this read after it
        // whatever stream it was (stats are not on any hot path)
        int v = 0;
        SQE_HIP(hipDeviceSynchronize());
        if (hipMemcpy(&v, ctx->unc_last.p, 4, hipMemcpyDeviceToHost) == hipSuccess) out->uncertified += v;
    }
    return SQE_OK;
}

int sqe_stats(sqe_ctx* ctx, sqe_stats_t* out) {
    if (!ctx) return fail(SQE_ERR_INVALID, "null handle");
    if (!out) return fail(SQE_ERR_INVALID, "sqe_stats: out is null");
    memset(out, 0, sizeof(*out));
    SQE_TRY(stats_accumulate(ctx, out));
    // A multi-device context: the shards of its indexes belong to the member contexts, which is where their
    // searches book timings and uncertified counts (r02: the leader reported shard 0 only).  Timings add up over
    // the members (device time, not wall time: the shards run side by side), so does the uncertified count (a query
    // can fail its certificate on one shard and pass on another: the sum counts collect passes, not queries).
    if (ctx->group) {
        const int n = group_member_count(ctx);
        for (int p = 1; p < n; ++p) SQE_TRY(stats_accumulate(group_member(ctx, p), out));
        SQE_HIP(hipSetDevice(ctx->device));
        out->scan_rows = 0; out->scan_flops = 0; out->scan_bytes = 0;
        for (int p = 0; p < n; ++p) {
            sqe_ctx* m = group_member(ctx, p);
            out->scan_rows += m->last_scan_rows.load();
            out->scan_flops += m->last_scan_flops.load();
            out->scan_bytes += m->last_scan_bytes.load();
        }
        out->search_calls = ctx->search_calls.load();
        return SQE_OK;
    }
    out->search_calls = ctx->search_calls.load();
    out->scan_rows = ctx->last_scan_rows.load();
    out->scan_flops = ctx->last_scan_flops.load();
    out->scan_bytes = ctx->last_scan_bytes.load();
    return SQE_OK;
}

static void stats_reset_one(sqe_ctx* ctx) {
    ctx->prof.drain();
    std::lock_guard<std::mutex> lk(ctx->prof.mu);
    for (int i = 0; i < ST_COUNT; ++i) { ctx->prof.ms[i] = 0; ctx->prof.calls[i] = 0; }
    ctx->search_calls.store(0);
    ctx->i8_valid.store(false);      // the int8 counters describe the last int8 search SINCE the reset
}

int sqe_stats_reset(sqe_ctx* ctx) {
    if (!ctx) return fail(SQE_ERR_INVALID, "null handle");
    stats_reset_one(ctx);
    if (ctx->group) {
        for (int p = 1; p < group_member_count(ctx); ++p) {
            sqe_ctx* m = group_member(ctx, p);
            (void)hipSetDevice(m->device);
            stats_reset_one(m);
        }
        (void)hipSetDevice(ctx->device);
    }
    return SQE_OK;
}

}  // extern "C"
