// group.hip -- one host process driving several MI355X: the multi-device form of the context.
//
// The reference is ONE uvicorn process (main.py:738-739) that calls add_embeddings from a pool thread and search
// on the event loop; a drop-in OpenSearchIndexer therefore needs all the GPUs of the node behind one handle:
// sqe_create(device_ids, n_dev > 1) returns a context that leads a GROUP of member contexts, one per device,
// and every flat index created on it is sharded over them.
//
//   * Placement.  Global row g lives on shard g % P at local row g / P (P = shards): appends stay balanced to
//     within one row whatever the call sizes, a row is located without a table, and an append of n rows is one
//     strided view per shard (host: hipMemcpy2DAsync, device: the normalise kernel reads every P-th row over xGMI).
//   * Search.  The query batch goes to every device (host -> each device, or leader -> peers by
//     hipMemcpyPeerAsync); each shard runs the whole single-device pipeline on its own stream; ONE exchange of
//     the packed [B, k] results (ids int64 | cosines fp32: 120 KB per shard at B = 1024, k = 10 -- latency-bound,
//     so a single step): RCCL ncclAllGather inside one ncclGroupStart/End over a communicator per device
//     (ncclCommInitAll -- single process, many devices), or peer copies to the leader where RCCL is not
//     available / the shards are logical shards of one device; then the merge kernel on the leader maps
//     shard-local ids to global ones (local * P + shard) and keeps the best k, ties to the lowest global id.
//   * RCCL is loaded with dlopen at group creation (librccl.so.1): single-device users never map it, and a
//     process that already holds RCCL through torch shares that copy.
//
// P logical shards may share one device (device_ids = {0, 0, 0}): the same code path, with the copy
// exchange -- that is how the one-GPU test box rehearses it.  IVF indexes are not sharded by this layer.
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <memory>
#include <new>
#include <thread>

#include "internal.h"

namespace sqe {

namespace {

// ---- the handful of RCCL entry points, resolved at run time
typedef void* rccl_comm_t;
struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(rccl_comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(rccl_comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, rccl_comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool load() {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && AllGather;
    }
};
constexpr int RCCL_CHAR = 0;      // ncclInt8 / ncclChar

size_t packed_part_bytes(int B, int k) { return ((size_t)B * k * 12 + 15) / 16 * 16; }

}  // namespace

// One enqueue thread per member beyond the leader (r03, r02 verdict item 8): a search enqueues ~15 launches, copies and
// events per shard; from ONE host thread that was 0.46 ms for 8 shards of 1.25 M rows -- a fifth of one shard's 2.3 ms step
// (profiles/r03_configs/group_host_cost.jsonl), and on P real GPUs the last shard would start that much late.  The
// calling thread keeps shard 0; worker p sets its device once and runs shard p's closure; errors come back with their text
// (sqe_last_error is thread-local).
struct ShardWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> task;
    bool has_task = false, done = false, stop = false;
    int rc = SQE_OK;
    std::string err;
    int dev = 0;
    void loop() {
        (void)hipSetDevice(dev);
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return has_task || stop; });
            if (stop) return;
            std::function<int()> fn = std::move(task);
            has_task = false;
            lk.unlock();
            const int r = fn();
            std::string e = r == SQE_OK ? std::string() : std::string(sqe_last_error());
            lk.lock();
            rc = r;
            err = std::move(e);
            done = true;
            cv.notify_all();
        }
    }
    void post(std::function<int()> fn) {
        std::lock_guard<std::mutex> lk(mu);
        task = std::move(fn);
        has_task = true;
        done = false;
        cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
        if (rc != SQE_OK) set_error(err);
        return rc;
    }
};

struct Group {
    int P = 0;
    std::vector<std::unique_ptr<ShardWorker>> workers;   // [P - 1]: worker p - 1 enqueues for member p
    std::vector<int> devs;
    std::vector<sqe_ctx*> members;        // members[0] is the leader itself
    std::vector<char> peer_ok;            // leader memory is directly addressable from member p
    int exchange = SQE_EXCHANGE_COPY;     // resolved mode
    Rccl rccl;
    std::vector<rccl_comm_t> comms;
    std::mutex coll_mu;                   // RCCL group calls of one process must not interleave
    std::mutex fan_mu;                    // the workers have ONE task slot each: a fan-out holds this from its first post() to its last
                                          // wait() (two indexes of one context searched from two threads share the workers; their
                                          // per-index locks do not exclude each other)
};

struct GroupIndex {
    std::vector<sqe_index*> shards;
    // per shard, on its device
    std::vector<std::unique_ptr<DevBuf>> qbuf, gather, stage;
    std::vector<hipEvent_t> ev;           // shard p's part is in the leader's gather buffer / its search is done
    hipEvent_t ev_q = nullptr;            // the leader's query batch is ready to be copied to the peers
    DevBuf out;                           // leader: merged [B,k] cos | ids for the host entry point
};

namespace {

int64_t shard_rows_of(int64_t n_total, int P, int p) { return (n_total - p + P - 1) / P; }   // rows g < n_total with g % P == p

// scopes of one group operation: the group index lock, then every shard's OpScope (device p, its stream)
struct GroupScope {
    sqe_index* gi;
    std::vector<std::unique_ptr<OpScope>> ops;
    GroupScope(sqe_index* idx, bool host_call) : gi(idx) {
        gi->ord.mu.lock();
        Group* g = idx->ctx->group;
        for (int p = 0; p < g->P; ++p)
            ops.emplace_back(new OpScope(g->members[p], idx->group->shards[p]->ord, p == 0 ? host_call : true));
    }
    hipStream_t s(int p) const { return ops[p]->s; }
    ~GroupScope() {
        Group* g = gi->ctx->group;
        for (int p = g->P - 1; p >= 0; --p) {
            (void)hipSetDevice(g->devs[p]);
            ops[p].reset();
        }
        (void)hipSetDevice(g->devs[0]);
        gi->ord.mu.unlock();
    }
};

int sync_all(const GroupScope& sc, Group* g) {
    for (int p = 0; p < g->P; ++p) {
        SQE_HIP(hipSetDevice(g->devs[p]));
        SQE_HIP(hipStreamSynchronize(sc.s(p)));
    }
    SQE_HIP(hipSetDevice(g->devs[0]));
    return SQE_OK;
}

}  // namespace

// ================================================================ group lifetime
int group_create(sqe_ctx* leader, const int* device_ids, int n, int exchange) {
    if (n == 1 && exchange == SQE_EXCHANGE_AUTO) return SQE_OK;      // a plain single-device context
    int count = 0;
    SQE_HIP(hipGetDeviceCount(&count));
    std::unique_ptr<Group> g(new (std::nothrow) Group);
    if (!g) return fail(SQE_ERR_OOM, "sqe_create: host allocation failed");
    g->P = n;
    g->devs.assign(device_ids, device_ids + n);
    bool distinct = true;
    for (int i = 0; i < n; ++i) {
        if (device_ids[i] < 0 || device_ids[i] >= count) return fail(SQE_ERR_INVALID, "sqe_create: no such HIP device");
        for (int j = 0; j < i; ++j) distinct = distinct && device_ids[i] != device_ids[j];
    }
    g->members.push_back(leader);
    g->peer_ok.assign(n, 1);
    leader->group = g.get();       // from here on sqe_destroy(leader) releases the members
    Group* gp = g.release();
    for (int p = 1; p < n; ++p) {
        sqe_ctx* m = nullptr;
        const int one[1] = {device_ids[p]};
        SQE_TRY(sqe_create(one, 1, &m));
        gp->members.push_back(m);
        if (device_ids[p] != device_ids[0]) {
            int can = 0;
            (void)hipDeviceCanAccessPeer(&can, device_ids[p], device_ids[0]);
            bool ok = can != 0;
            if (ok) {
                SQE_HIP(hipSetDevice(device_ids[p]));
                hipError_t e = hipDeviceEnablePeerAccess(device_ids[0], 0);
                ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
                (void)hipGetLastError();
                SQE_HIP(hipSetDevice(device_ids[0]));
                e = hipDeviceEnablePeerAccess(device_ids[p], 0);
                ok = ok && (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled);
                (void)hipGetLastError();
            }
            gp->peer_ok[p] = ok ? 1 : 0;
        }
    }
    SQE_HIP(hipSetDevice(device_ids[0]));
    // exchange step: RCCL needs one communicator per DISTINCT device
    gp->exchange = SQE_EXCHANGE_COPY;
    if (exchange != SQE_EXCHANGE_COPY && distinct) {
        if (gp->rccl.load()) {
            gp->comms.assign(n, nullptr);
            const int rc = gp->rccl.CommInitAll(gp->comms.data(), n, gp->devs.data());
            if (rc == 0) gp->exchange = SQE_EXCHANGE_RCCL;
            else if (exchange == SQE_EXCHANGE_RCCL)
                return fail(SQE_ERR_HIP, std::string("ncclCommInitAll: ") + (gp->rccl.GetErrorString ? gp->rccl.GetErrorString(rc) : "failed"));
            else gp->comms.clear();
        } else if (exchange == SQE_EXCHANGE_RCCL) {
            return fail(SQE_ERR_UNSUPPORTED, "sqe_create_sharded: librccl.so.1 could not be loaded");
        }
        SQE_HIP(hipSetDevice(device_ids[0]));
    } else if (exchange == SQE_EXCHANGE_RCCL) {
        return fail(SQE_ERR_INVALID, "sqe_create_sharded: the RCCL exchange needs distinct devices (logical shards of one device use the copy exchange)");
    }
    for (int p = 1; p < n; ++p) {
        gp->workers.emplace_back(new ShardWorker);
        ShardWorker* w = gp->workers.back().get();
        w->dev = device_ids[p];
        w->th = std::thread([w] { w->loop(); });
    }
    return SQE_OK;
}

void group_destroy(sqe_ctx* leader) {
    Group* g = leader->group;
    if (!g) return;
    for (auto& w : g->workers) {
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->stop = true;
            w->cv.notify_all();
        }
        if (w->th.joinable()) w->th.join();
    }
    g->workers.clear();
    for (size_t p = 0; p < g->comms.size(); ++p)
        if (g->comms[p]) (void)g->rccl.CommDestroy(g->comms[p]);
    for (size_t p = 1; p < g->members.size(); ++p) sqe_destroy(g->members[p]);
    leader->group = nullptr;
    delete g;
}

int group_member_count(const sqe_ctx* leader) { return leader->group ? (int)leader->group->members.size() : 1; }
sqe_ctx* group_member(sqe_ctx* leader, int p) { return leader->group ? leader->group->members[p] : leader; }

int group_describe(sqe_ctx* leader, int* n_shards, int* exchange, int* device_ids, int cap) {
    Group* g = leader->group;
    if (n_shards) *n_shards = g->P;
    if (exchange) *exchange = g->exchange;
    for (int p = 0; device_ids && p < g->P && p < cap; ++p) device_ids[p] = g->devs[p];
    return SQE_OK;
}

// ================================================================ group index
int group_index_create(sqe_ctx* leader, int dim, int kind, int nlist, sqe_index** out) {
    *out = nullptr;
    Group* g = leader->group;
    if (kind != SQE_INDEX_FLAT && kind != SQE_INDEX_IVF_FLAT) return fail(SQE_ERR_INVALID, "sqe_index_create: unknown index kind");
    if (kind == SQE_INDEX_IVF_FLAT && (nlist < 1 || nlist > (1 << 20)))
        return fail(SQE_ERR_INVALID, "sqe_index_create: IVF needs 1 <= nlist <= 2^20");
    std::unique_ptr<sqe_index> idx(new (std::nothrow) sqe_index);
    std::unique_ptr<GroupIndex> gi(new (std::nothrow) GroupIndex);
    if (!idx || !gi) return fail(SQE_ERR_OOM, "sqe_index_create: host allocation failed");
    idx->ctx = leader;
    idx->dim = dim;
    idx->kind = kind;
    idx->nlist = kind == SQE_INDEX_IVF_FLAT ? nlist : 0;
    SQE_HIP(hipSetDevice(leader->device));
    SQE_TRY(idx->ord.init());
    SQE_HIP(hipEventCreateWithFlags(&gi->ev_q, hipEventDisableTiming));
    idx->group = gi.release();
    sqe_index* raw = idx.release();
    const int idx_nlist = raw->nlist;
    for (int p = 0; p < g->P; ++p) {
        sqe_index* sh = nullptr;
        // IVF: every shard is a complete IVF index over ITS rows with the same (replicated) centroids --
        // lists are sharded by the group's row ownership, the [B,k] exchange and the merge are the flat index's
        int rc = index_create_impl(g->members[p], dim, kind, idx_nlist, false, &sh);
        if (rc != SQE_OK) { group_index_destroy(raw); return rc; }
        raw->group->shards.push_back(sh);
        raw->group->qbuf.emplace_back(new DevBuf);
        raw->group->gather.emplace_back(new DevBuf);
        raw->group->stage.emplace_back(new DevBuf);
        hipEvent_t e = nullptr;
        (void)hipSetDevice(g->devs[p]);
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { group_index_destroy(raw); return fail(SQE_ERR_HIP, "hipEventCreate"); }
        raw->group->ev.push_back(e);
    }
    SQE_HIP(hipSetDevice(leader->device));
    raw->pitch = raw->group->shards[0]->pitch;
    *out = raw;
    return SQE_OK;
}

void group_index_destroy(sqe_index* idx) {
    if (!idx || !idx->group) return;
    Group* g = idx->ctx->group;
    GroupIndex* gi = idx->group;
    {
        std::lock_guard<std::mutex> lk(idx->ord.mu);
        for (size_t p = 0; p < gi->shards.size(); ++p) {
            (void)hipSetDevice(g->devs[p]);
            {
                std::lock_guard<std::mutex> lk2(gi->shards[p]->ord.mu);
                gi->shards[p]->ord.quiesce();
            }
            gi->qbuf[p].reset();
            gi->gather[p].reset();
            gi->stage[p].reset();
            if (p < gi->ev.size() && gi->ev[p]) (void)hipEventDestroy(gi->ev[p]);
            sqe_index_destroy(gi->shards[p]);
        }
        (void)hipSetDevice(g->devs[0]);
        if (gi->ev_q) (void)hipEventDestroy(gi->ev_q);
        gi->out.release();
    }
    idx->ord.destroy();
    delete gi;
    idx->group = nullptr;
    delete idx;
}

int group_index_reserve(sqe_index* idx, int64_t rows) {
    Group* g = idx->ctx->group;
    GroupScope sc(idx, true);
    for (int p = 0; p < g->P; ++p) {
        SQE_HIP(hipSetDevice(g->devs[p]));
        SQE_TRY(index_grow(idx->group->shards[p], shard_rows_of(rows, g->P, p), sc.s(p)));
    }
    return SQE_OK;
}

int group_index_count(const sqe_index* idx, int64_t* out) {
    int64_t n = 0;
    for (sqe_index* sh : idx->group->shards) n += sh->n.load();
    *out = n;
    return SQE_OK;
}

// Append n rows (ids count .. count + n - 1).  x: host block, or a device block on the LEADER device.
int group_index_add(sqe_index* idx, const float* x, int64_t n, bool x_on_device, bool restore) {
    Group* g = idx->ctx->group;
    GroupIndex* gi = idx->group;
    const int P = g->P, dim = idx->dim;
    GroupScope sc(idx, !x_on_device);
    int64_t total = 0;
    for (sqe_index* sh : gi->shards) total += sh->n.load();
    // global row total + i goes to shard (total + i) % P; shard p's first input row is i0 = (p - total) mod P
    if (x_on_device) {
        // the block is the caller's work on the leader's context stream (= sc.s(0)): the peers wait for it
        SQE_HIP(hipSetDevice(g->devs[0]));
        SQE_HIP(hipEventRecord(gi->ev_q, sc.s(0)));
    }
    for (int p = 0; p < P; ++p) {
        const int64_t i0 = ((p - total) % P + P) % P;
        const int64_t m = i0 < n ? (n - i0 + P - 1) / P : 0;
        if (m == 0) continue;
        sqe_index* sh = gi->shards[p];
        hipStream_t s = sc.s(p);
        SQE_HIP(hipSetDevice(g->devs[p]));
        if (x_on_device) {
            if (p > 0) SQE_HIP(hipStreamWaitEvent(s, gi->ev_q, 0));
            const float* src = x + i0 * dim;
            int64_t stride = (int64_t)P * dim;
            if (!g->peer_ok[p]) {
                // no direct access to the leader's memory: bring the block over, then take every P-th row locally
                SQE_TRY(gi->stage[p]->ensure((size_t)n * dim * 4));
                SQE_HIP(hipMemcpyPeerAsync(gi->stage[p]->p, g->devs[p], x, g->devs[0], (size_t)n * dim * 4, s));
                src = gi->stage[p]->as<float>() + i0 * dim;
            }
            SQE_TRY(index_add_impl(sh, src, m, stride, restore, s));
        } else {
            const int64_t rows_per_step = std::max<int64_t>(1, (64ll << 20) / ((int64_t)dim * 4));
            SQE_TRY(gi->stage[p]->ensure((size_t)std::min(rows_per_step, m) * dim * 4));
            for (int64_t off = 0; off < m; off += rows_per_step) {
                const int64_t mm = std::min(rows_per_step, m - off);
                SQE_HIP(hipMemcpy2DAsync(gi->stage[p]->p, (size_t)dim * 4, x + (i0 + off * P) * dim, (size_t)P * dim * 4,
                                         (size_t)dim * 4, (size_t)mm, hipMemcpyHostToDevice, s));
                SQE_TRY(index_add_impl(sh, gi->stage[p]->as<float>(), mm, dim, restore, s));
            }
        }
    }
    if (!x_on_device) SQE_TRY(sync_all(sc, g));           // x is not retained past return
    else {
        // the caller may reuse its block once the leader's stream says so: order the peers' reads before that
        for (int p = 1; p < P; ++p) {
            SQE_HIP(hipSetDevice(g->devs[p]));
            SQE_HIP(hipEventRecord(gi->ev[p], sc.s(p)));
            SQE_HIP(hipSetDevice(g->devs[0]));
            SQE_HIP(hipStreamWaitEvent(sc.s(0), gi->ev[p], 0));
        }
    }
    SQE_HIP(hipSetDevice(g->devs[0]));
    return SQE_OK;
}

int group_index_update(sqe_index* idx, const int64_t* rows_host, const float* x_host, int64_t n) {
    Group* g = idx->ctx->group;
    GroupIndex* gi = idx->group;
    const int P = g->P, dim = idx->dim;
    GroupScope sc(idx, true);
    int64_t total = 0;
    for (sqe_index* sh : gi->shards) total += sh->n.load();
    for (int64_t i = 0; i < n; ++i)
        if (rows_host[i] < 0 || rows_host[i] >= total) return fail(SQE_ERR_INVALID, "sqe_index_update: row out of range");
    std::vector<std::vector<int64_t>> local(P);
    std::vector<std::vector<float>> xs(P);
    for (int64_t i = 0; i < n; ++i) {
        const int p = (int)(rows_host[i] % P);
        local[p].push_back(rows_host[i] / P);
        xs[p].insert(xs[p].end(), x_host + i * dim, x_host + (i + 1) * dim);
    }
    for (int p = 0; p < P; ++p) {
        const int64_t m = (int64_t)local[p].size();
        if (m == 0) continue;
        SQE_HIP(hipSetDevice(g->devs[p]));
        hipStream_t s = sc.s(p);
        const size_t xb = (size_t)m * dim * 4, rb = (size_t)m * 8;
        SQE_TRY(gi->stage[p]->ensure(xb + rb));
        SQE_HIP(hipMemcpyAsync(gi->stage[p]->p, xs[p].data(), xb, hipMemcpyHostToDevice, s));
        SQE_HIP(hipMemcpyAsync((char*)gi->stage[p]->p + xb, local[p].data(), rb, hipMemcpyHostToDevice, s));
        SQE_TRY(index_update_impl(gi->shards[p], (const int64_t*)((char*)gi->stage[p]->p + xb), gi->stage[p]->as<float>(), m, s));
    }
    return sync_all(sc, g);
}

int group_index_get_rows(sqe_index* idx, const int64_t* rows_host, int64_t n, float* out_host) {
    Group* g = idx->ctx->group;
    GroupIndex* gi = idx->group;
    const int P = g->P, dim = idx->dim;
    GroupScope sc(idx, true);
    int64_t total = 0;
    for (sqe_index* sh : gi->shards) total += sh->n.load();
    for (int64_t i = 0; i < n; ++i) {
        const int64_t r = rows_host[i];
        if (r < 0 || r >= total) return fail(SQE_ERR_INVALID, "sqe_index_get_rows: row out of range");
        const int p = (int)(r % P);
        SQE_HIP(hipSetDevice(g->devs[p]));
        SQE_HIP(hipMemcpyAsync(out_host + (size_t)i * dim, gi->shards[p]->master + (size_t)(r / P) * dim, (size_t)dim * 4,
                               hipMemcpyDeviceToHost, sc.s(p)));
    }
    return sync_all(sc, g);
}

int group_index_set_option(sqe_index* idx, const char* key, double value) {
    const std::string k(key);
    if (k == "id_base") {
        if (value < 0) return fail(SQE_ERR_INVALID, "id_base must be >= 0");
        std::lock_guard<std::mutex> lk(idx->ord.mu);
        idx->id_base = (int64_t)value;           // applied by the merge; shards return local ids
        return SQE_OK;
    }
    for (sqe_index* sh : idx->group->shards) SQE_TRY(sqe_index_set_option(sh, key, value));
    if (k == "certify") idx->certify = value != 0.0;
    return SQE_OK;
}

// ---------------------------------------------------------------- IVF on a device group
// Training runs ONCE, on the leader's shard (k-means over the sample), and the normalised centroids are then
// replicated: every other shard takes them bit for bit (ivf_restore) and assigns its own rows.  All shards therefore
// probe the same lists in the same order, a shard scans the rows it owns of every probed list, and the merged result is
// the IVF search over the global lists.  x: the training sample, host or LEADER-device memory.
int group_index_train(sqe_index* idx, const float* x, int64_t n, int iters, uint64_t seed, bool x_on_device) {
    Group* g = idx->ctx->group;
    GroupIndex* gi = idx->group;
    const int P = g->P, dim = idx->dim, nlist = idx->nlist;
    if (idx->kind != SQE_INDEX_IVF_FLAT) return fail(SQE_ERR_STATE, "sqe_index_train: not an IVF index");
    GroupScope sc(idx, !x_on_device);        // a device block is the caller's work on the leader's context stream (= sc.s(0))
    SQE_HIP(hipSetDevice(g->devs[0]));
    sqe_index* lead = gi->shards[0];
    {
        DevBuf tmp;
        const float* xd = x;
        if (!x_on_device) {
            SQE_TRY(tmp.ensure((size_t)n * dim * 4));
            SQE_HIP(hipMemcpyAsync(tmp.p, x, (size_t)n * dim * 4, hipMemcpyHostToDevice, sc.s(0)));
            xd = tmp.as<float>();
        }
        SQE_TRY(ivf_train(lead, lead->ivf, xd, n, iters, seed, sc.s(0)));
        SQE_HIP(hipStreamSynchronize(sc.s(0)));
    }
    // replicate: leader -> host -> every member (16 MB at nlist 4096 x 1024; once per training)
    const size_t cb = (size_t)nlist * dim * 4;
    std::vector<float> cent(cb / 4);
    SQE_HIP(hipMemcpyAsync(cent.data(), ivf_coarse(lead->ivf)->master, cb, hipMemcpyDeviceToHost, sc.s(0)));
    SQE_HIP(hipStreamSynchronize(sc.s(0)));
    for (int p = 1; p < P; ++p) {
        SQE_HIP(hipSetDevice(g->devs[p]));
        sqe_index* sh = gi->shards[p];
        SQE_TRY(gi->stage[p]->ensure(cb));
        SQE_HIP(hipMemcpyAsync(gi->stage[p]->p, cent.data(), cb, hipMemcpyHostToDevice, sc.s(p)));
        SQE_TRY(ivf_restore(sh, sh->ivf, gi->stage[p]->as<float>(), nullptr, 0, sc.s(p)));   // no assignments yet ...
        SQE_TRY(ivf_rows_added(sh, sh->ivf, sc.s(p)));                                          // ... the shard makes its own
    }
    return sync_all(sc, g);
}

// centroids [nlist, dim] (the leader's = everyone's) and the list of every stored row in GLOBAL row order
int group_index_ivf_export(sqe_index* idx, float* centroids_host, int32_t* assign_host) {
    Group* g = idx->ctx->group;
    GroupIndex* gi = idx->group;
    const int P = g->P;
    if (idx->kind != SQE_INDEX_IVF_FLAT) return fail(SQE_ERR_STATE, "sqe_index_ivf_export: not an IVF index");
    GroupScope sc(idx, true);
    std::vector<int32_t> part;
    for (int p = 0; p < P; ++p) {
        SQE_HIP(hipSetDevice(g->devs[p]));
        sqe_index* sh = gi->shards[p];
        const int64_t m = sh->n.load();
        part.resize((size_t)std::max<int64_t>(m, 1));
        SQE_TRY(ivf_export(sh, sh->ivf, p == 0 ? centroids_host : nullptr, assign_host ? part.data() : nullptr, sc.s(p)));
        if (assign_host)
            for (int64_t i = 0; i < m; ++i) assign_host[i * P + p] = part[(size_t)i];      // local row i of shard p = global row i * P + p
    }
    SQE_HIP(hipSetDevice(g->devs[0]));
    return SQE_OK;
}

bool group_index_ivf_trained(sqe_index* idx) {
    return idx->kind == SQE_INDEX_IVF_FLAT && idx->group->shards[0]->ivf && ivf_trained(idx->group->shards[0]->ivf);
}

// sqe_index_load: centroids and the per-row assignment in GLOBAL row order (host) -> every shard's share
int group_index_ivf_restore(sqe_index* idx, const float* centroids_host, const int32_t* assign_host, int64_t n) {
    Group* g = idx->ctx->group;
    GroupIndex* gi = idx->group;
    const int P = g->P, dim = idx->dim, nlist = idx->nlist;
    GroupScope sc(idx, true);
    const size_t cb = (size_t)nlist * dim * 4;
    std::vector<int32_t> part;
    for (int p = 0; p < P; ++p) {
        SQE_HIP(hipSetDevice(g->devs[p]));
        sqe_index* sh = gi->shards[p];
        const int64_t m = sh->n.load();
        if (m != shard_rows_of(n, P, p)) return fail(SQE_ERR_STATE, "sqe_index_load: shard sizes do not match the file");
        part.resize((size_t)std::max<int64_t>(m, 1));
        for (int64_t i = 0; i < m; ++i) part[(size_t)i] = assign_host[i * P + p];
        SQE_TRY(gi->stage[p]->ensure(cb + (size_t)std::max<int64_t>(m, 1) * 4));
        SQE_HIP(hipMemcpyAsync(gi->stage[p]->p, centroids_host, cb, hipMemcpyHostToDevice, sc.s(p)));
        SQE_HIP(hipMemcpyAsync(gi->stage[p]->as<char>() + cb, part.data(), (size_t)m * 4, hipMemcpyHostToDevice, sc.s(p)));
        SQE_TRY(ivf_restore(sh, sh->ivf, gi->stage[p]->as<float>(), (const int32_t*)(gi->stage[p]->as<char>() + cb), m, sc.s(p)));
        SQE_HIP(hipStreamSynchronize(sc.s(p)));           // `part` is reused
    }
    SQE_HIP(hipSetDevice(g->devs[0]));
    return SQE_OK;
}

// q: [B, dim] raw queries, host or LEADER-device memory; outputs likewise.
int group_index_search(sqe_index* idx, const float* q, int B, int k, int nprobe, float* cos_out, int64_t* id_out, bool on_device) {
    Group* g = idx->ctx->group;
    GroupIndex* gi = idx->group;
    const int P = g->P, dim = idx->dim;
    const size_t qbytes = (size_t)B * dim * 4, part = packed_part_bytes(B, k);
    const size_t cb = (size_t)B * k * 4, ib = (size_t)B * k * 8;
    GroupScope sc(idx, !on_device);
    const bool rccl = g->exchange == SQE_EXCHANGE_RCCL;
    if (on_device) {
        SQE_HIP(hipSetDevice(g->devs[0]));
        SQE_HIP(hipEventRecord(gi->ev_q, sc.s(0)));
    }
    // ---- every shard: queries in, local top-k into its slot of the gather buffer (shard 0 from this thread, shard p from worker p)
    auto shard_step = [&, q, B, k, nprobe, on_device, qbytes, part, ib, rccl](int p) -> int {
        SQE_HIP(hipSetDevice(g->devs[p]));
        hipStream_t s = sc.s(p);
        // RCCL: every device holds the whole gather buffer (in-place all-gather); copy exchange: only the leader does
        SQE_TRY(gi->gather[p]->ensure((rccl || p == 0) ? part * P : part));
        char* slot = gi->gather[p]->as<char>() + ((rccl || p == 0) ? part * p : 0);
        const float* qp = q;
        if (!on_device) {
            SQE_TRY(gi->qbuf[p]->ensure(qbytes));
            SQE_HIP(hipMemcpyAsync(gi->qbuf[p]->p, q, qbytes, hipMemcpyHostToDevice, s));
            qp = gi->qbuf[p]->as<float>();
        } else if (p > 0) {
            SQE_TRY(gi->qbuf[p]->ensure(qbytes));
            SQE_HIP(hipStreamWaitEvent(s, gi->ev_q, 0));
            SQE_HIP(hipMemcpyPeerAsync(gi->qbuf[p]->p, g->devs[p], q, g->devs[0], qbytes, s));
            qp = gi->qbuf[p]->as<float>();
        }
        SQE_TRY(index_search_impl(gi->shards[p], qp, B, k, nprobe, reinterpret_cast<float*>(slot + ib), reinterpret_cast<int64_t*>(slot), s));
        return SQE_OK;
    };
    {
        static const bool serial = [] { const char* e = knob_env("SQE_GROUP_SERIAL"); return e && e[0] == '1'; }();   // knobs build: the r02 form, for A/B
        int rc = SQE_OK;
        if (serial || g->workers.size() != (size_t)(P - 1)) {
            for (int p = 0; p < P && rc == SQE_OK; ++p) rc = shard_step(p);
        } else {
            std::lock_guard<std::mutex> fan(g->fan_mu);
            for (int p = 1; p < P; ++p) g->workers[p - 1]->post([&shard_step, p] { return shard_step(p); });
            rc = shard_step(0);
            for (int p = 1; p < P; ++p) {                    // every posted closure has run before this frame (shard_step, sc) goes away
                const int r = g->workers[p - 1]->wait();
                if (rc == SQE_OK) rc = r;
            }
        }
        SQE_HIP(hipSetDevice(g->devs[0]));
        if (rc != SQE_OK) return rc;
    }
    // ---- ONE exchange step
    if (rccl) {
        std::lock_guard<std::mutex> lk(g->coll_mu);
        int rc = g->rccl.GroupStart();
        for (int p = 0; p < P && rc == 0; ++p) {
            char* buf = gi->gather[p]->as<char>();
            rc = g->rccl.AllGather(buf + part * p, buf, part, RCCL_CHAR, g->comms[p], sc.s(p));
        }
        const int rc2 = g->rccl.GroupEnd();
        if (rc != 0 || rc2 != 0)
            return fail(SQE_ERR_HIP, std::string("ncclAllGather: ") + (g->rccl.GetErrorString ? g->rccl.GetErrorString(rc ? rc : rc2) : "failed"));
    } else {
        for (int p = 1; p < P; ++p) {
            SQE_HIP(hipSetDevice(g->devs[p]));
            SQE_HIP(hipMemcpyPeerAsync(gi->gather[0]->as<char>() + part * p, g->devs[0], gi->gather[p]->p, g->devs[p], part, sc.s(p)));
            SQE_HIP(hipEventRecord(gi->ev[p], sc.s(p)));
        }
        SQE_HIP(hipSetDevice(g->devs[0]));
        for (int p = 1; p < P; ++p) SQE_HIP(hipStreamWaitEvent(sc.s(0), gi->ev[p], 0));
    }
    // ---- merge on the leader: shard-local ids -> global (local * P + shard + id_base), ties to the lowest global id
    SQE_HIP(hipSetDevice(g->devs[0]));
    hipStream_t s0 = sc.s(0);
    float* cos_dev = cos_out;
    int64_t* id_dev = id_out;
    if (!on_device) {
        SQE_TRY(gi->out.ensure((size_t)round_up((int64_t)cb, 16) + ib));
        cos_dev = gi->out.as<float>();
        id_dev = reinterpret_cast<int64_t*>(gi->out.as<char>() + round_up((int64_t)cb, 16));
    }
    const char* gb = gi->gather[0]->as<char>();
    SQE_TRY(launch_merge_topk(reinterpret_cast<const float*>(gb + ib), reinterpret_cast<const int64_t*>(gb), (int64_t)part, P, B, k,
                              cos_dev, id_dev, P, 1, idx->id_base, s0));
    if (!on_device) {
        SQE_HIP(hipMemcpyAsync(cos_out, cos_dev, cb, hipMemcpyDeviceToHost, s0));
        SQE_HIP(hipMemcpyAsync(id_out, id_dev, ib, hipMemcpyDeviceToHost, s0));
        SQE_TRY(sync_all(sc, g));
    }
    return SQE_OK;
}

// rows in GLOBAL order into f (sqe_index_save): chunk by chunk, every shard's strided part of the chunk
int group_index_save_rows(sqe_index* idx, FILE* f, void* pinned, size_t pinned_bytes) {
    Group* g = idx->ctx->group;
    GroupIndex* gi = idx->group;
    const int P = g->P, dim = idx->dim;
    GroupScope sc(idx, true);
    int64_t total = 0;
    for (sqe_index* sh : gi->shards) total += sh->n.load();
    const size_t row_bytes = (size_t)dim * 4;
    const int64_t step = std::max<int64_t>(P, (int64_t)(pinned_bytes / row_bytes) / P * P);   // a multiple of P rows
    for (int64_t g0 = 0; g0 < total; g0 += step) {
        const int64_t g1 = std::min(total, g0 + step);
        for (int p = 0; p < P; ++p) {
            // g0 is a multiple of P: shard p's rows of the chunk are global g0 + p, g0 + p + P, ...
            const int64_t first = g0 + p;
            if (first >= g1) continue;
            const int64_t cnt = (g1 - first + P - 1) / P;
            SQE_HIP(hipSetDevice(g->devs[p]));
            SQE_HIP(hipMemcpy2DAsync((char*)pinned + (size_t)p * row_bytes, (size_t)P * row_bytes,
                                     gi->shards[p]->master + (size_t)(first / P) * dim, row_bytes, row_bytes, (size_t)cnt,
                                     hipMemcpyDeviceToHost, sc.s(p)));
        }
        SQE_TRY(sync_all(sc, g));
        const size_t bytes = (size_t)(g1 - g0) * row_bytes;
        if (fwrite(pinned, 1, bytes, f) != bytes) return fail(SQE_ERR_IO, "sqe_index_save: short write");
    }
    return SQE_OK;
}

}  // namespace sqe
