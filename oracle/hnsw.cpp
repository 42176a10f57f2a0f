// hnsw.cpp -- CPU HNSW (Malkov & Yashunin, "Efficient and robust approximate nearest neighbor search
// using Hierarchical Navigable Small World graphs", 2018) with the parameters of the reference's index
// mapping (app/main.py:272-276: hnsw, nmslib, cosinesimil, m = 64, ef_construction = 500).
//
// TEST / BASELINE INFRASTRUCTURE (oracle/): this is the ANN the reference's OpenSearch index actually
// runs, restated from the paper (Algorithms 1-5: layered insertion, greedy descent, ef-bounded best-first
// search on layer 0, neighbour selection by the heuristic of Algorithm 4 as nmslib's default
// delaunay_type = 2 does), so that bench.py can time a faithful CPU baseline and tests can state what
// recall that index has against the exact answer.  nmslib itself is not in this environment; parity with
// its implementation details (tie handling, level generator) is unpinned.
//
// Vectors must be L2-normalised: similarity = inner product = cosine.  Build is multi-threaded (std::thread workers
// pulling node ids from an atomic counter, per-node locks); search is one query per thread.
//
// Concurrency rule of the build (r03; the r02 build missed planted exact matches one run in three on 4 threads): a
// node must not be read before it is completely linked.  connect() back-links a node at layer l before its lists at
// the layers below exist, so a concurrent insertion could pick it as the entry point of layer l - 1, find an empty
// list there and link itself to that one isolated node only.  As hnswlib does, the inserting thread now holds the
// node's lock for the WHOLE insertion and every reader takes that lock to copy a list: a reader that reaches a
// half-linked node waits until it is finished.  No deadlock: for thread A to wait for b, b is linked at A's current
// layer, so B's current layer is lower; for B to wait for a at the same time a would have to be linked at B's layer,
// i.e. A's layer lower than B's.  `make -C oracle tsan` runs a 4-thread build under ThreadSanitizer (no OpenMP
// runtime in the picture: it is not instrumented and reports false races).
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <queue>
#include <random>
#include <vector>

#include <atomic>
#include <thread>

namespace {

struct Hnsw {
    const float* x = nullptr;      // [n, dim], not owned
    int64_t n = 0;
    int dim = 0, M = 0, M0 = 0, efc = 0;
    int max_level = -1;
    int64_t entry = -1;
    std::vector<int> level;                       // level of every node
    std::vector<int32_t> link0;                   // layer 0: [n][M0 + 1] (count, neighbours)
    std::vector<std::vector<int32_t>> link_up;    // per node: levels 1..level, [(M + 1) per level]
    std::vector<std::mutex> locks;
    std::mutex entry_lock;
};

inline float dot(const float* a, const float* b, int dim) {
    float s = 0.f;
#pragma omp simd reduction(+ : s)
    for (int i = 0; i < dim; ++i) s += a[i] * b[i];
    return s;
}

struct Visited {
    std::vector<uint32_t> mark;
    uint32_t epoch = 0;
    void reset(int64_t n) {
        if ((int64_t)mark.size() != n) { mark.assign(n, 0); epoch = 0; }
        if (++epoch == 0) { std::fill(mark.begin(), mark.end(), 0); epoch = 1; }
    }
    bool test_set(int64_t i) {
        if (mark[i] == epoch) return true;
        mark[i] = epoch;
        return false;
    }
};

typedef std::pair<float, int32_t> SimId;   // (similarity, node)

inline int32_t* links(Hnsw& h, int64_t node, int lvl) {
    if (lvl == 0) return h.link0.data() + (size_t)node * (h.M0 + 1);
    return h.link_up[node].data() + (size_t)(lvl - 1) * (h.M + 1);
}

// Algorithm 2: ef-bounded best-first search on one layer; returns up to ef (similarity, node), unordered
void search_layer(Hnsw& h, const float* q, int64_t ep, float ep_sim, int ef, int lvl, Visited& vis, std::vector<SimId>& out,
                  bool lock_nodes) {
    // candidates: max-heap by similarity; results: min-heap by similarity (worst on top)
    std::priority_queue<SimId> cand;
    std::priority_queue<SimId, std::vector<SimId>, std::greater<SimId>> res;
    cand.emplace(ep_sim, (int32_t)ep);
    res.emplace(ep_sim, (int32_t)ep);
    vis.test_set(ep);
    std::vector<int32_t> nb;
    while (!cand.empty()) {
        const SimId c = cand.top();
        if (c.first < res.top().first && (int)res.size() >= ef) break;
        cand.pop();
        {
            int32_t* l = links(h, c.second, lvl);
            if (lock_nodes) {
                std::lock_guard<std::mutex> g(h.locks[c.second]);
                nb.assign(l + 1, l + 1 + l[0]);
            } else {
                nb.assign(l + 1, l + 1 + l[0]);
            }
        }
        for (int32_t e : nb) {
            if (vis.test_set(e)) continue;
            const float s = dot(q, h.x + (size_t)e * h.dim, h.dim);
            if ((int)res.size() < ef || s > res.top().first) {
                cand.emplace(s, e);
                res.emplace(s, e);
                if ((int)res.size() > ef) res.pop();
            }
        }
    }
    out.clear();
    while (!res.empty()) { out.push_back(res.top()); res.pop(); }
}

// Algorithm 4 (heuristic): from candidates sorted by similarity to the base point (best first), keep a
// candidate only if it is closer to the base than to every neighbour already kept
void select_neighbours(Hnsw& h, std::vector<SimId>& cands, int m) {
    std::sort(cands.begin(), cands.end(), [](const SimId& a, const SimId& b) {
        return a.first > b.first || (a.first == b.first && a.second < b.second);
    });
    std::vector<SimId> keep;
    for (const SimId& c : cands) {
        if ((int)keep.size() >= m) break;
        bool good = true;
        for (const SimId& k : keep) {
            const float s = dot(h.x + (size_t)c.second * h.dim, h.x + (size_t)k.second * h.dim, h.dim);
            if (s > c.first) { good = false; break; }      // closer to a kept neighbour than to the base
        }
        if (good) keep.push_back(c);
    }
    cands.swap(keep);
}

void connect(Hnsw& h, int64_t node, std::vector<SimId>& sel, int lvl) {
    const int mmax = lvl == 0 ? h.M0 : h.M;
    {
        // the caller (insert) holds h.locks[node] for the whole insertion
        int32_t* l = links(h, node, lvl);
        l[0] = (int32_t)sel.size();
        for (size_t i = 0; i < sel.size(); ++i) l[1 + i] = sel[i].second;
    }
    for (const SimId& s : sel) {
        const int64_t o = s.second;
        std::lock_guard<std::mutex> g(h.locks[o]);
        int32_t* l = links(h, o, lvl);
        if (l[0] < mmax) {
            l[1 + l[0]++] = (int32_t)node;
        } else {
            // shrink: re-select among the old neighbours plus the new node
            std::vector<SimId> c;
            c.reserve(l[0] + 1);
            const float* xo = h.x + (size_t)o * h.dim;
            c.emplace_back(s.first, (int32_t)node);
            for (int i = 0; i < l[0]; ++i) c.emplace_back(dot(xo, h.x + (size_t)l[1 + i] * h.dim, h.dim), l[1 + i]);
            select_neighbours(h, c, mmax);
            l[0] = (int32_t)c.size();
            for (size_t i = 0; i < c.size(); ++i) l[1 + i] = c[i].second;
        }
    }
}

void insert(Hnsw& h, int64_t node, Visited& vis) {
    const int lvl = h.level[node];
    const float* q = h.x + (size_t)node * h.dim;
    int64_t ep;
    int top;
    // readers of this node wait until it is fully linked; taken AFTER the entry lock (lock order: entry, own node,
    // other nodes), released last
    std::unique_lock<std::mutex> whole_insertion(h.locks[node], std::defer_lock);
    {
        std::unique_lock<std::mutex> g(h.entry_lock);
        whole_insertion.lock();
        if (h.entry < 0) {
            h.entry = node;
            h.max_level = lvl;
            return;
        }
        ep = h.entry;
        top = h.max_level;
        if (lvl <= top) g.unlock();
        else {
            // this node becomes the new entry point: keep the lock until it is linked
            float ep_sim = dot(q, h.x + (size_t)ep * h.dim, h.dim);
            std::vector<SimId> w;
            for (int l = std::min(lvl, top); l >= 0; --l) {
                vis.reset(h.n);
                search_layer(h, q, ep, ep_sim, h.efc, l, vis, w, true);
                std::vector<SimId> sel = w;
                select_neighbours(h, sel, h.M);
                connect(h, node, sel, l);
                for (const SimId& s : w) if (s.first > ep_sim) { ep_sim = s.first; ep = s.second; }
            }
            h.entry = node;
            h.max_level = lvl;
            return;
        }
    }
    float ep_sim = dot(q, h.x + (size_t)ep * h.dim, h.dim);
    // Algorithm 1: greedy descent through the layers above the node's level
    for (int l = top; l > lvl; --l) {
        bool moved = true;
        while (moved) {
            moved = false;
            std::vector<int32_t> nb;
            {
                std::lock_guard<std::mutex> g(h.locks[ep]);
                int32_t* ll = links(h, ep, l);
                nb.assign(ll + 1, ll + 1 + ll[0]);
            }
            for (int32_t e : nb) {
                const float s = dot(q, h.x + (size_t)e * h.dim, h.dim);
                if (s > ep_sim) { ep_sim = s; ep = e; moved = true; }
            }
        }
    }
    std::vector<SimId> w;
    for (int l = std::min(lvl, top); l >= 0; --l) {
        vis.reset(h.n);
        search_layer(h, q, ep, ep_sim, h.efc, l, vis, w, true);
        std::vector<SimId> sel = w;
        select_neighbours(h, sel, h.M);
        connect(h, node, sel, l);
        for (const SimId& s : w) if (s.first > ep_sim) { ep_sim = s.first; ep = s.second; }
    }
}

// run body(i) for i in [begin, end) on `threads` workers, ids handed out in blocks (dynamic schedule)
template <class F>
void parallel_for(int64_t begin, int64_t end, int threads, int64_t block, F&& body) {
    if (threads <= 0) threads = (int)std::max(1u, std::thread::hardware_concurrency());
    threads = (int)std::min<int64_t>(threads, std::max<int64_t>(1, (end - begin + block - 1) / block));
    std::atomic<int64_t> next(begin);
    auto worker = [&]() {
        for (;;) {
            const int64_t b = next.fetch_add(block);
            if (b >= end) return;
            const int64_t e = std::min(end, b + block);
            body(b, e);
        }
    };
    if (threads == 1) { worker(); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back(worker);
    for (auto& t : th) t.join();
}

}  // namespace

extern "C" {

void* hnsw_build(const float* x_normalised, int64_t n, int dim, int M, int ef_construction, uint64_t seed, int threads) {
    Hnsw* h = new Hnsw;
    h->x = x_normalised; h->n = n; h->dim = dim; h->M = M; h->M0 = 2 * M; h->efc = ef_construction;
    h->level.resize(n);
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    const double mult = 1.0 / log((double)M);
    for (int64_t i = 0; i < n; ++i) h->level[i] = (int)(-log(std::max(u(rng), 1e-300)) * mult);
    h->link0.assign((size_t)n * (h->M0 + 1), 0);
    h->link_up.resize(n);
    for (int64_t i = 0; i < n; ++i)
        if (h->level[i] > 0) h->link_up[i].assign((size_t)h->level[i] * (M + 1), 0);
    h->locks = std::vector<std::mutex>(n);
    if (n > 0) {
        Visited v0;
        insert(*h, 0, v0);
    }
    parallel_for(1, n, threads, 64, [&](int64_t b, int64_t e) {
        thread_local Visited vis;
        for (int64_t i = b; i < e; ++i) insert(*h, i, vis);
    });
    return h;
}

void hnsw_search(void* handle, const float* q_normalised, int nq, int k, int ef, int threads, int64_t* ids_out, float* cos_out) {
    Hnsw& h = *static_cast<Hnsw*>(handle);
    if (ef < k) ef = k;
    parallel_for(0, nq, threads, 4, [&](int64_t qb, int64_t qe) {
        thread_local Visited vis;
        std::vector<SimId> w;
        for (int64_t qi = qb; qi < qe; ++qi) {
            const float* q = q_normalised + (size_t)qi * h.dim;
            for (int j = 0; j < k; ++j) { ids_out[(size_t)qi * k + j] = -1; cos_out[(size_t)qi * k + j] = -INFINITY; }
            if (h.entry < 0) continue;
            int64_t ep = h.entry;
            float ep_sim = dot(q, h.x + (size_t)ep * h.dim, h.dim);
            for (int l = h.max_level; l > 0; --l) {
                bool moved = true;
                while (moved) {
                    moved = false;
                    int32_t* ll = links(h, ep, l);
                    for (int i = 0; i < ll[0]; ++i) {
                        const int32_t e = ll[1 + i];
                        const float s = dot(q, h.x + (size_t)e * h.dim, h.dim);
                        if (s > ep_sim) { ep_sim = s; ep = e; moved = true; }
                    }
                }
            }
            vis.reset(h.n);
            search_layer(h, q, ep, ep_sim, ef, 0, vis, w, false);
            std::sort(w.begin(), w.end(), [](const SimId& a, const SimId& b) {
                return a.first > b.first || (a.first == b.first && a.second < b.second);
            });
            for (int j = 0; j < k && j < (int)w.size(); ++j) {
                ids_out[(size_t)qi * k + j] = w[j].second;
                cos_out[(size_t)qi * k + j] = w[j].first;
            }
        }
    });
}

int hnsw_max_level(void* handle) { return static_cast<Hnsw*>(handle)->max_level; }

void hnsw_free(void* handle) { delete static_cast<Hnsw*>(handle); }

}  // extern "C"
