// TEST INFRASTRUCTURE ONLY — CPU oracle for the DBSCAN leg of the EventCalib hot path.
//
// This file is a sequential CPU restatement of the reference algorithm.  It is the checker
// for the HIP path (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg); nothing
// in the product (eventcalib_amd/, include/) may link, import or call it.
//
// What it restates (paths relative to the reference tree):
//   * modules/camera_calibration/dbscan/src/kdtree.cpp:106-146  insertion-order, unbalanced
//     k-d tree (left iff pos[dir] < node.pos[dir], ties go right, direction alternates).
//   * modules/camera_calibration/dbscan/src/kdtree.cpp:148-179  range query: ball test is
//     inclusive (dist_sq <= range*range, summed dim 0 then dim 1 with plain mul/add), the near
//     child is always visited, the far child only when fabs(dx) < range (strict).
//   * modules/camera_calibration/dbscan/src/kdtree.cpp:469-486  every hit is prepended to a
//     singly linked list (one heap allocation per hit) => iteration order is reverse visit order.
//   * modules/camera_calibration/dbscan/include/dbscan.h:115-177  Run(): visited/assigned
//     bitmaps, clusters seeded in ascending pid order, Noise = every unassigned pid.
//   * modules/camera_calibration/dbscan/include/dbscan.h:198-227  regionQuery(): the query
//     point itself is dropped by pid, coincident other points are kept.
//   * modules/camera_calibration/dbscan/include/dbscan.h:229-265  expandCluster(): FIFO queue,
//     std::set "border set" reset once per outer pid, only core points are ever assigned.
//
// Pinning status: the k-d tree part (insert + range query incl. result order) is checked
// against the real reference kdtree.cpp compiled into oracle/_ref/libkdtree_ref.so
// (tests/test_oracle_dbscan.py).  dbscan.h itself cannot be built in this image (it includes
// <Eigen/Eigen>, which is absent), so the Run()/expandCluster() driver is pinned only through
// `oracle_dbscan_kdapi`, which runs this file's driver on top of the *reference's* kd_* C ABI
// and must agree with the fully restated path, plus a scikit-learn cross-check on inputs where
// the strict-pruning quirk cannot fire.  Build with -ffp-contract=off (see oracle/Makefile).

#include <cstdint>
#include <cstddef>
#include <cmath>
#include <vector>
#include <queue>
#include <set>
#include <dlfcn.h>
#include <string>

namespace {

struct Node {
    double pos[2];
    uint32_t pid;
    int dir;
    Node *lo_child, *hi_child;
};

struct Hit {
    Node *node;
    Hit *next;
};

struct Tree {
    Node *root = nullptr;
};

void tree_insert(Tree &t, const double *p, uint32_t pid) {
    Node **slot = &t.root;
    int dir = 0;
    while (*slot) {
        Node *cur = *slot;
        dir = (cur->dir + 1) % 2;
        slot = (p[cur->dir] < cur->pos[cur->dir]) ? &cur->lo_child : &cur->hi_child;
    }
    Node *n = new Node;
    n->pos[0] = p[0];
    n->pos[1] = p[1];
    n->pid = pid;
    n->dir = dir;
    n->lo_child = n->hi_child = nullptr;
    *slot = n;
}

void tree_free(Node *n) {
    if (!n) return;
    tree_free(n->lo_child);
    tree_free(n->hi_child);
    delete n;
}

// visit order: node, near subtree, (far subtree if fabs(dx) < range); hits are prepended.
int range_walk(Node *n, const double *q, double range, Hit *head) {
    if (!n) return 0;
    int found = 0;
    double d2 = 0;
    for (int i = 0; i < 2; i++) {
        double d = n->pos[i] - q[i];
        d2 += d * d;
    }
    if (d2 <= range * range) {
        Hit *h = new Hit;
        h->node = n;
        h->next = head->next;
        head->next = h;
        found = 1;
    }
    double dx = q[n->dir] - n->pos[n->dir];
    found += range_walk(dx <= 0.0 ? n->lo_child : n->hi_child, q, range, head);
    if (std::fabs(dx) < range) {
        found += range_walk(dx <= 0.0 ? n->hi_child : n->lo_child, q, range, head);
    }
    return found;
}

// abstract "give me the neighbours of pid in reference iteration order, self removed"
struct NeighbourSource {
    virtual ~NeighbourSource() {}
    virtual std::vector<uint32_t> query(uint32_t pid) = 0;
};

struct OwnTreeSource : NeighbourSource {
    const double *xy;
    double eps;
    Tree tree;
    OwnTreeSource(const double *xy_, uint32_t n, double eps_) : xy(xy_), eps(eps_) {
        for (uint32_t i = 0; i < n; i++) tree_insert(tree, xy + 2 * (size_t) i, i);
    }
    ~OwnTreeSource() override { tree_free(tree.root); }
    std::vector<uint32_t> query_with_self(uint32_t pid) {
        Hit head;
        head.node = nullptr;
        head.next = nullptr;
        range_walk(tree.root, xy + 2 * (size_t) pid, eps, &head);
        std::vector<uint32_t> out;
        for (Hit *h = head.next; h;) {
            out.push_back(h->node->pid);
            Hit *dead = h;
            h = h->next;
            delete dead;
        }
        return out;
    }
    std::vector<uint32_t> query(uint32_t pid) override {
        std::vector<uint32_t> all = query_with_self(pid), out;
        for (uint32_t v : all)
            if (v != pid) out.push_back(v);
        return out;
    }
};

// the reference's kd_* C ABI (dbscan/include/kdtree.h:69-140), resolved from a shared object
struct KdApi {
    void *(*create)(int);
    void (*free_tree)(void *);
    int (*insert)(void *, const double *, void *);
    void *(*nearest_range)(void *, const double *, double);
    void (*res_free)(void *);
    int (*res_end)(void *);
    int (*res_next)(void *);
    void *(*res_item)(void *, double *);
};

bool load_kdapi(const char *so_path, KdApi &api, void **handle) {
    void *h = dlopen(so_path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return false;
    *handle = h;
    api.create = (void *(*)(int)) dlsym(h, "kd_create");
    api.free_tree = (void (*)(void *)) dlsym(h, "kd_free");
    api.insert = (int (*)(void *, const double *, void *)) dlsym(h, "kd_insert");
    api.nearest_range = (void *(*)(void *, const double *, double)) dlsym(h, "kd_nearest_range");
    api.res_free = (void (*)(void *)) dlsym(h, "kd_res_free");
    api.res_end = (int (*)(void *)) dlsym(h, "kd_res_end");
    api.res_next = (int (*)(void *)) dlsym(h, "kd_res_next");
    api.res_item = (void *(*)(void *, double *)) dlsym(h, "kd_res_item");
    return api.create && api.free_tree && api.insert && api.nearest_range && api.res_free &&
           api.res_end && api.res_next && api.res_item;
}

struct KdApiSource : NeighbourSource {
    KdApi api;
    const double *xy;
    double eps;
    void *tree;
    KdApiSource(const KdApi &a, const double *xy_, uint32_t n, double eps_) : api(a), xy(xy_), eps(eps_) {
        tree = api.create(2);
        // dbscan.h:186-196 — data pointer = address of the element; pid by pointer arithmetic
        for (uint32_t i = 0; i < n; i++) api.insert(tree, xy + 2 * (size_t) i, (void *) (xy + 2 * (size_t) i));
    }
    ~KdApiSource() override { api.free_tree(tree); }
    std::vector<uint32_t> query_with_self(uint32_t pid) {
        double q[2] = {xy[2 * (size_t) pid], xy[2 * (size_t) pid + 1]};
        std::vector<uint32_t> out;
        void *res = api.nearest_range(tree, q, eps);
        while (!api.res_end(res)) {
            const double *item = (const double *) api.res_item(res, q);
            out.push_back((uint32_t) ((item - xy) / 2));
            api.res_next(res);
        }
        api.res_free(res);
        return out;
    }
    std::vector<uint32_t> query(uint32_t pid) override {
        std::vector<uint32_t> all = query_with_self(pid), out;
        for (uint32_t v : all)
            if (v != pid) out.push_back(v);
        return out;
    }
};

// dbscan.h:115-177, 229-265 restated on top of a NeighbourSource
int run_driver(NeighbourSource &src, uint32_t n, uint32_t minpts, int32_t *labels, uint32_t *n_clusters,
               uint32_t *members, uint32_t *member_off) {
    std::vector<bool> visited(n, false), assigned(n, false);
    std::vector<std::vector<uint32_t>> clusters;
    std::set<uint32_t> border;

    for (uint32_t pid = 0; pid < n; ++pid) {
        border.clear();
        if (visited[pid]) continue;
        visited[pid] = true;
        std::vector<uint32_t> nb = src.query(pid);
        if (nb.size() < minpts) continue;
        uint32_t cid = (uint32_t) clusters.size();
        clusters.emplace_back();
        border.insert(pid);
        clusters[cid].push_back(pid);
        assigned[pid] = true;

        std::queue<uint32_t> fifo;
        for (uint32_t v : nb) fifo.push(v);
        for (uint32_t v : nb) border.insert(v);
        while (!fifo.empty()) {
            uint32_t cur = fifo.front();
            fifo.pop();
            if (visited[cur]) continue;
            visited[cur] = true;
            std::vector<uint32_t> cnb = src.query(cur);
            if (cnb.size() >= minpts) {
                clusters[cid].push_back(cur);
                assigned[cur] = true;
                for (uint32_t w : cnb) {
                    if (border.find(w) == border.end()) {
                        fifo.push(w);
                        border.insert(w);
                    }
                }
            }
        }
    }

    for (uint32_t i = 0; i < n; i++) labels[i] = -1;  // Noise (dbscan.h:164-168)
    uint32_t cursor = 0;
    for (size_t c = 0; c < clusters.size(); c++) {
        if (member_off) member_off[c] = cursor;
        for (uint32_t v : clusters[c]) {
            labels[v] = (int32_t) c;
            if (members) members[cursor] = v;
            cursor++;
        }
    }
    if (member_off) member_off[clusters.size()] = cursor;
    *n_clusters = (uint32_t) clusters.size();
    return 0;
}

KdApi g_backend_api;
void *g_backend_handle = nullptr;
bool g_backend_ref = false;
std::string g_backend_path;

}  // namespace

extern "C" {

// Run() semantics on one point set.  Returns 0 (SUCCESS) or 1 (FAILED: n<1 or minpts<1),
// mirroring dbscan.h:120-123.  labels[i] = index into Clusters or -1 (Noise).
// members/member_off (optional, may be NULL) receive the concatenated Clusters[c] contents in
// the reference's in-cluster order; member_off has n_clusters+1 entries (size it n+1).
int oracle_dbscan(const double *xy, uint32_t n, double eps, uint32_t minpts, int32_t *labels,
                  uint32_t *n_clusters, uint32_t *members, uint32_t *member_off) {
    if (n < 1 || minpts < 1) {
        if (n_clusters) *n_clusters = 0;
        return 1;
    }
    if (g_backend_ref) {   // oracle_set_kd_backend: tree + range query are the reference's compiled kd_* functions
        KdApiSource src(g_backend_api, xy, n, eps);
        return run_driver(src, n, minpts, labels, n_clusters, members, member_off);
    }
    OwnTreeSource src(xy, n, eps);
    return run_driver(src, n, minpts, labels, n_clusters, members, member_off);
}

// Process-wide choice of the k-d tree under oracle_dbscan() and everything built on it (extraction, the window loops):
// so_path = oracle/_ref/libkdtree_ref.so -> the reference's own kdtree.cpp as compiled by oracle/Makefile (the library
// stays mapped until the backend is changed, so a loaded-library listing shows it); NULL or "" -> this file's restated
// tree.  Call it before any worker thread runs (kd_* on separate trees is thread-safe: plain malloc, no shared state).
// Returns 0, or -1 when the library cannot be loaded (the backend is then the restated tree).
int oracle_set_kd_backend(const char *so_path) {
    if (so_path && *so_path && g_backend_ref && g_backend_path == so_path) return 0;   // already on it: keep it mapped
    if (g_backend_handle) {
        dlclose(g_backend_handle);
        g_backend_handle = nullptr;
    }
    g_backend_ref = false;
    if (!so_path || !*so_path) return 0;
    if (!load_kdapi(so_path, g_backend_api, &g_backend_handle)) {
        g_backend_handle = nullptr;
        return -1;
    }
    g_backend_ref = true;
    g_backend_path = so_path;
    return 0;
}

// 1: oracle_dbscan runs on the reference's compiled kd-tree, 0: on the restated one
int oracle_kd_backend(void) { return g_backend_ref ? 1 : 0; }

// Same driver, but the tree and the range query are the reference's own kd_* functions taken
// from `so_path` (oracle/_ref/libkdtree_ref.so).  Returns -1 if the library cannot be loaded.
int oracle_dbscan_kdapi(const char *so_path, const double *xy, uint32_t n, double eps, uint32_t minpts,
                        int32_t *labels, uint32_t *n_clusters, uint32_t *members, uint32_t *member_off) {
    if (n < 1 || minpts < 1) {
        if (n_clusters) *n_clusters = 0;
        return 1;
    }
    KdApi api;
    void *handle = nullptr;
    if (!load_kdapi(so_path, api, &handle)) return -1;
    int rc;
    {
        KdApiSource src(api, xy, n, eps);
        rc = run_driver(src, n, minpts, labels, n_clusters, members, member_off);
    }
    dlclose(handle);
    return rc;
}

// Raw range query of point `pid` (self included), in result-list iteration order.
// out must hold n entries; returns the hit count.
int oracle_range_query(const double *xy, uint32_t n, double eps, uint32_t pid, uint32_t *out) {
    OwnTreeSource src(xy, n, eps);
    std::vector<uint32_t> r = src.query_with_self(pid);
    for (size_t i = 0; i < r.size(); i++) out[i] = r[i];
    return (int) r.size();
}

// All range queries at once (CSR): off[n+1], idx[cap]; returns total hits or -2 if cap too small.
long oracle_range_query_all(const double *xy, uint32_t n, double eps, uint64_t *off, uint32_t *idx, uint64_t cap) {
    OwnTreeSource src(xy, n, eps);
    uint64_t cur = 0;
    for (uint32_t p = 0; p < n; p++) {
        off[p] = cur;
        std::vector<uint32_t> r = src.query_with_self(p);
        if (cur + r.size() > cap) return -2;
        for (uint32_t v : r) idx[cur++] = v;
    }
    off[n] = cur;
    return (long) cur;
}

long oracle_range_query_all_kdapi(const char *so_path, const double *xy, uint32_t n, double eps, uint64_t *off,
                                  uint32_t *idx, uint64_t cap) {
    KdApi api;
    void *handle = nullptr;
    if (!load_kdapi(so_path, api, &handle)) return -1;
    long ret;
    {
        KdApiSource src(api, xy, n, eps);
        uint64_t cur = 0;
        ret = 0;
        for (uint32_t p = 0; p < n && ret >= 0; p++) {
            off[p] = cur;
            std::vector<uint32_t> r = src.query_with_self(p);
            if (cur + r.size() > cap) {
                ret = -2;
                break;
            }
            for (uint32_t v : r) idx[cur++] = v;
        }
        if (ret >= 0) {
            off[n] = cur;
            ret = (long) cur;
        }
    }
    dlclose(handle);
    return ret;
}

// Batched form used by the parity tests and by bench.py's cpu_baseline leg: one Run() per
// segment [seg_off[s], seg_off[s]+seg_cnt[s]).  Empty segments give n_clusters 0.
int oracle_dbscan_batch(const double *xy, const uint32_t *seg_off, const uint32_t *seg_cnt, uint32_t S, double eps,
                        uint32_t minpts, int32_t *labels, uint32_t *n_clusters) {
    for (uint32_t s = 0; s < S; s++) {
        uint32_t nc = 0;
        if (seg_cnt[s] > 0 && minpts >= 1)
            oracle_dbscan(xy + 2 * (size_t) seg_off[s], seg_cnt[s], eps, minpts, labels + seg_off[s], &nc, nullptr,
                          nullptr);
        n_clusters[s] = nc;
    }
    return 0;
}

}  // extern "C"
