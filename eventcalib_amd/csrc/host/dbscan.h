// Drop-in for the reference's dbscan/include/dbscan.h on top of libecal.so (include/ecal.h).
//
// Same class name, template parameters, Run() signature, return codes and public result members as
// DBSCAN<T,Float> in the reference (dbscan.h:42-113): a caller such as CirclesEventFrame.cpp:66-72
// compiles unchanged against this header.  What differs, and is documented in INTEGRATION.md:
//   * only dim == 2 runs on the GPU path (the one the reference uses); other dims return FAILED;
//   * Clusters[c] lists its members in the reference's order (expandCluster's pop order, ecal_cluster_order) for inputs of up
//     to 2^20 points (the member-order kernel's global-scratch tier, whose range-query arena holds 2^24 entries); only
//     beyond that — ecal_cluster_order reports status 1 for the input — in ascending pid (membership and numbering are
//     identical, bit for bit, either way);
//   * the distance-function argument is accepted and ignored (as in the reference's kd-tree build,
//     where it is only used under BRUTEFORCE, dbscan.h:64,203-206).
#ifndef ECAL_HOST_DBSCAN_H_
#define ECAL_HOST_DBSCAN_H_

#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/ecal.h"

typedef unsigned int uint;

namespace ecal_host {
// one context per host thread, created on first use (the reference uses one DBSCAN per worker thread)
inline ecal_ctx *thread_ctx(int device = 0) {
    struct Holder {
        ecal_ctx *ctx = nullptr;
        ~Holder() {
            if (ctx) ecal_destroy(ctx);
        }
    };
    static thread_local Holder h;
    if (!h.ctx) {
        const int rc = ecal_init(device, &h.ctx);
        if (rc != ECAL_OK) throw std::runtime_error(std::string("ecal_init: ") + ecal_strerror(rc));
    }
    return h.ctx;
}
}  // namespace ecal_host

template <typename T, typename Float>
class DBSCAN final {
    enum ERROR_TYPE { SUCCESS = 0, FAILED, COUNT };
    using DistanceFunc = std::function<Float(const T &, const T &)>;

public:
    DBSCAN() {}
    ~DBSCAN() {}

    // V: the points (borrowed, read once); dim must be 2; eps: radius; min: neighbours (self excluded)
    // a point needs to be a core point.  Returns SUCCESS (0) or FAILED (1) like the reference.
    template <typename Alloc>
    int Run(std::vector<T, Alloc> *V, const uint dim, const Float eps, const uint min,
            const DistanceFunc & = [](const T &, const T &) -> Float { return 0; }) {
        if (V->size() < 1) return FAILED;
        if (dim < 1) return FAILED;
        if (min < 1) return FAILED;
        Clusters.clear();
        Noise.clear();
        if (dim != 2) return FAILED;
        const size_t n = V->size();
        std::vector<double> xy(2 * n);
        for (size_t r = 0; r < n; ++r) {
            xy[2 * r] = (double) (*V)[r][0];
            xy[2 * r + 1] = (double) (*V)[r][1];
        }
        const uint32_t off[2] = {0u, (uint32_t) n};
        std::vector<int32_t> labels(n);
        uint32_t n_clusters = 0;
        const int rc = ecal_dbscan_batch(ecal_host::thread_ctx(), xy.data(), off, 1, (double) eps, min, labels.data(),
                                         &n_clusters);
        if (rc != ECAL_OK) return FAILED;
        Clusters.resize(n_clusters);
        for (size_t pid = 0; pid < n; ++pid) {
            if (labels[pid] >= 0) Clusters[labels[pid]].push_back((uint) pid);
            else Noise.push_back((uint) pid);
        }
        // member order = the reference's (dbscan.h:229-265): position of every core point in its cluster's pop order
        std::vector<int32_t> order(n);
        uint32_t status = 1;
        if (ecal_cluster_order(ecal_host::thread_ctx(), xy.data(), off, 1, (double) eps, labels.data(), &n_clusters, order.data(),
                               &status) == ECAL_OK && status == 0) {
            for (size_t pid = 0; pid < n; ++pid)
                if (labels[pid] >= 0) Clusters[labels[pid]][(size_t) order[pid]] = (uint) pid;
        }
        return SUCCESS;
    }

public:
    std::vector<std::vector<uint>> Clusters;
    std::vector<uint> Noise;
};

#endif  // ECAL_HOST_DBSCAN_H_
