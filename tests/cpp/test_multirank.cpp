// Multi-rank plumbing of libecal.so from plain C++ (no Python, no torch): one process per rank.
//   usage: test_multirank <mode> <world> views.bin [scratch_dir]
//     mode rccl : every rank joins the library's RCCL communicator (ecal_comm_unique_id / ecal_comm_init) on device
//                 (rank % device count) and ecal_calibrate_views all-reduces through it (options.allreduce = ecal_comm_allreduce, user = ctx).
//                 RCCL refuses two ranks on one GPU, so on a one-GPU box this runs with world = 1 (the call path is the
//                 same: communicator, ncclAllReduce on the calibration stream); with >= 2 GPUs it is a real 2-rank run.
//     mode shm  : world ranks on device 0, the all-reduce supplied as a CALLBACK that sums through POSIX shared memory
//                 (the ecal_allreduce_fn seam; any transport) — the sharded calibration protocol with 2 ranks on one GPU.
//   The launching process never touches the GPU: it forks + execs itself once per rank (argv[0] <mode> <world> views.bin dir
//   <rank>) and waits.  Every rank calibrates its shard of the views; rank 0 also calibrates ALL views on a second,
//   communicator-less context and requires the same intrinsics (views.bin: f64 V, n, fisheye, V x n x 2 pixels).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/ecal.h"

namespace {

struct Shm {                      // mode shm: a sense-reversing barrier and a sum buffer
    std::atomic<int> arrived, phase;
    double sum[1024];
};

struct ShmUser {
    Shm *shm;
    int rank, world;
    std::vector<double> host;
};

void shm_barrier(Shm *s, int world) {
    const int ph = s->phase.load();
    if (s->arrived.fetch_add(1) + 1 == world) {
        s->arrived.store(0);
        s->phase.store(ph + 1);
    } else {
        while (s->phase.load() == ph) usleep(50);
    }
}

// ecal_allreduce_fn: d_buf[0 .. n) summed over the ranks in place
int shm_allreduce(void *user, double *d_buf, size_t n, void *stream) {
    ShmUser *u = (ShmUser *) user;
    if (n > 1024) return -1;
    u->host.resize(n);
    if (hipMemcpyAsync(u->host.data(), d_buf, n * sizeof(double), hipMemcpyDeviceToHost, (hipStream_t) stream) != hipSuccess) return -1;
    if (hipStreamSynchronize((hipStream_t) stream) != hipSuccess) return -1;
    if (u->rank == 0) memset(u->shm->sum, 0, n * sizeof(double));
    shm_barrier(u->shm, u->world);
    for (int r = 0; r < u->world; r++) {   // rank order: the same sum on every rank, bit for bit
        if (r == u->rank)
            for (size_t i = 0; i < n; i++) u->shm->sum[i] += u->host[i];
        shm_barrier(u->shm, u->world);
    }
    memcpy(u->host.data(), u->shm->sum, n * sizeof(double));
    shm_barrier(u->shm, u->world);
    if (hipMemcpyAsync(d_buf, u->host.data(), n * sizeof(double), hipMemcpyHostToDevice, (hipStream_t) stream) != hipSuccess) return -1;
    return hipStreamSynchronize((hipStream_t) stream) == hipSuccess ? 0 : -1;
}

int run_rank(const std::string &mode, int world, const char *views, const std::string &dir, int rank) {
    std::ifstream f(views, std::ios::binary);
    double hdr[3];
    f.read(reinterpret_cast<char *>(hdr), sizeof(hdr));
    const uint32_t V = (uint32_t) hdr[0], n = (uint32_t) hdr[1];
    std::vector<double> img((size_t) V * n * 2);
    f.read(reinterpret_cast<char *>(img.data()), img.size() * sizeof(double));
    if (!f) return 3;
    std::vector<double> obj(3 * (size_t) n);   // the 9 x 4 asymmetric grid, square 5.5 (EventCalibIni.cpp:102-106)
    for (uint32_t i = 0; i < 9; i++)
        for (uint32_t j = 0; j < 4; j++) {
            obj[3 * (i * 4 + j)] = (2 * j + i % 2) * 5.5;
            obj[3 * (i * 4 + j) + 1] = i * 5.5;
            obj[3 * (i * 4 + j) + 2] = 0.0;
        }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return 4;
    const int device = mode == "rccl" ? rank % ndev : 0;
    ecal_ctx *ctx = nullptr;
    if (ecal_init(device, &ctx) != ECAL_OK) return 5;
    ecal_calib_options opt;
    ecal_calib_default_options(&opt);
    opt.flags = 455u;   // the shipped example.yaml: fixed aspect ratio 1, centred principal point, no tangential, K4..K6 fixed
    opt.aspect_ratio = 1.0;
    ShmUser su{nullptr, rank, world, {}};
    if (mode == "rccl") {
        unsigned char id[ECAL_COMM_ID_BYTES];
        const std::string idf = dir + "/comm_id.bin", tmp = idf + ".tmp";
        if (rank == 0) {
            if (ecal_comm_unique_id(id) != ECAL_OK) return 6;
            FILE *o = fopen(tmp.c_str(), "wb");
            fwrite(id, 1, sizeof(id), o);
            fclose(o);
            rename(tmp.c_str(), idf.c_str());
        } else {
            FILE *i = nullptr;
            for (int tries = 0; tries < 20000 && !(i = fopen(idf.c_str(), "rb")); tries++) usleep(1000);
            if (!i || fread(id, 1, sizeof(id), i) != sizeof(id)) return 6;
            fclose(i);
        }
        if (ecal_comm_init(ctx, id, rank, world) != ECAL_OK) {
            std::fprintf(stderr, "rank %d: ecal_comm_init: %s\n", rank, ecal_last_error(ctx));
            return 7;
        }
        if (ecal_comm_size(ctx) != world || ecal_comm_rank(ctx) != rank) return 8;
        // the bare collective first: rank r contributes r + 1, 2 (r + 1), ...
        double h[4], *d = nullptr;
        for (int i = 0; i < 4; i++) h[i] = (i + 1) * (rank + 1);
        if (hipMalloc((void **) &d, sizeof(h)) != hipSuccess) return 9;
        (void) hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
        if (ecal_comm_allreduce_sum_dev(ctx, d, 4, nullptr) != ECAL_OK) return 10;
        (void) hipDeviceSynchronize();
        (void) hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        (void) hipFree(d);
        for (int i = 0; i < 4; i++)
            if (h[i] != (i + 1) * world * (world + 1) / 2.0) return 11;
        opt.allreduce = ecal_comm_allreduce;   // collectives are explicit: NULL would be a rank-local calibration
        opt.allreduce_user = ctx;
    } else {
        const std::string name = dir + "/shm";
        const int fd = open(name.c_str(), O_RDWR);
        if (fd < 0) return 6;
        su.shm = (Shm *) mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (su.shm == MAP_FAILED) return 7;
        opt.allreduce = shm_allreduce;
        opt.allreduce_user = &su;
    }
    const uint32_t lo = (uint32_t) ((uint64_t) V * rank / world), hi = (uint32_t) ((uint64_t) V * (rank + 1) / world);
    ecal_calib_result res;
    std::vector<double> rv(3 * (size_t) (hi - lo)), tv(3 * (size_t) (hi - lo));
    int rc = ecal_calibrate_views(ctx, obj.data(), n, img.data() + (size_t) lo * n * 2, hi - lo, 346, 260, &opt, &res, rv.data(), tv.data(),
                                  nullptr);
    if (rc != ECAL_OK) {
        std::fprintf(stderr, "rank %d: ecal_calibrate_views: %s (%s)\n", rank, ecal_strerror(rc), ecal_last_error(ctx));
        return 12;
    }
    std::printf("rank %d/%d device %d views [%u, %u) rms %.12g fx %.12g fy %.12g cx %.12g cy %.12g iterations %d\n", rank, world, device,
                lo, hi, res.rms, res.intr[0], res.intr[1], res.intr[2], res.intr[3], res.iterations);
    int bad = 0;
    if (rank == 0) {   // the unsharded calibration on a context of its own
        ecal_ctx *one = nullptr;
        if (ecal_init(device, &one) != ECAL_OK) return 13;
        ecal_calib_options o1;
        ecal_calib_default_options(&o1);
        o1.flags = 455u;
        o1.aspect_ratio = 1.0;
        ecal_calib_result r1;
        std::vector<double> rv1(3 * (size_t) V), tv1(3 * (size_t) V);
        if (ecal_calibrate_views(one, obj.data(), n, img.data(), V, 346, 260, &o1, &r1, rv1.data(), tv1.data(), nullptr) != ECAL_OK) return 14;
        for (int k = 0; k < 12; k++)
            if (std::fabs(res.intr[k] - r1.intr[k]) > 1e-7 * (1 + std::fabs(r1.intr[k]))) bad++;
        if (std::fabs(res.rms - r1.rms) > 1e-9) bad++;   // (the iteration count may differ: the stop test compares changes with DBL_EPSILON and the sharded sums round differently)
        std::printf("single rms %.12g fx %.12g iterations %d -> %s\n", r1.rms, r1.intr[0], r1.iterations, bad ? "MISMATCH" : "same");
        ecal_destroy(one);
        if (mode == "rccl") {
            // a rank-local call (allreduce == NULL) on the context that HAS joined the communicator: it must not become a
            // collective (rank 0 alone is here: an implicit ncclAllReduce would hang or sum unrelated blocks)
            ecal_calib_result r2;
            if (ecal_calibrate_views(ctx, obj.data(), n, img.data(), V, 346, 260, &o1, &r2, rv1.data(), tv1.data(), nullptr) != ECAL_OK) return 15;
            if (r2.rms != r1.rms || r2.intr[0] != r1.intr[0] || r2.iterations != r1.iterations) bad++;
            std::printf("rank-local call on the communicator's context: rms %.12g -> %s\n", r2.rms, bad ? "MISMATCH" : "same");
        }
    }
    ecal_destroy(ctx);   // (destroys the communicator too)
    return bad ? 20 : 0;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    const std::string mode = argv[1];
    int world = std::atoi(argv[2]);
    const std::string dir = argc > 4 ? argv[4] : "/tmp";
    if (argc > 5) return run_rank(mode, world, argv[3], dir, std::atoi(argv[5]));
    // launcher: no HIP call in this process
    if (mode == "shm") {
        const std::string name = dir + "/shm";
        const int fd = open(name.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(Shm)) != 0) return 3;
        close(fd);
    } else {
        unlink((dir + "/comm_id.bin").c_str());
    }
    std::vector<pid_t> kids;
    for (int r = 0; r < world; r++) {
        const pid_t p = fork();
        if (p == 0) {
            const std::string rs = std::to_string(r), ws = std::to_string(world);
            execl(argv[0], argv[0], mode.c_str(), ws.c_str(), argv[3], dir.c_str(), rs.c_str(), (char *) nullptr);
            _exit(127);
        }
        kids.push_back(p);
    }
    int worst = 0;
    for (pid_t p : kids) {
        int st = 0;
        waitpid(p, &st, 0);
        const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128;
        if (code > worst) worst = code;
    }
    std::printf("%s world %d: %s\n", mode.c_str(), world, worst ? "FAILED" : "multirank ok");
    return worst;
}
