// Circle-candidate extraction of one window by one workgroup of DET_T threads, as device functions
// (CirclesEventFrame.cpp:89-312).  Shared by the extraction kernels (ecal_detect.hip) and the fused detection pass
// (ecal_fused.hip).
#pragma once
#include "ecal_ctx.hpp"
#include "block_utils.hpp"
#include "ref_nth_element.hpp"
#include <type_traits>

#pragma clang fp contract(off)

namespace ecal {

#ifdef ECAL_PHASE_PROF
static __device__ unsigned long long g_det_cycles[16];
#define DET_MARK(idx)                                                                    \
    do {                                                                                 \
        if (threadIdx.x == 0) {                                                          \
            const unsigned long long now__ = __builtin_readcyclecounter();               \
            atomicAdd(&g_det_cycles[idx], now__ - det_t__);                              \
            det_t__ = now__;                                                             \
        }                                                                                \
    } while (0)
#define DET_T0() unsigned long long det_t__ = __builtin_readcyclecounter()
#else
#define DET_MARK(idx) do { } while (0)
#define DET_T0() do { } while (0)
#endif
// debug builds (-DECAL_DET_STOP=k, tools/px_stop_probe.sh): leave extract_window after phase k so that instruction
// counters can be attributed to phases (results stay in LDS / scratch; nothing downstream may run)
#ifndef ECAL_DET_STOP
#define ECAL_DET_STOP 0
#endif

constexpr int DET_T = 256;
constexpr uint32_t DET_MAXC = 2048;       // DBSCAN clusters per polarity the kernel handles at all
constexpr uint32_t DET_LDS_PTS = 1408;    // points per window (both polarities) staged in LDS, packed to 4 bytes (6 workgroups/CU) ...
constexpr uint32_t DET_LDS_MAXC = 384;    // ... when neither polarity has more DBSCAN clusters than this and every
constexpr uint32_t DET_LDS_PTS2 = 2816, DET_LDS_MAXC2 = 512;   // second pass over the windows the first one lists (44 KB of LDS)
                                          // coordinate is an integer of |v| <= 32767 (event pixels)

struct DetectParams {
    uint32_t cluster_min;    // clusterMinSample
    uint32_t need_clusters;  // rows * cols
    double four_thr2;        // 4 * circleRadiusThreshold_^2
    double thr;              // circleRadiusThreshold_
    uint32_t fit_circle;     // Params::fitCircle
    uint32_t knn;            // Params::knn_num (<= DET_KNN_MAX)
    const uint32_t *xy16;    // packed points (ecal_packed_points) or null: windows marked seg_fmt & 1 are staged from these
    const uint32_t *seg_fmt; //   (a packed window that cannot be staged has had its doubles written before the launch)
    // the exact extraction's first pass (TDET): the kd-trees the pixel DBSCAN kernel exported (ecal_ctx::px_tree: child links of
    // point i of segment s at px_tree[seg_off[s] + i], valid where px_tree_flag[s] == px_tree_epoch) or null, and the range
    // query's two tests for integer pixels: d2 <= eps^2 <=> d2 <= tie_e2i, |dx| < eps <=> |dx| <= tie_prune
    const uint32_t *px_tree = nullptr, *px_tree_flag = nullptr;
    uint32_t px_tree_epoch = 0;
    int tie_e2i = 0, tie_prune = 0;
};
constexpr uint32_t DET_TIE_TREE_CAP = 768;   // points of a segment whose tree can be staged (the pixel DBSCAN kernel's first pass: PX_CAP)
constexpr size_t DET_TIE_INV_BYTES = DET_TIE_TREE_CAP;   // u8 per point: its position in its cluster's ascending-pid member list
constexpr uint32_t DET_KNN_MAX = 8;

__device__ __forceinline__ double norm_of(double2 p) { return __dsqrt_rn(p.x * p.x + p.y * p.y); }  // Vector2d::norm()

// Per-window working set: either staged in LDS (16-bit indices, points copied in) or in global scratch.
// All indices are window-local: point i of polarity pol lives at base[pol] + i, kept cluster k likewise.
struct DetGlobal {
    const double2 *pts;  // + window slot offset applied by the caller
    uint32_t *members, *sorted, *koff, *ksize, *rep;
    int32_t *kept;
    double *norms;
    __device__ __forceinline__ double2 pt(uint32_t li) const { return pts[li]; }
    __device__ __forceinline__ double norm(uint32_t li) const { return norms[li]; }
    __device__ __forceinline__ double key(uint32_t li) const { return norms[li]; }  // ordering key of the median
    __device__ __forceinline__ void set_norm(uint32_t li, double v) const { norms[li] = v; }
    static constexpr bool INT_PIXELS = false;
    static constexpr uint32_t REP_TIE = 0x80000000u;   // (ORD) flag on a representative whose rank has an equal-norm rival
    static constexpr uint32_t REP_BAD = 0x40000000u;   // (ORD) ... and whose cluster has no usable member order
    static constexpr uint32_t IDX_MASK = 0xFFFFFFFFu, IDXB = 0u;
    static constexpr int J = 1;
    static constexpr uint32_t MAXC = 0;
    using CIdx = uint32_t;  // renumbered cluster id / first member slot of a DBSCAN cluster
    static constexpr CIdx CNONE = 0xFFFFFFFFu;
    __device__ __forceinline__ bool composite() const { return false; }
    __device__ __forceinline__ uint32_t member_word(uint32_t, uint32_t i) const { return i; }
    __device__ __forceinline__ uint32_t ipt(uint32_t) const { return 0; }
    __device__ __forceinline__ int32_t label(uint32_t, const int32_t *lab, uint32_t i) const { return lab[i]; }  // DBSCAN label
};
template <uint32_t PTS_, uint32_t MAXC_>
struct DetLdsT {
    static constexpr uint32_t IDXB = PTS_ > 2048u ? 12u : 11u;   // bits of the window-local point index in a member word
    static constexpr uint32_t IDX_MASK = (1u << IDXB) - 1u;
    static constexpr uint32_t REP_TIE = 0x8000u;                  // (ORD) flag on a representative (u16, point indices < 2^12)
    static constexpr uint32_t REP_BAD = 0x4000u;
    static constexpr int J = (int) ((PTS_ + 255u) / 256u);        // points per thread at most
    static constexpr uint32_t MAXC = MAXC_;
    uint32_t *pts;  // x | y << 16, two's complement int16 each (exact: the staged path is taken for integer pixels only)
    uint32_t *members;  // member lists; when `small`: key << IDXB | point index (one compare orders by (norm, pid))
    uint16_t *sorted, *koff, *ksize, *rep;
    int16_t *kept;
    bool small;  // every |coordinate| <= 1023 (723 in the second pass): x^2 + y^2 < 2^(32 - IDXB) leaves IDXB bits for the index
    static constexpr bool INT_PIXELS = true;
    using CIdx = uint16_t;
    static constexpr CIdx CNONE = 0xFFFFu;
    __device__ __forceinline__ bool composite() const { return small; }
    __device__ __forceinline__ uint32_t member_word(uint32_t li, uint32_t i) const { return small ? (key(li) << IDXB) | i : i; }
    __device__ __forceinline__ uint32_t ipt(uint32_t li) const { return pts[li]; }
    // the DBSCAN labels were staged into kept[] with the points (one trip to HBM instead of four); the renumbered
    // label replaces the raw one in place
    __device__ __forceinline__ int32_t label(uint32_t li, const int32_t *, uint32_t) const { return kept[li]; }
    __device__ __forceinline__ double2 pt(uint32_t li) const {
        const uint32_t w = pts[li];
        return make_double2((double) (int) (short) (w & 0xFFFFu), (double) (((int) w) >> 16));
    }
    __device__ __forceinline__ double norm(uint32_t li) const { return norm_of(pt(li)); }
    // ordering key of the median: for integer pixels x^2 + y^2 (< 2^31, exact) orders exactly like Vector2d::norm() —
    // sqrt is monotone and two different integers below 2^53 never round to the same double root
    __device__ __forceinline__ uint32_t key(uint32_t li) const {
        const uint32_t w = pts[li];
        const int x = (int) (short) (w & 0xFFFFu), y = ((int) w) >> 16;
        return (uint32_t) (x * x + y * y);
    }
    __device__ __forceinline__ void set_norm(uint32_t, double) const {}
};

struct CircleFit {
    double err, radius, cx, cy;
};

// CirclesEventFrame::fitCircle (:361-415) over the + cluster kp and the - cluster kn (members in
// ascending pid; the reference sums in its BFS member order — last-bit differences only), then the
// error of :202-219.  3x3 system solved by Gaussian elimination with partial pivoting (Eigen's lu()).
template <typename ST>
__device__ __forceinline__ CircleFit fit_pair(const ST &st, const uint32_t (&base)[2], const uint32_t (&kb)[2], uint32_t kp,
                                              uint32_t kn, double thr) {
    double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0, sxxx = 0, syyy = 0, sxyy = 0, sxxy = 0;
    uint32_t cnt = 0;
    for (int pol = 0; pol < 2; pol++) {
        const uint32_t o = base[pol], kk = pol ? kn : kp;
        const uint32_t m = st.ksize[kb[pol] + kk], first = o + st.koff[kb[pol] + kk];
        for (uint32_t t = 0; t < m; t++) {
            const double2 e = st.pt(o + st.sorted[first + t]);
            sx += e.x;
            sy += e.y;
            const double xx = e.x * e.x, yy = e.y * e.y, xy = e.x * e.y;
            sxx += xx;
            syy += yy;
            sxy += xy;
            sxxx += xx * e.x;
            syyy += yy * e.y;
            sxyy += xy * e.y;
            sxxy += e.x * xy;
        }
        cnt += m;
    }
    double A[3][4] = {{2 * sx, 2 * sy, (double) cnt, sxx + syy},
                      {2 * sxx, 2 * sxy, sx, sxxx + sxyy},
                      {2 * sxy, 2 * syy, sy, sxxy + syyy}};
#pragma unroll
    for (int c = 0; c < 3; c++) {
        int piv = c;
#pragma unroll
        for (int r = c + 1; r < 3; r++)
            if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double a = A[c][k], b = A[piv][k];
            A[c][k] = b;
            A[piv][k] = (piv == c) ? b : a;
        }
#pragma unroll
        for (int r = c + 1; r < 3; r++) {
            const double f = A[r][c] / A[c][c];
#pragma unroll
            for (int k = c; k < 4; k++) A[r][k] -= f * A[c][k];
        }
    }
    double x[3];
    x[2] = A[2][3] / A[2][2];
    x[1] = (A[1][3] - A[1][2] * x[2]) / A[1][1];
    x[0] = (A[0][3] - A[0][1] * x[1] - A[0][2] * x[2]) / A[0][0];
    CircleFit f;
    f.cx = x[0];
    f.cy = x[1];
    f.radius = __dsqrt_rn(x[0] * x[0] + x[1] * x[1] + x[2]);
    f.err = 0.0;
    const double2 pr = st.pt(base[0] + st.rep[kb[0] + kp]), nr = st.pt(base[1] + st.rep[kb[1] + kn]);
    const double ax = pr.x - nr.x, ay = pr.y - nr.y;
    const double approx = __dsqrt_rn(ax * ax + ay * ay) / 2;
    if (f.radius > thr || f.radius > 2 * approx) {
        f.err = 1.79769313486231570e308;  // std::numeric_limits<double>::max(), :207
        return f;
    }
    for (int pol = 0; pol < 2; pol++) {
        const uint32_t o = base[pol], kk = pol ? kn : kp;
        const uint32_t m = st.ksize[kb[pol] + kk], first = o + st.koff[kb[pol] + kk];
        for (uint32_t t = 0; t < m; t++) {
            const double2 e = st.pt(o + st.sorted[first + t]);
            const double ex = e.x - f.cx, ey = e.y - f.cy;
            f.err += fabs(__dsqrt_rn(ex * ex + ey * ey) - f.radius);
        }
    }
    f.err /= (double) cnt * f.radius;
    return f;
}

// the k nearest representatives of polarity `pol` to q, ascending squared distance, ties to the smaller
// cluster index (nanoflann's order among ties is unpinned); returns how many survive the gates of
// :187-193 (d > 4 d0 or d > 4 thr^2 cuts the list)
template <typename ST>
__device__ __forceinline__ uint32_t knn_gated(const ST &st, uint32_t base_pol, uint32_t kb_pol, uint32_t nk, double2 q, uint32_t K,
                                              double gate, uint32_t (&idx)[DET_KNN_MAX]) {
    double d2[DET_KNN_MAX];
    for (uint32_t i = 0; i < DET_KNN_MAX; i++) {
        d2[i] = 1.79769313486231570e308;
        idx[i] = 0;
    }
    uint32_t have = 0;
    for (uint32_t k = 0; k < nk; k++) {
        const double2 c = st.pt(base_pol + st.rep[kb_pol + k]);
        const double dx = q.x - c.x, dy = q.y - c.y;
        const double d = dx * dx + dy * dy;
        // insertion into the sorted top-K (strict <: equal distances keep the earlier index first)
        uint32_t pos = have < K ? have : K;
        while (pos > 0 && d < d2[pos - 1]) pos--;
        if (pos < K) {
            for (uint32_t m = (have < K ? have : K - 1); m > pos; m--) {
                d2[m] = d2[m - 1];
                idx[m] = idx[m - 1];
            }
            d2[pos] = d;
            idx[pos] = k;
            if (have < K) have++;
        }
    }
    uint32_t real = K;
    for (uint32_t oi = 0; oi < K; oi++)
        if (d2[oi] > d2[0] * 4 || d2[oi] > gate) {
            real = oi;
            break;
        }
    return real;
}

// base[pol]: window-local offset of the polarity's points (and of its kept-cluster arrays).
// std::nth_element(a, a + nth, a + m, comp) of libstdc++ — ref_nth_element.hpp: __introselect with its heap-select branch, restated —
// as CirclesEventFrame.cpp:136-147 runs it over Clusters[c] with comp = "norm of the pixel is smaller": WHICH of two members of
// equal norm ends up at a + nth depends on the input order and on the library's data movements.  a[] = window-local point
// indices in the reference's member order (ecal_cluster_order_dev); key = the norm's ordering key.  Returns a[nth].
template <typename ST>
__device__ __forceinline__ uint32_t ref_nth_member(const ST &st, uint32_t o, uint32_t *a, uint32_t m, uint32_t nth) {
    ecal::ref_nth_element(a, m, nth, [&](uint32_t x, uint32_t y) { return st.key(o + x) < st.key(o + y); });
    return a[nth];
}

// The tie path INSIDE the first extraction pass (round 5).  A window whose tied clusters are small used to leave this pass on a
// list, have the member order of those clusters worked out by cluster_order_kernel (ecal_bfs.hip) and be extracted again from
// the start — two more launches that stage the same window twice more (0.36 ms of the 2.75 ms pass for ties in 17 % of the
// windows).  Here the workgroup that found the tie resolves it on the spot, a WAVE per tied cluster of <= 64 members:
//   the polarity's kd-tree (dbscan_pixel_kernel's child links, 4 bytes per point) staged where csize / newid / coff lay (dead
//   between the scatter and the pairing); a lane per member runs its range query — find_nearest's visiting order
//   (kdtree.cpp:148-179), the pending far subtrees in a 192-bit shift register, integer arithmetic (exact for pixels) — and
//   keeps the hits that are members of its own cluster, as positions in the cluster's ascending-pid list, in visiting order;
//   the wave then simulates expandCluster's queue (dbscan.h:229-265; the result list is the hits in REVERSE visiting order,
//   rlist_insert at the head, kdtree.cpp:469-486; a lane per hit of the popped member) and one lane runs libstdc++'s
//   nth_element over the members in that order (CirclesEventFrame.cpp:136-147).
// Scratch: members[] (free after the rank scan), 1408 bytes per wave.  Anything that does not fit — a cluster of more than 64
// members, more same-cluster hits than the wave's list slots hold, more than twelve pending subtrees, no exported tree — fails
// the whole window, which then goes on the list as before (nothing is committed until every tied cluster is resolved).
// Returns true (uniform) when every tied cluster's representative is now the reference's pick (REP_TIE cleared).
// an array of <= 64 words in the lanes of ONE vector register of the wave (element i in lane off + i), for ref_nth_element: the
// loops run wave-uniformly, an element access is a v_readlane / a select on the lane id instead of a trip to LDS
struct LaneArr {
    using value_type = uint32_t;
    uint32_t *v;
    uint32_t off;
    struct Ref {
        uint32_t *v;
        uint32_t lane;
        __device__ __forceinline__ operator uint32_t() const { return (uint32_t) __builtin_amdgcn_readlane((int) *v, (int) lane); }
        __device__ __forceinline__ Ref &operator=(uint32_t x) {   // (a compare + select: v_writelane_b32 has no builtin)
            *v = (threadIdx.x & 63u) == lane ? x : *v;
            return *this;
        }
        __device__ __forceinline__ Ref &operator=(const Ref &o) { return *this = (uint32_t) o; }
    };
    __device__ __forceinline__ Ref operator[](int64_t i) const { return Ref{v, (uint32_t) __builtin_amdgcn_readfirstlane((int) (off + (uint32_t) i))}; }
    __device__ __forceinline__ LaneArr operator+(uint32_t d) const { return LaneArr{v, off + d}; }
};

// tree_win: the tree words of the window's points (px_tree + the window's first slot: a staged window's two segments are
// contiguous).  Read HERE, by the windows that need them (17 % on the benchmark stream): asking for them at the head of the
// kernel with the points — so that a tied window need not go back to memory, ~15 us at this point — was measured: 0.560 ->
// 0.549 ms for the stage at 0.15 GB more traffic per pass, every window paying for the third load; not kept.
template <typename ST>
__device__ __forceinline__ bool resolve_ties_inline(const ST &st, const uint32_t (&base)[2], const uint32_t (&kb)[2], const uint32_t (&n_pol)[2],
                                                    const uint32_t (&nk)[2], uint32_t tied0, uint32_t tied1, const uint32_t *tree_win,
                                                    int e2i, int prune, uint32_t *tree_lds, unsigned char *scratch,
                                                    uint8_t *inv, uint32_t *fail_word) {
    constexpr uint32_t WS = 1408, LIST_OFF = 200, LIST_BYTES = WS - LIST_OFF, NIL = 0xFFFFu;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    unsigned char *const ws = scratch + wave * WS;
    uint8_t *const queue = ws + 128;                                 // [64] positions in the ascending-pid member list, in Clusters[c]'s order
    uint32_t *const inq = reinterpret_cast<uint32_t *>(ws + 192);    // [2] member is (or was) in the queue
    uint8_t *const lists = ws + LIST_OFF;
    if (tid == 0) *fail_word = 0;
    for (int pol = 0; pol < 2; pol++) {
        __syncthreads();   // the other polarity's tree is done with
        if (!(pol ? tied1 : tied0)) continue;
        const uint32_t o = base[pol];
#ifdef ECAL_PHASE_PROF
        const unsigned long long tie_t0__ = __builtin_readcyclecounter();
#endif
        for (uint32_t i = tid; i < n_pol[pol]; i += DET_T) tree_lds[i] = tree_win[o + i];   // (staged windows: base = {0, n_pol[0]})
        __syncthreads();
#ifdef ECAL_PHASE_PROF
        if (tid == 0) atomicAdd(&g_det_cycles[9], __builtin_readcyclecounter() - tie_t0__);
#endif
        for (uint32_t k = wave; k < nk[pol]; k += DET_T / 64) {   // (uniform in the wave)
            const uint32_t rv = st.rep[kb[pol] + k];
            if (!(rv & ST::REP_TIE)) continue;
            const uint32_t m = st.ksize[kb[pol] + k], first = o + st.koff[kb[pol] + k];
            if (m > 64u || !st.composite()) {   // (composite member words: key << IDXB | index, what the nth_element below orders)
                if (lane == 0) *fail_word = 1;
                continue;
            }
#ifdef ECAL_PHASE_PROF
            unsigned long long tie_t__ = __builtin_readcyclecounter();
#define TIE_MARK(i) do { if (lane == 0) { const unsigned long long n__ = __builtin_readcyclecounter(); atomicAdd(&g_det_cycles[i], n__ - tie_t__); tie_t__ = n__; } } while (0)
            if (lane == 0) {
                atomicAdd(&g_det_cycles[13], 1ull);
                atomicAdd(&g_det_cycles[14], (unsigned long long) m);
            }
#else
#define TIE_MARK(i) do { } while (0)
#endif
            const uint32_t slot = LIST_BYTES / m > 63u ? 63u : LIST_BYTES / m;
            const uint32_t p = lane < m ? st.sorted[first + lane] : 0u;
            if (lane < m) inv[p] = (uint8_t) lane;
            if (lane < 2u) inq[lane] = lane == 0 ? 1u : 0u;   // the seed = the smallest pid = position 0 (dbscan.h:140-158)
            if (lane == 0) queue[0] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- the members' range queries ----
            // (a cluster of <= 16 members — most are: half an arc of a circle's edge — keeps a member's hits in ONE 64-bit register,
            // four bits a hit, and runs the queue below on scalars: no list in LDS, no atomics)
            const bool tiny = m <= 16u;
            unsigned long long hl = 0;
            uint32_t cnt = 0;
            bool bad = false;
            if (lane < m) {
                const uint32_t pw = st.ipt(o + p);
                const int qx = (int) (short) (pw & 0xFFFFu), qy = ((int) pw) >> 16;
                uint8_t *const out = lists + lane * slot;
                unsigned long long s0 = 0, s1 = 0, s2 = 0;   // pending far subtrees: twelve entries of node | dir << 15
                uint32_t sp = 0, node = 0, dir = 0;
                for (;;) {
                    while (node != NIL) {
                        const uint32_t nw = st.ipt(o + node);
                        const int nx = (int) (short) (nw & 0xFFFFu), ny = ((int) nw) >> 16;
                        const int ddx = nx - qx, ddy = ny - qy;
                        if (ddx * ddx + ddy * ddy <= e2i && node != p && st.kept[o + node] == (int) k) {   // (regionQuery drops the query point, dbscan.h:218)
                            if (tiny) hl |= (unsigned long long) inv[node] << (4u * cnt);   // (at most 15 other members)
                            else if (cnt < slot) out[cnt] = inv[node];
                            cnt++;
                        }
                        const int dx = dir ? (qy - ny) : (qx - nx);
                        const uint32_t t = tree_lds[node], l = t & 0xFFFFu, r = t >> 16;
                        const uint32_t nearc = dx <= 0 ? l : r, farc = dx <= 0 ? r : l;
                        if ((dx < 0 ? -dx : dx) <= prune && farc != NIL) {
                            if (sp < 12u) {
                                s2 = (s2 << 16) | (s1 >> 48);
                                s1 = (s1 << 16) | (s0 >> 48);
                                s0 = (s0 << 16) | (farc | ((dir ^ 1u) << 15));
                            }
                            sp++;
                        }
                        node = nearc;
                        dir ^= 1u;
                    }
                    if (sp == 0 || sp > 12u) break;
                    sp--;
                    const uint32_t e = (uint32_t) (s0 & 0xFFFFu);
                    s0 = (s0 >> 16) | (s1 << 48);
                    s1 = (s1 >> 16) | (s2 << 48);
                    s2 >>= 16;
                    node = e & 0x7FFFu;
                    dir = e >> 15;
                }
                bad = (!tiny && cnt > slot) || sp > 12u;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            TIE_MARK(10);
            // ---- expandCluster's queue: a lane per hit of the popped member, the hits taken from the END of its list ----
            uint32_t head = 0, tail = 1;
            const bool any_bad = __any(bad);
            unsigned long long qv = 0;   // tiny: the queue, four bits a member (the seed, position 0, first)
            if (tiny && !any_bad) {
                uint32_t inqm = 1u;
                while (head < tail) {
                    const int q = __builtin_amdgcn_readfirstlane((int) ((qv >> (4u * head)) & 15u));
                    head++;
                    const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) hl, q);
                    const uint32_t hi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (hl >> 32), q);
                    const unsigned long long hq = ((unsigned long long) hi << 32) | lo;
                    const uint32_t mq = (uint32_t) __builtin_amdgcn_readlane((int) cnt, q);
                    for (uint32_t t = mq; t-- > 0;) {   // the result list = the hits in REVERSE visiting order
                        const uint32_t j = (uint32_t) (hq >> (4u * t)) & 15u;
                        if (!((inqm >> j) & 1u)) {
                            inqm |= 1u << j;
                            qv |= (unsigned long long) j << (4u * tail);
                            tail++;
                        }
                    }
                }
            }
            while (!tiny && !any_bad && head < tail) {
                const uint32_t q = reinterpret_cast<volatile uint8_t *>(queue)[head];
                head++;
                const uint32_t mq = (uint32_t) __shfl((int) cnt, (int) q, 64);
                bool take = false;
                uint32_t j = 0;
                if (lane < mq) {
                    j = lists[q * slot + (mq - 1u - lane)];
                    const uint32_t bit = 1u << (j & 31u);
                    take = !(atomicOr(&inq[j >> 5], bit) & bit);   // (the hits of one query are distinct points)
                }
                const unsigned long long mask = __ballot(take);
                if (take) queue[tail + __popcll(mask & ((1ull << lane) - 1ull))] = (uint8_t) j;
                tail += (uint32_t) __popcll(mask);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if (any_bad || tail != m) {   // (tail != m cannot happen with labels and tree of the same DBSCAN call: defensive)
                if (lane == 0) *fail_word = 1;
                continue;
            }
            TIE_MARK(11);
            // the members in Clusters[c]'s order, one per lane, as key << IDXB | point index; libstdc++'s nth_element by the whole
            // wave on that register (the comparison looks at the key alone: members of equal norm are equivalent, and which of
            // them ends up in the middle is the library's data movement)
            uint32_t aw = 0;
            if (lane < m) {
                const uint32_t pid = st.sorted[first + (tiny ? (uint32_t) (qv >> (4u * lane)) & 15u : (uint32_t) queue[lane])];
                aw = st.member_word(o + pid, pid);
            }
            ecal::ref_nth_element(LaneArr{&aw, 0u}, m, m / 2u, [](uint32_t x, uint32_t y) { return (x >> ST::IDXB) < (y >> ST::IDXB); });
            const uint32_t pick = (uint32_t) __builtin_amdgcn_readlane((int) aw, (int) (m / 2u)) & ST::IDX_MASK;
            if (lane == 0)
                st.rep[kb[pol] + k] = (typename std::remove_reference<decltype(st.rep[0])>::type) (pick | ST::REP_TIE | ST::REP_BAD);   // resolved, not yet committed
            TIE_MARK(12);
#undef TIE_MARK
        }
    }
    __syncthreads();
    return *fail_word == 0;
}

// ORD: ord0 / ord1 = the points' positions inside the reference's Clusters[label] (ecal_cluster_order_dev), per polarity:
// the representative of a cluster whose median rank has an equal-norm rival is then the reference's own pick.
// TDET (without ORD): the window is appended to tie_list when some kept cluster's median is tied — the representatives stay
// the smaller-pid ones; a later pass over that list (ORD) replaces them (ecal_extract_batch_exact_dev).
template <bool FIT, bool ORD, bool TDET, typename ST>
__device__ __forceinline__ void extract_window(const ST &st, const uint32_t (&base)[2], const uint32_t (&kb)[2],
                                               const uint32_t (&n_pol)[2], const int32_t *lab0, const int32_t *lab1,
                                               const int32_t *ord0, const int32_t *ord1, uint32_t *tie_list, uint32_t *tie_count,
                                               uint32_t tie_token, int32_t *mark0, int32_t *mark1,
                                               const uint32_t (&nc_pol)[2], const DetectParams &prm, uint32_t *csize,
                                               typename ST::CIdx *newid, typename ST::CIdx *coff,
                                               unsigned long long *red, uint32_t *nk_sh, uint32_t *info,
                                               uint32_t *cand_pair, double *cand_xyr, const uint32_t *tie_tree, const uint32_t *tie_flags,
                                               uint8_t *tie_inv) {
    const uint32_t tid = threadIdx.x;
    DET_T0();
    for (int pol = 0; pol < 2; pol++) {
        const uint32_t o = base[pol], ko = kb[pol], n = n_pol[pol], nc = nc_pol[pol];
        const int32_t *lab = pol ? lab1 : lab0;
        for (uint32_t c = tid; c < nc; c += DET_T) csize[c] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += DET_T) {
            const int32_t l = st.label(o + i, lab, i);
            if (l >= 0) atomicAdd(&csize[l], 1u);
        }
        __syncthreads();
        // kept clusters (:89-117): renumber, and lay their member lists out back to back
        {
            const uint32_t per = (nc + DET_T - 1) / DET_T, c0 = tid * per;
            uint32_t k = 0, m = 0;
            for (uint32_t c = c0; c < c0 + per && c < nc; c++) {
                if (csize[c] >= prm.cluster_min) {
                    k++;
                    m += csize[c];
                }
            }
            uint32_t ek, em, tk, tm;
            if constexpr (ST::INT_PIXELS) block_exscan_pair16<DET_T>(k, m, red, &ek, &em, &tk, &tm);   // staged: <= 2816 points, <= 512 clusters
            else block_exscan_pair<DET_T>(k, m, red, &ek, &em, &tk, &tm);
            for (uint32_t c = c0; c < c0 + per && c < nc; c++) {
                if (csize[c] >= prm.cluster_min) {
                    newid[c] = (typename ST::CIdx) ek;
                    coff[c] = (typename ST::CIdx) em;
                    st.koff[ko + ek] = em;
                    st.ksize[ko + ek] = csize[c];
                    ek++;
                    em += csize[c];
                } else {
                    newid[c] = ST::CNONE;
                }
            }
            if (tid == 0) {
                nk_sh[pol] = tk;
                nk_sh[2 + pol] = tm;  // members of kept clusters
            }
        }
        __syncthreads();
        // per-point renumbered label; member lists (arbitrary order first)
        for (uint32_t i = tid; i < n; i += DET_T) {
            const int32_t l = st.label(o + i, lab, i);
            int32_t kl = -1;
            if (l >= 0 && newid[l] != ST::CNONE) {
                kl = (int32_t) newid[l];
                const uint32_t at = atomicSub(&csize[l], 1u) - 1u;
                st.members[o + coff[l] + at] = st.member_word(o + i, i);
                st.set_norm(o + i, norm_of(st.pt(o + i)));
            }
            st.kept[o + i] = kl;
        }
        __syncthreads();
    }
    DET_MARK(1);
    if (ECAL_DET_STOP == 2) return;
    const uint32_t nk[2] = {nk_sh[0], nk_sh[1]};
    if (nk[0] < prm.need_clusters || nk[1] < prm.need_clusters) {  // :127-129
        if (tid == 0) {
            info[0] = 0;
            info[1] = nk[0];
            info[2] = nk[1];
            info[3] = 1;
        }
        return;
    }
    // One scan of its cluster per kept point gives (a) its rank in the order (norm, pid): rank
    // size/2 is the representative (:136-147), and (b) its position in ascending-pid order, which
    // turns the scattered member list into a sorted one.
    // The loop runs over the member lists, not over the points: neighbouring lanes then work on the same cluster —
    // equal trip counts (a wave pays for its longest scan) and broadcast LDS reads.  Both polarities in one loop.
    const uint32_t tm0 = nk_sh[2], tm1 = nk_sh[3];
    for (uint32_t idx = tid; idx < tm0 + tm1; idx += DET_T) {
        const int pol = idx >= tm0 ? 1 : 0;
        const uint32_t o = base[pol];
        const uint32_t wq = st.members[o + idx - (pol ? tm0 : 0u)];
        const uint32_t i = st.composite() ? (wq & ST::IDX_MASK) : wq;
        const int32_t kl = st.kept[o + i];
        const uint32_t m = st.ksize[kb[pol] + kl], first = o + st.koff[kb[pol] + kl];
        uint32_t rank = 0, at = 0, eq = 0;   // eq (ORD): members of the same norm, itself included
        if (st.composite()) {
            const uint32_t wi = wq;
            for (uint32_t t = 0; t < m; t++) {
                const uint32_t wj = st.members[first + t];
                rank += (wj < wi) ? 1u : 0u;
                at += ((wj & ST::IDX_MASK) < i) ? 1u : 0u;
                if constexpr (ORD || TDET) eq += ((wj >> ST::IDXB) == (wi >> ST::IDXB)) ? 1u : 0u;
            }
        } else {
            const auto ni = st.key(o + i);
            for (uint32_t t = 0; t < m; t++) {
                const uint32_t j = st.members[first + t];
                const auto nj = st.key(o + j);
                rank += (nj < ni || (nj == ni && j < i)) ? 1u : 0u;
                at += (j < i) ? 1u : 0u;
                if constexpr (ORD || TDET) eq += (nj == ni) ? 1u : 0u;
            }
        }
        if (rank == m / 2) st.rep[kb[pol] + kl] = ((ORD || TDET) && eq > 1u) ? (i | ST::REP_TIE) : i;
        st.sorted[first + at] = i;
    }
    __syncthreads();
    if constexpr (TDET && !ORD) {
        for (int pol = 0; pol < 2; pol++)
            for (uint32_t k = tid; k < nk[pol]; k += DET_T)
                if (st.rep[kb[pol] + k] & ST::REP_TIE) nk_sh[pol] |= 0x80000000u;   // (nk[] was read into registers above; every writer stores the same bit)
        __syncthreads();
        const uint32_t tied0 = nk_sh[0] >> 31, tied1 = nk_sh[1] >> 31;
        const bool window_tied = (tied0 | tied1) != 0;
        __syncthreads();
        if (tid == 0) {
            nk_sh[0] &= 0x7FFFFFFFu;
            nk_sh[1] &= 0x7FFFFFFFu;
        }
        bool resolved = false;
        if constexpr (ST::INT_PIXELS) {   // staged integer pixels: the tied clusters' member order worked out here (resolve_ties_inline)
            if (window_tied && tie_tree) {
                // First pass (<= 1408 points, 384 clusters): the tree goes where csize / newid / coff lay (768 words: the pixel DBSCAN
                // kernel's first-pass segments), the members' positions behind the staging (DET_TIE_INV_BYTES, added by the launch).
                // Second pass (<= 2816 points): the waves' scratch takes the first half of members[], the tree its second half
                // (1408 words), the positions the place of csize — segments of up to 1408 points, whose trees the DBSCAN kernel's
                // second pass exports.
                constexpr bool FIRST_LAYOUT = ST::IDXB == 11u;   // (the 1408-point staging; the second pass's 2816 points take 12 index bits)
                constexpr uint32_t TREE_CAP = FIRST_LAYOUT ? DET_TIE_TREE_CAP : 1408u;
                static_assert(FIRST_LAYOUT ? 8u * ST::MAXC >= 4u * DET_TIE_TREE_CAP : 8u * ST::MAXC >= 1408u, "the tree / the positions fit where csize, newid and coff lay");
                uint32_t *const tree_lds = FIRST_LAYOUT ? csize : st.members + 1408;
                uint8_t *const inv = FIRST_LAYOUT ? tie_inv : reinterpret_cast<uint8_t *>(csize);
                // the trees of the tied polarities must be at hand: flagged with the DBSCAN call's number
                const bool ok0 = !tied0 || (n_pol[0] <= TREE_CAP && tie_flags[0] == prm.px_tree_epoch);
                const bool ok1 = !tied1 || (n_pol[1] <= TREE_CAP && tie_flags[1] == prm.px_tree_epoch);
                if (ok0 && ok1 && inv)
                    resolved = resolve_ties_inline(st, base, kb, n_pol, nk, tied0, tied1, tie_tree, prm.tie_e2i, prm.tie_prune, tree_lds,
                                                   reinterpret_cast<unsigned char *>(st.members), inv, &nk_sh[2]);
            }
        }
        for (int pol = 0; pol < 2; pol++)
            for (uint32_t k = tid; k < nk[pol]; k += DET_T) {
                const uint32_t rv = st.rep[kb[pol] + k];
                if (!(rv & ST::REP_TIE)) continue;
                const uint32_t idx = rv & ~(ST::REP_TIE | ST::REP_BAD);
                if (resolved) {
                    st.rep[kb[pol] + k] = idx;    // the reference's pick
                } else {
                    // the window goes on the list: the smaller-pid member stays for now, and the cluster is named to
                    // ecal_cluster_order_list_dev by a mark on the slot of one of its points (idx is one, resolved inline or not)
                    st.rep[kb[pol] + k] = idx;
                    int32_t *mk = pol ? mark1 : mark0;
                    if (mk) mk[idx] = -3;
                }
            }
        __syncthreads();
        if (tid == 0 && window_tied && !resolved) {
            // the list entry: the window, bits 30 / 31 set = its + / - segment holds no tied cluster (ecal_cluster_order_list_dev skips it)
            if (tie_list) tie_list[atomicAdd(tie_count, 1u)] = tie_token | ((tied0 ^ 1u) << 30) | ((tied1 ^ 1u) << 31);
        }
        __syncthreads();
        // a listed window is extracted again from the start with the reference's picks (ORD): its pairing here would be
        // thrown away
        if (window_tied && !resolved && tie_list) return;
    }
    if constexpr (ORD) {
        // the flagged clusters: members into the reference's order (members[] is free from here on; a thread per POINT
        // scatters itself to its position), then the library's nth_element on them, a thread per cluster.  Without a usable
        // order (segment not taken by ecal_cluster_order_dev) the smaller pid stays — and the window says so in its status word
        // (ECAL_WIN_TIE_FALLBACK): that pick is not guaranteed to be the reference's.
        if (tid == 0) nk_sh[2] = 0;   // (the member totals were read above; the word now collects the fallbacks)
        // (two steps: a cluster without a usable order keeps its member list as it is — ascending pid, which the circle test's
        // sums run over —, so nobody scatters before every member's position has been looked at)
        for (int pol = 0; pol < 2; pol++) {
            const int32_t *ord = pol ? ord1 : ord0;
            for (uint32_t i = tid; i < n_pol[pol]; i += DET_T) {
                const int32_t kl = st.kept[base[pol] + i];
                if (kl < 0) continue;
                const uint32_t rv = st.rep[kb[pol] + kl];
                if (!(rv & ST::REP_TIE)) continue;
                const int32_t p = ord ? ord[i] : -1;
                if (p < 0 || (uint32_t) p >= st.ksize[kb[pol] + kl]) st.rep[kb[pol] + kl] = rv | ST::REP_BAD;   // (every writer: the same value)
            }
        }
        __syncthreads();
        for (int pol = 0; pol < 2; pol++) {
            const int32_t *ord = pol ? ord1 : ord0;
            for (uint32_t i = tid; i < n_pol[pol]; i += DET_T) {
                const int32_t kl = st.kept[base[pol] + i];
                if (kl < 0) continue;
                const uint32_t rv = st.rep[kb[pol] + kl];
                if ((rv & (ST::REP_TIE | ST::REP_BAD)) != ST::REP_TIE) continue;
                st.members[base[pol] + st.koff[kb[pol] + kl] + (uint32_t) ord[i]] = i;
            }
        }
        __syncthreads();
        for (int pol = 0; pol < 2; pol++) {
            for (uint32_t k = tid; k < nk[pol]; k += DET_T) {
                const uint32_t rv = st.rep[kb[pol] + k];
                if (!(rv & ST::REP_TIE)) continue;
                uint32_t pick = rv & ~(ST::REP_TIE | ST::REP_BAD);
                const uint32_t m = st.ksize[kb[pol] + k], first = base[pol] + st.koff[kb[pol] + k];
                if (!(rv & ST::REP_BAD)) pick = ref_nth_member(st, base[pol], &st.members[first], m, m / 2u);
                else nk_sh[2] = 1;   // no member order for this cluster (every writer: the same value)
                st.rep[kb[pol] + k] = pick;
            }
        }
        __syncthreads();
    }
    DET_MARK(2);
    if (ECAL_DET_STOP == 3) return;
    // mutual nearest +/- representatives and the circle test (:283-311); candidates in + cluster order
    // (Dealing the ~40 + clusters of a window round-robin over the four waves was tried: 0.93 -> 1.06 ms.  The kernel
    // is issue bound, and four waves with 10 active lanes issue four times the instructions of one wave with 40.)
    uint32_t carry = 0;
    auto emit = [&](uint32_t pi, bool ok, uint32_t ni_best, double cx, double cy, double r) {
        uint32_t ex, dummy, tot, dummy2;
        block_exscan_pair16<DET_T>(ok ? 1u : 0u, 0u, red, &ex, &dummy, &tot, &dummy2);   // <= 256 candidates per batch
        if (ok) {
            const size_t at = (size_t) carry + ex;
            cand_pair[2 * at] = pi;
            cand_pair[2 * at + 1] = ni_best;
            cand_xyr[3 * at] = cx;
            cand_xyr[3 * at + 1] = cy;
            cand_xyr[3 * at + 2] = r;
        }
        carry += tot;
    };
    bool paired = false;
    if constexpr (!FIT && ST::INT_PIXELS) {
        // Staged integer pixels, <= 64 + clusters (the usual window): the searches stay on wave 0 (a lane per + cluster),
        // but the circle test's |distance - r| terms — a correctly rounded f64 square root per member, ~900 instructions
        // when one lane walks both clusters of its pair — are computed by all four waves, a term per point, and only
        // their summation (in the reference's order: + members, then - members, ascending pid) goes back to the lane.
        // The terms take the place of members[] and pts[], which nothing reads after this.
        if (st.composite() && nk[0] <= 64u && nk[1] <= 64u) {
            paired = true;
            static_assert(4 * ST::MAXC >= 64 * 3 * sizeof(double), "the pair circles take the place of csize[]");
            double *const circ = reinterpret_cast<double *>(csize);          // [64][3]: cx, cy, r of + cluster pi's pair
            typename ST::CIdx *const pair_p = newid, *const pair_n = coff;   // cluster -> its pair (+ cluster index) or none
            double *const term = reinterpret_cast<double *>(st.members);     // [n_all]
            uint32_t *const rp = st.members;  // until the terms arrive: the representatives' packed pixels, [0..63] +, [64..127] -
            for (uint32_t k = tid; k < nk[0]; k += DET_T) {
                pair_p[k] = ST::CNONE;
                rp[k] = st.ipt(base[0] + st.rep[kb[0] + k]);
            }
            for (uint32_t k = tid; k < nk[1]; k += DET_T) {
                pair_n[k] = ST::CNONE;
                rp[64 + k] = st.ipt(base[1] + st.rep[kb[1] + k]);
            }
            __syncthreads();
            const uint32_t pi = tid;
            bool cand = false;
            uint32_t ni_best = 0;
            double cx = 0, cy = 0, r = 0;
            if (pi < nk[0]) {
                const uint32_t pw = rp[pi];
                const int px = (int) (short) (pw & 0xFFFFu), py = ((int) pw) >> 16;
                uint32_t bi = 0xFFFFFFFFu;
                for (uint32_t k = 0; k < nk[1]; k++) {  // nanoflann 1-NN, metric_L2_Simple
                    const uint32_t cw = rp[64 + k];
                    const int dx = px - (int) (short) (cw & 0xFFFFu), dy = py - (((int) cw) >> 16);
                    const uint32_t d = (uint32_t) (dx * dx) + (uint32_t) (dy * dy);
                    if (d < bi) {
                        bi = d;
                        ni_best = k;
                    }
                }
                if (bi != 0xFFFFFFFFu && !((double) bi > prm.four_thr2)) {  // :286
                    const uint32_t nw = rp[64 + ni_best];
                    const int nx = (int) (short) (nw & 0xFFFFu), ny = ((int) nw) >> 16;
                    uint32_t bi2 = 0xFFFFFFFFu, back = 0;
                    for (uint32_t k = 0; k < nk[0]; k++) {
                        const uint32_t cw = rp[k];
                        const int dx = nx - (int) (short) (cw & 0xFFFFu), dy = ny - (((int) cw) >> 16);
                        const uint32_t d = (uint32_t) (dx * dx) + (uint32_t) (dy * dy);
                        if (d < bi2) {
                            bi2 = d;
                            back = k;
                        }
                    }
                    if (back == pi) {
                        cand = true;
                        const double2 pc = st.pt(base[0] + st.rep[kb[0] + pi]), nc = st.pt(base[1] + st.rep[kb[1] + ni_best]);
                        cx = (pc.x + nc.x) / 2;
                        cy = (pc.y + nc.y) / 2;
                        const double ddx = pc.x - nc.x, ddy = pc.y - nc.y;
                        r = __dsqrt_rn(ddx * ddx + ddy * ddy) / 2;
                        circ[3 * pi] = cx;
                        circ[3 * pi + 1] = cy;
                        circ[3 * pi + 2] = r;
                        pair_p[pi] = (typename ST::CIdx) pi;
                        pair_n[ni_best] = (typename ST::CIdx) pi;
                    }
                }
            }
            __syncthreads();
            constexpr int J = ST::J;
            const uint32_t n_all = n_pol[0] + n_pol[1];
            double tv[J];
#pragma unroll
            for (int j = 0; j < J; j++) {
                const uint32_t li = tid + j * DET_T;   // staged: base = {0, n_pol[0]}, so li is the window-local point index
                tv[j] = 0.0;
                if (li < n_all) {
                    const int pol = li >= n_pol[0] ? 1 : 0;
                    const int32_t kl = st.kept[li];
                    if (kl >= 0) {
                        const uint32_t pr = pol ? pair_n[kl] : pair_p[kl];
                        if (pr != ST::CNONE) {
                            const double2 e = st.pt(li);
                            const double ex = e.x - circ[3 * pr], ey = e.y - circ[3 * pr + 1];
                            tv[j] = fabs(__dsqrt_rn(ex * ex + ey * ey) - circ[3 * pr + 2]);
                        }
                    }
                }
            }
            __syncthreads();  // every point has been read: the terms may take the place of members[] and pts[]
#pragma unroll
            for (int j = 0; j < J; j++) {
                const uint32_t li = tid + j * DET_T;
                if (li < n_all) term[li] = tv[j];
            }
            __syncthreads();
            bool ok = false;
            if (cand) {
                double fit = 0;
                uint32_t cnt = 0;
                for (int pol = 0; pol < 2; pol++) {
                    const uint32_t o = base[pol], kk = pol ? ni_best : pi;
                    const uint32_t m = st.ksize[kb[pol] + kk], first = o + st.koff[kb[pol] + kk];
                    for (uint32_t t = 0; t < m; t++) fit += term[o + st.sorted[first + t]];  // ascending pid
                    cnt += m;
                }
                fit /= (double) cnt * r;
                ok = fit < 10 / r;
            }
            emit(pi, ok, ni_best, cx, cy, r);
        }
    }
    for (uint32_t p0 = 0; !paired && p0 < nk[0]; p0 += DET_T) {
        const uint32_t pi = p0 + tid;
        bool ok = false;
        uint32_t ni_best = 0;
        double cx = 0, cy = 0, r = 0;
        if (FIT && pi < nk[0]) {  // :180-281
            uint32_t n_idx[DET_KNN_MAX], p_idx[DET_KNN_MAX];
            uint32_t real = knn_gated(st, base[1], kb[1], nk[1], st.pt(base[0] + st.rep[kb[0] + pi]), prm.knn, prm.four_thr2,
                                      n_idx);
            if (real > 0) {
                CircleFit best = fit_pair(st, base, kb, pi, n_idx[0], prm.thr);
                uint32_t nmin = 0;
                for (uint32_t j = 1; j < real; j++) {
                    const CircleFit f = fit_pair(st, base, kb, pi, n_idx[j], prm.thr);
                    if (f.err < best.err) {  // std::min_element: first minimum
                        best = f;
                        nmin = j;
                    }
                }
                if (best.err < 2 / best.radius) {  // :229-230
                    const uint32_t nsel = n_idx[nmin];
                    real = knn_gated(st, base[0], kb[0], nk[0], st.pt(base[1] + st.rep[kb[1] + nsel]), prm.knn, prm.four_thr2,
                                     p_idx);
                    if (real > 0) {
                        CircleFit bb = fit_pair(st, base, kb, p_idx[0], nsel, prm.thr);
                        uint32_t pmin = 0;
                        for (uint32_t i = 1; i < real; i++) {
                            const CircleFit f = fit_pair(st, base, kb, p_idx[i], nsel, prm.thr);
                            if (f.err < bb.err) {
                                bb = f;
                                pmin = i;
                            }
                        }
                        if (p_idx[pmin] == pi) {  // :275
                            ok = true;
                            ni_best = nsel;
                            cx = bb.cx;
                            cy = bb.cy;
                            r = bb.radius;
                        }
                    }
                }
            }
        } else if (pi < nk[0]) {
            const double2 pc = st.pt(base[0] + st.rep[kb[0] + pi]);
            double bd = 1.79769313486231570e308;
            uint32_t back = 0;
            bool near = false;
            bool int_nn = false;
            if constexpr (ST::INT_PIXELS) int_nn = st.composite();  // |v| <= 1023: no 32-bit overflow below
            if (int_nn) {
                // integer pixels: dx^2 + dy^2 is exact in 32-bit integers and orders like the reference's doubles
                const uint32_t pw = st.ipt(base[0] + st.rep[kb[0] + pi]);
                const int px = (int) (short) (pw & 0xFFFFu), py = ((int) pw) >> 16;
                uint32_t bi = 0xFFFFFFFFu;
                for (uint32_t k = 0; k < nk[1]; k++) {  // nanoflann 1-NN, metric_L2_Simple
                    const uint32_t cw = st.ipt(base[1] + st.rep[kb[1] + k]);
                    const int dx = px - (int) (short) (cw & 0xFFFFu), dy = py - (((int) cw) >> 16);
                    const uint32_t d = (uint32_t) (dx * dx) + (uint32_t) (dy * dy);
                    if (d < bi) {
                        bi = d;
                        ni_best = k;
                    }
                }
                if (bi != 0xFFFFFFFFu) bd = (double) bi;
                if (!(bd > prm.four_thr2)) {  // :286
                    near = true;
                    const uint32_t nw = st.ipt(base[1] + st.rep[kb[1] + ni_best]);
                    const int nx = (int) (short) (nw & 0xFFFFu), ny = ((int) nw) >> 16;
                    uint32_t bi2 = 0xFFFFFFFFu;
                    for (uint32_t k = 0; k < nk[0]; k++) {
                        const uint32_t cw = st.ipt(base[0] + st.rep[kb[0] + k]);
                        const int dx = nx - (int) (short) (cw & 0xFFFFu), dy = ny - (((int) cw) >> 16);
                        const uint32_t d = (uint32_t) (dx * dx) + (uint32_t) (dy * dy);
                        if (d < bi2) {
                            bi2 = d;
                            back = k;
                        }
                    }
                }
            } else {
                for (uint32_t k = 0; k < nk[1]; k++) {  // nanoflann 1-NN, metric_L2_Simple
                    const double2 c = st.pt(base[1] + st.rep[kb[1] + k]);
                    const double dx = pc.x - c.x, dy = pc.y - c.y;
                    const double d = dx * dx + dy * dy;
                    if (d < bd) {
                        bd = d;
                        ni_best = k;
                    }
                }
                if (!(bd > prm.four_thr2)) {  // :286
                    near = true;
                    const double2 nc = st.pt(base[1] + st.rep[kb[1] + ni_best]);
                    double bd2 = 1.79769313486231570e308;
                    for (uint32_t k = 0; k < nk[0]; k++) {
                        const double2 c = st.pt(base[0] + st.rep[kb[0] + k]);
                        const double dx = nc.x - c.x, dy = nc.y - c.y;
                        const double d = dx * dx + dy * dy;
                        if (d < bd2) {
                            bd2 = d;
                            back = k;
                        }
                    }
                }
            }
            if (near) {
                const double2 nc = st.pt(base[1] + st.rep[kb[1] + ni_best]);
                if (back == pi) {
                    cx = (pc.x + nc.x) / 2;
                    cy = (pc.y + nc.y) / 2;
                    const double ddx = pc.x - nc.x, ddy = pc.y - nc.y;
                    r = __dsqrt_rn(ddx * ddx + ddy * ddy) / 2;
                    double fit = 0;
                    uint32_t cnt = 0;
                    for (int pol = 0; pol < 2; pol++) {
                        const uint32_t o = base[pol], kk = pol ? ni_best : pi;
                        const uint32_t m = st.ksize[kb[pol] + kk], first = o + st.koff[kb[pol] + kk];
                        for (uint32_t t = 0; t < m; t++) {  // ascending pid
                            const double2 e = st.pt(o + st.sorted[first + t]);
                            const double ex = e.x - cx, ey = e.y - cy;
                            fit += fabs(__dsqrt_rn(ex * ex + ey * ey) - r);
                        }
                        cnt += m;
                    }
                    fit /= (double) cnt * r;
                    ok = fit < 10 / r;
                }
            }
        }
        emit(pi, ok, ni_best, cx, cy, r);
    }
    DET_MARK(3);
    if (tid == 0) {
        info[0] = carry;
        info[1] = nk[0];
        info[2] = nk[1];
        info[3] = (ORD && nk_sh[2]) ? ECAL_WIN_TIE_FALLBACK : 0u;
    }
}

// LDS of the staged path: per DBSCAN cluster (csize u32: atomics; newid, coff), per kept cluster and polarity (koff,
// ksize, rep), per point (sorted, kept, members, pts)
template <uint32_t PTS, uint32_t MAXC>
struct DetLdsLayoutT {
    static constexpr size_t csize_off = 0;                                          // u32[MAXC]
    static constexpr size_t newid_off = csize_off + 4 * MAXC;              // u16[MAXC]
    static constexpr size_t coff_off = newid_off + 2 * MAXC;               // u16[MAXC]
    static constexpr size_t koff_off = coff_off + 2 * MAXC;                // u16[2 * MAXC]
    static constexpr size_t ksize_off = koff_off + 4 * MAXC;               // u16[2 * MAXC]
    static constexpr size_t rep_off = ksize_off + 4 * MAXC;                // u16[2 * MAXC]
    static constexpr size_t sorted_off = rep_off + 4 * MAXC;               // u16[PTS]
    static constexpr size_t kept_off = sorted_off + 2 * PTS;               // i16[PTS]
    static constexpr size_t members_off = kept_off + 2 * PTS;              // u32[PTS]
    static constexpr size_t pts_off = members_off + 4 * PTS;               // u32[PTS]
    static constexpr size_t bytes = pts_off + 4 * PTS;
};

// One window.  PTS / MAXC: capacity of the LDS staging.  FIRST: the first pass hands windows that do not fit its staging but
// fit the second pass's (DET_LDS_PTS2 points, DET_LDS_MAXC2 clusters) to the to-do list instead of taking the global path.
// KNOWN: offsets, counts and cluster counts of the window's two segments are handed in (the fused pass, ecal_fused.hip: the
// workgroup wrote them itself a moment ago; a scalar load might find a stale line in the constant cache).
template <bool FIT, uint32_t PTS, uint32_t MAXC, bool FIRST, bool KNOWN = false, bool ORD = false, bool TDET = false>
__device__ __forceinline__ void extract_one(
    unsigned char *smem, unsigned long long *red, uint32_t *nk_sh, const uint32_t s, const double *__restrict__ xy,
    const uint32_t *__restrict__ seg_off, const uint32_t *__restrict__ seg_cnt, const int32_t *__restrict__ labels,
    const uint32_t *__restrict__ n_clusters, const DetectParams &prm, uint32_t *__restrict__ win_info,
    uint32_t *__restrict__ cand_pair, double *__restrict__ cand_xyr, int32_t *__restrict__ kept_labels,
    uint32_t *__restrict__ rep, uint32_t *__restrict__ members, uint32_t *__restrict__ koff, uint32_t *__restrict__ ksize,
    uint32_t *__restrict__ sorted, double *__restrict__ norms, uint32_t *__restrict__ todo, uint32_t *__restrict__ todo_count,
    const uint32_t *known = nullptr, const int32_t *__restrict__ order = nullptr, uint32_t *tie_list = nullptr,
    uint32_t *tie_count = nullptr, int32_t *tie_mark = nullptr) {
    using LL = DetLdsLayoutT<PTS, MAXC>;
    // csize: members per DBSCAN cluster, later a scatter cursor; newid: renumbered id of a kept cluster;
    // coff: first member slot of a kept cluster.  Sized for the global path; the LDS path uses the first
    // DET_LDS_MAXC entries and the rest of the block for its staged arrays.
    const uint32_t tid = threadIdx.x;
    const double2 *pts = reinterpret_cast<const double2 *>(xy);
    uint32_t *info = win_info + 4 * (size_t) s;
    const uint32_t o_pol[2] = {KNOWN ? known[0] : seg_off[2 * s], KNOWN ? known[1] : seg_off[2 * s + 1]};
    const uint32_t n_pol[2] = {KNOWN ? known[2] : seg_cnt[2 * s], KNOWN ? known[3] : seg_cnt[2 * s + 1]};
    const uint32_t nc_pol[2] = {KNOWN ? known[4] : n_clusters[2 * s], KNOWN ? known[5] : n_clusters[2 * s + 1]};

    if (n_pol[0] == 0 || n_pol[1] == 0 || nc_pol[0] > DET_MAXC || nc_pol[1] > DET_MAXC) {
        // empty polarity: CirclesEventFrame.cpp:62-64; otherwise capacity exceeded (status 4)
        for (int pol = 0; pol < 2; pol++)
            for (uint32_t i = tid; i < n_pol[pol]; i += DET_T) kept_labels[o_pol[pol] + i] = -1;
        if (tid == 0) {
            info[0] = 0;
            info[1] = 0;
            info[2] = 0;
            info[3] = (n_pol[0] == 0 || n_pol[1] == 0) ? 1 : 4;
        }
        return;
    }
    // the window's slots are contiguous: positives then negatives (ecal_slice_events_dev layout) — if not,
    // or if the window is too large, work in global scratch
    const bool contiguous = o_pol[1] == o_pol[0] + n_pol[0];
    const uint32_t n_all = n_pol[0] + n_pol[1];
    bool staged = contiguous && n_all <= PTS && nc_pol[0] <= MAXC && nc_pol[1] <= MAXC;
    if (FIRST && !staged && contiguous && n_all <= DET_LDS_PTS2 && nc_pol[0] <= DET_LDS_MAXC2 && nc_pol[1] <= DET_LDS_MAXC2 && todo) {
        if (tid == 0) todo[atomicAdd(todo_count, 1u)] = s;   // the second pass stages it in LDS
        return;
    }
    uint32_t *csize = reinterpret_cast<uint32_t *>(smem);
    uint32_t *const lds_pts = reinterpret_cast<uint32_t *>(smem + LL::pts_off);
    DET_T0();
    bool small_px = false;
    // the composite member word holds key = x^2 + y^2 in 32 - IDXB bits
    constexpr double KEY_LIM = PTS > 2048u ? 723.0 : 1023.0;
    if (staged) {  // stage the points packed; a coordinate that does not pack exactly sends the window to the global path
        bool fits = true, large = false;
        // all of the thread's loads are issued before the first is used (index clamped instead of a branch around the load):
        // inside `if (i < n_all)` the compiler waits for each one in turn — five trips to HBM in a row at the head of every
        // workgroup instead of one
        constexpr int JS = (int) ((PTS + DET_T - 1) / DET_T);
        // (packed points: the window's pixels are read as the 4-byte words they were sliced into — a quarter of the bytes)
        const bool packed = !KNOWN && prm.xy16 && (prm.seg_fmt[2 * s] & 1u);
        double2 vin[JS];
        uint32_t win[JS];
        int32_t lin[JS];
#pragma unroll
        for (int j = 0; j < JS; j++) {
            const uint32_t i = min(tid + j * DET_T, n_all - 1u);
            if (packed) win[j] = prm.xy16[o_pol[0] + i];
            else vin[j] = pts[o_pol[0] + i];
            lin[j] = labels[o_pol[0] + i];
        }
#pragma unroll
        for (int j = 0; j < JS; j++) {
            const uint32_t i = tid + j * DET_T;
            if (i < n_all) {
                double2 v;
                if (packed) v = make_double2((double) (int) (short) (win[j] & 0xFFFFu), (double) (((int) win[j]) >> 16));
                else v = vin[j];
                fits = fits && v.x == floor(v.x) && v.y == floor(v.y) && fabs(v.x) <= 32767.0 && fabs(v.y) <= 32767.0;
                large = large || !(fabs(v.x) <= KEY_LIM && fabs(v.y) <= KEY_LIM);
                lds_pts[i] = ((uint32_t) (int) v.x & 0xFFFFu) | ((uint32_t) (int) v.y << 16);
                // labels < MAXC (checked above) or -1: they fit the int16 table that later holds the renumbered ones
                reinterpret_cast<int16_t *>(smem + LL::kept_off)[i] = (int16_t) lin[j];
            }
        }
        // bit 0: some coordinate does not pack; bit 1: some coordinate is beyond the composite-key range
        if (tid == 0) nk_sh[0] = 0;
        __syncthreads();
        if (!fits || large) atomicOr(&nk_sh[0], (fits ? 0u : 1u) | (large ? 2u : 0u));
        __syncthreads();
        const uint32_t verdict = nk_sh[0];
        staged = !(verdict & 1);
        small_px = !(verdict & 2);
        DET_MARK(0);
        if (ECAL_DET_STOP == 1) {
            for (uint32_t i = tid; i < n_all; i += DET_T) kept_labels[o_pol[0] + i] = (int32_t) lds_pts[i];
            return;
        }
    }
    if (staged) {
        DetLdsT<PTS, MAXC> st;
        st.sorted = reinterpret_cast<uint16_t *>(smem + LL::sorted_off);
        st.koff = reinterpret_cast<uint16_t *>(smem + LL::koff_off);
        st.ksize = reinterpret_cast<uint16_t *>(smem + LL::ksize_off);
        st.rep = reinterpret_cast<uint16_t *>(smem + LL::rep_off);
        st.kept = reinterpret_cast<int16_t *>(smem + LL::kept_off);
        st.members = reinterpret_cast<uint32_t *>(smem + LL::members_off);
        st.pts = lds_pts;
        st.small = small_px;
        const uint32_t base[2] = {0u, n_pol[0]};
        const uint32_t kb[2] = {0u, MAXC};  // per-cluster arrays: one block of MAXC per polarity
        // TDET, first pass: the kd-trees of the window's two segments, where the pixel DBSCAN kernel exported them (resolve_ties_inline)
        // TDET, first pass: where the pixel DBSCAN kernel exported the kd-trees of the window's two segments (resolve_ties_inline)
        uint8_t *tie_inv = nullptr;
        const uint32_t *tie_tree = nullptr, *tie_flags = nullptr;
        if constexpr (TDET && !KNOWN) {
            if (prm.px_tree) {
                if constexpr (FIRST) tie_inv = smem + (LL::bytes > 3 * DET_MAXC * sizeof(uint32_t) ? LL::bytes : 3 * DET_MAXC * sizeof(uint32_t));   // (behind DET_LDS_BYTES: the launch adds DET_TIE_INV_BYTES)
                tie_tree = prm.px_tree + o_pol[0];
                tie_flags = prm.px_tree_flag + 2 * (size_t) s;
            }
        }
        extract_window<FIT, ORD, TDET>(st, base, kb, n_pol, labels + o_pol[0], labels + o_pol[1], order ? order + o_pol[0] : nullptr,
                                       order ? order + o_pol[1] : nullptr, tie_list, tie_count, s, tie_mark ? tie_mark + o_pol[0] : nullptr,
                                       tie_mark ? tie_mark + o_pol[1] : nullptr, nc_pol, prm, csize,
                       reinterpret_cast<uint16_t *>(smem + LL::newid_off), reinterpret_cast<uint16_t *>(smem + LL::coff_off), red, nk_sh, info, cand_pair + 2 * (size_t) o_pol[0],
                       cand_xyr + 3 * (size_t) o_pol[0], tie_tree, tie_flags, tie_inv);
        __syncthreads();
#ifdef ECAL_PHASE_PROF
        det_t__ = __builtin_readcyclecounter();
#endif
        for (uint32_t i = tid; i < n_all; i += DET_T) kept_labels[o_pol[0] + i] = st.kept[i];
        if (nk_sh[0] >= prm.need_clusters && nk_sh[1] >= prm.need_clusters) {  // representatives exist only then
            for (int pol = 0; pol < 2; pol++)
                for (uint32_t k = tid; k < nk_sh[pol]; k += DET_T) rep[o_pol[pol] + k] = st.rep[kb[pol] + k];
        }
        DET_MARK(4);
        if (tid == 0) {
#ifdef ECAL_PHASE_PROF
            atomicAdd(&g_det_cycles[8], 1ull);
#endif
        }
    } else {
        DetGlobal st;
        const uint32_t w0 = o_pol[0] < o_pol[1] ? o_pol[0] : o_pol[1];  // window-local indices relative to w0
        st.pts = pts + w0;
        st.members = members + w0;
        st.sorted = sorted + w0;
        st.koff = koff + w0;
        st.ksize = ksize + w0;
        st.rep = rep + w0;
        st.kept = kept_labels + w0;
        st.norms = norms + w0;
        const uint32_t base[2] = {o_pol[0] - w0, o_pol[1] - w0};
        extract_window<FIT, ORD, TDET>(st, base, base, n_pol, labels + o_pol[0], labels + o_pol[1], order ? order + o_pol[0] : nullptr,
                                       order ? order + o_pol[1] : nullptr, tie_list, tie_count, s, tie_mark ? tie_mark + o_pol[0] : nullptr,
                                       tie_mark ? tie_mark + o_pol[1] : nullptr, nc_pol, prm, csize, csize + DET_MAXC,
                       csize + 2 * DET_MAXC, red, nk_sh, info, cand_pair + 2 * (size_t) o_pol[0],
                       cand_xyr + 3 * (size_t) o_pol[0], nullptr, nullptr, nullptr);
    }
}
constexpr size_t DET_LDS_BYTES_GLOBAL = 3 * DET_MAXC * sizeof(uint32_t);
constexpr size_t DET_LDS_BYTES_STAGED = DetLdsLayoutT<DET_LDS_PTS, DET_LDS_MAXC>::bytes;
static_assert(DET_LDS_BYTES_STAGED + DET_TIE_INV_BYTES + 64 <= 26624, "six workgroups per CU");
constexpr size_t DET_LDS_BYTES2 = DetLdsLayoutT<DET_LDS_PTS2, DET_LDS_MAXC2>::bytes > DET_LDS_BYTES_GLOBAL
                                      ? DetLdsLayoutT<DET_LDS_PTS2, DET_LDS_MAXC2>::bytes : DET_LDS_BYTES_GLOBAL;
constexpr size_t DET_LDS_BYTES = DET_LDS_BYTES_STAGED > DET_LDS_BYTES_GLOBAL ? DET_LDS_BYTES_STAGED : DET_LDS_BYTES_GLOBAL;

}  // namespace ecal
