// The member ORDER of the reference's clusters.
//
// DBSCAN<T,Float>::Clusters[c] (dbscan/include/dbscan.h:92) lists a cluster's core points in the order expandCluster's queue
// pops them (:229-265): the seed (the cluster's smallest pid, Run's outer loop :140-158), then breadth first, every popped
// core point pushing its not yet queued neighbours in the order regionQuery returns them (:198-227) — the kd-tree's result
// list, which kd_nearest_range fills by inserting each hit at the HEAD (rlist_insert with dist_sq = -1, kdtree.cpp:469-486),
// i.e. the REVERSE of find_nearest's visiting order (:148-179: the node, then the near child, then the far child if
// fabs(dx) < range).  ecal_dbscan_batch_dev's labels say which cluster a point belongs to; this pass adds where in
// Clusters[c] it stands.  extractFeatures' medians (std::nth_element over Clusters[c], CirclesEventFrame.cpp:136-147) depend
// on that order when two members tie in norm.
//
// One workgroup per segment (three LDS tiers up to 4096 points, then a global-scratch tier for anything larger or anything a
// tier's fixed list / stack sizes cannot hold; coordinates as the doubles they are, no pixel assumption):
//   1. the insertion-order kd-tree rebuilt level-synchronously: every unplaced point stands at a node of the current depth
//      and bids its pid with ds_min for the child slot on its side — the smallest pid wins, as sequential insertion
//      (kdtree.cpp:106-146) would place it —, the others go on below the winner; one level and one barrier per round;
//   2. one range query per core point of the wanted clusters, all in parallel (a thread per query): find_nearest's
//      traversal with an explicit stack, hits written in visiting order to a list in LDS or global scratch;
//   3. one queue simulation per wanted cluster, all clusters in parallel (a thread per small cluster, a wave per large one):
//      pops in Clusters[c]'s order.
// Which clusters are wanted: all, those with a tied median (own test), or those the caller marked (only_tied = 0 / 1 / 2).
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "ecal_ctx.hpp"
#include "block_utils.hpp"

#pragma clang fp contract(off)

namespace ecal {

// Three launches.  The first takes segments of <= 768 points (the pixel DBSCAN's first tier too) and <= 256 clusters in 20 KB
// of LDS with 256 threads: 8 workgroups per CU — the pass is latency bound, workgroups in flight pay.  The second (<= 2048
// points) and third (<= 4096 points, <= 2048 clusters: 141 KB, one workgroup per CU) run 1024 threads: the few segments that
// reach them (windows grown by the adaptive policy) are on the critical path of a lock-step pass, where the latency of ONE
// segment counts.
#ifndef ECAL_BO_T1B
#define ECAL_BO_T1B 512
#endif
#ifndef ECAL_BO_WG2
#define ECAL_BO_WG2 2
#endif
constexpr int BO_T1 = 256, BO_T1B = ECAL_BO_T1B, BO_T2 = 1024;
constexpr uint32_t BO_CAP1 = 768, BO_CAP2 = 2048, BO_CAP3 = 4096;
constexpr uint32_t BO_WG1 = 8, BO_WG2 = ECAL_BO_WG2, BO_WG3 = 1;   // workgroups per CU
constexpr uint32_t BO_WAVE_MIN = 32;     // clusters of at least this many members: a wave runs the queue (a lane per neighbour)
constexpr uint32_t BO_MAXN = 64;         // hits per range query kept (a disc of radius 4 holds 48 other pixels)
// Range-query lists kept in LDS (handed out first come first served; the rest go to global scratch).  The first tier keeps a
// token 8: there more workgroups per CU beat lists in LDS (profiles/r02_notes.md).  The later tiers run one workgroup per CU
// and fill the LDS with lists: their queue is a dependent chain of pops, and a list in global memory is a round trip per pop.
#ifndef ECAL_BO_STACK
#define ECAL_BO_STACK 96
#endif
constexpr uint32_t BO_STACK = ECAL_BO_STACK;   // pending far subtrees of one traversal
constexpr uint32_t BO_NONE = 0xFFFFFFFFu;

template <uint32_t CAP>
struct BoLayout {
    static constexpr uint32_t NCAP = CAP == BO_CAP1 ? 256u : (CAP == BO_CAP2 && ECAL_BO_WG2 > 1 ? (ECAL_BO_WG2 > 2 ? 512u : 1024u) : 2048u);   // clusters per segment
    static constexpr uint32_t POOL = CAP == BO_CAP1 ? 8u : (CAP == BO_CAP2 ? (ECAL_BO_WG2 > 1 ? (ECAL_BO_WG2 > 2 ? 12u : 200u) : 640u) : 168u);
    // coordinates: the first tier holds them as floats and takes only segments whose doubles ARE floats (pixels are) — read
    // back and widened they are the same numbers, and 6 KB less LDS is two more workgroups per CU; the rest keep doubles
    static constexpr size_t CB = (CAP == BO_CAP1 || (CAP == BO_CAP2 && ECAL_BO_WG2 > 1)) ? 4 : 8;
    static constexpr size_t px_off = 0;                                   // f32 | f64 [CAP]
    static constexpr size_t py_off = px_off + CB * CAP;                   // f32 | f64 [CAP]
    static constexpr size_t child_off = py_off + CB * CAP;                // u32[2 CAP]: children (left, right) of node i
    static constexpr size_t lab_off = child_off + 8 * CAP;                // i16[CAP]
    static constexpr size_t queue_off = lab_off + 2 * CAP;                // u16[CAP]: the clusters' queues, back to back
    static constexpr size_t qbase_off = queue_off + 2 * CAP;              // u32[NCAP + 1]: members per cluster, then offsets
    static constexpr size_t inq_off = qbase_off + 4 * (NCAP + 1) + 12;    // u32[CAP / 32]: point is (or was) in its cluster's queue
    static constexpr size_t seed_off = inq_off + 4 * (CAP / 32);          // u32[NCAP]: smallest pid per cluster
    static constexpr size_t slot_off = seed_off + 4 * NCAP;               // u16[CAP]: LDS list slot of a point's range query, 0xFFFF: global
    static constexpr size_t pool_off = slot_off + 2 * CAP;                // u16[POOL][BO_MAXN] + u16[POOL] counts
    static constexpr size_t red_off = pool_off + 2 * POOL * (BO_MAXN + 1) + (2 * POOL * (BO_MAXN + 1)) % 16;   // u32[16]
    static constexpr size_t bytes = red_off + 64;
    static_assert(bytes <= 160 * 1024, "LDS of a CU");
};

#ifdef ECAL_PHASE_PROF
#ifndef ECAL_BO_PROF_CAP
#define ECAL_BO_PROF_CAP 0   // marks of one tier only (its CAP)
#endif
static __device__ unsigned long long g_bo_cycles[16];
#define BO_MARK(i)                                                   \
    do {                                                             \
        if (threadIdx.x == 0 && (ECAL_BO_PROF_CAP == 0 || CAP == ECAL_BO_PROF_CAP)) { \
            const unsigned long long now__ = __builtin_amdgcn_s_memtime(); \
            atomicAdd(&g_bo_cycles[i], now__ - bo_t__);              \
            bo_t__ = now__;                                          \
        }                                                            \
    } while (0)
#else
#define BO_MARK(i)
#endif

__device__ __forceinline__ bool bo_block_any(bool v, uint32_t *flag, uint32_t &round) {
    // three rotating flag words: the word cleared now is read next in the round after the next barrier
    uint32_t *cur = flag + round % 3u, *nxt = flag + (round + 1u) % 3u;
    if (threadIdx.x == 0) *nxt = 0;
    if (v) *cur = 1;
    __syncthreads();
    const bool any = *cur != 0;
    round++;
    return any;
}

// status: 0 taken; 1 not taken at all (see ecal.h).  TIER 0, 1, 2: the three launches (a segment's size names its launch)
template <uint32_t CAP, int T, int TIER>
__global__ __launch_bounds__(T, (T == BO_T1 ? 8 : 4)) void cluster_order_kernel(const double *__restrict__ xy, const uint32_t *__restrict__ seg_off,
                                                            const uint32_t *__restrict__ seg_cnt, uint32_t S, double eps,
                                                            const int32_t *__restrict__ labels,
                                                            const uint32_t *__restrict__ n_clusters, int32_t *__restrict__ order,
                                                            uint32_t *__restrict__ status, uint16_t *__restrict__ lists,
                                                            uint8_t *__restrict__ list_cnt, int only_tied,
                                                            const uint32_t *__restrict__ win_list,
                                                            const uint32_t *__restrict__ win_count,
                                                            uint32_t *__restrict__ defer_list, uint32_t *__restrict__ defer_cnt,
                                                            const uint32_t *__restrict__ xy16, const uint32_t *__restrict__ seg_fmt,
                                                            int lean /* TIER 2 only: this launch also takes the second launch's list (ecal_ctx::tail_seen) */,
                                                            const uint32_t *__restrict__ px_tree = nullptr /* TIER 0: the pixel DBSCAN kernel's trees */,
                                                            const uint32_t *__restrict__ px_tree_flag = nullptr, uint32_t px_tree_epoch = 0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using Coord = typename std::conditional<BoLayout<CAP>::CB == 4, float, double>::type;
    Coord *const px = reinterpret_cast<Coord *>(smem + BoLayout<CAP>::px_off);
    Coord *const py = reinterpret_cast<Coord *>(smem + BoLayout<CAP>::py_off);
    auto X = [&](uint32_t i) { return (double) px[i]; };
    auto Y = [&](uint32_t i) { return (double) py[i]; };
    uint32_t *const child = reinterpret_cast<uint32_t *>(smem + BoLayout<CAP>::child_off);
    int16_t *const lab = reinterpret_cast<int16_t *>(smem + BoLayout<CAP>::lab_off);
    uint16_t *const queue = reinterpret_cast<uint16_t *>(smem + BoLayout<CAP>::queue_off);
    uint32_t *const qbase = reinterpret_cast<uint32_t *>(smem + BoLayout<CAP>::qbase_off);
    uint32_t *const inq = reinterpret_cast<uint32_t *>(smem + BoLayout<CAP>::inq_off);
    uint32_t *const seed = reinterpret_cast<uint32_t *>(smem + BoLayout<CAP>::seed_off);
    uint32_t *const red = reinterpret_cast<uint32_t *>(smem + BoLayout<CAP>::red_off);
    uint16_t *const slotmap = reinterpret_cast<uint16_t *>(smem + BoLayout<CAP>::slot_off);
    uint16_t *const pool = reinterpret_cast<uint16_t *>(smem + BoLayout<CAP>::pool_off);
    constexpr uint32_t BO_POOL = BoLayout<CAP>::POOL;
    uint16_t *const pool_cnt = pool + BO_POOL * BO_MAXN;
    constexpr int BO_PPT = (int) ((CAP + T - 1) / T);
    constexpr uint32_t BO_T = T;
    const uint32_t tid = threadIdx.x;
    // this workgroup's slice of the neighbour lists
    uint16_t *const my_lists = lists + (size_t) blockIdx.x * CAP * BO_MAXN;
    uint8_t *const my_cnt = list_cnt + (size_t) blockIdx.x * CAP;
    const double eps2 = eps * eps;   // SQ(range), kdtree.cpp:155-159

    // win_list: only the two segments (2 w, 2 w + 1) of the listed windows w.  The first launch looks at all of them and lists
    // the ones it leaves to the second (defer_list[0 .. S)) and third (defer_list[S .. 2 S)), which look at nothing else.
    const uint32_t n_first = (TIER == 2 && lean) ? defer_cnt[0] : 0u;   // (lean: list 0, then this launch's own list 1)
    const uint32_t n_work = TIER > 0 ? defer_cnt[TIER - 1] + n_first : (win_list ? 2u * *win_count : S);
    // A segment starts with a chain of dependent global reads — which segment, its extent, then its points — that is a third of
    // the time a workgroup spends on it: the first two links are fetched one and two segments ahead.
    // (a list entry with bit 30 / 31 set: the caller has no use for the window's first / second segment — BO_NONE, skipped)
    auto seg_of = [&](uint32_t w) -> uint32_t {
        if (TIER > 0) return w < n_first ? defer_list[w] : defer_list[(size_t) (TIER - 1) * S + (w - n_first)];
        if (!win_list) return w;
        const uint32_t t = win_list[w >> 1];
        return ((t >> (30u + (w & 1u))) & 1u) ? BO_NONE : 2u * (t & 0x3FFFFFFFu) + (w & 1u);
    };
    const uint32_t G = gridDim.x;
    uint32_t s_cur = blockIdx.x < n_work ? seg_of(blockIdx.x) : 0u, n_cur = 0, base_cur = 0, nc_cur = 0;
    if (blockIdx.x < n_work && s_cur != BO_NONE) {
        n_cur = seg_cnt[s_cur];
        base_cur = seg_off[s_cur];
        nc_cur = n_clusters[s_cur];
    }
    uint32_t s_nxt = blockIdx.x + G < n_work ? seg_of(blockIdx.x + G) : 0u;
    for (uint32_t wk = blockIdx.x; wk < n_work; wk += G) {
        const uint32_t s = s_cur, n = n_cur, base = base_cur, nc = nc_cur;
        s_cur = s_nxt;
        if (wk + G < n_work && s_cur != BO_NONE) {
            n_cur = seg_cnt[s_cur];
            base_cur = seg_off[s_cur];
            nc_cur = n_clusters[s_cur];
        }
        s_nxt = wk + 2u * G < n_work ? seg_of(wk + 2u * G) : 0u;
#ifdef ECAL_PHASE_PROF
        unsigned long long bo_t__ = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();   // the previous segment's LDS is dead
        if (s == BO_NONE) continue;
        if (n == 0) {
            if (tid == 0) status[s] = 0;
            continue;
        }
        // which launch takes a segment follows from its size alone (the launches run side by side)
        const int tier = (n <= BO_CAP1 && nc <= BoLayout<BO_CAP1>::NCAP) ? 0
                         : (n <= BO_CAP2 && nc <= BoLayout<BO_CAP2>::NCAP) ? 1
                         : (n <= BO_CAP3 && nc <= BoLayout<BO_CAP3>::NCAP) ? 2 : 3;
        if (TIER > 0 ? (n > CAP || nc > BoLayout<CAP>::NCAP) : tier != 0) {   // (the later launches take what was listed for them)
            if constexpr (TIER == 0) {   // (tier 3: the global-scratch launch, cluster_order_big_kernel)
                if (tid == 0) defer_list[(size_t) (tier - 1) * S + atomicAdd(&defer_cnt[tier - 1], 1u)] = s;
            }
            continue;
        }
        if (tid < 8) red[tid] = 0;   // [0 .. 2]: block_any flags; [3]: failure; [4]: LDS list slots handed out; [5]: not floats
        bool inexact = false;
        uint32_t marked = 0;   // only_tied == 2: bit u = the caller's mark on point tid + u BO_T (read here, with the point)
        // (packed points, first launch: a segment sliced into 4-byte pixel words is read as those; the later launches' segments
        // have had their doubles written, ecal_cluster_order_sized)
        const bool packed = (TIER == 0 || (TIER == 2 && lean)) && xy16 && (seg_fmt[s] & 1u);
        for (uint32_t i = tid, u = 0; i < n; i += BO_T, u++) {
            if (only_tied == 2 && order[base + i] == -3) marked |= 1u << u;
            double2 p;
            if (packed) {
                const uint32_t w = xy16[base + i];
                p = make_double2((double) (int) (short) (w & 0xFFFFu), (double) (((int) w) >> 16));
            } else {
                p = reinterpret_cast<const double2 *>(xy)[base + i];
            }
            px[i] = (Coord) p.x;
            py[i] = (Coord) p.y;
            if (BoLayout<CAP>::CB == 4 && ((double) px[i] != p.x || (double) py[i] != p.y)) inexact = true;
            lab[i] = (int16_t) labels[base + i];   // (-1 or < n_clusters <= CAP)
            child[2 * i] = BO_NONE;
            child[2 * i + 1] = BO_NONE;
        }
        for (uint32_t c = tid; c <= nc; c += BO_T) qbase[c] = 0;
        for (uint32_t c = tid; c < nc; c += BO_T) seed[c] = BO_NONE;
        for (uint32_t w = tid; w < (n + 31u) / 32u; w += BO_T) inq[w] = 0;
        __syncthreads();
        if constexpr (BoLayout<CAP>::CB == 4) {   // coordinates that are not floats: the second launch keeps doubles
            if (inexact) red[5] = 1;
            __syncthreads();
            if (red[5]) {
                if (tid == 0) defer_list[(size_t) TIER * S + atomicAdd(&defer_cnt[TIER], 1u)] = s;   // (to the next launch)
                continue;
            }
        }
        BO_MARK(0);
        // members per cluster, the clusters' seeds (= smallest pid), the clusters' member lists (in queue[], for the tie test)
        for (uint32_t i = tid; i < n; i += BO_T) {
            if (lab[i] >= 0) {
                atomicAdd(&qbase[lab[i] + 1], 1u);
                atomicMin(&seed[lab[i]], i);
            }
        }
        __syncthreads();
        if (nc < 64u) {   // offsets of the clusters' queues: one wave scans the (usually few dozen) counts
            if (tid < 64u) {
                const uint32_t v = tid <= nc ? qbase[tid] : 0u;
                uint32_t inc = v;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t o = __shfl_up(inc, d, 64);
                    if ((int) tid >= d) inc += o;
                }
                if (tid <= nc) qbase[tid] = inc;
            }
        } else if (tid == 0) {
            uint32_t run = 0;
            for (uint32_t c = 0; c <= nc; c++) {
                run += qbase[c];
                qbase[c] = run;
            }
        }
        __syncthreads();
        // only_tied: the order is wanted for the clusters whose median has an equal-norm rival (what std::nth_element's
        // result depends on it for, CirclesEventFrame.cpp:136-147) — the test ecal_extract_batch_ordered_dev applies: the
        // member of rank size / 2 in the order (norm, pid) shares its norm with another member.  tie[c] lives in the upper
        // half of seed[]'s word... kept apart: cursor[] = child[] (not yet in use), tie flags = bit 31 of qbase's copy.
        bool seg_has_tie = !only_tied;
        if (only_tied == 2) {
            // the caller named the clusters: a -3 in `order` on the slot of one of their points (ecal_extract_batch_exact_dev's plain
            // pass marks the representative of every cluster whose median it found tied)
            uint32_t *const tie = child + CAP;            // [nc] (the tree is built later)
            for (uint32_t c = tid; c < nc; c += BO_T) tie[c] = 0;
            __syncthreads();
            for (uint32_t i = tid, u = 0; i < n; i += BO_T, u++)
                if (lab[i] >= 0 && ((marked >> u) & 1u)) tie[lab[i]] = 1;
            __syncthreads();
            bool any_tie = false;
            for (uint32_t c = tid; c < nc; c += BO_T) {
                if (tie[c]) any_tie = true;
                else seed[c] = BO_NONE - 1u;
            }
            uint32_t rr = 0;
            seg_has_tie = bo_block_any(any_tie, red, rr);
            __syncthreads();
            BO_MARK(1);
            if (!seg_has_tie) {
                for (uint32_t i = tid; i < n; i += BO_T) order[base + i] = lab[i] < 0 ? -1 : -2;
                if (tid == 0) status[s] = 0;
                BO_MARK(2);
                continue;
            }
            for (uint32_t c = tid; c < nc; c += BO_T) tie[c] = BO_NONE;   // child[] goes back to the tree
            if (tid < 4) red[tid] = 0;
            __syncthreads();
        } else if (only_tied) {
            uint32_t *const cursor = child;               // [nc] scatter cursors (the tree is built later)
            uint32_t *const tie = child + CAP;            // [nc]
            for (uint32_t c = tid; c < nc; c += BO_T) {
                cursor[c] = 0;
                tie[c] = 0;
            }
            __syncthreads();
            for (uint32_t i = tid; i < n; i += BO_T)
                if (lab[i] >= 0) queue[qbase[lab[i]] + atomicAdd(&cursor[lab[i]], 1u)] = (uint16_t) i;
            __syncthreads();
            for (uint32_t i = tid; i < n; i += BO_T) {
                if (lab[i] < 0) continue;
                const uint32_t c = (uint32_t) lab[i], qb = qbase[c], m = qbase[c + 1] - qb;
                // the norms' order (Vector2d::norm() = sqrt(x^2 + y^2)): where both squared norms are whole numbers (integer
                // pixels) they order exactly like their roots — the square root is injective on integers below 2^53 —, and the
                // two square roots per pair are only taken otherwise
                const double di = X(i) * X(i) + Y(i) * Y(i);
                const bool whole_i = di == floor(di);
                const double ki = __dsqrt_rn(di);
                uint32_t rank = 0, eq = 0;
                for (uint32_t t = 0; t < m; t++) {
                    const uint32_t j = queue[qb + t];
                    const double dj = X(j) * X(j) + Y(j) * Y(j);
                    bool lt, same;
                    if (whole_i && dj == floor(dj)) {
                        lt = dj < di;
                        same = dj == di;
                    } else {
                        const double kj = __dsqrt_rn(dj);
                        lt = kj < ki;
                        same = kj == ki;
                    }
                    rank += (lt || (same && j < i)) ? 1u : 0u;
                    eq += same ? 1u : 0u;
                }
                if (rank == m / 2u && eq > 1u) tie[c] = 1;
            }
            __syncthreads();
            bool any_tie = false;
            for (uint32_t c = tid; c < nc; c += BO_T) {
                if (tie[c]) any_tie = true;
                else seed[c] = BO_NONE - 1u;              // not wanted: no queue simulation, members get -2
            }
            uint32_t rr = 0;
            seg_has_tie = bo_block_any(any_tie, red, rr);
            __syncthreads();
            BO_MARK(1);
            if (!seg_has_tie) {
                for (uint32_t i = tid; i < n; i += BO_T) order[base + i] = lab[i] < 0 ? -1 : -2;
                if (tid == 0) status[s] = 0;
                BO_MARK(2);
                continue;
            }
            for (uint32_t i = tid; i < n; i += BO_T) {    // child[] goes back to the tree
                child[2 * i] = BO_NONE;
                child[2 * i + 1] = BO_NONE;
            }
            if (tid < 4) red[tid] = 0;
            __syncthreads();
        }
        // ---- 1. the insertion-order kd-tree (root = point 0 splits on x, children alternate: kdtree.cpp:127) ----
        // Sequential insertion puts at every empty child slot the FIRST (smallest pid) of the points whose descent reaches it.
        // Level by level: in round r every unplaced point stands at a node of depth r and bids (atomicMin of its pid) for the
        // child slot on its side — a slot of depth r + 1, which nobody can have filled before —; the winner is that node, the
        // others go on below it.  One barrier a round: the slots read after it (depth r + 1) are not the ones bid for next
        // (depth r + 2), and the "anybody still unplaced" vote rides on the same barrier (three rotating flag words).
        // (Round 4: a segment whose tree the pixel DBSCAN kernel has just built — the same insertions, the same rule — gets its
        // child links from there, 4 bytes per point, instead of replaying them: the build was a third of this kernel's time.)
        const bool have_tree = TIER == 0 && px_tree && px_tree_flag[s] == px_tree_epoch;
        if (have_tree) {
            for (uint32_t i = tid; i < n; i += BO_T) {
                const uint32_t w = px_tree[base + i], l = w & 0xFFFFu, r = w >> 16;
                child[2 * i] = l == 0xFFFFu ? BO_NONE : l;
                child[2 * i + 1] = r == 0xFFFFu ? BO_NONE : r;
            }
        }
        uint32_t cur[BO_PPT], dep[BO_PPT];
        bool placed[BO_PPT];
#pragma unroll
        for (int u = 0; u < BO_PPT; u++) {
            const uint32_t i = tid + u * BO_T;
            cur[u] = 0;
            dep[u] = 0;
            placed[u] = i == 0 || i >= n || have_tree;
        }
        for (uint32_t round = 0; !have_tree; round++) {
            uint32_t at[BO_PPT];
            bool active = false;
            if (tid == 0) red[(round + 1u) % 3u] = 0;
#pragma unroll
            for (int u = 0; u < BO_PPT; u++) {
                if (placed[u]) continue;
                const uint32_t i = tid + u * BO_T, c = cur[u];
                const bool dy = dep[u] & 1u;
                const double v = dy ? Y(i) : X(i), cv = dy ? Y(c) : X(c);
                at[u] = 2 * c + (v < cv ? 0u : 1u);   // left iff pos[dir] < node->pos[dir] (kdtree.cpp:128-131)
                atomicMin(&child[at[u]], i);
                active = true;
            }
            if (active) red[round % 3u] = 1;
            __syncthreads();
            if (red[round % 3u] == 0) break;
#pragma unroll
            for (int u = 0; u < BO_PPT; u++) {
                if (placed[u]) continue;
                const uint32_t w = child[at[u]];
                if (w == tid + u * BO_T) {
                    placed[u] = true;
                } else {
                    cur[u] = w;
                    dep[u]++;
                }
            }
        }
        __syncthreads();
        BO_MARK(3);
        // ---- 2. one range query per core point: find_nearest's visiting order (kdtree.cpp:148-179) ----
        bool fail = false;
#pragma unroll 1
        for (int u = 0; u < BO_PPT; u++) {
            const uint32_t i = tid + u * BO_T;
            if (i >= n || lab[i] < 0 || seed[lab[i]] == BO_NONE - 1u) continue;
            const double qx = X(i), qy = Y(i);
            const uint32_t slot = atomicAdd(&red[4], 1u);
            const bool in_lds = slot < BO_POOL;
            slotmap[i] = in_lds ? (uint16_t) slot : (uint16_t) 0xFFFFu;
            uint16_t *out = in_lds ? pool + slot * BO_MAXN : my_lists + (size_t) i * BO_MAXN;
            uint32_t cnt = 0, sp = 0;
            // the pending far subtrees: a stack of (node | dir << 15) entries.  First tier: kept in REGISTERS as a 192-bit shift
            // register of twelve 16-bit entries (an indexed array would live in scratch memory: 400 bytes per lane); a traversal
            // with more pending subtrees sends the segment to the global-scratch launch.  Later tiers: an array.
            constexpr bool REG_STACK = TIER == 0;
            constexpr uint32_t STACK_CAP = REG_STACK ? 12u : BO_STACK;
            unsigned long long s0 = 0, s1 = 0, s2 = 0;
            uint32_t stk[REG_STACK ? 1 : BO_STACK];
            uint32_t node = 0, dir = 0;
            for (;;) {
                while (node != BO_NONE) {
                    const double ddx = X(node) - qx, ddy = Y(node) - qy;
                    double d2 = 0;
                    d2 += ddx * ddx;   // dist_sq += SQ(node->pos[i] - pos[i]), i ascending
                    d2 += ddy * ddy;
                    if (d2 <= eps2 && node != i) {   // regionQuery drops the query point itself (dbscan.h:218)
                        if (cnt < BO_MAXN) out[cnt] = (uint16_t) node;
                        cnt++;
                    }
                    const double dx = dir ? (qy - Y(node)) : (qx - X(node));
                    const uint32_t l = child[2 * node], r = child[2 * node + 1];
                    const uint32_t nearc = dx <= 0.0 ? l : r, farc = dx <= 0.0 ? r : l;
                    if (fabs(dx) < eps && farc != BO_NONE) {
                        if (sp < STACK_CAP) {
                            const uint32_t e = farc | ((dir ^ 1u) << 15);
                            if constexpr (REG_STACK) {
                                s2 = (s2 << 16) | (s1 >> 48);
                                s1 = (s1 << 16) | (s0 >> 48);
                                s0 = (s0 << 16) | e;
                            } else {
                                stk[sp] = e;
                            }
                        }
                        sp++;
                    }
                    node = nearc;
                    dir ^= 1u;
                }
                if (sp == 0 || sp > STACK_CAP) break;
                sp--;
                uint32_t e;
                if constexpr (REG_STACK) {
                    e = (uint32_t) (s0 & 0xFFFFu);
                    s0 = (s0 >> 16) | (s1 << 48);
                    s1 = (s1 >> 16) | (s2 << 48);
                    s2 >>= 16;
                } else {
                    e = stk[sp];
                }
                node = e & 0x7FFFu;
                dir = e >> 15;
            }
            if (cnt > BO_MAXN || sp > STACK_CAP) fail = true;
            if (in_lds) pool_cnt[slot] = (uint16_t) (cnt > BO_MAXN ? BO_MAXN : cnt);
            else my_cnt[i] = (uint8_t) (cnt > BO_MAXN ? BO_MAXN : cnt);
        }
        if (fail) red[3] = 1;
        __threadfence_block();
        __syncthreads();
        if (red[3]) {   // more hits in one eps-ball or more pending subtrees than this tier's fixed sizes: the global-scratch launch
            if (tid == 0) defer_list[(size_t) 2 * S + atomicAdd(&defer_cnt[2], 1u)] = s;
            continue;   // (the caller's marks in `order` are still in place: nothing has been written yet)
        }
        BO_MARK(4);
        // ---- 3. expandCluster's queue, one thread per cluster (dbscan.h:229-265) ----
        for (uint32_t i = tid; i < n; i += BO_T)
            if (lab[i] < 0) order[base + i] = -1;
            else if (seed[lab[i]] == BO_NONE - 1u) order[base + i] = -2;
        // (a thread per small cluster; a wave per large one, its lanes taking one neighbour of the popped point each: the pops
        // of one cluster are a dependent chain, and in a lock-step pass of the adaptive policy that chain is the kernel's latency)
        // Round 4: when FEW clusters are wanted — the exact extraction marks one or two per segment — every one of them gets a
        // wave, whatever its size: a thread walks a popped point's ~10 neighbours one after the other (each a chain of four LDS
        // operations), a wave takes them side by side; with one or two threads at work and 254 idle that walk was 43 % of
        // the kernel (tools/phase_prof.py).
        {
            uint32_t mine = 0;
            for (uint32_t c = tid; c < nc; c += BO_T) mine += seed[c] < BO_NONE - 1u ? 1u : 0u;
            if (mine) atomicAdd(&red[6], mine);   // (red[6] = 0 since the segment's start)
            __syncthreads();
        }
        const bool few_wanted = red[6] <= 2u * (BO_T / 64u);
        for (uint32_t c = tid; c < nc; c += BO_T) {
            const uint32_t qb = qbase[c], sd = seed[c];
            if (sd >= BO_NONE - 1u || qbase[c + 1] - qb >= BO_WAVE_MIN || few_wanted) continue;   // not wanted (or, impossible, without members)
            uint32_t head = 0, tail = 1;
            queue[qb] = (uint16_t) sd;
            atomicOr(&inq[sd >> 5], 1u << (sd & 31u));
            while (head < tail) {
                const uint32_t q = queue[qb + head];
                order[base + q] = (int32_t) head;
                head++;
                const uint32_t sl = slotmap[q];
                const uint32_t m = sl != 0xFFFFu ? (uint32_t) pool_cnt[sl] : (uint32_t) my_cnt[q];
                const uint16_t *lst = sl != 0xFFFFu ? pool + sl * BO_MAXN : my_lists + (size_t) q * BO_MAXN;
                for (uint32_t k = m; k-- > 0;) {   // the result list = hits in REVERSE visiting order (rlist_insert at the head)
                    const uint32_t j = lst[k];
                    if ((uint32_t) lab[j] != c) continue;   // not a core point of this cluster: never expands, never listed
                    const uint32_t bit = 1u << (j & 31u);
                    if (atomicOr(&inq[j >> 5], bit) & bit) continue;   // already in the border set
                    queue[qb + tail] = (uint16_t) j;
                    tail++;
                }
            }
        }
        {
            const uint32_t lane = tid & 63u;
            for (uint32_t c = tid >> 6; c < nc; c += BO_T / 64u) {   // (uniform in the wave)
                const uint32_t qb = qbase[c], sd = seed[c];
                if (sd >= BO_NONE - 1u || (qbase[c + 1] - qb < BO_WAVE_MIN && !few_wanted)) continue;
                uint32_t head = 0, tail = 1;
                if (lane == 0) {
                    queue[qb] = (uint16_t) sd;
                    atomicOr(&inq[sd >> 5], 1u << (sd & 31u));
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                while (head < tail) {
                    const uint32_t q = reinterpret_cast<volatile uint16_t *>(queue)[qb + head];
                    if (lane == 0) order[base + q] = (int32_t) head;
                    head++;
                    const uint32_t sl = slotmap[q];
                    const uint32_t m = sl != 0xFFFFu ? (uint32_t) pool_cnt[sl] : (uint32_t) my_cnt[q];
                    const uint16_t *lst = sl != 0xFFFFu ? pool + sl * BO_MAXN : my_lists + (size_t) q * BO_MAXN;
                    // lane l takes the l-th entry from the END of the list (BO_MAXN = the wave's width); the hits of one query
                    // are distinct points, so the lanes' bits in inq[] never coincide
                    bool take = false;
                    uint32_t j = 0;
                    if (lane < m) {
                        j = lst[m - 1u - lane];
                        if ((uint32_t) lab[j] == c) {
                            const uint32_t bit = 1u << (j & 31u);
                            take = !(atomicOr(&inq[j >> 5], bit) & bit);
                        }
                    }
                    const unsigned long long mask = __ballot(take);
                    if (take) queue[qb + tail + __popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t) j;
                    tail += (uint32_t) __popcll(mask);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (tid == 0) status[s] = 0;
        __syncthreads();
        BO_MARK(5);
    }
}

// ---- the global-scratch launch: segments of any size -------------------------------------------------------------------------
// What the LDS tiers leave over — more than 4096 points or 2048 clusters (BASELINE configs[0]: 100 k events in ONE window), more
// than BO_MAXN hits in one eps-ball, more than BO_STACK pending subtrees — is taken here with the same three steps on a workspace
// in global memory, without fixed list or stack sizes:
//   * the tree is rebuilt level by level as above (all unplaced points of round r stand at depth r) and keeps PARENT links;
//   * a range query walks the tree without a stack, and directly in the order of the reference's RESULT list: that list is
//     the reverse of find_nearest's visiting order (node, near subtree, far subtree; kdtree.cpp:148-179,469-486), i.e.
//     [far subtree reversed, if fabs(dx) < range] [near subtree reversed] [node] — a post-order walk that needs only "where did
//     I come from" (parent, far child or near child) to go on;
//   * every wanted core point's query runs twice, in parallel: once to count its hits, once (after a scan of the counts) to
//     write them into the workgroup's arena; then a wave per wanted cluster simulates expandCluster's queue.
// Latency does not matter here (the segments are rare and large); what matters is that no size is refused: status 1 is left
// for segments beyond the workspace (BO_BIG_W points) or whose hit lists overflow the arena.
constexpr int BO_TB = 1024;
constexpr uint32_t BO_BIG_GRID = 2;
constexpr uint32_t BO_BIG_W = 1u << 20;        // points per segment the workspace is sized for (at most; the caller's n_points if smaller)
constexpr uint32_t BO_BIG_ARENA = 1u << 24;    // hit-list entries per workgroup (at most; 64 per point of the workspace if smaller)
// workspace words per point of capacity: child 2 | parent | cur | off (+1) | queue | qbase (+1) | seed | tie | inq (1/32)
__host__ __device__ constexpr size_t bo_big_words(size_t W) { return 10 * W + W / 32 + 16; }

__device__ __forceinline__ uint32_t bo_ld(const uint32_t *p) {   // a word other waves wrote with atomics: read at L2
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One range query in result-list order.  emit(j) is called for every hit (d2 <= eps^2, the query point itself excluded,
// dbscan.h:218) in the order the reference's list holds them.
template <typename F>
__device__ __forceinline__ void bo_query_reversed(const double2 *__restrict__ pts, const uint32_t *child, const uint32_t *parent,
                                                  uint32_t q, double eps, double eps2, F emit) {
    const double2 pq = pts[q];
    uint32_t node = 0, from = BO_NONE, dir = 0;   // from: BO_NONE = arrived from the parent, else the child just finished
    for (;;) {
        const double2 pn = pts[node];
        const double dx = dir ? (pq.y - pn.y) : (pq.x - pn.x);
        const uint32_t l = bo_ld(&child[2 * node]), r = bo_ld(&child[2 * node + 1]);   // (written by atomics: read at L2)
        const uint32_t nearc = dx <= 0.0 ? l : r, farc = dx <= 0.0 ? r : l;
        uint32_t next = BO_NONE;
        if (from == BO_NONE) {                       // first arrival: the far subtree comes first in the list, then the near one
            if (fabs(dx) < eps && farc != BO_NONE) next = farc;
            else if (nearc != BO_NONE) next = nearc;
        } else if (from == farc && nearc != BO_NONE) {
            next = nearc;                            // (from == nearc: both subtrees are done)
        }
        if (next != BO_NONE) {
            node = next;
            from = BO_NONE;
            dir ^= 1u;
            continue;
        }
        {   // the node itself closes its subtree's part of the list
            const double ddx = pn.x - pq.x, ddy = pn.y - pq.y;
            double d2 = 0;
            d2 += ddx * ddx;   // dist_sq += SQ(node->pos[i] - pos[i]), i ascending (kdtree.cpp:155-159)
            d2 += ddy * ddy;
            if (d2 <= eps2 && node != q) emit(node);
        }
        const uint32_t up = parent[node];
        if (up == BO_NONE) return;
        from = node;
        node = up;
        dir ^= 1u;
    }
}

__global__ __launch_bounds__(BO_TB) void cluster_order_big_kernel(const double *__restrict__ xy, const uint32_t *__restrict__ seg_off,
                                                                  const uint32_t *__restrict__ seg_cnt, uint32_t S, double eps,
                                                                  const int32_t *__restrict__ labels, const uint32_t *__restrict__ n_clusters,
                                                                  int32_t *__restrict__ order, uint32_t *__restrict__ status, int only_tied,
                                                                  const uint32_t *__restrict__ defer_list, const uint32_t *__restrict__ defer_cnt,
                                                                  uint32_t *__restrict__ ws, uint32_t W, uint32_t *__restrict__ arena_all,
                                                                  uint32_t arena_cap, const uint32_t *__restrict__ xy16, uint32_t *__restrict__ seg_fmt,
                                                                  uint32_t *seen) {
    __shared__ unsigned long long red[BO_TB / 64 + 2];
    __shared__ uint32_t flag[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t *const w0 = ws + (size_t) blockIdx.x * bo_big_words(W);
    uint32_t *const child = w0, *const parent = child + 2 * (size_t) W, *const cur = parent + W, *const off = cur + W,
                    *const queue = off + W + 1, *const qbase = queue + W, *const seed = qbase + W + 1, *const tie = seed + W,
                    *const inq = tie + W;
    uint32_t *const arena = arena_all + (size_t) blockIdx.x * arena_cap;
    const double eps2 = eps * eps;
    const uint32_t n_work = defer_cnt[2];
    if (seen && blockIdx.x == 0 && tid < 3) seen[tid] = defer_cnt[tid];   // (the stage's last launch: what its lists held)
    for (uint32_t wk = blockIdx.x; wk < n_work; wk += gridDim.x) {
        const uint32_t s = defer_list[(size_t) 2 * S + wk];
        const uint32_t n = seg_cnt[s], base = seg_off[s], nc = n_clusters[s];
        const double2 *const pts = reinterpret_cast<const double2 *>(xy) + base;
        const int32_t *const lab = labels + base;
        __syncthreads();
        if (xy16 && seg_fmt[s] == 1u) {   // (lean form: a segment that exists packed only gets its doubles here; fmt 1 -> 3)
            double2 *out = const_cast<double2 *>(pts);
            for (uint32_t i = tid; i < n; i += BO_TB) {
                const uint32_t v = xy16[base + i];
                out[i] = make_double2((double) (int) (short) (v & 0xFFFFu), (double) (((int) v) >> 16));
            }
            __threadfence();
            __syncthreads();
            if (tid == 0) seg_fmt[s] = 3u;
        }
        auto refuse = [&]() {
            for (uint32_t i = tid; i < n; i += BO_TB) order[base + i] = -1;
            if (tid == 0) status[s] = 1;
        };
        if (n > W || nc > W) {
            refuse();
            continue;
        }
        if (tid < 4) flag[tid] = 0;
        for (uint32_t i = tid; i < n; i += BO_TB) {
            // (only_tied == 2) the caller's marks, read before anything is written to `order`
            tie[i] = (only_tied == 2 && order[base + i] == -3) ? 1u : 0u;
            child[2 * i] = BO_NONE;
            child[2 * i + 1] = BO_NONE;
            parent[i] = BO_NONE;
            cur[i] = i == 0 ? BO_NONE : 0u;   // the node the point stands at; BO_NONE once it is placed (the root is)
        }
        for (uint32_t c = tid; c <= nc; c += BO_TB) qbase[c] = 0;
        for (uint32_t c = tid; c < nc; c += BO_TB) seed[c] = BO_NONE;
        for (uint32_t w = tid; w < n / 32u + 1u; w += BO_TB) inq[w] = 0;
        __threadfence();
        __syncthreads();
        // members per cluster, seeds; the marks move from the points to their clusters (queue[] = scratch for them)
        for (uint32_t i = tid; i < n; i += BO_TB)
            if (lab[i] >= 0) {
                atomicAdd(&qbase[lab[i] + 1], 1u);
                atomicMin(&seed[lab[i]], i);
            }
        for (uint32_t c = tid; c < nc; c += BO_TB) queue[c] = 0;
        __threadfence();
        __syncthreads();
        if (only_tied == 2)
            for (uint32_t i = tid; i < n; i += BO_TB)
                if (lab[i] >= 0 && tie[i]) queue[lab[i]] = 1;   // (every writer: the same value)
        {   // offsets of the clusters' queues: inclusive scan of the counts, a contiguous chunk per thread
            const uint32_t per = (nc + 1u + BO_TB - 1u) / BO_TB, c0 = tid * per;
            uint32_t sum = 0;
            for (uint32_t c = c0; c < c0 + per && c <= nc; c++) sum += bo_ld(&qbase[c]);
            uint32_t ex, d0, tot, d1;
            block_exscan_pair<BO_TB>(sum, 0u, red, &ex, &d0, &tot, &d1);
            for (uint32_t c = c0; c < c0 + per && c <= nc; c++) {
                ex += bo_ld(&qbase[c]);
                qbase[c] = ex;
            }
        }
        __threadfence();
        __syncthreads();
        // which clusters are wanted (seed[c] = BO_NONE - 1: not wanted, its members get -2)
        if (only_tied == 2) {
            bool any = false;
            for (uint32_t c = tid; c < nc; c += BO_TB) {
                if (queue[c]) any = true;
                else seed[c] = BO_NONE - 1u;
            }
            if (any) flag[0] = 1;
        } else if (only_tied) {
            // own test (as the LDS tiers): the member of rank size / 2 in the order (norm, pid) shares its norm with another
            for (uint32_t c = tid; c < nc; c += BO_TB) tie[c] = 0;   // (the marks are not in use in this mode)
            for (uint32_t c = tid; c < nc; c += BO_TB) cur[c] = 0;   // scatter cursors (cur[] is set up again below)
            __threadfence();
            __syncthreads();
            uint32_t *const members = arena;   // [n] (the arena is free until the hit lists are written)
            if (n <= arena_cap) {
                for (uint32_t i = tid; i < n; i += BO_TB)
                    if (lab[i] >= 0) members[qbase[lab[i]] + atomicAdd(&cur[lab[i]], 1u)] = i;
                __threadfence();
                __syncthreads();
                for (uint32_t i = tid; i < n; i += BO_TB) {
                    if (lab[i] < 0) continue;
                    const uint32_t c = (uint32_t) lab[i], qb = qbase[c], m = qbase[c + 1] - qb;
                    const double di = pts[i].x * pts[i].x + pts[i].y * pts[i].y;
                    const bool whole_i = di == floor(di);
                    const double ki = __dsqrt_rn(di);
                    uint32_t rank = 0, eq = 0;
                    for (uint32_t t = 0; t < m; t++) {
                        const uint32_t j = members[qb + t];
                        const double dj = pts[j].x * pts[j].x + pts[j].y * pts[j].y;
                        bool lt, same;
                        if (whole_i && dj == floor(dj)) {
                            lt = dj < di;
                            same = dj == di;
                        } else {
                            const double kj = __dsqrt_rn(dj);
                            lt = kj < ki;
                            same = kj == ki;
                        }
                        rank += (lt || (same && j < i)) ? 1u : 0u;
                        eq += same ? 1u : 0u;
                    }
                    if (rank == m / 2u && eq > 1u) tie[c] = 1;
                }
            } else if (tid == 0) {
                flag[3] = 1;
            }
            __threadfence();
            __syncthreads();
            bool any = false;
            for (uint32_t c = tid; c < nc; c += BO_TB) {
                if (tie[c]) any = true;
                else seed[c] = BO_NONE - 1u;
            }
            if (any) flag[0] = 1;
            __syncthreads();
            for (uint32_t i = tid; i < n; i += BO_TB) cur[i] = i == 0 ? BO_NONE : 0u;
        } else if (tid == 0) {
            flag[0] = 1;
        }
        __threadfence();
        __syncthreads();
        if (flag[3]) {
            refuse();
            continue;
        }
        if (!flag[0]) {   // no wanted cluster in this segment
            for (uint32_t i = tid; i < n; i += BO_TB) order[base + i] = lab[i] < 0 ? -1 : -2;
            if (tid == 0) status[s] = 0;
            continue;
        }
        // ---- 1. the insertion-order kd-tree, level by level (kdtree.cpp:106-146; see cluster_order_kernel) ----
        for (uint32_t round = 0;; round++) {
            const bool dy = round & 1u;
            bool active = false;
            if (tid == 0) flag[1 + ((round + 1u) & 1u)] = 0;
            for (uint32_t i = tid; i < n; i += BO_TB) {
                const uint32_t c = cur[i];
                if (c == BO_NONE) continue;
                const double v = dy ? pts[i].y : pts[i].x, cv = dy ? pts[c].y : pts[c].x;
                atomicMin(&child[2 * c + (v < cv ? 0u : 1u)], i);   // left iff pos[dir] < node->pos[dir] (kdtree.cpp:128-131)
                active = true;
            }
            if (active) flag[1 + (round & 1u)] = 1;
            __threadfence();
            __syncthreads();
            if (flag[1 + (round & 1u)] == 0) break;
            for (uint32_t i = tid; i < n; i += BO_TB) {
                const uint32_t c = cur[i];
                if (c == BO_NONE) continue;
                const double v = dy ? pts[i].y : pts[i].x, cv = dy ? pts[c].y : pts[c].x;
                const uint32_t w = bo_ld(&child[2 * c + (v < cv ? 0u : 1u)]);
                if (w == i) {
                    parent[i] = c;
                    cur[i] = BO_NONE;
                } else {
                    cur[i] = w;
                }
            }
            __syncthreads();   // (flag[] of this round is cleared by thread 0 at the top of the round after the next)
        }
        __threadfence();
        __syncthreads();
        // ---- 2. the range queries of the wanted clusters' core points: count, scan, fill ----
        // (child[] was written by atomics: the walks read it at L2; parent[] by plain stores of this workgroup)
        auto wanted = [&](uint32_t i) { return lab[i] >= 0 && bo_ld(&seed[lab[i]]) != BO_NONE - 1u; };
        for (uint32_t i = tid; i < n; i += BO_TB) {
            uint32_t cnt = 0;
            if (wanted(i)) bo_query_reversed(pts, child, parent, i, eps, eps2, [&](uint32_t) { cnt++; });
            cur[i] = cnt;
        }
        __syncthreads();
        {
            const uint32_t per = (n + BO_TB - 1u) / BO_TB, i0 = tid * per;
            unsigned long long sum = 0;
            for (uint32_t i = i0; i < i0 + per && i < n; i++) sum += cur[i];
            // (two 32-bit lanes of the pair scan carry the low and high half: totals beyond 2^32 must be seen as overflow)
            uint32_t ex, exh, tot, toth;
            block_exscan_pair<BO_TB>((uint32_t) (sum & 0xFFFFFu), (uint32_t) (sum >> 20), red, &ex, &exh, &tot, &toth);
            const unsigned long long total = ((unsigned long long) toth << 20) + tot;
            if (total > arena_cap) {
                if (tid == 0) flag[3] = 1;
            } else {
                uint32_t run = (uint32_t) (((unsigned long long) exh << 20) + ex);
                for (uint32_t i = i0; i < i0 + per && i < n; i++) {
                    off[i] = run;
                    run += cur[i];
                }
                if (tid == 0) off[n] = (uint32_t) total;
            }
        }
        __syncthreads();
        if (flag[3]) {
            refuse();
            continue;
        }
        for (uint32_t i = tid; i < n; i += BO_TB) {
            if (!wanted(i)) continue;
            uint32_t *out = arena + off[i];
            bo_query_reversed(pts, child, parent, i, eps, eps2, [&](uint32_t j) { *out++ = j; });
        }
        for (uint32_t i = tid; i < n; i += BO_TB)
            if (lab[i] < 0) order[base + i] = -1;
            else if (bo_ld(&seed[lab[i]]) == BO_NONE - 1u) order[base + i] = -2;
        __threadfence();
        __syncthreads();
        // ---- 3. expandCluster's queue (dbscan.h:229-265), a wave per wanted cluster, a lane per neighbour of the popped point ----
        for (uint32_t c = wave; c < nc; c += BO_TB / 64u) {   // (uniform in the wave)
            const uint32_t qb = qbase[c], sd = bo_ld(&seed[c]);
            if (sd >= BO_NONE - 1u) continue;
            uint32_t head = 0, tail = 1;
            if (lane == 0) {
                queue[qb] = sd;
                atomicOr(&inq[sd >> 5], 1u << (sd & 31u));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            while (head < tail) {
                const uint32_t q = reinterpret_cast<volatile uint32_t *>(queue)[qb + head];
                if (lane == 0) order[base + q] = (int32_t) head;
                head++;
                const uint32_t lo = off[q], m = off[q + 1] - lo;   // (a wanted cluster's members all have their lists)
                for (uint32_t k0 = 0; k0 < m; k0 += 64u) {
                    bool take = false;
                    uint32_t j = 0;
                    if (k0 + lane < m) {
                        j = arena[lo + k0 + lane];
                        if ((uint32_t) lab[j] == c) {   // a core point of this cluster (others never expand, never get listed)
                            const uint32_t bit = 1u << (j & 31u);
                            take = !(atomicOr(&inq[j >> 5], bit) & bit);   // not yet in the border set
                        }
                    }
                    const unsigned long long mask = __ballot(take);
                    if (take) queue[qb + tail + __popcll(mask & ((1ull << lane) - 1ull))] = j;
                    tail += (uint32_t) __popcll(mask);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (tid == 0) status[s] = 0;
    }
}

}  // namespace ecal

using namespace ecal;

extern "C" int ecal_cluster_order_dev(ecal_ctx *ctx, const double *d_xy, const uint32_t *d_seg_off, const uint32_t *d_seg_cnt,
                                      uint32_t S, double eps, const int32_t *d_labels, const uint32_t *d_n_clusters,
                                      int32_t *d_order, uint32_t *d_status, int only_tied_medians, void *stream) {
    return ecal_cluster_order_sized(ctx, d_xy, d_seg_off, d_seg_cnt, S, 0xFFFFFFFFu, eps, d_labels, d_n_clusters, d_order, d_status,
                                    only_tied_medians, nullptr, nullptr, stream);
}

// d_win_list != NULL: only the segments 2 w and 2 w + 1 of the windows w = d_win_list[0 .. *d_win_count) (S = 2 x windows in all)
extern "C" int ecal_cluster_order_list_dev(ecal_ctx *ctx, const double *d_xy, const uint32_t *d_seg_off, const uint32_t *d_seg_cnt,
                                           uint32_t S, double eps, const int32_t *d_labels, const uint32_t *d_n_clusters, int32_t *d_order,
                                           uint32_t *d_status, int only_tied_medians, const uint32_t *d_win_list,
                                           const uint32_t *d_win_count, void *stream) {
    return ecal_cluster_order_sized(ctx, d_xy, d_seg_off, d_seg_cnt, S, 0xFFFFFFFFu, eps, d_labels, d_n_clusters, d_order, d_status,
                                    only_tied_medians, d_win_list, d_win_count, stream);
}

// n_points: an upper bound on any segment's size (the callers inside the library know their point capacity) — it sizes the
// global-scratch launch's workspace
int ecal_cluster_order_sized(ecal_ctx *ctx, const double *d_xy, const uint32_t *d_seg_off, const uint32_t *d_seg_cnt, uint32_t S,
                             uint32_t n_points, double eps, const int32_t *d_labels, const uint32_t *d_n_clusters, int32_t *d_order,
                             uint32_t *d_status, int only_tied_medians, const uint32_t *d_win_list, const uint32_t *d_win_count,
                             void *stream, const ecal_packed_points *pk) {
    const ecal_range range__(ctx, "ecal_cluster_order");
    if (!ctx) return ECAL_ERR_INVALID;
    if (S == 0) return ECAL_OK;
    if (pk && (!pk->d_xy16 || !pk->d_seg_fmt)) pk = nullptr;
    const uint32_t *const xy16 = pk ? pk->d_xy16 : nullptr, *const sfmt = pk ? pk->d_seg_fmt : nullptr;
    if (!d_xy || !d_seg_off || !d_seg_cnt || !d_labels || !d_n_clusters || !d_order || !d_status) {
        ctx->last_error = "null pointer";
        return ECAL_ERR_INVALID;
    }
    if (!(eps > 0.0) || !(eps < 1.0e300)) {
        ctx->last_error = "eps must be a positive finite number";
        return ECAL_ERR_INVALID;
    }
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t) stream;
    const uint32_t grid1 = std::min<uint32_t>(S, BO_WG1 * ctx->n_cu), grid2 = std::min<uint32_t>(S, BO_WG2 * ctx->n_cu),
                   grid3 = std::min<uint32_t>(S, BO_WG3 * ctx->n_cu);
    int rc;
    // global lists: a slice of CAP rows per workgroup; the launches follow each other on the stream and share the buffer
    const size_t rows = std::max({(size_t) grid1 * BO_CAP1, (size_t) grid2 * BO_CAP2, (size_t) grid3 * BO_CAP3});
    if ((rc = ecal_ensure(ctx, ctx->bfs_lists, rows * (BO_MAXN * sizeof(uint16_t) + 1)))) return rc;
    uint16_t *lists = (uint16_t *) ctx->bfs_lists.ptr;
    uint8_t *cnt = (uint8_t *) (lists + rows * BO_MAXN);
    // the segments the first launch leaves to the later ones: two counters, two lists of up to S entries
    if ((rc = ecal_ensure(ctx, ctx->bfs_defer, (3 * (size_t) S + 4) * sizeof(uint32_t)))) return rc;
    uint32_t *dcnt = (uint32_t *) ctx->bfs_defer.ptr, *dlist = dcnt + 4;
    if (uint32_t *z = ecal_zero_words(ctx, st, 3)) dcnt = z;
    else ECAL_HIP_TRY(ctx, hipMemsetAsync(dcnt, 0, 3 * sizeof(uint32_t), st));
    // the global-scratch launch's workspace and hit-list arena, a slice per workgroup
    const uint32_t bigW = std::max<uint32_t>(std::min<uint32_t>(n_points, BO_BIG_W), 64u);
    uint32_t big_arena = (uint32_t) std::min<uint64_t>(BO_BIG_ARENA, std::max<uint64_t>(64ull * bigW, std::min<uint64_t>((uint64_t) bigW * bigW, 1ull << 22)));
    if (ctx->sw.bo_big_arena)   // (ECAL_BO_BIG_ARENA; tests: an arena too small for the segment = the flagged fallback, status 1)
        big_arena = (uint32_t) std::min<uint64_t>(big_arena, std::max<uint64_t>(64, ctx->sw.bo_big_arena));
    const uint32_t grid_big = std::min<uint32_t>(S, BO_BIG_GRID);
    if ((rc = ecal_ensure(ctx, ctx->bfs_big, (size_t) grid_big * (bo_big_words(bigW) + big_arena) * sizeof(uint32_t)))) return rc;
    uint32_t *big_ws = (uint32_t *) ctx->bfs_big.ptr, *big_hits = big_ws + (size_t) grid_big * bo_big_words(bigW);
    if (!ctx->bfs_attr_set) {
        ECAL_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&cluster_order_kernel<BO_CAP1, BO_T1, 0>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int) BoLayout<BO_CAP1>::bytes));
        ECAL_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&cluster_order_kernel<BO_CAP2, BO_T1B, 1>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int) BoLayout<BO_CAP2>::bytes));
        ECAL_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&cluster_order_kernel<BO_CAP3, BO_T2, 2>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int) BoLayout<BO_CAP3>::bytes));
        ctx->bfs_attr_set = true;
    }
    // (The three launches are independent — a segment's size names its launch —, but starting the later ones on streams of
    // their own beside the first cost more in cross-stream waits than it saved: 0.74 against 0.65 ms per lock-step pass.)
    // the trees of the pixel DBSCAN kernel, when these labels and segments are the ones its last call on this context produced
    const bool trees = ctx->px_tree_labels && ctx->px_tree_labels == (const void *) d_labels && ctx->px_tree_seg_off == (const void *) d_seg_off &&
                       ctx->px_tree_S == S;
    hipLaunchKernelGGL((cluster_order_kernel<BO_CAP1, BO_T1, 0>), dim3(grid1), dim3(BO_T1), BoLayout<BO_CAP1>::bytes, st, d_xy, d_seg_off,
                       d_seg_cnt, S, eps, d_labels, d_n_clusters, d_order, d_status, lists, cnt, only_tied_medians, d_win_list, d_win_count, dlist, dcnt,
                       xy16, sfmt, 0, trees ? (const uint32_t *) ctx->px_tree.ptr : nullptr,
                       trees ? (const uint32_t *) ctx->px_tree_flag.ptr : nullptr, ctx->px_tree_epoch);
    // lean (ecal_ctx::tail_seen: the first launch listed nothing when this stage last ran): the third launch takes the second's
    // list with its own, reading packed segments as they are, and the global-scratch launch unpacks what it is given — two
    // launches behind the first instead of six
    const bool lean = ecal_tail_lean(ctx, ECAL_TAIL_ORDER, 3);
    if (pk && !lean) {   // the later launches read doubles: the segments the first one listed for them are unpacked
        int rcu;
        for (int k = 0; k < 3; k++)
            if ((rcu = ecal_unpack_listed(ctx, pk, dlist + (size_t) k * S, dcnt + k, S, d_seg_off, d_seg_cnt, const_cast<double *>(d_xy), 0, st)))
                return rcu;
    }
    if (!lean)
    hipLaunchKernelGGL((cluster_order_kernel<BO_CAP2, BO_T1B, 1>), dim3(grid2), dim3(BO_T1B), BoLayout<BO_CAP2>::bytes, st, d_xy, d_seg_off,
                       d_seg_cnt, S, eps, d_labels, d_n_clusters, d_order, d_status, lists, cnt, only_tied_medians, d_win_list, d_win_count, dlist, dcnt,
                       nullptr, nullptr, 0);
    hipLaunchKernelGGL((cluster_order_kernel<BO_CAP3, BO_T2, 2>), dim3(grid3), dim3(BO_T2), BoLayout<BO_CAP3>::bytes, st, d_xy, d_seg_off,
                       d_seg_cnt, S, eps, d_labels, d_n_clusters, d_order, d_status, lists, cnt, only_tied_medians, d_win_list, d_win_count, dlist, dcnt,
                       lean ? xy16 : nullptr, lean ? sfmt : nullptr, lean ? 1 : 0);
    hipLaunchKernelGGL(cluster_order_big_kernel, dim3(grid_big), dim3(BO_TB), 0, st, d_xy, d_seg_off, d_seg_cnt, S, eps, d_labels, d_n_clusters,
                       d_order, d_status, only_tied_medians, (const uint32_t *) dlist, (const uint32_t *) dcnt, big_ws, bigW, big_hits, big_arena,
                       lean ? xy16 : nullptr, lean && pk ? pk->d_seg_fmt : nullptr, ctx->tail_seen_dev ? ctx->tail_seen_dev + ECAL_TAIL_ORDER : nullptr);
    ECAL_HIP_TRY(ctx, hipGetLastError());
    return ECAL_OK;
}

/* host-buffer form: slices [slice_off[s], slice_off[s + 1]) of xy as ecal_dbscan_batch takes them */
extern "C" int ecal_cluster_order(ecal_ctx *ctx, const double *xy, const uint32_t *slice_off, uint32_t S, double eps,
                                  const int32_t *labels, const uint32_t *n_clusters, int32_t *order, uint32_t *status) {
    if (!ctx) return ECAL_ERR_INVALID;
    if (S == 0) return ECAL_OK;
    if (!slice_off || !n_clusters || !status) {
        ctx->last_error = "null pointer";
        return ECAL_ERR_INVALID;
    }
    for (uint32_t s = 0; s < S; s++)
        if (slice_off[s + 1] < slice_off[s]) {
            ctx->last_error = "slice_off must be non-decreasing";
            return ECAL_ERR_INVALID;
        }
    const uint32_t base0 = slice_off[0];
    const size_t N = (size_t) slice_off[S] - base0;
    if (N && (!xy || !labels || !order)) {
        ctx->last_error = "null pointer";
        return ECAL_ERR_INVALID;
    }
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    int rc;
    // staging: xy | labels | order | off | cnt | n_clusters | status
    const size_t bytes = (N + 1) * 16 + (N + 1) * 4 * 2 + (size_t) S * 4 * 4 + 64;
    if ((rc = ecal_ensure(ctx, ctx->bfs_host, bytes))) return rc;
    unsigned char *p = (unsigned char *) ctx->bfs_host.ptr;
    double *d_xy = (double *) p;
    int32_t *d_lab = (int32_t *) (p + (N + 1) * 16), *d_ord = d_lab + (N + 1);
    uint32_t *d_off = (uint32_t *) (d_ord + (N + 1)), *d_cnt = d_off + S, *d_ncl = d_cnt + S, *d_st = d_ncl + S;
    std::vector<uint32_t> h(2 * (size_t) S);
    for (uint32_t s = 0; s < S; s++) {
        h[s] = slice_off[s] - base0;
        h[S + s] = slice_off[s + 1] - slice_off[s];
    }
    if (N) {
        ECAL_HIP_TRY(ctx, hipMemcpyAsync(d_xy, xy + 2 * (size_t) base0, N * 16, hipMemcpyHostToDevice, st));
        ECAL_HIP_TRY(ctx, hipMemcpyAsync(d_lab, labels + base0, N * 4, hipMemcpyHostToDevice, st));
    }
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(d_off, h.data(), 2 * (size_t) S * 4, hipMemcpyHostToDevice, st));
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(d_ncl, n_clusters, (size_t) S * 4, hipMemcpyHostToDevice, st));
    ECAL_HIP_TRY(ctx, hipStreamSynchronize(st));   // h is pageable: consumed
    if ((rc = ecal_cluster_order_sized(ctx, d_xy, d_off, d_cnt, S, (uint32_t) std::min<size_t>(N, 0xFFFFFFFFu), eps, d_lab, d_ncl, d_ord, d_st, 0,
                                       nullptr, nullptr, st)))
        return rc;
    std::vector<int32_t> to(N);
    std::vector<uint32_t> ts(S);
    if (N) ECAL_HIP_TRY(ctx, hipMemcpyAsync(to.data(), d_ord, N * 4, hipMemcpyDeviceToHost, st));
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(ts.data(), d_st, (size_t) S * 4, hipMemcpyDeviceToHost, st));
    ECAL_HIP_TRY(ctx, hipStreamSynchronize(st));
    if (N) memcpy(order + base0, to.data(), N * 4);   // caller buffers are written only on success
    memcpy(status, ts.data(), (size_t) S * 4);
    return ECAL_OK;
}

#ifdef ECAL_PHASE_PROF
extern "C" int ecal_debug_bo_cycles(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(ecal::g_bo_cycles), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(ecal::g_bo_cycles), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
