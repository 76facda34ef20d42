// The reference driver's adaptive windowing + keyframe gate with the policy ON THE DEVICE.
//   MultiProcess::process     event_camera_calib/test/eventCameraCalib.cpp:34-97  (success / slide / grow rule :49-95)
//   piece construction        event_camera_calib/test/eventCameraCalib.cpp:168-179
//   EventCalibIni::track      event_camera_calib/src/EventCalibIni.cpp:23-97      (row-line directions :46-57, gate :82)
// Same deterministic per-piece policy as host/multi_process.hpp and eventcalib_amd/adaptive.py, which drive one
// ecal_detect_pass per lock-step pass from the host (upload of the window bounds, download of verdicts + circles, the
// rule in host code: ~2 ms per pass, of which the GPU works ~0.5).  Here a pass is the five detection stages + grid
// ordering + ONE policy kernel (a thread per piece: gate, keyframe record, next window), all enqueued back to back on
// one stream; the host only reads a 4-byte "pieces still active" counter every few passes.  Every pass covers ALL
// pieces: a finished piece has the empty window (+inf, -inf), which every stage skips.
//
// Two gate modes (ecal_adaptive_params.gate_mode):
//   ECAL_GATE_OWN_PIECE   a window is checked against the previous keyframe of ITS OWN piece, every piece's first successful
//                         window is accepted like the reference's very first frame — deterministic whatever the schedule;
//   ECAL_GATE_SHARED_MAP  the reference with ONE worker thread: one keyframe map for all pieces, pieces taken in pop_back
//                         order = ascending time (eventCameraCalib.cpp:40-41,177-179), so keyframes().lower_bound(t) is always
//                         end() and the reference frame is the map's LAST keyframe (EventCalibIni.cpp:26-36) — the previous
//                         keyframe of the own piece, or, before the piece's first acceptance, the last keyframe of the nearest
//                         earlier piece that has one; only the first success of the whole run is ungated (TrackingBase.cpp:18-27).
//                         A piece therefore depends on its predecessors only through ONE frame R (time + row directions), and
//                         only until its first acceptance: F(R) = "reject successes until one passes the gate against R, then
//                         go on as usual".  All pieces run speculatively with R = none (= the own-piece run); then rounds of
//                         { verify every piece against its predecessor's current last keyframe — from the recorded successes
//                         up to the first acceptance —, re-run the pieces whose verdict sequence changes, with that frame as
//                         the initial reference } until nothing changes.  The earliest unsettled piece is settled by every
//                         round, so the result is the sequential single-worker run's; on the benchmark stream a few rounds.
#include "ecal_ctx.hpp"
#include "ref_nth_element.hpp"
#include "row_direction.hpp"

#include <math.h>
#include <atomic>
#include <chrono>

#include <algorithm>
#include <numeric>

namespace {

constexpr int AD_MAX_ROWS = 32;

using ecal::row_direction;   // (row_direction.hpp: shared with the grid finder's epilogue)

constexpr uint32_t AD_NREJ = 4;   // successes rejected before a piece's first acceptance that are kept for the verification
struct AdaptiveArrays {
    double *first, *second, *bound_hi, *ref_t, *ref_dir;  // [P], [P], [P], [P], [P][rows][2]
    uint32_t *active, *have_ref, *levels;                 // [P]; levels: windows the piece has gone through
    uint32_t *slot0, *depth, *lay;                        // [P]: the piece's window slots of this pass (adaptive_alloc_kernel): first slot, ChainTree's two words
    uint32_t *want;                                       // [P]: the chain length the piece asks for (live form: grows while its chains hold)
    uint32_t *counters;  // 0: pieces active after this pass, 1: keyframes, 2: passes that evaluated a window, 3: slots overflowed,
                         // 8: pieces to run again (shared-map verification)
    unsigned long long *windows;                          // windows evaluated
    // shared-map mode: the run a piece's current results come from, and what a change of its initial reference frame can alter
    uint32_t *gen, *nacc, *nrej, *init_has, *rerun;       // [P]: run number, accepted / rejected-before-the-first-acceptance successes
    double *init_t, *init_dir;                            // [P], [P][rows][2]: the reference frame the run started with (init_has)
    double *facc_t, *facc_dir;                            // [P], [P][rows][2]: the first accepted success
    double *rej_t, *rej_dir;                              // [P][AD_NREJ], [P][AD_NREJ][rows][2]: the rejected ones before it
    double start_time, end_time;
    uint32_t p_total, p_first;   // the pieces at work are p_first .. p_first + P - 1 of the p_total pieces of [start_time, end_time]
    // a search over a SUBSET of the pieces under the shared-map gate (ecal_detect_keyframes_sharded): the frame the pieces before the
    // subset (larger indices: earlier in time, another rank's) hand over — ext[0] = 0: not known yet, 1: known, there is none,
    // 2: known; ext_frame = its time stamp, then its row directions [rows][2].  NULL: the subset starts the run (no predecessor)
    const uint32_t *ext;
    const double *ext_frame;
};

// EventCalibIni::track's test (EventCalibIni.cpp:73-82): the median — what libstdc++'s std::nth_element leaves at position
// rows / 2, NaNs included: an angle is NaN when the cosine rounds above 1 (no clamp in the reference) — of the angles between
// corresponding pattern rows, over the time distance, below (5e-4 pi) / MotionTimeStep.  (Eigen's norm() = sqrt of the sum of
// squares, EventCalibIni.cpp:73-77 — not hypot(): one ulp in the norm decides whether a cosine of near-parallel rows rounds above
// 1, i.e. whether the angle is 0 or NaN, and the NaN is kept.)
// By a whole wave: lane i computes row i's angle, the library's nth_element runs wave-uniformly on the LANES of that register (an
// element access is two v_readlane / a select on the lane id) — the same operations on the same values in the same order as a
// thread with the angle array in scratch memory, without a trip to memory per access of the selection.
struct LaneArrF64 {
    using value_type = double;
    double *v;
    uint32_t off;
    struct Ref {
        double *v;
        uint32_t lane;
        __device__ __forceinline__ operator double() const {
            const unsigned long long b = (unsigned long long) __double_as_longlong(*v);
            const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) b, (int) lane);
            const uint32_t hi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (b >> 32), (int) lane);
            return __longlong_as_double((long long) (((unsigned long long) hi << 32) | lo));
        }
        __device__ __forceinline__ Ref &operator=(double x) {
            *v = (threadIdx.x & 63u) == lane ? x : *v;
            return *this;
        }
        __device__ __forceinline__ Ref &operator=(const Ref &o) { return *this = (double) o; }
    };
    __device__ __forceinline__ Ref operator[](int64_t i) const { return Ref{v, (uint32_t) __builtin_amdgcn_readfirstlane((int) (off + (uint32_t) i))}; }
    __device__ __forceinline__ LaneArrF64 operator+(uint32_t d) const { return LaneArrF64{v, off + d}; }
};
// Both frames on registers: lane i holds row i of the reference frame (rx, ry) and of the window's (dx, dy); the same expression,
// operand for operand, as gate_accepts.
__device__ __forceinline__ bool gate_accepts_regs(double rx, double ry, double ref_t, double dx, double dy, double t_mid, uint32_t rows, double mts) {
    const uint32_t i = threadIdx.x & 63u;
    double theta = 0.0;
    if (i < rows) {
        const double c = (rx * dx + ry * dy) / (sqrt(rx * rx + ry * ry) * sqrt(dx * dx + dy * dy));
        theta = acos(c);
    }
    ecal::ref_nth_element(LaneArrF64{&theta, 0u}, rows, rows / 2u, [](double x, double y) { return x < y; });
    const double med = (double) LaneArrF64{&theta, 0u}[rows / 2u];
    return med / fabs(t_mid - ref_t) < (5e-4 * M_PI) / mts;
}

// The window that follows (f, s2) under outcome o of eventCameraCalib.cpp:61-62 (0: keyframe accepted), :67-69,75-77 (1: slide),
// :70-71,78-79 (2: grow) — the ONE place these sums are written, so that a window evaluated ahead of time is bit for bit the
// window the policy arrives at.
__device__ __forceinline__ void next_window(int o, double f, double s2, double mts, double &nf, double &ns) {
    const double ln = 3 * mts, gap = 5 * mts;
    if (o == 0) {
        nf = s2 + gap;
        ns = nf + ln;
    } else if (o == 1) {
        nf = f + mts;
        ns = nf + ln;
    } else {
        nf = f;
        ns = s2 + mts;
    }
}

// Window slots of a pass: a few per piece in all, dealt out among the pieces still at work (adaptive_alloc_kernel)
// up to AD_DEPTH_MAX windows of chain each.  Measured on the benchmark stream (Mev/s at 1270 / 4096 pieces) with the same
// chain length for every piece: 1 window 145 / 490, 2: 246 / 690, 3: 307 / 781, 4: 358 / 800, 5: 406 / 837, 6: 432 / 840,
// 8 and more: less again (the work of a pass grows with the chain, the likely verdicts get rarer along it); with the slots
// of the finished pieces going to the others: 6 slots per piece and chains of <= 12: 533 / 922, <= 24: 592 / 943,
// <= 48: 633 / 962, <= 64: 626 / 973; 4, 5, 8 slots per piece (<= 64, <= 48, <= 64): 559 / 1001, 582 / 1017, 541 / 823.
constexpr uint32_t AD_DEPTH_MAX = 48;
// pieces at work up to which a pass's grid finder runs in its latency form (a wave per start, ecal_grid.hip).  Measured, shared-map
// gate, seconds at 1270 / 4096 pieces: never 0.137 / 0.132; 32: 0.133 / 0.133; 64: 0.133 / 0.135; 128: 0.132 / 0.130; 256: 0.131 /
// 0.122; 512: 0.132 / 0.128
constexpr uint32_t AD_GRID_LATENCY_PIECES = 256;
#ifndef ECAL_AD_STAGE_LATENCY_PIECES
#define ECAL_AD_STAGE_LATENCY_PIECES 2048
#endif
// Pieces at work up to which the stages take their latency forms (ecal_ctx::latency_pass): one launch per stage takes a window through
// the tier its size asks for — no to-do list, no second pass.  Measured at 1270 pieces (shared-map search, 3 runs each, s):
// 96: 0.112 - 0.113 | 256: 0.109 - 0.110 | 640: 0.108 | 1100: 0.107 | always: 0.105; at 4096 pieces 0.116 - 0.137 for every value
// (run-to-run noise is larger than the differences) — the forms stay off above 2048 windows, where the first pass's small blocks fill
// the chip and the occupancy of the wide forms starts to cost.
constexpr uint32_t AD_STAGE_LATENCY_PIECES = ECAL_AD_STAGE_LATENCY_PIECES;
// pieces at work up to which the slicer's latency form takes the third pass's windows (4096 .. 5119 events) in its one launch too
#ifndef ECAL_AD_THIRD_IN_ONE_PIECES
#define ECAL_AD_THIRD_IN_ONE_PIECES 600    // (with the two-pass form at two workgroups per CU — 64 / 256 / 700 / always: 0.093 - 0.094 / 0.092 / 0.092 / 0.090 - 0.091 s at 1270 pieces; with it compiled for three: 600 / 1000 / always: 0.0850 - 0.0852 / 0.0854 - 0.0856 / 0.0862 - 0.0865)
#endif
constexpr uint32_t AD_THIRD_IN_ONE_PIECES = ECAL_AD_THIRD_IN_ONE_PIECES;
// the search's tail: with at most AD_TAIL_PIECES pieces at work a pass is launched over AD_TAIL_SLOTS window slots (run_passes)
constexpr uint32_t AD_TAIL_PIECES = 64, AD_TAIL_SLOTS = 4096;
static uint32_t adaptive_slots_per_piece(uint32_t pieces, int forced = 0) {
    if (forced >= 1 && forced <= 64) return (uint32_t) forced;   // (ECAL_ADAPTIVE_SHAPE depth: debug / measurement switch; the result does not depend on it)
    // (many pieces fill the GPU by themselves.  Round 6, with the passes' fixed cost down: a pass of 8 - 16 thousand slots is the
    // optimum — shared-map search, s, at slots per piece 1 / 2 / 3 / 4 / 5 / 6: 2048 pieces - / - / 0.119-0.129 / 0.116-0.122 / 0.123-0.127 /
    // 0.124-0.133; 4096 pieces 0.103 / 0.099 / 0.120 / - / 0.114-0.127 / -; 8192 pieces 0.086-0.096 / 0.065 / 0.086-0.092 / 0.090-0.097 /
    // 0.094-0.098 / -; 16384 pieces 0.152 / 0.168-0.179 / 0.165; 1270 pieces 5 / 6 / 7: 0.107 / 0.105 / 0.104; a second box: 1800 pieces
    // 4 / 6: 0.123-0.131 / 0.140; 2300 pieces 4 / 5 / 6: 0.132-0.135 / 0.122-0.128 / 0.150; 3000 pieces 2 / 4: 0.121-0.148 / 0.108 —
    // tools/depth_sweep.sh; the boxes differ by more than neighbouring settings do)
    // (with the third hash pass and the look-ahead of round 6's end, 1270 pieces 6 / 7 / 8 / 9: 0.0848 - 0.0856 / 0.0839 - 0.0840 / 0.0877 /
    // 0.0966: seven where that makes a pass of 8 - 9 thousand slots)
    return pieces <= 1100u ? 6u : (pieces <= 1400u ? 7u : (pieces <= 1536u ? 6u : (pieces <= 3500u ? 4u : (pieces <= 8192u ? 2u : 1u))));
}
static uint32_t adaptive_depth_max(int forced = 0) {
    if (forced >= 1 && forced <= 64) return (uint32_t) forced;   // (ECAL_ADAPTIVE_SHAPE depth_max: debug / measurement switch)
    return AD_DEPTH_MAX;
}

// The verdict that is to be expected of a window: not a keyframe (one window in thirteen is), and then the rule's choice
// between slide and grow as far as the window's length decides it (a window longer than three lengths slides; a shorter one
// grows unless it holds more than the event threshold, which is rare).
__device__ __forceinline__ int likely_outcome(double f, double s2, double mts) { return (s2 - f) > 3 * (3 * mts) ? 1 : 2; }


// A piece's window slots as a TREE of chains (round 5).  The verdicts that do not accept are predictable (likely_outcome: on the
// benchmark stream without a miss), so a chain only ever ends at an ACCEPTANCE or at its last slot — and where a window is
// accepted cannot be known ahead: 4 to 10 steps long, most often 5.  Level 0 is the main chain (len0 windows from the piece's
// current one, every verdict the likely one).  Behind positions [a1, a1 + c1) of it hang the c1 chains of level 1 — what follows
// if THAT window is accepted, len1 windows —, behind positions [a2, a2 + c2) of every level-1 chain the chains of level 2, and so
// on: a walk goes down one level per keyframe.  Slots: level 0, then the chains of level 1 in the order of their positions, then
// level 2 (chain i of level 1, position q: number i * c2 + (q - a2)), then level 3.  c1 = 0: a chain; c2 = 0: the main chain
// and side chains of round 4.
struct ChainTree {
    uint32_t len0, len1, len2, len3, a1, a2, a3, c1, c2, c3;
    __device__ __forceinline__ uint32_t off1() const { return len0; }
    __device__ __forceinline__ uint32_t off2() const { return len0 + c1 * len1; }
    __device__ __forceinline__ uint32_t off3() const { return len0 + c1 * len1 + c1 * c2 * len2; }
    __device__ __forceinline__ uint32_t slots() const { return len0 + c1 * (len1 + c2 * (len2 + c3 * len3)); }
    __device__ __forceinline__ uint32_t chains() const { return 1u + c1 * (1u + c2 * (1u + c3)); }
    __device__ __forceinline__ uint32_t longest_walk() const { return len0 + (c1 ? len1 : 0u) + (c2 ? len2 : 0u) + (c3 ? len3 : 0u); }
    // the two words a piece's layout is kept in (AdaptiveArrays::depth, ::lay)
    __device__ __forceinline__ uint32_t word0() const { return len0 | (c1 << 8) | (a1 << 16) | (len1 << 24); }   // (len0 <= 255, c1 <= 255)
    __device__ __forceinline__ uint32_t word1() const { return a2 | (c2 << 4) | (len2 << 8) | (a3 << 16) | (c3 << 20) | (len3 << 24); }
    __device__ __forceinline__ static ChainTree unpack(uint32_t w0, uint32_t w1) {
        ChainTree y;
        y.len0 = w0 & 0xFFu, y.c1 = (w0 >> 8) & 0xFFu, y.a1 = (w0 >> 16) & 0xFFu, y.len1 = w0 >> 24;
        y.a2 = w1 & 0xFu, y.c2 = (w1 >> 4) & 0xFu, y.len2 = (w1 >> 8) & 0xFFu, y.a3 = (w1 >> 16) & 0xFu, y.c3 = (w1 >> 20) & 0xFu, y.len3 = w1 >> 24;
        return y;
    }
};
// The tree a piece gets in the regime where keyframes are to be expected (its last chain ended at one, or it starts): `tree` =
// main | len << 8 | c1 << 16 | c2 << 20 | c3 << 24 | from << 28 at most, as much of it as `slots` hold — level by level, every
// chain of a level or none.  Fewer than main + len slots: no tree (c1 = 0, the caller's chain).
__device__ __forceinline__ ChainTree tree_for(uint32_t slots, uint32_t tree) {
    ChainTree y = {};
    const uint32_t TM = tree & 0xFFu, TL = (tree >> 8) & 0xFFu, C1 = (tree >> 16) & 0xFu, C2 = (tree >> 20) & 0xFu, C3 = (tree >> 24) & 0xFu, TF = tree >> 28;
    // (a level's chains hang below positions TF .. TF + c - 1 of the chain above: a word whose hang positions run past that chain —
    // only reachable through ECAL_ADAPTIVE_SHAPE tree= — is clamped; write_tree_level writes child chains for positions < len only,
    // and a counted but unwritten chain would keep the previous pass's windows)
    if (!tree || TL == 0 || slots < TM + TL || TF >= TM) return y;
    const uint32_t cap1 = TM - TF, cap23 = TF < TL ? TL - TF : 0u;
    y.len0 = TM;
    y.len1 = y.len2 = y.len3 = TL;
    y.a1 = y.a2 = y.a3 = TF;
    uint32_t used = TM;
    y.c1 = (slots - used) / TL;
    y.c1 = y.c1 < C1 ? y.c1 : C1;
    y.c1 = y.c1 < cap1 ? y.c1 : cap1;
    used += y.c1 * TL;
    y.c2 = y.c1 ? (slots - used) / (y.c1 * TL) : 0u;
    y.c2 = y.c2 < C2 ? y.c2 : C2;
    y.c2 = y.c2 < cap23 ? y.c2 : cap23;
    used += y.c1 * y.c2 * TL;
    y.c3 = y.c2 ? (slots - used) / (y.c1 * y.c2 * TL) : 0u;
    y.c3 = y.c3 < C3 ? y.c3 : C3;
    y.c3 = y.c3 < cap23 ? y.c3 : cap23;
    return y;
}
__device__ __forceinline__ uint32_t tree_slots_max(uint32_t tree) {
    const uint32_t TM = tree & 0xFFu, TL = (tree >> 8) & 0xFFu, C1 = (tree >> 16) & 0xFu, C2 = (tree >> 20) & 0xFu, C3 = (tree >> 24) & 0xFu;
    return TM + TL * C1 * (1u + C2 * (1u + C3));
}
// chain `idx` of level LV: its windows from (f, s2) on as if every verdict were the likely one (empty behind the piece's end, as
// write_chain), and below the positions that carry them the chains of the next level, from the window that follows an acceptance
template <int LV>
__device__ __forceinline__ void write_tree_level(const ChainTree &y, uint32_t idx, double f, double s2, bool act, double hi, double mts, double *t0, double *t1) {
    const uint32_t len = LV == 0 ? y.len0 : (LV == 1 ? y.len1 : (LV == 2 ? y.len2 : y.len3));
    const uint32_t off = LV == 0 ? 0u : (LV == 1 ? y.off1() : (LV == 2 ? y.off2() : y.off3()));
    const uint32_t ca = LV == 0 ? y.a1 : (LV == 1 ? y.a2 : y.a3), cc = LV == 0 ? y.c1 : (LV == 1 ? y.c2 : (LV == 2 ? y.c3 : 0u));
    double *c0 = t0 + off + (size_t) idx * len, *c1 = t1 + off + (size_t) idx * len;
    for (uint32_t q = 0; q < len; q++) {
        c0[q] = act ? f : INFINITY;
        c1[q] = act ? s2 : -INFINITY;
        if constexpr (LV < 3) {
            if (q >= ca && q < ca + cc) {
                double af = 0, as = 0;
                bool sact = act;
                if (sact) {
                    next_window(0, f, s2, mts, af, as);
                    sact = as < hi;
                }
                write_tree_level<LV + 1>(y, idx * cc + (q - ca), af, as, sact, hi, mts, t0, t1);
            }
        }
        if (act) {
            double nf, ns;
            next_window(likely_outcome(f, s2, mts), f, s2, mts, nf, ns);
            f = nf;
            s2 = ns;
            act = ns < hi;
        }
    }
}

// ONE chain of the tree — number c in the order level 0, the chains of level 1, of level 2, of level 3 — written by itself: its first
// window follows from the piece's current one by the walk down to it (the likely verdicts along each chain above, an acceptance
// at the position it hangs below: the very sequence of next_window calls write_tree_level makes on its way there, so the same
// doubles), then its windows as write_tree_level writes them.
__device__ __forceinline__ void write_tree_chain(const ChainTree &y, uint32_t c, double f, double s2, double hi, double mts, double *t0, double *t1) {
    uint32_t lv = 0, idx = 0;
    if (c >= 1u) {
        if (c < 1u + y.c1) lv = 1u, idx = c - 1u;
        else if (c < 1u + y.c1 + y.c1 * y.c2) lv = 2u, idx = c - 1u - y.c1;
        else lv = 3u, idx = c - 1u - y.c1 - y.c1 * y.c2;
    }
    // positions of the descents: in the main chain, in the level-1 chain, in the level-2 chain
    uint32_t p[3] = {0u, 0u, 0u};
    if (lv == 1u) p[0] = y.a1 + idx;
    else if (lv == 2u) p[0] = y.a1 + idx / y.c2, p[1] = y.a2 + idx % y.c2;
    else if (lv == 3u) p[0] = y.a1 + (idx / y.c3) / y.c2, p[1] = y.a2 + (idx / y.c3) % y.c2, p[2] = y.a3 + idx % y.c3;
    bool act = true;
    for (uint32_t m = 0; m < lv; m++) {
        for (uint32_t q = 0; q < p[m]; q++)
            if (act) {
                double nf, ns;
                next_window(likely_outcome(f, s2, mts), f, s2, mts, nf, ns);
                f = nf;
                s2 = ns;
                act = ns < hi;
            }
        if (act) {
            double af, as;
            next_window(0, f, s2, mts, af, as);
            f = af;
            s2 = as;
            act = as < hi;
        }
    }
    const uint32_t len = lv == 0u ? y.len0 : (lv == 1u ? y.len1 : (lv == 2u ? y.len2 : y.len3));
    const uint32_t off = lv == 0u ? 0u : (lv == 1u ? y.off1() : (lv == 2u ? y.off2() : y.off3()));
    double *c0 = t0 + off + (size_t) idx * len, *c1 = t1 + off + (size_t) idx * len;
    for (uint32_t q = 0; q < len; q++) {
        c0[q] = act ? f : INFINITY;
        c1[q] = act ? s2 : -INFINITY;
        if (act) {
            double nf, ns;
            next_window(likely_outcome(f, s2, mts), f, s2, mts, nf, ns);
            f = nf;
            s2 = ns;
            act = ns < hi;
        }
    }
}

// The window slots of the next pass.  The pieces' chains differ a lot in length (the benchmark's longest has 559 windows, the
// average 83), so most passes see few pieces still at work: the B slots of a pass are dealt out evenly among THOSE, up to
// d_max windows of chain each — the fewer pieces remain, the further each one looks ahead.  One workgroup.
__device__ void restart_piece(uint32_t k, uint32_t P, uint32_t rows, double mts, const AdaptiveArrays &st);
constexpr int AD_ALLOC_T = 1024;
constexpr int AD_ALLOC_TAB = 2 * AD_ALLOC_T;   // pieces at work up to which adaptive_alloc_kernel writes the slots a chain per thread
// live form, few pieces at work: a main chain and side chains (write_chain): first main-chain window with one, how many, their
// length, the main chain's length.  Measured (ECAL_ADAPTIVE_SHAPE side, tools/side_sweep.sh): 24 side chains of 12 behind a main chain
// of 24 take the search from 108 to 79 passes at 1270 pieces, but the passes of the tail grow by what they save — 0.151 against
// 0.159 s at 1270 pieces, 0.153 / 0.156 at 4096, 0.289 / 0.271 at 254; eight other layouts within 3 % of that — so the default
// is none, and the form stays behind the switch with its test.
constexpr uint32_t AD_SIDE_NONE = 0u;
constexpr uint32_t AD_SIDE_MEASURED = 0u | (24u << 8) | (12u << 16) | (24u << 24);
// Round 5: with the passes cheaper (ties resolved inside the extraction passes, the step kernel's chain walk on registers) the
// saved passes now outweigh the wider ones — shared-map gate 0.145 -> 0.138 s at 1270 pieces, 0.148 -> 0.137 s at 4096 — and the
// measured layout is the default (ECAL_ADAPTIVE_SHAPE side=0: none).
constexpr uint32_t AD_SIDE_DEFAULT = AD_SIDE_MEASURED;
// tree_for's word: main chain 8, chains of 8 behind positions 1 .. 5 / 1 .. 4 / 1 .. 3 of the levels above (a keyframe's window is
// 4 - 10 steps long, position = steps - 3 after a fresh start: 5 steps 32 %, 4: 21 %, 6: 20 %, 7: 12 %, 8: 8 %)
// Round 6 (cheaper passes; tools/tree_ab.sh, shared-map search at 1270 pieces, s): chains of 8 / 5 / 4 / 3 as above 0.0903 - 0.0906;
// chains of 7: 0.0890; 7 with 5 / 5 / 3 chains: 0.0884 - 0.0885; 6 / 4 / 3, 5 / 4 / 4 and others within 0.001 of those
constexpr uint32_t AD_TREE_DEFAULT = 8u | (7u << 8) | (5u << 16) | (5u << 20) | (3u << 24) | (1u << 28);
// report (pinned host memory, may be NULL) + seq: the pass that ends with this launch tells the host how it went — pieces still
// active, keyframe records so far, capacity overflow, and LAST the pass's number, which the host polls for (a copy engine
// transfer + an event per pass between the kernels of a launch-bound chain cost more than the kernels they sat between) —
// and takes the active-pieces counter back to zero for the next pass.
__global__ __launch_bounds__(AD_ALLOC_T) void adaptive_alloc_kernel(uint32_t P, uint32_t B, uint32_t deal, uint32_t d_max, AdaptiveArrays st,
                                                                    double mts, double *t0, double *t1, uint32_t *report, uint32_t seq, uint32_t live_base, uint32_t live_floor,
                                                                    uint32_t side /* from | count << 8 | length << 16 | main chain << 24 */,
                                                                    uint32_t tree /* tree_for's word; 0: no trees */,
                                                                    uint32_t restart_rows /* > 0: restart_piece first (the shared-map gate's re-runs; the pattern's rows) */) {
    __shared__ uint32_t red[AD_ALLOC_T / 64 + 1];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t per = (P + AD_ALLOC_T - 1) / AD_ALLOC_T, k0 = tid * per;
    // (the re-runs the verification asked for start here: a launch of its own cost 6 us of every pass)
    if (restart_rows)
        for (uint32_t k = k0; k < k0 + per && k < P; k++) restart_piece(k, P, restart_rows, mts, st);
    // this thread's pieces' words, read once (up to AD_ALLOC_REG pieces a thread — 4096 pieces — on registers; beyond: from memory
    // at every use, as it used to be for all: a chain of dependent reads through the three sums below)
    constexpr uint32_t AD_ALLOC_REG = 4;
    uint32_t r_act[AD_ALLOC_REG], r_want[AD_ALLOC_REG];
    const bool regs = per <= AD_ALLOC_REG;
#pragma unroll
    for (uint32_t i = 0; i < AD_ALLOC_REG; i++) {
        const uint32_t k = k0 + i;
        const bool in = regs && i < per && k < P;
        r_act[i] = in ? st.active[k] : 0u;
        r_want[i] = in ? st.want[k] : 0u;
    }
    auto active_of = [&](uint32_t k) -> uint32_t {
        if (!regs) return st.active[k];
        uint32_t v = 0;
#pragma unroll
        for (uint32_t i = 0; i < AD_ALLOC_REG; i++) v = k - k0 == i ? r_act[i] : v;
        return v;
    };
    auto want_of = [&](uint32_t k) -> uint32_t {
        if (!regs) return st.want[k];
        uint32_t v = 0;
#pragma unroll
        for (uint32_t i = 0; i < AD_ALLOC_REG; i++) v = k - k0 == i ? r_want[i] : v;
        return v;
    };
    uint32_t mine = 0;
    for (uint32_t k = k0; k < k0 + per && k < P; k++) mine += active_of(k) ? 1u : 0u;
    uint32_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if ((int) lane >= d) inc += o;
    }
    if (lane == 63) red[wave] = inc;
    __syncthreads();
    uint32_t n_act = 0, j0 = inc - mine;   // j0: active pieces before this thread's
    for (uint32_t w = 0; w < AD_ALLOC_T / 64; w++) {
        n_act += red[w];
        if (w < wave) j0 += red[w];
    }
    // `deal` of the B slots are dealt out (a run of few pieces — a verification round of the shared-map gate — does not get
    // the whole pass: beyond what fills the GPU a pass's time grows with its windows, and most of a long look-ahead is thrown away)
    // live_base (the live form of the shared-map search, where first runs and re-runs share the passes): every piece gets the
    // chain length it asks for (want: twice its last length while its chains hold, else the base), scaled down together when the
    // pass cannot hold them all
    {
        const uint32_t SF = side & 0xFFu, SC = (side >> 8) & 0xFFu, SL = (side >> 16) & 0xFFu, SM = side >> 24;
        __shared__ uint32_t red2[AD_ALLOC_T / 64 + 1];
        const uint32_t share = n_act ? live_floor / n_act : 0u;   // few pieces at work: the pass has slots to spare for all of them
        // slots of a piece: a chain of up to d_max windows; with slots to spare beyond that (few pieces at work), a main chain
        // of AD_SIDE_MAIN windows + side chains of AD_SIDE_LEN windows for as many of them as fit (write_chain)
        // a piece whose last chain ended at a keyframe (or that starts) is where keyframes are to be expected: its slots are a tree
        // (tree_for) as far as the share goes; one whose chains hold — a stretch without the pattern — asks for a chain as before
        const uint32_t tree_max = tree ? tree_slots_max(tree) : 0u;
        auto wanted = [&](uint32_t k) -> uint32_t {
            uint32_t w = want_of(k) > live_base ? want_of(k) : live_base;
            w = w > share ? w : share;
            if (tree && want_of(k) == 0u && share >= (tree & 0xFFu) + ((tree >> 8) & 0xFFu)) return w < tree_max ? w : tree_max;
            const uint32_t cap = SC && share >= SM + 2u * SL ? SM + SC * SL : d_max;
            return w < cap ? w : cap;
        };
        uint32_t mw = 0;
        for (uint32_t k = k0; k < k0 + per && k < P; k++) mw += active_of(k) ? wanted(k) : 0u;
        uint32_t tw = mw;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) tw += __shfl_xor(tw, d, 64);
        __syncthreads();
        if (lane == 0) red2[wave] = tw;
        __syncthreads();
        uint32_t total_want = 0;
        for (uint32_t w = 0; w < AD_ALLOC_T / 64; w++) total_want += red2[w];
        auto depth_of = [&](uint32_t k) -> uint32_t {
            const uint32_t w = wanted(k);
            if (total_want <= B) return w;
            // one slot each (B >= P), the rest in proportion to what is asked beyond that, rounded down: the sum stays within B
            return 1u + (uint32_t) ((unsigned long long) (w - 1u) * (B - n_act) / (total_want - n_act));
        };
        uint32_t md = 0;
        for (uint32_t k = k0; k < k0 + per && k < P; k++) md += active_of(k) ? depth_of(k) : 0u;
        uint32_t incd = md;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incd, d, 64);
            if ((int) lane >= d) incd += o;
        }
        __syncthreads();
        if (lane == 63) red2[wave] = incd;
        __syncthreads();
        uint32_t pred = 0, total_depth = 0;
        for (uint32_t w = 0; w < AD_ALLOC_T / 64; w++) {
            if (w < wave) pred += red2[w];
            total_depth += red2[w];
        }
        // The windows of the slots are written by ALL threads, a chain of a piece's tree each (a tree has up to 86 chains, 688
        // slots: a thread per PIECE walked them one after the other — 40 to 90 us of every pass of the search's tail, where few
        // pieces hold deep trees).  The pieces' layouts go through LDS for that; more pieces at work than the table holds (their
        // chains are short then): a thread per piece as before.
        __shared__ uint32_t pk_k[AD_ALLOC_TAB], pk_at[AD_ALLOC_TAB], pk_slots[AD_ALLOC_TAB], pk_w0[AD_ALLOC_TAB], pk_w1[AD_ALLOC_TAB], pk_off[AD_ALLOC_TAB + 1];
        const bool tabled = n_act <= AD_ALLOC_TAB;
        uint32_t at = pred + incd - md;   // slots before this thread's pieces
        uint32_t j = j0;
        for (uint32_t k = k0; k < k0 + per && k < P; k++) {
            if (active_of(k)) {
                const uint32_t slots = depth_of(k);
                ChainTree y = {};
                if (tree && want_of(k) == 0u && share >= (tree & 0xFFu) + ((tree >> 8) & 0xFFu)) y = tree_for(slots, tree);
                if (y.c1 == 0u) {   // a chain, with round 4's side chains when there are slots for them
                    y = ChainTree{};
                    y.len0 = slots < d_max ? slots : d_max;
                    if (slots > d_max && SC) {
                        y.len0 = SM;
                        y.c1 = (slots - SM) / SL;
                        y.c1 = y.c1 < SC ? y.c1 : SC;
                        y.a1 = SF;
                        y.len1 = SL;
                    }
                }
                st.slot0[k] = at;
                st.depth[k] = y.word0();
                st.lay[k] = y.word1();
                if (tabled) {
                    pk_k[j] = k;
                    pk_at[j] = at;
                    pk_slots[j] = slots;
                    pk_w0[j] = y.word0();
                    pk_w1[j] = y.word1();
                    j++;
                } else {
                    write_tree_level<0>(y, 0u, st.first[k], st.second[k], true, st.bound_hi[k], mts, t0 + at, t1 + at);
                    for (uint32_t q = y.slots(); q < slots; q++) {   // (the remainder of its slots)
                        t0[at + q] = INFINITY;
                        t1[at + q] = -INFINITY;
                    }
                }
                at += slots;
            } else {
                st.depth[k] = 0;
                st.lay[k] = 0;
            }
        }
        if (tabled) {
            __syncthreads();
            // chains before table entry e: entries 2 tid, 2 tid + 1 by thread tid, then the scan over the threads
            static_assert(AD_ALLOC_TAB == 2 * AD_ALLOC_T, "two table entries per thread");
            uint32_t c2[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const uint32_t e = 2u * tid + (uint32_t) i;
                c2[i] = e < n_act ? ChainTree::unpack(pk_w0[e], pk_w1[e]).chains() : 0u;
            }
            uint32_t incc = c2[0] + c2[1];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incc, d, 64);
                if ((int) lane >= d) incc += o;
            }
            if (lane == 63) red2[wave] = incc;
            __syncthreads();
            uint32_t before = incc - (c2[0] + c2[1]), n_tasks = 0;
            for (uint32_t w = 0; w < AD_ALLOC_T / 64; w++) {
                if (w < wave) before += red2[w];
                n_tasks += red2[w];
            }
            pk_off[2u * tid] = before;
            pk_off[2u * tid + 1u] = before + c2[0];
            if (tid == 0) pk_off[AD_ALLOC_TAB] = n_tasks;
            __syncthreads();
            for (uint32_t t = tid; t < n_tasks; t += AD_ALLOC_T) {
                uint32_t lo = 0, hi_e = n_act;   // the entry whose chains hold task t: pk_off[e] <= t < pk_off[e + 1]
                while (hi_e - lo > 1u) {
                    const uint32_t mid = (lo + hi_e) / 2u;
                    if (pk_off[mid] <= t) lo = mid;
                    else hi_e = mid;
                }
                const uint32_t e = lo, k = pk_k[e], pat = pk_at[e];
                const ChainTree y = ChainTree::unpack(pk_w0[e], pk_w1[e]);
                write_tree_chain(y, t - pk_off[e], st.first[k], st.second[k], st.bound_hi[k], mts, t0 + pat, t1 + pat);
                if (t == pk_off[e])
                    for (uint32_t q = y.slots(); q < pk_slots[e]; q++) {   // (the remainder of the piece's slots)
                        t0[pat + q] = INFINITY;
                        t1[pat + q] = -INFINITY;
                    }
            }
        }
        for (uint32_t i = total_depth + tid; i < B; i += AD_ALLOC_T) {   // slots nobody got
            t0[i] = INFINITY;
            t1[i] = -INFINITY;
        }
    }
    __syncthreads();   // (the restarts' additions to the counter)
    if (report && tid == 0) {
        const uint32_t active = __hip_atomic_load(&st.counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st.counters[0] = 0;
        __hip_atomic_store(&report[0], active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&report[1], (uint32_t) *st.windows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (windows the rule has been applied to so far: the trace's)
        __hip_atomic_store(&report[3], st.counters[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        __hip_atomic_store(&report[2], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// piece k at the start of a run: its first window, and the reference frame it starts with (none, or init_*)
__device__ void piece_start(uint32_t k, uint32_t P, uint32_t rows, double mts, const AdaptiveArrays &st) {
    const double ln = 3 * mts;
    const double step = (st.end_time - st.start_time) / (double) st.p_total;  // eventCameraCalib.cpp:168-179
    const uint32_t kg = st.p_first + k;
    const double hi = st.end_time - step * (double) kg, first = st.end_time - step * (double) (kg + 1), second = first + ln;
    st.bound_hi[k] = hi;
    st.first[k] = first;
    st.second[k] = second;
    st.active[k] = second < hi ? 1u : 0u;
    st.levels[k] = 0;
    st.nacc[k] = 0;
    st.nrej[k] = 0;
    st.want[k] = 0;   // (0: the base length)
    st.have_ref[k] = st.init_has[k];
    st.ref_t[k] = st.init_t[k];
    for (uint32_t i = 0; i < 2 * rows; i++) st.ref_dir[(size_t) k * rows * 2 + i] = st.init_dir[(size_t) k * rows * 2 + i];
}

__global__ void adaptive_init_kernel(uint32_t P, uint32_t rows, double mts, AdaptiveArrays st) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) {
        for (int i = 0; i < 16; i++) st.counters[i] = 0;
        *st.windows = 0;
    }
    if (k >= P) return;
    st.gen[k] = 0;
    st.rerun[k] = 0;
    st.init_has[k] = 0;
    st.init_t[k] = 0;
    for (uint32_t i = 0; i < 2 * rows; i++) st.init_dir[(size_t) k * rows * 2 + i] = 0;
    piece_start(k, P, rows, mts, st);
}

__device__ void restart_piece(uint32_t k, uint32_t P, uint32_t rows, double mts, const AdaptiveArrays &st) {
    if (!st.rerun[k]) return;
    st.rerun[k] = 0;
    st.gen[k]++;   // the keyframe records of the earlier runs of this piece are dead
    piece_start(k, P, rows, mts, st);
    if (st.active[k]) atomicAdd(&st.counters[0], 1u);   // (the live form: the pass reports pieces with windows still to go)
}

// The verification of adaptive_verify_kernel after EVERY pass instead of after every set of runs (shared-map mode, the default
// form): piece k is looked at as soon as the pieces between it and its nearest earlier keyframe have finished their runs, whether
// k itself is finished or not — a set of runs lasts as long as its longest piece (16 passes where the average run takes 9), and
// most re-runs can start long before that.  Same rule, same fixed point (the sequential run: the earliest unsettled piece is
// settled by the time its predecessors are); what a piece is judged against is the state of FINISHED predecessors only, and a
// predecessor that starts again later hands its successor a new frame when it is finished again.
//   k finished, or running with its first acceptance behind it: as adaptive_verify_kernel (same verdicts -> the frame is noted,
//   nothing else; different -> run again).
//   k running, nothing accepted yet: the rejected successes so far must stay rejected (else: run again); then the run so far is
//   the run it would have been with the new frame, and the frame becomes its reference for the successes to come.
// the same by a WAVE per piece (the default's kernel): the scan for the predecessor 64 pieces at a time, the frames a lane per row,
// the gate on registers (gate_accepts_regs) — a thread per piece went through the scan, the comparison and up to five gates
// with their angle arrays in scratch memory as chains of dependent reads: 36 us of every pass
__global__ __launch_bounds__(256) void adaptive_verify_live_kernel(uint32_t P, uint32_t rows, double mts, AdaptiveArrays st) {
    const uint32_t k = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (k >= P) return;
    uint32_t j = k + 1;
    bool has = false;
    while (j < P) {
        const uint32_t jl = j + lane;
        const uint32_t a = jl < P ? st.active[jl] : 0u, n = jl < P ? st.nacc[jl] : 0u;
        const unsigned long long hit = __ballot(a != 0u || n > 0u);
        if (hit) {
            const uint32_t first = (uint32_t) __builtin_ctzll(hit);
            if (__shfl((int) a, (int) first, 64)) return;   // not finished: its keyframes (or their absence) are not known yet
            j += first;
            has = true;
            break;
        }
        j += 64u;
    }
    // no keyframe in the pieces before k that this call runs: the frame comes from the pieces before THEM (another rank's) — not
    // decidable until that frame has arrived (k's results stand as a speculation with "none" till then)
    bool from_ext = false;
    if (!has && st.ext) {
        const uint32_t es = __hip_atomic_load(st.ext, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (es == 0u) return;
        from_ext = has = es == 2u;
    }
    const double r_t = has ? (from_ext ? st.ext_frame[0] : st.ref_t[j]) : 0.0;
    double rx = 0.0, ry = 0.0, ix = 0.0, iy = 0.0;
    if (lane < rows) {
        if (has) {
            rx = from_ext ? st.ext_frame[1 + 2 * lane] : st.ref_dir[((size_t) j * rows + lane) * 2];
            ry = from_ext ? st.ext_frame[2 + 2 * lane] : st.ref_dir[((size_t) j * rows + lane) * 2 + 1];
        }
        ix = st.init_dir[((size_t) k * rows + lane) * 2];
        iy = st.init_dir[((size_t) k * rows + lane) * 2 + 1];
    }
    bool same = (st.init_has[k] != 0) == has;
    if (same && has) same = st.init_t[k] == r_t && __ballot(lane < rows && !(ix == rx && iy == ry)) == 0ull;
    if (same) return;
    const uint32_t nrej_k = st.nrej[k], nacc_k = st.nacc[k];
    bool again = nrej_k > AD_NREJ;   // more rejected successes than were kept: not decidable here
    const uint32_t nr = nrej_k < AD_NREJ ? nrej_k : AD_NREJ;
    for (uint32_t i = 0; i < nr && !again; i++) {
        double dx = 0.0, dy = 0.0;
        if (lane < rows) {
            dx = st.rej_dir[(((size_t) k * AD_NREJ + i) * rows + lane) * 2];
            dy = st.rej_dir[(((size_t) k * AD_NREJ + i) * rows + lane) * 2 + 1];
        }
        again = !has || gate_accepts_regs(rx, ry, r_t, dx, dy, st.rej_t[(size_t) k * AD_NREJ + i], rows, mts);
    }
    if (!again && nacc_k > 0) {
        double dx = 0.0, dy = 0.0;
        if (lane < rows) {
            dx = st.facc_dir[((size_t) k * rows + lane) * 2];
            dy = st.facc_dir[((size_t) k * rows + lane) * 2 + 1];
        }
        again = has && !gate_accepts_regs(rx, ry, r_t, dx, dy, st.facc_t[k], rows, mts);
    }
    if (lane == 0) {
        st.init_has[k] = has ? 1u : 0u;   // (the frame the piece's results are now known to be right for)
        st.init_t[k] = r_t;
    }
    if (lane < rows) {
        st.init_dir[((size_t) k * rows + lane) * 2] = rx;   // (0 without a frame)
        st.init_dir[((size_t) k * rows + lane) * 2 + 1] = ry;
    }
    if (again) {
        if (lane == 0) st.rerun[k] = 1;
    } else if (st.active[k] && nacc_k == 0) {   // still before its first acceptance: the new frame is what its successes meet from now on
        if (lane == 0) {
            st.have_ref[k] = has ? 1u : 0u;
            st.ref_t[k] = r_t;
        }
        if (lane < rows) {
            st.ref_dir[((size_t) k * rows + lane) * 2] = rx;
            st.ref_dir[((size_t) k * rows + lane) * 2 + 1] = ry;
        }
    }
}

// The rows' line fits of every window of the pass that produced a grid (a 3 x 3 Jacobi eigen-decomposition per row: the bulk of
// the policy's arithmetic), a workgroup per window slot, a lane per row: all chains' windows side by side, so that the walk
// along a chain below is left with the gate's few operations per window.
__global__ __launch_bounds__(64) void adaptive_dir_kernel(uint32_t rows, uint32_t cols, const uint32_t *__restrict__ win_info,
                                                          const uint32_t *__restrict__ seg_off, const double *__restrict__ cand_xyr,
                                                          const int32_t *__restrict__ order, const uint32_t *__restrict__ found,
                                                          double *__restrict__ dirs /*[S][rows][2]*/) {
    const uint32_t w = blockIdx.x, lane = threadIdx.x;
    if (!(ECAL_WIN_STATUS(win_info[4 * w + 3]) == 0 && found[w]) || lane >= rows) return;
    const double *xyr = cand_xyr + 3 * (size_t) seg_off[2 * w];
    const int32_t *ord = order + (size_t) w * rows * cols;
    double dx, dy;
    row_direction(xyr, ord + lane * cols, cols, dx, dy);
    dirs[2 * ((size_t) w * rows + lane)] = dx;
    dirs[2 * ((size_t) w * rows + lane) + 1] = dy;
}

// One wave per piece: verdict of the pass -> gate -> keyframe record -> next window, up to n_levels times: as long as the
// verdict is the likely one, the next window is the next slot of the chain that this pass evaluated ahead of time.
// (A lock-step pass costs the latency of one workgroup through ~25 kernels plus the work of its windows — 0.63 ms for 1270
// windows of ~6 steps, 0.84 ms for three times the events —, and a piece's windows are a dependent chain: 2.7 links per pass
// at nine verdicts in ten as expected.)
// Round 5: the walk along the chain is wave-uniform arithmetic on registers.  It used to be lane 0 alone, two workgroup barriers
// and four dependent global reads per window (events of the window, status, grid found) — 100 - 140 us of every pass for chains
// of up to 48 windows, nothing but memory latency.  Now lane l reads those words of the chain's l-th window up front (one round
// trip for the whole chain) and every lane follows the same verdicts from them; the rare window that holds a grid (one in
// thirteen) goes through the gate as before (the rows' line fits come from adaptive_dir_kernel), its circles copied a lane each.
__global__ __launch_bounds__(64) void adaptive_step_kernel(uint32_t P, uint32_t rows, uint32_t cols, uint32_t max_levels,
                                     const uint32_t *__restrict__ win_info,
                                     const uint32_t *__restrict__ seg_off, const uint32_t *__restrict__ seg_cnt,
                                     const double *__restrict__ cand_xyr, const int32_t *__restrict__ order,
                                     const uint32_t *__restrict__ found, AdaptiveArrays st, double mts, uint32_t thr_events,
                                     uint32_t max_keys, double *__restrict__ kf_time, double *__restrict__ kf_dur,
                                     int32_t *__restrict__ kf_events, double *__restrict__ kf_feat, uint32_t *__restrict__ kf_piece,
                                     uint32_t *__restrict__ kf_gen, double *__restrict__ t0,
                                     double *__restrict__ t1, const int *__restrict__ overflow, const double *__restrict__ dirs) {
    const uint32_t k = blockIdx.x, lane = threadIdx.x;
    if (k == 0 && lane == 0 && *overflow) st.counters[3] = 1;  // keep it until the host looks
    if (k >= P || !st.active[k]) return;
    const uint32_t M = rows * cols;
    const double ln = 3 * mts;
    double f = st.first[k], s2 = st.second[k];
    const double hi = st.bound_hi[k];
    const uint32_t slot0 = st.slot0[k];
    uint32_t w = slot0;   // slot of the window under evaluation
    // the piece's slots of this pass: a tree of chains (ChainTree; adaptive_alloc_kernel) — the walk starts on the main chain and goes
    // down a level at every acceptance that has a chain behind it
    const ChainTree y = ChainTree::unpack(st.depth[k], st.lay[k]);
    const uint32_t D = y.len0;
    uint32_t lv = 0, cidx = 0, cbase = 0;   // level, number (within its level) and first slot (within the piece's) of the chain being walked
    uint32_t pos = 0, chain_len = D;   // position in that chain
    bool act = true, held = true;
    const uint32_t lev0 = st.levels[k];   // (max_levels: the caller's bound on the windows of a piece)
    const uint32_t d_all = y.longest_walk();
    const uint32_t n_levels = max_levels - lev0 < d_all ? max_levels - lev0 : d_all;
    if (n_levels == 0) return;
    // every slot's verdict words into LDS, all lanes at once (one round trip for the whole tree): events of the window
    // (EventFrame::eventsNum()) and bit 31 = extractFeatures() == true
    constexpr uint32_t AD_STEP_SLOTS = 1024;
    __shared__ uint32_t vw[AD_STEP_SLOTS];
    const uint32_t n_slots = y.slots() < AD_STEP_SLOTS ? y.slots() : AD_STEP_SLOTS;
    uint32_t pre_ok = 0;   // (this lane's main-chain window: D <= 48)
    for (uint32_t i = lane; i < n_slots; i += 64) {
        const uint32_t wl = slot0 + i;
        const uint32_t c_ = seg_cnt[2 * wl] + seg_cnt[2 * wl + 1];
        const uint32_t ok_ = (ECAL_WIN_STATUS(win_info[4 * wl + 3]) == 0 && found[wl]) ? 1u : 0u;
        vw[i] = (c_ & 0x7FFFFFFFu) | (ok_ << 31);
        if (i == lane && lane < D) pre_ok = ok_;
    }
    uint32_t have_ref = st.have_ref[k], nacc = st.nacc[k], nrej = st.nrej[k];
    // the reference frame on registers (lane i: row i) for the whole walk — it used to be read back from memory at every gate and
    // written at every acceptance; the first AD_STEP_PF grid-bearing windows' line fits with it, in the same round trip
    double rx = 0.0, ry = 0.0, ref_t = st.ref_t[k];
    if (lane < rows) {
        rx = st.ref_dir[(size_t) k * rows * 2 + 2 * lane];
        ry = st.ref_dir[(size_t) k * rows * 2 + 2 * lane + 1];
    }
    constexpr int AD_STEP_PF = 4;
    const unsigned long long okmask = __ballot(pre_ok != 0u);
    double pfx[AD_STEP_PF], pfy[AD_STEP_PF];
    {
        unsigned long long m = okmask;
#pragma unroll
        for (int j = 0; j < AD_STEP_PF; j++) {
            pfx[j] = pfy[j] = 0.0;
            if (m) {
                const uint32_t p = (uint32_t) __builtin_ctzll(m);
                m &= m - 1ull;
                if (lane < rows) {
                    pfx[j] = dirs[2 * ((size_t) (slot0 + p) * rows + lane)];
                    pfy[j] = dirs[2 * ((size_t) (slot0 + p) * rows + lane) + 1];
                }
            }
        }
    }
    // the accepted windows of this walk: their keyframe records are written behind it, all at once (an atomic with a return
    // value and two dependent reads per record were on the walk's path)
    constexpr uint32_t ACC_CAP = 96;
    __shared__ uint32_t acc_w[ACC_CAP], acc_cnt[ACC_CAP];
    __shared__ double acc_f[ACC_CAP], acc_s[ACC_CAP];
    uint32_t n_new = 0;
    uint32_t levels_done = 0;
    for (uint32_t level = 0; level < n_levels; level++) {
        if (cbase + pos >= AD_STEP_SLOTS) {   // (beyond what was staged: cannot happen with the layouts adaptive_alloc_kernel writes)
            held = false;
            break;
        }
        const uint32_t vword = vw[cbase + pos];
        const uint32_t cnt = vword & 0x7FFFFFFFu, okw = vword >> 31;
        bool accepted = false;
        if (okw) {
            double dx = 0.0, dy = 0.0;   // lane i: row i of this window's line fits
            const uint32_t nth_ok = lv ? (uint32_t) AD_STEP_PF : (uint32_t) __popcll(okmask & ((1ull << pos) - 1ull));
            if (nth_ok < (uint32_t) AD_STEP_PF) {
#pragma unroll
                for (int j = 0; j < AD_STEP_PF; j++)
                    if (nth_ok == (uint32_t) j) {
                        dx = pfx[j];
                        dy = pfy[j];
                    }
            } else if (lane < rows) {
                dx = dirs[2 * ((size_t) w * rows + lane)];
                dy = dirs[2 * ((size_t) w * rows + lane) + 1];
            }
            const double t_mid = (f + s2) / 2;  // eventCameraCalib.cpp:58
            accepted = true;
            if (have_ref)   // EventCalibIni::track: median row angle / time distance
                accepted = gate_accepts_regs(rx, ry, ref_t, dx, dy, t_mid, rows, mts);
            if (accepted) {
                if (lane == 0) {
                    acc_w[n_new] = w;
                    acc_cnt[n_new] = cnt;
                    acc_f[n_new] = f;
                    acc_s[n_new] = s2;
                }
                n_new++;
                // the new reference frame (and, for the piece's first acceptance, what the shared-map verification looks at)
                rx = dx;
                ry = dy;
                ref_t = t_mid;
                if (nacc == 0) {
                    if (lane < rows) {
                        st.facc_dir[(size_t) k * rows * 2 + 2 * lane] = dx;
                        st.facc_dir[(size_t) k * rows * 2 + 2 * lane + 1] = dy;
                    }
                    if (lane == 0) st.facc_t[k] = t_mid;
                }
                have_ref = 1;
                nacc++;
            } else if (nacc == 0) {   // a success rejected before the first acceptance
                if (nrej < AD_NREJ) {
                    if (lane == 0) st.rej_t[(size_t) k * AD_NREJ + nrej] = t_mid;
                    if (lane < rows) {
                        st.rej_dir[((size_t) k * AD_NREJ + nrej) * rows * 2 + 2 * lane] = dx;
                        st.rej_dir[((size_t) k * AD_NREJ + nrej) * rows * 2 + 2 * lane + 1] = dy;
                    }
                }
                nrej++;
            }
        }
        const int o = accepted ? 0 : ((cnt > thr_events || (s2 - f) > 3 * ln) ? 1 : 2);
        const int likely = likely_outcome(f, s2, mts);
        double nf, ns;
        next_window(o, f, s2, mts, nf, ns);
        f = nf;
        s2 = ns;
        act = ns < hi;  // :50
#ifdef ECAL_ADAPTIVE_STATS
        if (lane == 0) {
            if (o == 0) atomicAdd(&st.counters[4], 1u);
            else if (o != likely) atomicAdd(&st.counters[5], 1u);
            else if (level + 1 == n_levels) atomicAdd(&st.counters[6], 1u);
            if (!act) atomicAdd(&st.counters[7], 1u);
        }
#endif
        levels_done++;
        if (!act) {
            held = false;
            break;
        }
        if (n_new == ACC_CAP) {   // (the list is full: the piece goes on from here in the next pass)
            held = false;
            break;
        }
        if (o == likely) {   // the chain evaluated ahead holds the likely successor …
            pos++;
            if (pos >= chain_len) {
                held = lv == 0u;
                break;
            }
            w++;
            continue;
        }
        if (o == 0 && lv < 3u) {   // … and, behind some of its windows, the chain of the next level: what follows an acceptance
            const uint32_t ca = lv == 0u ? y.a1 : (lv == 1u ? y.a2 : y.a3), cc = lv == 0u ? y.c1 : (lv == 1u ? y.c2 : y.c3);
            if (pos >= ca && pos < ca + cc) {
                const uint32_t nlen = lv == 0u ? y.len1 : (lv == 1u ? y.len2 : y.len3);
                const uint32_t noff = lv == 0u ? y.off1() : (lv == 1u ? y.off2() : y.off3());
                cidx = cidx * cc + (pos - ca);
                cbase = noff + cidx * nlen;
                w = slot0 + cbase;
                lv++;
                pos = 0;
                chain_len = nlen;
                continue;
            }
        }
        held = false;
        break;
    }
    if (n_new) {
        // the frame the piece goes on with, and the walk's keyframe records: one reservation, the circles of all of them side by side
        if (lane < rows) {
            st.ref_dir[(size_t) k * rows * 2 + 2 * lane] = rx;
            st.ref_dir[(size_t) k * rows * 2 + 2 * lane + 1] = ry;
        }
        uint32_t base = 0;
        if (lane == 0) {
            st.ref_t[k] = ref_t;
            base = atomicAdd(&st.counters[1], n_new);
        }
        base = (uint32_t) __shfl((int) base, 0, 64);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // (lane 0's list, read by the others)
        __builtin_amdgcn_wave_barrier();
        const uint32_t gen = st.gen[k];
        for (uint32_t a = lane; a < n_new; a += 64) {
            const uint32_t at = base + a;
            if (at < max_keys) {
                kf_time[at] = (acc_f[a] + acc_s[a]) / 2;
                kf_dur[2 * at] = acc_f[a];
                kf_dur[2 * at + 1] = acc_s[a];
                kf_events[at] = (int32_t) acc_cnt[a];
                kf_piece[at] = k;
                kf_gen[at] = gen;
            }
        }
        for (uint32_t i = lane; i < n_new * M; i += 64) {
            const uint32_t a = i / M, c = i - a * M, at = base + a;
            if (at >= max_keys) continue;
            const uint32_t wa = acc_w[a];
            const double *xyr = cand_xyr + 3 * (size_t) seg_off[2 * wa];
            const size_t src = 3 * (size_t) order[(size_t) wa * M + c];
            kf_feat[3 * ((size_t) at * M + c)] = xyr[src];
            kf_feat[3 * ((size_t) at * M + c) + 1] = xyr[src + 1];
            kf_feat[3 * ((size_t) at * M + c) + 2] = xyr[src + 2];
        }
    }
    if (lane != 0) return;
    st.have_ref[k] = have_ref;
    st.nacc[k] = nacc;
    st.nrej[k] = nrej;
    st.levels[k] = lev0 + levels_done;
    atomicAdd(st.windows, (unsigned long long) levels_done);
    st.first[k] = f;
    st.second[k] = s2;
    st.active[k] = act ? 1u : 0u;
    // a piece whose whole chain held asks for twice the length next time, one whose chain broke for the base length again (a
    // stretch without the pattern is hundreds of windows whose verdicts are all the likely one: it is the search's longest path)
    st.want[k] = held ? 2u * D : 0u;
    if (act && lev0 + levels_done < max_levels) atomicAdd(&st.counters[0], 1u);   // pieces with windows still to go
}

}  // namespace

extern "C" uint64_t ecal_detect_keyframes_cap_hint(const ecal_adaptive_params *ap, uint64_t n_events) {
    if (!ap || ap->piece_num == 0 || !(ap->motion_time_step > 0) || !(ap->end_time > ap->start_time)) return 0;
    if ((uint64_t) ap->piece_first + ap->piece_count > ap->piece_num) return 0;
    const uint32_t P = ap->piece_count ? ap->piece_count : ap->piece_num;
    const uint32_t D = adaptive_slots_per_piece(P);
    // a pass holds D windows per piece, of three to ten time steps, grown along the chain: eight steps each at the stream's
    // mean rate and some room (too little costs one aborted attempt of a few passes: ECAL_ERR_RANGE, doubled, again)
    const double per_step = (double) n_events * ap->motion_time_step / (ap->end_time - ap->start_time);
    const double want = (double) P * D * (8.0 * per_step + 256.0);
    const double most = (double) D * (double) n_events + 4096.0;   // (the windows of one slot index are disjoint)
    const double cap = want < most ? want : most;
    return cap > 4294967232.0 ? 4294967232ull : (uint64_t) cap;
}

// the same estimate from the events that lie in [start_time, end_time] (two binary searches on the resident stream): a search
// over part of a stream is sized for that part
extern "C" uint64_t ecal_detect_keyframes_cap_hint_dev(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events,
                                                       const ecal_adaptive_params *ap) {
    if (!ctx || !ap || (n_events && !d_events)) return 0;
    if (n_events == 0) return ecal_detect_keyframes_cap_hint(ap, 0);
    if (hipSetDevice(ctx->device) != hipSuccess) return 0;
    ecal_devbuf *B = ctx->host_pipe;
    if (ecal_ensure(ctx, B[0], 2 * sizeof(double)) || ecal_ensure(ctx, B[2], 4) || ecal_ensure(ctx, B[3], 4) || ecal_ensure(ctx, B[4], 8)) return 0;
    const double t[2] = {ap->start_time, ap->end_time};
    uint32_t lo = 0, hi = 0;
    hipStream_t st = ctx->stream;
    if (hipMemcpyAsync(B[0].ptr, t, sizeof(t), hipMemcpyHostToDevice, st) != hipSuccess) return 0;
    if (ecal_window_bounds_dev(ctx, d_events, n_events, (const double *) B[0].ptr, (const double *) B[0].ptr + 1, 1, (uint32_t *) B[2].ptr,
                               (uint32_t *) B[3].ptr, (uint32_t *) B[4].ptr, st))
        return 0;
    if (hipMemcpyAsync(&lo, B[2].ptr, 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(&hi, B[3].ptr, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return 0;
    const uint64_t hint = ecal_detect_keyframes_cap_hint(ap, hi > lo ? hi - lo : 0);
    return hint < 4096 ? 4096 : hint;
}

static int detect_keyframes_impl(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, const ecal_adaptive_params *ap,
                                 const ecal_detect_params *prm, uint32_t cap_points, uint32_t max_keyframes, double *kf_time,
                                 double *kf_duration, int32_t *kf_events_num, double *kf_features, uint32_t *n_keyframes,
                                 uint32_t *passes, uint64_t *windows, const ecal_adaptive_handover *ho);
static int detect_keyframes_entry(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, const ecal_adaptive_params *ap,
                                  const ecal_detect_params *prm, uint32_t cap_points, uint32_t max_keyframes, double *kf_time,
                                  double *kf_duration, int32_t *kf_events_num, double *kf_features, uint32_t *n_keyframes,
                                  uint32_t *passes, uint64_t *windows, const ecal_adaptive_handover *ho);
extern "C" int ecal_detect_keyframes(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, const ecal_adaptive_params *ap,
                                     const ecal_detect_params *prm, uint32_t cap_points, uint32_t max_keyframes, double *kf_time,
                                     double *kf_duration, int32_t *kf_events_num, double *kf_features, uint32_t *n_keyframes,
                                     uint32_t *passes, uint64_t *windows) {
    return detect_keyframes_entry(ctx, d_events, n_events, ap, prm, cap_points, max_keyframes, kf_time, kf_duration, kf_events_num,
                                  kf_features, n_keyframes, passes, windows, nullptr);
}
// The shared-map search of ONE stream cut over several callers (ranks): this call runs the pieces ap->piece_first ..
// piece_first + piece_count - 1 (contiguous in time; piece 0 is the last in time) with the gate of the reference's single-worker
// run.  What those pieces need from the pieces before them (larger indices, earlier in time: another caller's) is ONE frame — the
// last keyframe's time stamp and row directions (EventCalibIni.cpp:26-36: the map's last keyframe) —, so the callers form a chain in
// time: ho->recv delivers the frame of everything before this call's pieces (polled between passes with wait = 0, then waited
// for; never called when the subset holds the run's first piece), ho->send passes on the frame behind this call's pieces (its own
// last keyframe, else the received frame) once it is final.  Until the frame arrives the pieces run as a speculation without one;
// the pass-by-pass verification (adaptive_verify_live_kernel) then re-runs exactly the pieces whose verdicts it changes — as it
// does for the pieces of one call.  The union of the callers' keyframes is the single call's over all pieces, record for record
// (tests/test_gpu_bench_multirank.py).
extern "C" int ecal_detect_keyframes_sharded(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, const ecal_adaptive_params *ap,
                                             const ecal_detect_params *prm, uint32_t cap_points, uint32_t max_keyframes, double *kf_time,
                                             double *kf_duration, int32_t *kf_events_num, double *kf_features, uint32_t *n_keyframes,
                                             uint32_t *passes, uint64_t *windows, const ecal_adaptive_handover *ho) {
    if (!ho || !ho->send || !ho->recv || !ap || ap->gate_mode != ECAL_GATE_SHARED_MAP || ap->piece_count == 0) return ECAL_ERR_INVALID;
    return detect_keyframes_entry(ctx, d_events, n_events, ap, prm, cap_points, max_keyframes, kf_time, kf_duration, kf_events_num,
                                  kf_features, n_keyframes, passes, windows, ho);
}
static int detect_keyframes_entry(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, const ecal_adaptive_params *ap,
                                     const ecal_detect_params *prm, uint32_t cap_points, uint32_t max_keyframes, double *kf_time,
                                     double *kf_duration, int32_t *kf_events_num, double *kf_features, uint32_t *n_keyframes,
                                  uint32_t *passes, uint64_t *windows, const ecal_adaptive_handover *ho) {
    const ecal_range range__(ctx, "ecal_detect_keyframes");
    if (!ctx) return ECAL_ERR_INVALID;
    // the search's windows are three to ten steps long: second-tier work by design — its passes never take the lean form of a
    // stage (one slow general launch behind the first pass), but they do drop the tiers behind the second pass while those
    // find nothing (ecal_tail_plan: eight launches of ~5 us + their gaps in a chain of ~27, every pass)
    const int was = ctx->tail_mode;
    const bool was_no_lean = ctx->tail_no_lean;
    ctx->tail_no_lean = true;   // (AUTO stays AUTO: the tiers behind the second may still be dropped, ecal_tail_plan)
    const int rc = detect_keyframes_impl(ctx, d_events, n_events, ap, prm, cap_points, max_keyframes, kf_time, kf_duration, kf_events_num,
                                         kf_features, n_keyframes, passes, windows, ho);
    ctx->tail_mode = was;
    ctx->tail_no_lean = was_no_lean;
    ctx->latency_pass = 0;
    ctx->overflow_sticky = nullptr;
    return rc;
}
static int detect_keyframes_impl(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, const ecal_adaptive_params *ap,
                                 const ecal_detect_params *prm, uint32_t cap_points, uint32_t max_keyframes, double *kf_time,
                                 double *kf_duration, int32_t *kf_events_num, double *kf_features, uint32_t *n_keyframes,
                                 uint32_t *passes, uint64_t *windows, const ecal_adaptive_handover *ho) {
    if (!ctx || !ap || !prm || !n_keyframes || (n_events && !d_events)) return ECAL_ERR_INVALID;
    *n_keyframes = 0;
    if (passes) *passes = 0;
    if (windows) *windows = 0;
    const uint32_t P = ap->piece_count ? ap->piece_count : ap->piece_num, M = prm->rows * prm->cols, rows = prm->rows;
    const bool subset = ap->piece_count != 0 && ap->piece_count != ap->piece_num;
    if ((uint64_t) ap->piece_first + ap->piece_count > ap->piece_num || (!ap->piece_count && ap->piece_first) ||
        (subset && ap->gate_mode == ECAL_GATE_SHARED_MAP && !ho)) {
        ctx->last_error = "ecal_detect_keyframes: piece_first + piece_count beyond piece_num, or a subset of the pieces under the shared-map gate "
                          "(a piece's gate frame comes from the pieces before it: ecal_detect_keyframes_sharded hands it over)";
        return ECAL_ERR_INVALID;
    }
    if (P == 0 || M == 0 || M > 128 || prm->rows > (uint32_t) AD_MAX_ROWS || !(ap->motion_time_step > 0) ||
        !(ap->end_time > ap->start_time) || (max_keyframes && (!kf_time || !kf_duration || !kf_events_num || !kf_features)) ||
        (ap->gate_mode != ECAL_GATE_OWN_PIECE && ap->gate_mode != ECAL_GATE_SHARED_MAP)) {
        ctx->last_error = "ecal_detect_keyframes: invalid parameters";
        return ECAL_ERR_INVALID;
    }
    const bool shared = ap->gate_mode == ECAL_GATE_SHARED_MAP;
    ECAL_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    int rc;
    const uint32_t D = adaptive_slots_per_piece(P, ctx->sw.adaptive_depth), d_max = adaptive_depth_max(ctx->sw.adaptive_depth_max);
    const uint32_t S = D * P;   // window slots per pass, dealt out among the pieces still at work: the current window and the likely chain after it
    const size_t cap = (size_t) cap_points + 16;
    ecal_devbuf *B = ctx->host_pipe;  // roles as in ecal_detect_pass; 0 holds t0 and t1 back to back
    const auto t_scratch0 = std::chrono::steady_clock::now();
    const size_t sizes[17] = {2ul * S * sizeof(double), 16, S * 4ul, S * 4ul, (S + 1) * 4ul, cap * 16, 2ul * S * 4, 2ul * S * 4, cap * 4,
                              cap * 4, 2ul * S * 4, cap * 4, cap * 4, 4ul * S * 4, cap * 8, cap * 24, 16};
    for (int i = 0; i < 17; i++)
        if ((rc = ecal_ensure(ctx, B[i], sizes[i]))) return rc;
    if ((rc = ecal_ensure(ctx, ctx->host_grid_order, (size_t) S * M * sizeof(int32_t)))) return rc;
    if ((rc = ecal_ensure(ctx, ctx->host_grid_found, (size_t) S * sizeof(uint32_t)))) return rc;
    // per piece: 7 doubles + (3 + AD_NREJ) x rows x 2 doubles of row directions + AD_NREJ doubles + 10 words
    const size_t state_bytes = (size_t) P * (8 * (7 + AD_NREJ + 2 * (size_t) rows * (3 + AD_NREJ)) + 4 * 12) + 128 + 8 * (2 + 2 * (size_t) AD_MAX_ROWS);
    if ((rc = ecal_ensure(ctx, ctx->adaptive_state, state_bytes))) return rc;
    if ((rc = ecal_ensure(ctx, ctx->adaptive_dirs, (size_t) S * rows * 2 * sizeof(double)))) return rc;
    // keyframe records: in shared-map mode the pieces that run again leave dead records behind
    const size_t rec_cap = shared ? 2 * (size_t) max_keyframes + P + 64 : (size_t) max_keyframes;
    if (rec_cap > 0xFFFFFFF0ull) return ECAL_ERR_RANGE;
    const size_t key_stride = 8 + 16 + 8 + 24 * (size_t) M + 8;  // time, duration, events (padded), features, piece + run
    if ((rc = ecal_ensure(ctx, ctx->adaptive_keys, rec_cap * key_stride + 64))) return rc;
    if (ctx->pass_pinned_cap < 64) {
        if (ctx->pass_pinned) (void) hipHostFree(ctx->pass_pinned);
        ctx->pass_pinned = nullptr;
        ctx->pass_pinned_cap = 0;
        ECAL_HIP_TRY(ctx, hipHostMalloc((void **) &ctx->pass_pinned, 4096, hipHostMallocDefault));
        ctx->pass_pinned_cap = 4096;
    }
    if (ctx->sw.adaptive_trace)
        fprintf(stderr, "ecal_detect_keyframes: scratch buffers ready after %.4f s (cap_points %u)\n",
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_scratch0).count(), cap_points);
    AdaptiveArrays a;
    {
        unsigned char *p = (unsigned char *) ctx->adaptive_state.ptr;
        a.windows = (unsigned long long *) p;
        a.counters = (uint32_t *) (p + 16);   // [16]
        p += 128;
        double *d = (double *) p;
        a.first = d, d += P;
        a.second = d, d += P;
        a.bound_hi = d, d += P;
        a.ref_t = d, d += P;
        a.init_t = d, d += P;
        a.facc_t = d, d += P;
        a.rej_t = d, d += (size_t) P * AD_NREJ;
        a.ref_dir = d, d += (size_t) P * rows * 2;
        a.init_dir = d, d += (size_t) P * rows * 2;
        a.facc_dir = d, d += (size_t) P * rows * 2;
        a.rej_dir = d, d += (size_t) P * AD_NREJ * rows * 2;
        uint32_t *u = (uint32_t *) d;
        a.active = u, u += P;
        a.have_ref = u, u += P;
        a.levels = u, u += P;
        a.slot0 = u, u += P;
        a.depth = u, u += P;
        a.lay = u, u += P;
        a.gen = u, u += P;
        a.nacc = u, u += P;
        a.nrej = u, u += P;
        a.init_has = u, u += P;
        a.rerun = u, u += P;
        a.want = u, u += P;
        // (the handed-over frame behind everything else, 8-byte aligned: P words x 12 are a multiple of 8 bytes only for even P)
        double *xf = (double *) (((uintptr_t) u + 7u) & ~(uintptr_t) 7u);
        const bool has_pred = ho && (uint64_t) ap->piece_first + ap->piece_count < ap->piece_num;
        a.ext = has_pred ? (const uint32_t *) xf : nullptr;
        a.ext_frame = has_pred ? xf + 1 : nullptr;
        a.start_time = ap->start_time;
        a.end_time = ap->end_time;
        a.p_total = ap->piece_num;
        a.p_first = ap->piece_count ? ap->piece_first : 0u;
    }
    const uint32_t max_keys = (uint32_t) rec_cap;
    double *d_kt = (double *) ctx->adaptive_keys.ptr, *d_kd = d_kt + max_keys, *d_kf = d_kd + 2 * (size_t) max_keys;
    int32_t *d_ke = (int32_t *) (d_kf + 3 * (size_t) max_keys * M);
    uint32_t *d_kp = (uint32_t *) (d_ke + max_keys + (max_keys & 1u)), *d_kg = d_kp + max_keys;
    double *d_t0 = (double *) B[0].ptr, *d_t1 = d_t0 + S;
    hipLaunchKernelGGL(adaptive_init_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, rows, ap->motion_time_step, a);
    ECAL_HIP_TRY(ctx, hipMemsetAsync(B[16].ptr, 0, sizeof(int), st));
    ctx->overflow_sticky = (const int *) B[16].ptr;   // (wiped here, once: the passes' slicing calls leave it alone)
    if (a.ext) ECAL_HIP_TRY(ctx, hipMemsetAsync((void *) a.ext, 0, 8 * (2 + 2 * (size_t) AD_MAX_ROWS), st));   // (the frame from before: not known yet)
    uint32_t *h = reinterpret_cast<uint32_t *>(ctx->pass_pinned);  // [0..15] counters
    // The host runs `ahead` passes ahead of the device's counters: before it enqueues pass p it waits for the counters that pass
    // p - ahead left (a copy to pinned memory + an event behind every pass) — the device always has work queued, and at most
    // `ahead` passes run with nothing left to do.
    const uint32_t ahead = ap->check_every ? (ap->check_every < 8u ? ap->check_every : 8u) : 2u;
    // [8][4] in pinned memory: what the last kernel of a pass reports (adaptive_alloc_kernel) — active pieces, records, the
    // pass's number, overflow.  The host polls the number: no copy, no event between the kernels.
    volatile uint32_t *ring = h + 16;
    uint32_t *d_ring = nullptr;
    ECAL_HIP_TRY(ctx, hipHostGetDevicePointer((void **) &d_ring, (void *) (h + 16), 0));
    for (int i = 0; i < 32; i++) ring[i] = 0;
    std::vector<uint32_t> trace_active, trace_windows;   // ECAL_TRACE=adaptive: pieces still at work after every pass, windows gone through so far
    uint32_t seq = 0;   // passes enqueued in this call (over all its sets of runs): the number a pass reports
    uint32_t last_active = P;   // pieces at work after the last pass that has reported
    // max_passes bounds the windows a piece goes through (the lock-step passes of the one-window-per-pass form); a pass here
    // takes a piece through up to D of them
    const uint32_t max_levels = ap->max_passes ? ap->max_passes : 0xFFFFFFFFu;
    uint32_t n_passes = 0;
    const char *const range_msg = "cap_points is smaller than the number of events covered by the windows of one pass";
    // every error return below leaves nothing in flight on the stream (passes write the pinned ring and the context's scratch)
#define AD_TRY(call)                                  \
    do {                                              \
        if ((rc = (call))) {                          \
            (void) hipStreamSynchronize(st);          \
            return rc;                                \
        }                                             \
    } while (0)
    auto hip_rc = [&](hipError_t e, const char *what) -> int {
        if (e == hipSuccess) return ECAL_OK;
        ctx->last_error = std::string(what) + ": " + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? ECAL_ERR_NOMEM : ECAL_ERR_HIP;
    };
    // the frame handed over from the pieces before this call's (ecal_detect_keyframes_sharded): asked for between the passes, staged
    // in pinned memory, copied in on the search's stream — a pass sees all of it or none of it
    bool ext_known = a.ext == nullptr;
    ecal_keyframe_frame ext_f;
    memset(&ext_f, 0, sizeof(ext_f));
    // the first pass that sees the frame: a report "no piece at work" from an EARLIER pass does not end the search — the frame may
    // start pieces again that had finished (without it such a report is final: nothing restarts a search whose pieces are all
    // finished and verified)
    uint32_t ext_first_pass = 0;
    auto take_ext = [&](int wait, uint32_t next_pass) -> int {
        if (ext_known) return ECAL_OK;
        ecal_keyframe_frame f;
        memset(&f, 0, sizeof(f));
        const int r = ho->recv(ho->user, &f, wait);
        if (r < 0) {
            ctx->last_error = "ecal_detect_keyframes_sharded: the recv callback failed";
            return ECAL_ERR_INVALID;
        }
        if (r == 0) return ECAL_OK;
        double *stage = reinterpret_cast<double *>(ctx->pass_pinned) + 64;   // (bytes 512 ..: behind the counters and the report ring)
        const uint32_t word[2] = {f.has ? 2u : 1u, 0u};
        memcpy(stage, word, sizeof(word));
        stage[1] = f.time;
        for (uint32_t i = 0; i < 2 * rows; i++) stage[2 + i] = f.dir[i];
        const hipError_t e = hipMemcpyAsync((void *) a.ext, stage, 8 * (2 + 2 * (size_t) rows), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return hip_rc(e, "hipMemcpyAsync");
        ext_f = f;
        ext_known = true;
        ext_first_pass = next_pass;
        return ECAL_OK;
    };
    // the lock-step passes of one set of runs: until no piece has a window left
    uint32_t deal = S;   // window slots dealt out per pass (all of them in the first set of runs)
    // shared-map gate: the verification runs after every pass (adaptive_verify_live_kernel).  Slots are dealt piece by piece
    // (adaptive_alloc_kernel: a piece's chain doubles while it holds) under both gates.  (Until round 5 the library also kept the
    // verification after every SET of runs and a uniform deal behind switches: 0.164 against 0.116 s, 0.065 against 0.059 s at 1270
    // pieces — profiles/experiments/r06_adaptive_alternatives.patch.)
    const uint32_t live_side = ctx->sw.adaptive_side < 0 ? AD_SIDE_DEFAULT : (ctx->sw.adaptive_side == 0 ? AD_SIDE_NONE : (ctx->sw.adaptive_side == 1 ? AD_SIDE_MEASURED : (uint32_t) ctx->sw.adaptive_side));   // (ECAL_ADAPTIVE_SHAPE side: from | count << 8 | length << 16 | main << 24; 1 = the measured layout, 0 = none, unset = the default)
    // (side chains hang behind a main chain of (live_side >> 24) windows: a search whose chains are capped below that — the debug
    // switch ECAL_ADAPTIVE_SHAPE depth_max — runs without them)
    const uint32_t live_side_eff = d_max >= (live_side >> 24) ? live_side : AD_SIDE_NONE;
    // the tree of chains a piece gets where keyframes are to be expected (tree_for; ECAL_ADAPTIVE_SHAPE tree: 0 = none, else the word)
    const uint32_t live_tree_asked = ctx->sw.adaptive_tree < 0 ? AD_TREE_DEFAULT : (uint32_t) ctx->sw.adaptive_tree;
    const uint32_t live_tree = d_max >= (live_tree_asked & 0xFFu) ? live_tree_asked : 0u;   // (as the side chains: not under a chain cap below its main chain)
#ifndef ECAL_AD_LIVE_FLOOR
#define ECAL_AD_LIVE_FLOOR 640u
#endif
    // slots a pass has to spare for few pieces at work.  1024 until round 6 (swept in round 5, profiles/r05_notes.md); with the passes
    // of round 6 (s at 1270 / 4096 pieces, the tree above): 384: 0.0877 / 0.112 - 0.118; 512: 0.0864 - 0.0867 / 0.0875 - 0.0893;
    // 640: 0.0864 / 0.0855 - 0.0859; 768: 0.0866 - 0.0870 / 0.0862; 1024: 0.0885 / 0.086; 1536, 2048: 0.093 - 0.094 — a tail pass's time
    // is the sum over the stages of their SLOWEST window, and every window more is another draw
    const uint32_t live_floor = ECAL_AD_LIVE_FLOOR;
    auto run_passes = [&]() -> int {
        // the window slots of this set of runs: `deal` of the S there are (a verification round of a few pieces launches its
        // kernels over the slots it deals out, not over all S: thousands of workgroups that find an empty window still cost
        // tens of microseconds per kernel, and a round is a chain of ~10 passes of ~25 kernels)
        const uint32_t Sr = deal < S ? deal : S;
        hipLaunchKernelGGL(adaptive_alloc_kernel, dim3(1), dim3(AD_ALLOC_T), 0, st, P, Sr, deal, d_max, a, ap->motion_time_step, d_t0, d_t1,
                           (uint32_t *) nullptr, 0u, D, live_floor, live_side_eff, live_tree, 0u);
        const uint32_t seq0 = seq;   // this set's pass `pass` reports seq0 + pass + 1 into slot (seq0 + pass) % 8
        // the slots of the pass being enqueued: all of them, or AD_TAIL_SLOTS in the search's tail — a launch over 7620 window slots
        // of which a few hundred hold a window still pays for the empty workgroups (~1 us per thousand and kernel), and the few
        // pieces at work there cannot ask for more slots than that (adaptive_alloc_kernel: live_floor shared out, trees of <= 688)
        uint32_t Sw = Sr;
        for (uint32_t pass = 0; pass < max_levels; pass++) {
            if (pass >= ahead) {
                const uint32_t want = seq0 + pass - ahead + 1u, q = (want - 1u) % 8u;
                auto t_wait = std::chrono::steady_clock::now();
                for (uint32_t spin = 0; ring[4 * q + 2] != want; spin++) {
                    for (int i = 0; i < 8; i++) __builtin_ia32_pause();
                    if ((spin & 0x3FFFu) == 0x3FFFu && std::chrono::steady_clock::now() - t_wait > std::chrono::seconds(2)) {
                        // overdue: a stream that has stopped (an error) will never report
                        const hipError_t qe = hipStreamQuery(st);
                        if (qe != hipErrorNotReady && ring[4 * q + 2] != want) {
                            AD_TRY(hip_rc(qe == hipSuccess ? hipErrorUnknown : qe, "a pass of the keyframe search did not report"));
                        }
                        t_wait = std::chrono::steady_clock::now();
                    }
                }
                std::atomic_thread_fence(std::memory_order_acquire);
                last_active = (uint32_t) ring[4 * q];
                if (ctx->sw.adaptive_trace) {
                    trace_active.push_back((uint32_t) ring[4 * q]);
                    trace_windows.push_back((uint32_t) ring[4 * q + 1]);
                }
                if (ring[4 * q + 3]) {
                    (void) hipStreamSynchronize(st);
                    ctx->last_error = range_msg;
                    if (ctx->sw.adaptive_trace) fprintf(stderr, "ecal_detect_keyframes: cap_points %u too small after %u passes\n", cap_points, n_passes);
                    return ECAL_ERR_RANGE;
                }
                if (ring[4 * q] == 0 && want >= ext_first_pass) break;   // (the passes enqueued since find nothing to do)
            }
            n_passes++;
            seq++;
            AD_TRY(take_ext(0, seq));
            // (few pieces still at work — known two passes late —: the stages' latency forms, ecal_ctx::latency_pass)
            ctx->latency_pass = last_active <= AD_THIRD_IN_ONE_PIECES ? 2 : (last_active <= AD_STAGE_LATENCY_PIECES ? 1 : 0);
            AD_TRY(ecal_window_bounds_dev(ctx, d_events, n_events, d_t0, d_t1, Sw, (uint32_t *) B[2].ptr, (uint32_t *) B[3].ptr,
                                          (uint32_t *) B[4].ptr, st));
            AD_TRY(ecal_slice_events_dev(ctx, d_events, n_events, (uint32_t *) B[2].ptr, (uint32_t *) B[3].ptr, (uint32_t *) B[4].ptr, Sw, 0,
                                         cap_points, (double *) B[5].ptr, (uint32_t *) B[6].ptr, (uint32_t *) B[7].ptr,
                                         (int32_t *) B[8].ptr, (int *) B[16].ptr, st));
            AD_TRY(ecal_dbscan_batch_dev(ctx, (double *) B[5].ptr, (uint32_t *) B[6].ptr, (uint32_t *) B[7].ptr, 2 * Sw, cap_points, 0,
                                         prm->dbscan_eps, prm->dbscan_min_samples, (int32_t *) B[9].ptr, (uint32_t *) B[10].ptr, st));
            AD_TRY(ecal_extract_for_ctx(ctx, (double *) B[5].ptr, (uint32_t *) B[6].ptr, (uint32_t *) B[7].ptr, (int32_t *) B[9].ptr,
                                        (uint32_t *) B[10].ptr, Sw, cap_points, prm->dbscan_eps, prm->cluster_min_sample, prm->need_clusters,
                                        prm->circle_radius_threshold, prm->fit_circle, prm->knn_num, (uint32_t *) B[13].ptr,
                                        (uint32_t *) B[14].ptr, (double *) B[15].ptr, (int32_t *) B[11].ptr, (uint32_t *) B[12].ptr, st));
            // (few pieces still at work — known two passes late —: the grid finder's latency form, a wave per start, ecal_grid.hip)
            ctx->grid_hint_windows = last_active <= AD_GRID_LATENCY_PIECES ? 1u : Sw;
            // (the rows' line fits of the windows that hold a grid: by the grid finder's own workgroups, under the launch's slowest
            // failing window; patterns of more than 64 rows: by adaptive_dir_kernel behind it — same values)
            const bool dirs_in_grid = prm->rows <= 64;
            rc = ecal_grid_order_dirs_dev(ctx, (uint32_t *) B[13].ptr, (uint32_t *) B[6].ptr, (double *) B[15].ptr, Sw, prm->rows, prm->cols,
                                          (int32_t *) ctx->host_grid_order.ptr, (uint32_t *) ctx->host_grid_found.ptr,
                                          dirs_in_grid ? (double *) ctx->adaptive_dirs.ptr : nullptr, st);
            ctx->grid_hint_windows = 0;
            AD_TRY(rc);
            if (!dirs_in_grid)
                hipLaunchKernelGGL(adaptive_dir_kernel, dim3(Sw), dim3(64), 0, st, prm->rows, prm->cols, (const uint32_t *) B[13].ptr,
                                   (const uint32_t *) B[6].ptr, (const double *) B[15].ptr, (const int32_t *) ctx->host_grid_order.ptr,
                                   (const uint32_t *) ctx->host_grid_found.ptr, (double *) ctx->adaptive_dirs.ptr);
            hipLaunchKernelGGL(adaptive_step_kernel, dim3(P), dim3(64), 0, st, P, prm->rows, prm->cols, max_levels,
                               (const uint32_t *) B[13].ptr, (const uint32_t *) B[6].ptr, (const uint32_t *) B[7].ptr,
                               (const double *) B[15].ptr, (const int32_t *) ctx->host_grid_order.ptr,
                               (const uint32_t *) ctx->host_grid_found.ptr, a, ap->motion_time_step, ap->frame_event_num_threshold,
                               max_keys, d_kt, d_kd, d_ke, d_kf, d_kp, d_kg, d_t0, d_t1, (const int *) B[16].ptr,
                               (const double *) ctx->adaptive_dirs.ptr);
            // shared-map gate: verification and restarts pass by pass
            // (... and the re-runs start inside the slot allocation's launch: restart_piece)
            if (shared)
                hipLaunchKernelGGL(adaptive_verify_live_kernel, dim3((P + 3) / 4), dim3(256), 0, st, P, rows, ap->motion_time_step, a);
            const uint32_t Sn = last_active <= AD_TAIL_PIECES && Sr > AD_TAIL_SLOTS ? AD_TAIL_SLOTS : Sr;   // (the next pass's slots)
            hipLaunchKernelGGL(adaptive_alloc_kernel, dim3(1), dim3(AD_ALLOC_T), 0, st, P, Sn, deal, d_max, a, ap->motion_time_step, d_t0, d_t1,
                               d_ring + 4 * ((seq - 1u) % 8u), seq, D, live_floor, live_side_eff, live_tree, shared ? rows : 0u);
            Sw = Sn;
        }
        AD_TRY(hip_rc(hipStreamSynchronize(st), "hipStreamSynchronize"));
        AD_TRY(hip_rc(hipMemcpy(h, a.counters, 16 * sizeof(uint32_t), hipMemcpyDeviceToHost), "hipMemcpy"));
        if (h[3]) {   // (an overflow may sit in any of the last passes)
            ctx->last_error = range_msg;
            if (ctx->sw.adaptive_trace) fprintf(stderr, "ecal_detect_keyframes: cap_points %u too small (seen at the end, %u passes)\n", cap_points, n_passes);
            return ECAL_ERR_RANGE;
        }
        return ECAL_OK;
    };
    if ((rc = run_passes())) return rc;
    if (!ext_known) {
        // every piece has run to its end on the speculation that nothing comes before it: now the frame is needed.  With it the
        // verification of the next pass re-runs the pieces whose verdicts it changes (usually the first one or two in time)
        AD_TRY(take_ext(1, seq + 1u));
        if (!ext_known) {
            ctx->last_error = "ecal_detect_keyframes_sharded: recv(wait = 1) returned without a frame";
            return ECAL_ERR_INVALID;
        }
        if ((rc = run_passes())) return rc;
    }
    const bool trace = ctx->sw.adaptive_trace;
#undef AD_TRY
#ifdef ECAL_ADAPTIVE_STATS
    fprintf(stderr, "chain ends: keyframe %u, other verdict %u, chain used up %u, piece finished %u\n", h[4], h[5], h[6], h[7]);
#endif
    if (trace) {
        fprintf(stderr, "ecal_detect_keyframes: %u pieces, %u window slots per pass (chains of <= %u), %u passes\n", P, S,
                d_max, n_passes);
        fprintf(stderr, "  pieces at work after each pass:");
        for (uint32_t v : trace_active) fprintf(stderr, " %u", v);
        fprintf(stderr, "\n  windows gone through in each pass:");
        for (size_t i = 0; i < trace_windows.size(); i++) fprintf(stderr, " %u", trace_windows[i] - (i ? trace_windows[i - 1] : 0u));
        fprintf(stderr, "\n");
    }
    ECAL_HIP_TRY(ctx, hipGetLastError());
    // windows the rule was applied to / the longest chain of a piece: from the pieces' final runs
    std::vector<uint32_t> lev(P), gen(P);
    ECAL_HIP_TRY(ctx, hipMemcpy(lev.data(), a.levels, (size_t) P * sizeof(uint32_t), hipMemcpyDeviceToHost));
    ECAL_HIP_TRY(ctx, hipMemcpy(gen.data(), a.gen, (size_t) P * sizeof(uint32_t), hipMemcpyDeviceToHost));
    unsigned long long nwin = 0;
    uint32_t longest = 0;
    for (uint32_t k = 0; k < P; k++) {
        nwin += lev[k];
        longest = std::max(longest, lev[k]);
    }
    if (passes) *passes = longest;
    if (windows) *windows = nwin;
    if (ho) {
        // the frame behind this call's pieces: the last keyframe of its latest piece (smallest index) that has one, else the frame
        // it was handed
        std::vector<uint32_t> nacc(P);
        ECAL_HIP_TRY(ctx, hipMemcpy(nacc.data(), a.nacc, (size_t) P * sizeof(uint32_t), hipMemcpyDeviceToHost));
        ecal_keyframe_frame out = ext_f;
        for (uint32_t k = 0; k < P; k++)
            if (nacc[k] > 0) {
                out.has = 1;
                ECAL_HIP_TRY(ctx, hipMemcpy(&out.time, a.ref_t + k, sizeof(double), hipMemcpyDeviceToHost));
                ECAL_HIP_TRY(ctx, hipMemcpy(out.dir, a.ref_dir + (size_t) k * rows * 2, 2 * (size_t) rows * sizeof(double), hipMemcpyDeviceToHost));
                break;
            }
        if (ho->send(ho->user, &out) < 0) {
            ctx->last_error = "ecal_detect_keyframes_sharded: the send callback failed";
            return ECAL_ERR_INVALID;
        }
    }
    const uint32_t K_all = h[1];   // records written, dead ones included
    if (K_all > max_keys) {
        *n_keyframes = K_all;
        ctx->last_error = "ecal_detect_keyframes: more keyframes than max_keyframes (n_keyframes holds the count)";
        return ECAL_ERR_RANGE;
    }
    if (K_all == 0) return ECAL_OK;
    // the records arrive in completion order: keep the live ones (their piece's final run), sort by time stamp (the
    // reference's keyframe map is ordered by time)
    // (through the context's pinned staging: a hipMemcpy into pageable memory the runtime has not seen before — these vectors in the
    // first calls of a process — runs at ~0.3 GB/s: 25 - 30 ms for the 8 MB of a 50 M-event search, a third of the search itself)
    const size_t nb_t = K_all * sizeof(double), nb_d = 2 * nb_t, nb_f = 3 * (size_t) K_all * M * sizeof(double), nb_u = (size_t) K_all * 4;
    const size_t off_d = nb_t, off_f = off_d + nb_d, off_e = off_f + nb_f, off_p = off_e + nb_u, off_g = off_p + nb_u, nb_all = off_g + nb_u;
    std::vector<unsigned char> pageable;
    unsigned char *stage = ecal_fetch_pinned(ctx, nb_all);
    if (!stage) {
        pageable.resize(nb_all);
        stage = pageable.data();
    }
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(stage, d_kt, nb_t, hipMemcpyDeviceToHost, st));
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(stage + off_d, d_kd, nb_d, hipMemcpyDeviceToHost, st));
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(stage + off_f, d_kf, nb_f, hipMemcpyDeviceToHost, st));
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(stage + off_e, d_ke, nb_u, hipMemcpyDeviceToHost, st));
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(stage + off_p, d_kp, nb_u, hipMemcpyDeviceToHost, st));
    ECAL_HIP_TRY(ctx, hipMemcpyAsync(stage + off_g, d_kg, nb_u, hipMemcpyDeviceToHost, st));
    ECAL_HIP_TRY(ctx, hipStreamSynchronize(st));
    const double *const t = reinterpret_cast<const double *>(stage), *const d = reinterpret_cast<const double *>(stage + off_d),
                 *const ft = reinterpret_cast<const double *>(stage + off_f);
    const int32_t *const e = reinterpret_cast<const int32_t *>(stage + off_e);
    const uint32_t *const kp = reinterpret_cast<const uint32_t *>(stage + off_p), *const kg = reinterpret_cast<const uint32_t *>(stage + off_g);
    std::vector<uint32_t> perm;
    perm.reserve(K_all);
    for (uint32_t i = 0; i < K_all; i++)
        if (kg[i] == gen[kp[i]]) perm.push_back(i);
    const uint32_t K = (uint32_t) perm.size();
    *n_keyframes = K;
    if (K > max_keyframes) {
        ctx->last_error = "ecal_detect_keyframes: more keyframes than max_keyframes (n_keyframes holds the count)";
        return ECAL_ERR_RANGE;
    }
    std::sort(perm.begin(), perm.end(), [&](uint32_t x, uint32_t y) { return t[x] < t[y] || (t[x] == t[y] && d[2 * x] < d[2 * y]); });
    for (uint32_t i = 0; i < K; i++) {
        const uint32_t j = perm[i];
        kf_time[i] = t[j];
        kf_duration[2 * i] = d[2 * j];
        kf_duration[2 * i + 1] = d[2 * j + 1];
        kf_events_num[i] = e[j];
        memcpy(kf_features + 3 * (size_t) i * M, ft + 3 * (size_t) j * M, 3 * M * sizeof(double));
    }
    return ECAL_OK;
}

// The gate and the restated std::nth_element alone, for verification against the oracle (host; no GPU involved)
extern "C" void ecal_ref_nth_element_f64(double *a, uint32_t n, uint32_t nth) {
    if (a) ecal::ref_nth_element(a, n, nth, [](double x, double y) { return x < y; });
}
