/*
 * ecal.h — C ABI of the MI355X-native EventCalib hot path (libecal.so).
 *
 * Plain C: pointers and sizes only, no C++/torch types.  Every entry point names the reference
 * interface it replaces (paths relative to the reference tree, MobilePerceptionLab/EventCalib).
 * The reference has no FFI layer of its own; its seams are the C++ class APIs listed here, and
 * the C++ shims in eventcalib_amd/csrc/host/ (same class names and members) sit on top of this
 * ABI so that unit_test_eventCameraCalib-style callers can switch with a re-link
 * (INTEGRATION.md shows the binding).
 *
 * Conventions: returns ECAL_OK (0) or a negative ecal_status; never throws; all buffers are
 * caller-owned; outputs are written only on success.  Entry points ending in _dev take DEVICE
 * pointers and enqueue asynchronously on `stream` (a hipStream_t passed as void*; NULL is HIP's
 * default stream, exactly as in a kernel launch); the others take HOST pointers, run on the
 * context's own stream and return when the result is in place.
 * One ecal_ctx per host thread (matches the reference's one-DBSCAN-instance-per-worker use,
 * event_camera_calib/test/eventCameraCalib.cpp:181-190).
 */
#ifndef ECAL_H_
#define ECAL_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ECAL_ABI_VERSION 3

typedef enum ecal_status {
    ECAL_OK = 0,
    ECAL_ERR_INVALID = -1,    /* bad argument (NULL pointer, minpts < 1, ...) */
    ECAL_ERR_NO_DEVICE = -2,  /* no usable HIP device / HIP runtime failure at init */
    ECAL_ERR_HIP = -3,        /* a HIP call failed; ecal_last_error() has the text */
    ECAL_ERR_NOMEM = -4,      /* device or host allocation failed */
    ECAL_ERR_UNSORTED = -5,   /* event timestamps are not non-decreasing */
    ECAL_ERR_RANGE = -6,      /* a size exceeds what the ABI can index (2^32-1 points) */
    ECAL_ERR_COMM = -7        /* RCCL error (ecal_last_error has the text) */
} ecal_status;

typedef struct ecal_ctx ecal_ctx;

/* ---- lifecycle -------------------------------------------------------------------------- */
int ecal_abi_version(void);
int ecal_init(int device, ecal_ctx **out);
void ecal_destroy(ecal_ctx *ctx);
const char *ecal_strerror(int status);
const char *ecal_last_error(const ecal_ctx *ctx);
/* blocks until everything enqueued on the context's own stream has finished */
int ecal_sync(ecal_ctx *ctx);

/* ---- DBSCAN ------------------------------------------------------------------------------
 * Replaces DBSCAN<Eigen::Vector2d,double>::Run(&V, 2, eps, minpts)
 *   (dbscan/include/dbscan.h:115-177, regionQuery :198-227, expandCluster :229-259) and the
 *   kd-tree underneath it (dbscan/src/kdtree.cpp:106-179, 344-365), called twice per
 *   time-slice from CirclesEventFrame::extractFeatures (event_camera_calib/src/CirclesEventFrame.cpp:66-72).
 *
 * One Run() per slice s over the points xy[slice_off[s] .. slice_off[s+1]) (pid = index inside
 * the slice, dim = 2).  labels[i] = index into the reference's `Clusters` vector for that slice
 * (clusters are numbered in order of their smallest member pid, dbscan.h:154-155), or -1 for a
 * member of `Noise`.  Only core points (>= minpts neighbours, self excluded, dbscan.h:151,218,247)
 * are ever assigned; border points are Noise exactly as in the reference.  The neighbour
 * relation reproduces the reference kd-tree's inclusive-ball / strict-plane-pruning behaviour
 * bit for bit (DESIGN.md "the quirk").  Empty slices give n_clusters[s] = 0 (Run would return
 * FAILED, dbscan.h:121); minpts < 1 is ECAL_ERR_INVALID (Run's FAILED, dbscan.h:123).
 */
int ecal_dbscan_batch(ecal_ctx *ctx, const double *xy /*[N][2]*/, const uint32_t *slice_off /*[S+1]*/,
                      uint32_t S, double eps, uint32_t minpts, int32_t *labels /*[N]*/,
                      uint32_t *n_clusters /*[S]*/);

/* Device-resident form: segment s is d_xy[d_seg_off[s] .. d_seg_off[s]+d_seg_cnt[s]) (segments
 * may leave gaps but must not overlap); n_points = upper bound on any d_seg_off+d_seg_cnt
 * (size of d_xy/d_labels in points).  max_seg_points: upper bound on any d_seg_cnt, or 0 if
 * unknown (then every size tier is launched). */
int ecal_dbscan_batch_dev(ecal_ctx *ctx, const double *d_xy, const uint32_t *d_seg_off, const uint32_t *d_seg_cnt,
                          uint32_t S, uint32_t n_points, uint32_t max_seg_points, double eps, uint32_t minpts,
                          int32_t *d_labels, uint32_t *d_n_clusters, void *stream);

/* ecal_cluster_order_dev: the member ORDER of DBSCAN<T,Float>::Clusters (dbscan.h:92): d_order[slot] = position of the
 * point inside Clusters[label] — the order expandCluster's queue pops the cluster's core points in (dbscan.h:229-265; the
 * seed = the cluster's smallest pid first), which follows the result order of the kd-tree's range query (hits in reverse
 * visiting order, kdtree.cpp:148-179,469-486) —, -1 for Noise.  d_xy / d_seg_off / d_seg_cnt as ecal_dbscan_batch_dev
 * takes them, d_labels / d_n_clusters as it returned them, same eps.  d_status[s] = 0, or 1 for a segment this pass does
 * not take (more than 2^20 points, or range-query result lists of more than 2^24 entries in all — segments of up to 4096
 * points run in LDS, anything beyond or denser in a global workspace): its d_order entries are -1.  What it is for: Clusters[c] in the reference's order for callers that index into it, and
 * extractFeatures' medians (std::nth_element over Clusters[c], CirclesEventFrame.cpp:136-147), which depend on that order
 * when two members tie in norm.  only_tied_medians != 0: the order is worked out only for the clusters that need it for that
 * purpose — those whose member of rank size / 2 in the order (norm, pid) shares its norm with another member (the test
 * ecal_extract_batch_ordered_dev applies) —, the members of all other clusters get -2; segments without such a cluster do
 * not even have their tree rebuilt.  (only_tied_medians == 2, used by ecal_extract_batch_exact_dev: the caller has named
 * those clusters itself by storing -3 in d_order on the slot of one member of each.)
 * Four launches: segments of up to 768 points and 256 clusters, then up to 2048 points, then up to 4096, then the rest.
 * Contract on the hand-over of kd-trees: the last ecal_dbscan_batch*_dev call of the context leaves the trees of its first-pass
 * segments behind (4 bytes per point in context scratch); they are replayed here instead of being rebuilt when d_labels,
 * d_seg_off and S are that call's.  Any slicing entry point and ecal_copy_dev on the context drop them; a caller that
 * rewrites the points or labels of those buffers by other means must call ecal_dbscan_batch*_dev again before this. */
int ecal_cluster_order_dev(ecal_ctx *ctx, const double *d_xy, const uint32_t *d_seg_off, const uint32_t *d_seg_cnt, uint32_t S,
                           double eps, const int32_t *d_labels, const uint32_t *d_n_clusters, int32_t *d_order /*[n_points]*/,
                           uint32_t *d_status /*[S]*/, int only_tied_medians, void *stream);
/* host-buffer form (what host/dbscan.h calls): slices as ecal_dbscan_batch takes them, labels / n_clusters as it returned them */
int ecal_cluster_order(ecal_ctx *ctx, const double *xy, const uint32_t *slice_off /*[S+1]*/, uint32_t S, double eps,
                       const int32_t *labels, const uint32_t *n_clusters /*[S]*/, int32_t *order /*[N]*/, uint32_t *status /*[S]*/);

/* ---- ingest + time-slicing ----------------------------------------------------------------
 * The event stream is the reference's .bin image: packed 25-byte little-endian records
 * {f64 t_sec, f64 x, f64 y, u8 polarity} (event/include/opengv2/event/Event.hpp:41-47,
 * README.md:7-13), time-sorted as the reference's std::multimap<double,Event_loc_pol> keeps them
 * (event_camera_calib/test/eventCameraCalib.cpp:154-163).  At most 2^32-1 events per stream.
 *
 * ecal_window_bounds_dev: for every window s the index range [lo, hi) of the events with
 *   t0[s] <= t <= t1[s] — container.lower_bound(duration.first) .. upper_bound(duration.second)
 *   of EventFrame::EventFrame (event/src/EventFrame.cpp:14-15) — and win_base[s] = sum of the
 *   sizes of the windows before s (win_base[S] = total), the slot where window s writes.
 * ecal_check_sorted_dev: *d_flag = 1 if some timestamp is smaller than its predecessor.
 * ecal_slice_events_dev: the rest of the EventFrame constructor (EventFrame.cpp:10-36) for all
 *   windows at once: per polarity the set of unique pixel locations (operator== on the doubles),
 *   minus every location present in both sets.  Window s owns slots [win_base[s], win_base[s+1])
 *   of d_xy / d_event_point (capacity cap_points slots):
 *     segment 2s   = positiveEvents_ : d_xy[d_seg_off[2s]   ..+d_seg_cnt[2s]]
 *     segment 2s+1 = negativeEvents_ : d_xy[d_seg_off[2s+1] ..+d_seg_cnt[2s+1]]
 *   in the element order ecal_set_point_order selected: by default THE REFERENCE'S — the iteration
 *   order of its std::unordered_set<Vector2d, EigenMatrixHash> (EventFrame.cpp:12-13,34-35;
 *   core/utility/include/opengv2/utility/utility.hpp:38-51; libstdc++), which DBSCAN's cluster
 *   assignments depend on (insertion-order kd-tree, kdtree.cpp:128-131,169) — or first occurrence.
 *   d_event_point[win_base[s]+k] = index of event k's pixel inside its polarity's segment, or -1
 *   if the pixel was erased — an output of this library's own (the reference's EventFrame keeps no such map; the
 *   association stage uses it); d_event_point == NULL: not wanted, not written (4 bytes per event less to write,
 *   one table phase less in the slicer).  The segment arrays feed ecal_dbscan_batch_dev directly (S' = 2S).
 *   *d_overflow = 1 if some window did not fit cap_points (its segments are then empty).
 *   max_win_events: upper bound on any window size, 0 = unknown.
 */
/* Element order of the pixel sets every slicing entry point of this context emits (ecal_slice_events_dev and everything
 * built on it: ecal_detect_fused_dev, ecal_detect_batch, ecal_detect_pass, ecal_detect_keyframes, ecal_detect_stream_tiled).  ECAL_ORDER_REFERENCE (default): positiveEvents_ / negativeEvents_ exactly
 * as the reference's EventFrame constructor leaves them on libstdc++ — same `.bin` => same point order => same DBSCAN
 * labels as the reference.  ECAL_ORDER_FIRST_OCCURRENCE: ascending first occurrence of the pixel inside the window (the
 * same sets, a cheaper order; labels then equal the reference's only up to the order-dependent effects). */
enum { ECAL_ORDER_REFERENCE = 0, ECAL_ORDER_FIRST_OCCURRENCE = 1 };
int ecal_set_point_order(ecal_ctx *ctx, int order);
int ecal_get_point_order(const ecal_ctx *ctx);
/* the restated libstdc++ tables behind ECAL_ORDER_REFERENCE, exported for verification (no GPU needed): bucket count of
 * epoch e (13, 29, 59, ...; 0 past the table) and EigenMatrixHash of a pixel */
uint64_t ecal_ref_bucket_step(int epoch);
uint64_t ecal_ref_pixel_hash(double x, double y);

/* (One stream per context runs the single-launch form — the first stream that calls; calls on other streams of the same
 * context are correct too but take two launches: the look-back table of the fused scan is context scratch.) */
int ecal_window_bounds_dev(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, const double *d_t0,
                           const double *d_t1, uint32_t S, uint32_t *d_win_lo, uint32_t *d_win_hi,
                           uint32_t *d_win_base /*[S+1]*/, void *stream);
int ecal_check_sorted_dev(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, int *d_flag, void *stream);
/* d_sorted (n_events * 25 bytes, not overlapping d_events) = the records in the order the reference's
 * std::multimap<double, Event_loc_pol> iterates in (eventCameraCalib.cpp:154-163): ascending time stamp, records with
 * equal time stamps in their input order (stable).  For streams that do not arrive in time order; the _dev entry points
 * above take time-ordered records, ecal_stream_create sorts by itself when it has to. */
int ecal_sort_events_dev(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, uint8_t *d_sorted, void *stream);
int ecal_slice_events_dev(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, const uint32_t *d_win_lo,
                          const uint32_t *d_win_hi, const uint32_t *d_win_base, uint32_t S, uint32_t max_win_events,
                          uint32_t cap_points, double *d_xy, uint32_t *d_seg_off /*[2S]*/,
                          uint32_t *d_seg_cnt /*[2S]*/, int32_t *d_event_point, int *d_overflow, void *stream);

/* ---- packed points -----------------------------------------------------------------------------------------------------
 * Event pixels are small integers: between the stages a point needs 4 bytes (x | y << 16, two's complement int16 each), not
 * the 16 of the reference's Vector2d.  The *_packed_dev forms of the three stage entry points take an ecal_packed_points
 * (caller-owned device buffers: d_xy16 [cap_points], d_seg_fmt [2S]; NULL = the plain forms): the slicer writes the windows
 * its pixel kernels take (sensor pixels 0 <= x <= 2047, 0 <= y <= 1023, up to 2047 events; x <= 1023 up to 4095 events; in
 * reference order x <= 511 up to 5119 events) to d_xy16 only and marks their two
 * segments d_seg_fmt = 1; every other window goes to d_xy as doubles (d_seg_fmt = 0).  DBSCAN and the extraction read
 * whichever form a segment has (and write its doubles themselves, d_seg_fmt 1 -> 3, before a segment goes to one of their
 * general tiers); results are bit for bit those of the plain forms.  ecal_unpack_points_dev writes the doubles of every
 * segment that has none yet — for callers that want d_xy itself (positiveEvents_ / negativeEvents_ as doubles). */
typedef struct ecal_packed_points {
    uint32_t *d_xy16;     /* [cap_points] */
    uint32_t *d_seg_fmt;  /* [2S]: 0 doubles only, 1 packed only, 3 both */
} ecal_packed_points;
int ecal_slice_events_packed_dev(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, const uint32_t *d_win_lo,
                                 const uint32_t *d_win_hi, const uint32_t *d_win_base, uint32_t S, uint32_t max_win_events,
                                 uint32_t cap_points, double *d_xy, uint32_t *d_seg_off, uint32_t *d_seg_cnt, int32_t *d_event_point,
                                 int *d_overflow, const ecal_packed_points *pk, void *stream);
int ecal_dbscan_batch_packed_dev(ecal_ctx *ctx, double *d_xy, const uint32_t *d_seg_off, const uint32_t *d_seg_cnt, uint32_t S,
                                 uint32_t n_points, uint32_t max_seg_points, double eps, uint32_t minpts, int32_t *d_labels,
                                 uint32_t *d_n_clusters, const ecal_packed_points *pk, void *stream);
/* (the extraction the context's ecal_set_median_ties setting asks for: ecal_extract_batch_exact_dev or ecal_extract_batch_dev) */
int ecal_extract_batch_packed_dev(ecal_ctx *ctx, double *d_xy, const uint32_t *d_seg_off, const uint32_t *d_seg_cnt,
                                  const int32_t *d_labels, const uint32_t *d_n_clusters, uint32_t S /*windows*/, uint32_t n_points,
                                  double eps, uint32_t cluster_min, uint32_t need_clusters, double radius_threshold, int fit_circle,
                                  uint32_t knn_num, uint32_t *d_win_info, uint32_t *d_cand_pair, double *d_cand_xyr,
                                  int32_t *d_kept_labels, uint32_t *d_rep, const ecal_packed_points *pk, void *stream);
int ecal_unpack_points_dev(ecal_ctx *ctx, const ecal_packed_points *pk, const uint32_t *d_seg_off, const uint32_t *d_seg_cnt,
                           uint32_t n_segments, double *d_xy, void *stream);

/* ---- circle-candidate extraction -----------------------------------------------------------
 * Replaces CirclesEventFrame::extractFeatures between its DBSCAN::Run calls and cv::findCirclesGrid
 * (event_camera_calib/src/CirclesEventFrame.cpp:89-312) for all windows at once: fit_circle == 0 is the
 * mutual-nearest / midpoint path of :283-311 (the shipped example.yaml), fit_circle != 0 the knn_num-nearest
 * / algebraic CirclesEventFrame::fitCircle (:361-415) / double-direction path of :180-281 (knn_num <= 8).  Inputs are the outputs of ecal_slice_events_dev and ecal_dbscan_batch_dev with
 * S' = 2S segments (2s = positive, 2s+1 = negative polarity of window s).
 *
 * ecal_circle_radius_threshold: circleRadiusThreshold_ (CirclesEventFrame.cpp:16-33); pure host math.
 * ecal_extract_batch_dev, per window s (all outputs indexed like the points: window s owns the
 *   slots from d_seg_off[2s] on):
 *     d_kept_labels[slot of point] = cluster id after erasing clusters with fewer than cluster_min
 *        (clusterMinSample) members and renumbering (:89-117), or -1  (= pClusters_/nClusters_)
 *     d_rep[d_seg_off[2s+pol] + k]  = pid of the representative of kept cluster k: the member of
 *        rank size/2 ordered by (Vector2d::norm(), pid)  (:136-147; the reference's nth_element
 *        picks the same pixel unless two members tie in norm at that rank)
 *     candidates j = 0..n-1 in + cluster order (:283-311): d_cand_pair[2*(d_seg_off[2s]+j)..] =
 *        (+ cluster, - cluster), d_cand_xyr[3*(d_seg_off[2s]+j)..] = centre x, centre y, radius
 *     d_win_info[4s..] = { n candidates, kept + clusters, kept - clusters, status } with status
 *        0 = ok, 1 = extractFeatures would return false before pairing (an empty polarity :62-64
 *        or fewer than need_clusters = rows*cols kept clusters :127-129), 4 = more than 2048 DBSCAN
 *        clusters in one polarity (not handled; no candidates).  The exact extraction (ecal_extract_batch_exact_dev / _ordered_dev
 *        and every composite entry point under ECAL_TIES_REFERENCE) ORs ECAL_WIN_TIE_FALLBACK into the word when some tied
 *        median of the window was decided by the smaller-pid rule because the reference's pick could not be worked out (see
 *        ecal_cluster_order_dev's status 1: a segment beyond its workspace): everything else about the window
 *        is as for status 0, the representative of that cluster may differ from the reference's.  ECAL_WIN_STATUS(w) strips it.
 *   The ordering of the candidates into the pattern grid (cv::findCirclesGrid, :332-353) is not
 *   part of this entry point.
 */
#define ECAL_WIN_TIE_FALLBACK 0x100u
#define ECAL_WIN_STATUS(word) ((word) & 0xFFu)
double ecal_circle_radius_threshold(double width, double height, int rows, int cols, int asymmetric,
                                    double square_size, double circle_radius);
int ecal_extract_batch_dev(ecal_ctx *ctx, const double *d_xy, const uint32_t *d_seg_off, const uint32_t *d_seg_cnt,
                           const int32_t *d_labels, const uint32_t *d_n_clusters, uint32_t S /*windows*/,
                           uint32_t n_points, uint32_t cluster_min, uint32_t need_clusters, double radius_threshold,
                           int fit_circle, uint32_t knn_num, uint32_t *d_win_info /*[S][4]*/, uint32_t *d_cand_pair /*[n_points][2]*/,
                           double *d_cand_xyr /*[n_points][3]*/, int32_t *d_kept_labels /*[n_points]*/,
                           uint32_t *d_rep /*[n_points]*/, void *stream);

/* ---- the whole per-window body in one call ---------------------------------------------------
 * ecal_detect_fused_dev = ecal_slice_events_dev + ecal_dbscan_batch_dev (over the 2S segments) + ecal_extract_batch_dev
 * with the same arguments, the same output arrays and the same results bit for bit — the body of the reference's worker
 * loop per window (event_camera_calib/test/eventCameraCalib.cpp:49-56: EventFrame constructor, extractFeatures with its two
 * DBSCAN::Run calls).  In the shipped configuration (reference point order, eps < 16, fitCircle == 0) ONE kernel carries a
 * window through all three stages on one compute unit; what that kernel cannot take goes through the stages' later passes
 * as before.  Any other configuration runs the three stage functions one after the other. */
int ecal_detect_fused_dev(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, const uint32_t *d_win_lo,
                          const uint32_t *d_win_hi, const uint32_t *d_win_base, uint32_t S, uint32_t max_win_events,
                          uint32_t max_seg_points, uint32_t cap_points, double eps, uint32_t minpts, uint32_t cluster_min,
                          uint32_t need_clusters, double radius_threshold, int fit_circle, uint32_t knn_num,
                          double *d_xy, uint32_t *d_seg_off /*[2S]*/, uint32_t *d_seg_cnt /*[2S]*/, int32_t *d_event_point,
                          int *d_overflow, int32_t *d_labels, uint32_t *d_n_clusters /*[2S]*/, uint32_t *d_win_info /*[S][4]*/,
                          uint32_t *d_cand_pair, double *d_cand_xyr, int32_t *d_kept_labels, uint32_t *d_rep, void *stream);

/* ecal_extract_batch_ordered_dev = ecal_extract_batch_dev with the reference's own choice where a cluster's median rank has an
 * equal-norm rival: d_cluster_order = ecal_cluster_order_dev's output for the same segments (the members' positions inside
 * Clusters[c]); the representative of such a cluster is what libstdc++'s std::nth_element leaves at Clusters[c][size / 2]
 * (CirclesEventFrame.cpp:136-147; introselect restated), everything downstream follows.  Clusters of segments
 * ecal_cluster_order_dev did not take (status 1) keep ecal_extract_batch_dev's rule (the smaller pid) and their window's status
 * word carries ECAL_WIN_TIE_FALLBACK. */
int ecal_extract_batch_ordered_dev(ecal_ctx *ctx, const double *d_xy, const uint32_t *d_seg_off, const uint32_t *d_seg_cnt,
                                   const int32_t *d_labels, const uint32_t *d_n_clusters, const int32_t *d_cluster_order,
                                   uint32_t S /*windows*/, uint32_t n_points, uint32_t cluster_min, uint32_t need_clusters,
                                   double radius_threshold, int fit_circle, uint32_t knn_num, uint32_t *d_win_info,
                                   uint32_t *d_cand_pair, double *d_cand_xyr, int32_t *d_kept_labels, uint32_t *d_rep,
                                   void *stream);

/* What the composite entry points (ecal_detect_batch, ecal_detect_pass, ecal_detect_keyframes, ecal_detect_stream_tiled) do where
 * a kept cluster's median rank has an equal-norm rival: ECAL_TIES_REFERENCE (default) = the reference's own pick
 * (ecal_extract_batch_exact_dev), ECAL_TIES_SMALLER_PID = the cheaper rule of ecal_extract_batch_dev. */
#define ECAL_TIES_REFERENCE 0
#define ECAL_TIES_SMALLER_PID 1
int ecal_set_median_ties(ecal_ctx *ctx, int mode);
int ecal_get_median_ties(const ecal_ctx *ctx);

/* How the stage calls (slicing, DBSCAN, extraction, member order) schedule their later size tiers.  Every stage runs a first-pass
 * kernel that takes what the shipped configuration produces and lists the rest for tiers of growing capacity; on most data
 * those lists stay empty, and a launch that finds its list empty still costs ~5 us (twenty of them: 0.1 ms of every pass).
 * ECAL_TAIL_AUTO (default): a stage whose previous call on this context saw empty lists launches ONE kernel that takes
 * whatever is listed through its most general tier; slicing and DBSCAN, when the first pass listed work but the later hash /
 * pixel passes left none, run those passes + that one kernel; a stage that saw work behind the second pass launches every tier.
 * ECAL_TAIL_TIERED / ECAL_TAIL_LEAN force one form (tests).  The adaptive search (ecal_detect_keyframes), whose windows are
 * second-tier work by design, keeps AUTO but never takes the one-kernel form.
 * The choice moves time only: every listed window is processed either way, results are bit-identical
 * (tests/test_gpu_tail_modes.py). */
/* roctx ranges around the stage entry points (window bounds, slicing, DBSCAN, extraction, member order, grid ordering, keyframe
 * search, init calibration, solver evaluations and solves): on != 0 looks the marker library up at run time
 * (librocprofiler-sdk-roctx.so, else libroctx64.so; ECAL_ERR_INVALID when neither is there), `rocprofv3 --marker-trace
 * --kernel-trace` then attributes every kernel to its stage call.  ECAL_ROCTX=1 in the environment at ecal_init does the same. */
int ecal_set_profile_ranges(ecal_ctx *ctx, int on);

#define ECAL_TAIL_AUTO 0
#define ECAL_TAIL_TIERED 1
#define ECAL_TAIL_LEAN 2
int ecal_set_tail_mode(ecal_ctx *ctx, int mode);
/* the mode in force (ECAL_TAIL_*), or ECAL_ERR_INVALID: lets a caller that switches modes for one stage put back what it found */
int ecal_get_tail_mode(const ecal_ctx *ctx);

/* ecal_extract_batch_exact_dev: the exact extraction in one call (eps = the DBSCAN radius the labels were made with): the plain
 * pass lists the windows in which some kept cluster's median is tied in norm (a third of them on recorded-like data, all of
 * them on the dense benchmark stream),
 * ecal_cluster_order_list_dev works out the reference's member order for the tied clusters of those windows only, and the
 * listed windows are extracted again with it.  Results = ecal_extract_batch_ordered_dev's = the reference's own. */
int ecal_extract_batch_exact_dev(ecal_ctx *ctx, const double *d_xy, const uint32_t *d_seg_off, const uint32_t *d_seg_cnt,
                                 const int32_t *d_labels, const uint32_t *d_n_clusters, uint32_t S /*windows*/, uint32_t n_points,
                                 double eps, uint32_t cluster_min, uint32_t need_clusters, double radius_threshold, int fit_circle,
                                 uint32_t knn_num, uint32_t *d_win_info, uint32_t *d_cand_pair, double *d_cand_xyr,
                                 int32_t *d_kept_labels, uint32_t *d_rep, void *stream);
/* ecal_cluster_order_dev restricted to the segments 2 w, 2 w + 1 of the windows w = d_win_list[0 .. *d_win_count) (device memory;
 * an entry with bit 30 / bit 31 set: without segment 2 w / 2 w + 1, whose d_order and d_status entries are then left as they are) */
int ecal_cluster_order_list_dev(ecal_ctx *ctx, const double *d_xy, const uint32_t *d_seg_off, const uint32_t *d_seg_cnt, uint32_t S,
                                double eps, const int32_t *d_labels, const uint32_t *d_n_clusters, int32_t *d_order,
                                uint32_t *d_status, int only_tied_medians, const uint32_t *d_win_list, const uint32_t *d_win_count,
                                void *stream);

/* ---- host-buffer conveniences (what the C++ shims in eventcalib_amd/csrc/host/ call) ----------
 * ecal_stream: the event stream uploaded once and kept in HBM — the counterpart of the reference's
 *   EventContainer (event/include/opengv2/event/EventContainer.hpp:25-30), filled once by the driver
 *   (event_camera_calib/test/eventCameraCalib.cpp:154-163) and shared read-only by all workers.
 *   `events` = packed 25-byte records, any order: like the reference's multimap (eventCameraCalib.cpp:154-163) the
 *   stream is brought into time order (stable for equal time stamps; one device sort, only when it is not sorted already).
 * ecal_detect_batch: for every window [t0[s], t1[s]] the EventFrame constructor followed by
 *   extractFeatures up to the candidate circles (the four device entry points above, in order) and
 *   copies the requested results to host memory.  Every pointer of ecal_detect_result may be NULL
 *   (not copied); the layouts are those documented at the device entry points; cap_points must be
 *   >= the total number of events covered by the windows (ECAL_ERR_RANGE otherwise).
 */
typedef struct ecal_stream ecal_stream;
typedef struct ecal_detect_params {
    double dbscan_eps;               /* example.yaml: dbscan_eps */
    uint32_t dbscan_min_samples;     /* dbscan_startMinSample */
    uint32_t cluster_min_sample;     /* clusterMinSample */
    uint32_t need_clusters;          /* BoardSize_Rows * BoardSize_Cols */
    double circle_radius_threshold;  /* ecal_circle_radius_threshold(...) */
    int fit_circle;                  /* fitCircle */
    uint32_t knn_num;                /* knn_num (used when fit_circle != 0) */
    uint32_t rows, cols;             /* BoardSize_Rows / BoardSize_Cols: grid ordering runs when both are > 0 */
} ecal_detect_params;
typedef struct ecal_detect_result {
    uint32_t *win_lo, *win_hi; /* [S] */
    uint32_t *win_base;        /* [S+1] */
    double *xy;                /* [cap_points][2] */
    uint32_t *seg_off, *seg_cnt; /* [2S] */
    int32_t *event_point;      /* [cap_points] */
    int32_t *labels;           /* [cap_points] */
    uint32_t *n_clusters;      /* [2S] */
    int32_t *kept_labels;      /* [cap_points] */
    uint32_t *rep;             /* [cap_points] */
    uint32_t *win_info;        /* [S][4] */
    uint32_t *cand_pair;       /* [cap_points][2] */
    double *cand_xyr;          /* [cap_points][3] */
    int32_t *grid_order;       /* [S][rows*cols] (ecal_grid_order_dev), when params.rows * params.cols > 0 */
    uint32_t *grid_found;      /* [S] */
} ecal_detect_result;
int ecal_stream_create(ecal_ctx *ctx, const uint8_t *events, uint64_t n_events, ecal_stream **out);
/* The same from a .bin file of 25-byte records (EventStream's format): the loop of eventCameraCalib.cpp:154-163 — records with
 * timeStamp >= start_time are kept, and with has_end != 0 the first record with timeStamp >= end_time ends the reading —
 * without a host copy of the file: chunks are read by a few threads into pinned buffers while the previous chunk uploads. */
int ecal_stream_create_from_file(ecal_ctx *ctx, const char *path, double start_time, int has_end, double end_time, ecal_stream **out);
/* time stamps of the first and the last record (both 0 for an empty stream) */
int ecal_stream_times(const ecal_stream *s, double *first, double *last);
void ecal_stream_destroy(ecal_stream *s);
uint64_t ecal_stream_size(const ecal_stream *s);
/* DEVICE pointer of the packed records (what the _dev entry points take as d_events) */
const uint8_t *ecal_stream_data(const ecal_stream *s);
/* device-to-device copy on `stream` (optionally followed by a stream synchronise) */
int ecal_copy_dev(ecal_ctx *ctx, void *d_dst, const void *d_src, size_t bytes, void *stream, int sync);
int ecal_detect_batch(ecal_ctx *ctx, const ecal_stream *es, const double *t0, const double *t1, uint32_t S,
                      const ecal_detect_params *prm, uint32_t cap_points, ecal_detect_result *res);

/* ---- double-buffered ingest (BASELINE configs[4]: hipMemcpyAsync double-buffered ingest) -----------------
 * ecal_detect_stream_tiled: detection over a HOST-resident event file without uploading it first.  Tiled windows
 *   s = [t_start + s len, nextafter(t_start + (s + 1) len, -inf)] (policy P1 of SURVEY 8d: every event in exactly one
 *   window; the reference's piece loop `eventCameraCalib.cpp:172-190` with non-overlapping frames).  Chunks of
 *   windows_per_chunk windows are uploaded with hipMemcpyAsync on a copy stream into one of two device buffers while
 *   the detection kernels (bounds -> slice -> DBSCAN -> candidates -> grid order) of the previous chunk run; pass pinned
 *   host memory (ecal_pin_host, or any hipHostMalloc'ed buffer) — pageable memory makes the copies synchronous.
 *   Outputs (host, any may be NULL): win_info [S][4] as ecal_extract_batch_dev, grid_found [S], features
 *   [S][rows*cols][3] = the ordered circles (centre x, y, radius; NaN where no grid was found).
 *   *n_windows = S = floor((t_last - t_start) / len) + 1 (ECAL_ERR_RANGE if > max_windows). */
typedef struct ecal_ingest_stats {
    uint32_t chunks;
    uint64_t max_chunk_events, bytes_uploaded;
    double seconds;   /* wall time of the whole call */
} ecal_ingest_stats;
/* d_feat [S][M][3] = the ordered circles of every window (candidate d_order[s][k] of window s: centre x, y, radius),
 * NaN where status != 0 or no grid was found; inputs as written by ecal_extract_batch_dev / ecal_grid_order_dev */
int ecal_gather_features_dev(ecal_ctx *ctx, const uint32_t *d_win_info, const uint32_t *d_seg_off, const double *d_cand_xyr,
                             const int32_t *d_order /*[S][M]*/, const uint32_t *d_found, uint32_t S, uint32_t M /*rows*cols*/,
                             double *d_feat, void *stream);
/* ecal_detect_pass: one lock-step pass of the adaptive-window driver (MultiProcess::process, event_camera_calib/test/
 * eventCameraCalib.cpp:49-95, for every active piece at once): the five detection stages + grid ordering over S windows of a
 * DEVICE-resident stream, one upload of the window bounds and ONE download of what the driver's control flow needs:
 * packed [S][3 + 3 rows*cols] doubles = { status (win_info[3]), 1 if extractFeatures() would return true, unique pixels
 * (EventFrame::eventsNum()), then the ordered circles x y r (NaN if none) }.  Synchronous. */
int ecal_detect_pass(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, const double *t0, const double *t1, uint32_t S,
                     const ecal_detect_params *prm, uint32_t cap_points, double *packed);
/* ecal_detect_keyframes: the whole adaptive-window driver with the policy on the device — the reference's worker loop
 * (MultiProcess::process, event_camera_calib/test/eventCameraCalib.cpp:34-97: success / slide / grow rule :49-95) over
 * piece_num pieces of [start_time, end_time] (:168-179) with the keyframe gate of EventCalibIni::track (event_camera_calib/
 * src/EventCalibIni.cpp:23-97) against the previous keyframe of the window's own piece or, in the reference's single-worker
 * semantics, of the one shared map (ecal_adaptive_params.gate_mode).  Every lock-step pass = bounds, slicing, DBSCAN, candidates, grid ordering over a CHAIN of
 * windows per piece — its current one and the windows that follow it if every verdict is the likely one: no keyframe, grow
 * (or slide once the window is longer than three lengths); the window slots of a pass, a few per piece in all, go to the
 * pieces still at work — + one policy kernel that applies the rule along that chain for as long as the verdicts are the
 * likely ones (nine in ten are): many windows of a piece's chain per pass, same keyframes as one by one.  All enqueued back to back; the host follows a 4-byte counter check_every passes behind.
 * d_events: DEVICE-resident stream.  cap_points >= the events covered by the windows of any one pass
 * (ecal_detect_keyframes_cap_hint; ECAL_ERR_RANGE otherwise: call again with more).  Outputs (host), sorted by time stamp: kf_time [K], kf_duration
 * [K][2], kf_events_num [K] (EventFrame::eventsNum()), kf_features [K][rows*cols][3] (x, y, radius in grid order);
 * *n_keyframes = K (ECAL_ERR_RANGE with the needed count if K > max_keyframes); *passes = the longest chain of windows
 * a piece went through (the lock-step passes of the one-window-per-pass form; max_passes bounds it), *windows = windows the
 * rule was applied to (the ones evaluated ahead and not taken do not count).  Synchronous; runs on the context's own stream: d_events must be
 * complete when the call is made (no pending writes on other streams). */
typedef struct ecal_adaptive_params {
    double motion_time_step;             /* MotionTimeStep: window = 3 steps, gap after a keyframe = 5 steps */
    uint32_t frame_event_num_threshold;  /* FrameEventNumThreshold */
    uint32_t piece_num;                  /* the reference: 5 * (hardware threads - 2) */
    double start_time, end_time;         /* StartTime / EndTime */
    uint32_t max_passes;                 /* 0 = unlimited */
    uint32_t check_every;                /* passes the host may run ahead of the device's active-piece counter (0 = 2, at most 8) */
    int gate_mode;                       /* ECAL_GATE_OWN_PIECE / ECAL_GATE_SHARED_MAP (below) */
    /* piece_count != 0: only the pieces piece_first .. piece_first + piece_count - 1 of the piece_num pieces (piece 0 is the last
     * in time, eventCameraCalib.cpp:172-179), with exactly the bounds they have in the whole run.  Pieces are independent under
     * ECAL_GATE_OWN_PIECE, so several calls — one context and host thread each — share one search: their kernels overlap on the
     * GPU (a lock-step pass is latency bound) and the union of their keyframes is the whole run's.  With ECAL_GATE_SHARED_MAP a
     * subset needs the frame of the pieces before it: ecal_detect_keyframes_sharded. */
    uint32_t piece_first, piece_count;
} ecal_adaptive_params;
/* Which keyframe a successful window is gated against (EventCalibIni::track, EventCalibIni.cpp:26-36: the map's
 * lower_bound(time stamp), else its last keyframe; TrackingBase.cpp:18-27: only the very first frame is ungated):
 *   ECAL_GATE_OWN_PIECE   the previous keyframe of the window's own piece; every piece's first success is accepted ungated.
 *                         Deterministic and schedule-free, but not what any run of the reference computes.
 *   ECAL_GATE_SHARED_MAP  the reference run with ONE worker thread (threadNum = 1): one map, pieces in pop_back order =
 *                         ascending time, so the reference frame is the map's last keyframe — across piece boundaries too; only
 *                         the first success of the whole run is ungated.  == oracle/policy_oracle.cpp mode 1.  (With several
 *                         workers the reference's result depends on the thread schedule: no deterministic counterpart.) */
#define ECAL_GATE_OWN_PIECE 0
#define ECAL_GATE_SHARED_MAP 1
/* libstdc++'s std::nth_element on doubles with operator< (NaNs compare false), restated — what the gate's median is
 * (EventCalibIni.cpp:78); exported for verification against the real library (host only, no GPU) */
void ecal_ref_nth_element_f64(double *a, uint32_t n, uint32_t nth);
int ecal_detect_keyframes(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, const ecal_adaptive_params *ap,
                          const ecal_detect_params *prm, uint32_t cap_points, uint32_t max_keyframes, double *kf_time,
                          double *kf_duration, int32_t *kf_events_num, double *kf_features, uint32_t *n_keyframes,
                          uint32_t *passes, uint64_t *windows);
/* ecal_detect_keyframes_sharded: the shared-map search of ONE stream cut over several callers (one per GPU): this call runs the
 * pieces ap->piece_first .. piece_first + piece_count - 1 (contiguous in time) under ECAL_GATE_SHARED_MAP.  What they need from
 * the pieces before them (larger indices = earlier in time, another caller's) is one frame — the map's last keyframe as
 * EventCalibIni::track reads it (EventCalibIni.cpp:26-36): time stamp + the pattern rows' line directions —, so the callers form
 * a chain in time: recv delivers the frame behind all earlier pieces (return 1: *frame filled — has = 0: no keyframe before —,
 * 0: not there yet (only when wait == 0), < 0: error; polled between passes, then waited for; never called by the caller that
 * holds the run's first piece), send is called once with the frame behind this caller's pieces (return < 0: error).  The union
 * of the callers' keyframes == ecal_detect_keyframes over all pieces, record for record.  N - 1 messages of 8 (2 + 2 rows)
 * bytes per search; a caller's pieces run speculatively until its frame arrives (docs: design/11_keyframe_gate.md). */
#define ECAL_FRAME_MAX_ROWS 32
typedef struct ecal_keyframe_frame {
    int has;                              /* 0: no keyframe */
    double time;                          /* KeyFrame time stamp */
    double dir[2 * ECAL_FRAME_MAX_ROWS];  /* [rows][2]: direction of the line fitted to every pattern row (EventCalibIni.cpp:46-57) */
} ecal_keyframe_frame;
typedef int (*ecal_frame_recv_fn)(void *user, ecal_keyframe_frame *frame, int wait);
typedef int (*ecal_frame_send_fn)(void *user, const ecal_keyframe_frame *frame);
typedef struct ecal_adaptive_handover {
    ecal_frame_recv_fn recv;
    ecal_frame_send_fn send;
    void *user;
} ecal_adaptive_handover;
int ecal_detect_keyframes_sharded(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, const ecal_adaptive_params *ap,
                                  const ecal_detect_params *prm, uint32_t cap_points, uint32_t max_keyframes, double *kf_time,
                                  double *kf_duration, int32_t *kf_events_num, double *kf_features, uint32_t *n_keyframes,
                                  uint32_t *passes, uint64_t *windows, const ecal_adaptive_handover *ho);
/* a cap_points that ecal_detect_keyframes will usually find sufficient for a stream of n_events events (0: invalid
 * parameters); ECAL_ERR_RANGE still says when it was not — double it and call again */
uint64_t ecal_detect_keyframes_cap_hint(const ecal_adaptive_params *ap, uint64_t n_events);
/* ... from the events of the resident stream that lie in [start_time, end_time] (n_events above = events IN that range: a
 * search over part of a stream is sized for that part).  Synchronous on the context's stream; 0 on an error */
uint64_t ecal_detect_keyframes_cap_hint_dev(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, const ecal_adaptive_params *ap);
int ecal_pin_host(ecal_ctx *ctx, void *ptr, size_t bytes);   /* hipHostRegister */
int ecal_unpin_host(ecal_ctx *ctx, void *ptr);
int ecal_detect_stream_tiled(ecal_ctx *ctx, const uint8_t *events /*host*/, uint64_t n_events, double t_start, double window_len,
                             uint32_t windows_per_chunk, const ecal_detect_params *prm, uint32_t max_windows,
                             uint32_t *win_info, uint32_t *grid_found, double *features, uint32_t *n_windows,
                             ecal_ingest_stats *stats);

/* ---- grid ordering of the candidates -----------------------------------------------------------------
 * Replaces cv::findCirclesGrid(points, Size(cols, rows), centers, CALIB_CB_ASYMMETRIC_GRID[|CLUSTERING]) and the
 * nearest-candidate lookup after it (event_camera_calib/src/CirclesEventFrame.cpp:332-353; the finder is the
 * reference's vendored OpenCV code, cv_calib/src/circlesgrid.cpp) for all windows at once.
 * Inputs: d_win_info / d_seg_off / d_cand_xyr as written by ecal_extract_batch_dev.  Per window s:
 *   d_found[s] = 1 and d_order[s*rows*cols + i*cols + j] = index (into the window's candidate list) of the
 *   circle at model point ((2j + i%2) s, i s, 0) (EventCalibIni.cpp:102-106) — the reference's orderIdxs —
 *   or d_found[s] = 0 and -1 entries when no complete grid is found (extractFeatures returns false).
 * Deterministic lattice walk, not OpenCV's randomised (kmeans) search: same ordering whenever a complete
 * grid is present and seen from its front; parity with the third-party finder is otherwise unpinned.
 * Up to 128 candidates per window, rows*cols <= 128. */
int ecal_grid_order_dev(ecal_ctx *ctx, const uint32_t *d_win_info, const uint32_t *d_seg_off, const double *d_cand_xyr,
                        uint32_t S, uint32_t rows, uint32_t cols, int32_t *d_order /*[S][rows*cols]*/,
                        uint32_t *d_found /*[S]*/, void *stream);
/* Host-buffer form for one candidate list — the call cv::findCirclesGrid(points, Size(cols, rows), centers,
 * CALIB_CB_ASYMMETRIC_GRID [| CALIB_CB_CLUSTERING]) of CirclesEventFrame.cpp:332-336 (cv_calib/include/cv_calib.hpp:19-21):
 * cand_xyr [n][3] (x, y, radius; the radius is not used), order [rows * cols] = candidate index of every pattern point
 * (-1 when *found == 0).  eventcalib_amd/csrc/host/cv_calib.hpp wraps it in that very signature.  Synchronous. */
int ecal_grid_order(ecal_ctx *ctx, const double *cand_xyr, uint32_t n, uint32_t rows, uint32_t cols, int32_t *order, uint32_t *found);

/* ---- re-detection of the circles around their predicted projections ---------------------------------
 * Replaces CirclesEventFrame::rectifyFeatures(outlierIdxs, Rcw, tcw) (event_camera_calib/src/
 * CirclesEventFrame.cpp:417-638; caller EventCalibIni.cpp:294) for F keyframes at once.  Keyframe f is window
 * d_frame_window[f] of the slicing / DBSCAN / extraction outputs (d_xy, d_seg_off, d_seg_cnt, d_kept_labels =
 * pClusters_/nClusters_ membership, d_win_info); d_pose[12f..] = Rcw row-major (9) then tcw (3), the pose
 * solvePnPRansac gave it (EventCalibIni.cpp:258-272); d_landmarks [rows*cols][3] = landmark positions in grid
 * order (EventCalibIni.cpp:102-106; narrowed to float as the reference's cv::Point3f).  Per circle k:
 *   d_feat_valid[f*n + k] = 0 if the reference erases the feature (projection outside the image :457-461, fewer
 *   than 5 events of either polarity :560-563, refit too far from the prediction :572-576), else 1 and
 *   d_feat_xyr[3(f*n + k)..] = rectified centre x, y, radius (NaN when erased).
 *   d_frame_info[2f..] = { return value of rectifyFeatures (border score :587-622 with fit_circle == 0, 20 % rule
 *   :625-627), number of erased features }.
 * The outlierIdxs argument of the reference is unused there and has no counterpart.  d_feat_xyr of the accepted
 * keyframes is what ecal_associate_dev takes as d_kf_circles (NaN rows never match an event).
 * cv::projectPoints (OpenCV, third party) is restated: 5 distortion coefficients k1 k2 p1 p2 k3. */
typedef struct ecal_rectify_params {
    double fx, fy, cx, cy;     /* camera->K() */
    double dist[5];            /* camera->distCoeffs(): k1 k2 p1 p2 k3 */
    double width, height;      /* camera->size() */
    uint32_t rows, cols;       /* BoardSize_Rows / BoardSize_Cols, rows*cols <= 128 */
    int asymmetric;            /* pattern_->isAsymmetric */
    double circle_radius;      /* Circles_Radius (world units) */
    int fit_circle;            /* params_.fitCircle: != 0 skips the border-score test */
    int model;                 /* 0: cv::projectPoints with dist = k1 k2 p1 p2 k3 (the reference); 1: cv::fisheye::projectPoints
                                  with dist[0..3] = k1..k4 (BASELINE configs[4]; the reference hands its fisheye coefficients to
                                  the pinhole projection, EventCalibIni.cpp:258 / CirclesEventFrame.cpp:449 — not reproduced) */
} ecal_rectify_params;
int ecal_rectify_batch_dev(ecal_ctx *ctx, const double *d_xy, const uint32_t *d_seg_off, const uint32_t *d_seg_cnt,
                           const int32_t *d_kept_labels, const uint32_t *d_win_info, const uint32_t *d_frame_window /*[F]*/,
                           const double *d_pose /*[F][12]*/, uint32_t F, const double *d_landmarks /*[rows*cols][3]*/,
                           const ecal_rectify_params *prm, double *d_feat_xyr /*[F][rows*cols][3]*/,
                           uint32_t *d_feat_valid /*[F][rows*cols]*/, uint32_t *d_frame_info /*[F][2]*/, void *stream);
/* host-buffer form: keyframe f owns segments 2f (positiveEvents_) and 2f+1 (negativeEvents_) of xy / kept_labels */
int ecal_rectify_batch(ecal_ctx *ctx, const double *xy /*[n_points][2]*/, const uint32_t *seg_off /*[2F]*/,
                       const uint32_t *seg_cnt /*[2F]*/, const int32_t *kept_labels /*[n_points]*/, uint32_t n_points,
                       const double *pose /*[F][12]*/, uint32_t F, const double *landmarks, const ecal_rectify_params *prm,
                       double *feat_xyr, uint32_t *feat_valid, uint32_t *frame_info);

/* rectifyFeatures for F keyframes named by their time windows over a resident ecal_stream (durations [F][2]; what the batched
 * path of host/event_calib_ini.hpp calls after solvePnPRansac, EventCalibIni.cpp:281-302): the EventFrame constructor and
 * extractFeatures up to the kept clusters run for every window on the device, then ecal_rectify_batch_dev — only the poses go
 * up and the rectified circles (feat_xyr [F][rows*cols][3], feat_valid, frame_info [F][2], layouts as above) come back. */
int ecal_rectify_keyframes(ecal_ctx *ctx, const ecal_stream *es, const double *durations /*[F][2]*/, uint32_t F,
                           const ecal_detect_params *detect_prm, const double *poses /*[F][12]*/, const double *landmarks,
                           const ecal_rectify_params *prm, double *feat_xyr, uint32_t *feat_valid, uint32_t *frame_info);

/* ---- event -> residual association ----------------------------------------------------------------
 * Replaces the association loop of EventCalibSpline::optimize (event_camera_calib/src/EventCalibSpline.cpp:
 * 140-192) and CirclesEventFrame::findCenter (include/opengv2/event_camera_calib/CirclesEventFrame.hpp:50-65):
 * every event with t_min <= t <= t_max (the spline's range) looks up the keyframe nearest in time
 * (d_kf_time ascending, accepted if |dt| < max_dt = 5 * MotionTimeStep) and that keyframe's nearest circle
 * centre (d_kf_circles [K][n_circles][3] = cx, cy, radius in pixels, grid order), accepted if
 * | |pixel - centre| - radius | < edge_tol (5 px).  Accepted events are written in event order:
 * d_obs[j] = pixel, d_time[j] = t, d_lm_id[j] = circle index (= landmark index); *d_count = how many
 * (outputs need room for n_events entries).  These arrays are ecal_spline_problem's obs/time/lm_id. */
int ecal_associate_dev(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, const double *d_kf_time,
                       const double *d_kf_circles, uint32_t n_keyframes, uint32_t n_circles, double t_min, double t_max,
                       double max_dt, double edge_tol, double *d_obs, double *d_time, uint32_t *d_lm_id,
                       uint32_t *d_count, void *stream);

/* The same for ALL spline segments of a calibration in one pass over the stream: d_ranges [R][2] = (t_min, t_max) of segment r
 * (REQUIRED ascending and disjoint — t_min[r] <= t_max[r] < t_min[r + 1]; EventCalibSpline.cpp:318-345 cuts the keyframes at
 * gaps, so the reference's are —: the device finds an event's range by bisection and cannot report a violation from a device
 * table; ecal_solver_create_from_stream, which takes the ranges from the host, checks and returns ECAL_ERR_INVALID);
 * an event inside range r that passes the two
 * gates becomes a residual of segment r: d_seg_id[j] = r.  Outputs in event order = sorted by (segment, time), which is what
 * ecal_solver_create[_dev] takes; *d_count stays on the device (ecal_solver_create_dev reads it there). */
int ecal_associate_ranges_dev(ecal_ctx *ctx, const uint8_t *d_events, uint64_t n_events, const double *d_kf_time,
                              const double *d_kf_circles, uint32_t n_keyframes, uint32_t n_circles, const double *d_ranges,
                              uint32_t n_ranges, double max_dt, double edge_tol, double *d_obs, double *d_time, uint32_t *d_lm_id,
                              uint32_t *d_seg_id, uint32_t *d_count, void *stream);

/* host-buffer form: events from an ecal_stream (resident in HBM), keyframe tables and results in host memory;
 * obs/time/lm_id need room for `capacity` records, *count = records found (ECAL_ERR_RANGE if more than capacity) */
int ecal_associate(ecal_ctx *ctx, const ecal_stream *es, const double *kf_time, const double *kf_circles, uint32_t n_keyframes,
                   uint32_t n_circles, double t_min, double t_max, double max_dt, double edge_tol, uint64_t capacity, double *obs,
                   double *time, uint32_t *lm_id, uint64_t *count);

/* ---- continuous-time calibration solve ---------------------------------------------------------
 * Replaces the Ceres problem of EventCalibSpline::optimize (event_camera_calib/src/EventCalibSpline.cpp:
 * 196-247) for both rotation-spline variants (ecal_spline_problem.use_so3): one residual per associated event,
 *   r = | Xw(event pixel; intrinsics, pose(t)) - landmark | - circle_radius
 * (CalibReprojectionError::operator(), EventCalibSpline.hpp:158-229; unDistort :36-63), HuberLoss(huber_a)
 * with huber_a = 0.2 * circle_radius (:205), EigenQuaternionParameterization on the rotation control
 * points (:116-135), Levenberg-Marquardt with the options of :238-243.
 *
 * Parameter vector (doubles): [ fx fy cx cy k1..k5 | q_c (x y z w) for c < n_cp | t_c (x y z) for c < n_cp ],
 * n_cp = seg_cp_off[n_segments] control points; segment g owns control points
 * [seg_cp_off[g], seg_cp_off[g+1]) and the clamped knot vector knots[seg_cp_off[g] + 4g ..] of
 * n_cp_g + 4 entries (degree 3: BsplineReal<4>/<3> of one segment share it, EventCalibSpline.cpp:68-91).
 * Residual records (the output of the association step, EventCalibSpline.cpp:158-192) must be sorted
 * by (segment, time); 32 bytes each on the device (obs 16 + time 8 + landmark 4 + segment 4): spans and
 * basis values are recomputed from the time (BsplineReal.hpp:107-145,208-231).
 *
 * Normal-equation buffer of ecal_solver_evaluate[_dev] (ecal_solver_normal_size doubles, tangent space,
 * Huber-corrected, SUMS over this process's residuals — ranks add theirs with an all-reduce):
 *   [0] cost = sum rho/2 | [1..9] (J^T r)_intr | [10..90] (J^T J)_intr,intr (9x9 row-major, upper part)
 *   then per control point c, 204 doubles: (J^T r)_c [6: d_rot 3, d_trans 3] | (J^T J)_c,intr [6][9] |
 *   (J^T J)_c,c+d [4][6][6] for d = 0..3 (d = 0: upper part).  with_jacobian = 0 fills only [0].
 */
typedef struct ecal_solver ecal_solver;
typedef struct ecal_spline_problem {
    uint32_t n_segments;
    const uint32_t *seg_cp_off; /* [n_segments + 1] */
    const double *knots;        /* [n_cp + 4 n_segments] */
    uint64_t n_res;
    const double *obs;          /* [n_res][2] event pixel */
    const double *time;         /* [n_res] */
    const uint32_t *lm_id;      /* [n_res] index into landmarks */
    const uint32_t *seg_id;     /* [n_res] or NULL (= all segment 0) */
    uint32_t n_landmarks;
    const double *landmarks;    /* [n_landmarks][3] */
    double circle_radius;       /* Circles_Radius */
    double huber_a;             /* 0.2 * circle_radius in the reference */
    int use_so3;                /* useSO3 (eventCameraCalib.cpp:204-208): 0 = quaternion spline + EigenQuaternion-
                                   Parameterization; 1 = cumulative SO3 spline (CalibReprojectionError_SO3,
                                   EventCalibSpline.hpp:65-135) + LocalParameterizationSO3 (q <- q * exp(delta)) */
    int camera_model;           /* ECAL_CAMERA_RADIAL (the reference's unDistort, EventCalibSpline.hpp:36-63: k1..k5 = the
                                   inverse radial polynomial) or ECAL_CAMERA_FISHEYE (BASELINE configs[4], new: the reference's
                                   solver refuses anything else, EventCalibSpline.cpp:97-99): Kannala-Brandt in the same inverse
                                   form — theta = theta_d (1 + k1 theta_d^2 + .. + k5 theta_d^10) with theta_d = |((u-cx)/fx,
                                   (v-cy)/fy)|, ray = (x, y) tan(theta) / theta_d; k1..k5 are initialised from cv::fisheye's
                                   forward k1..k4 by the same series reversion (ecal_inverse_radial_distortion) */
} ecal_spline_problem;
#define ECAL_CAMERA_RADIAL 0
#define ECAL_CAMERA_FISHEYE 1
/* ---- multi-GPU: one process per GPU, one RCCL communicator per context ---------------------------------------
 * The reference has no distributed backend; north_star shards calibration views and spline residuals one batch per GPU
 * and sums the per-view / per-rank normal-equation blocks with an RCCL all-reduce over xGMI.  Rank 0 calls
 * ecal_comm_unique_id and hands the ECAL_COMM_ID_BYTES bytes to the other ranks by any means (a file, a socket, MPI,
 * torch.distributed); then EVERY rank calls ecal_comm_init on its own context (collective: returns when all world_size
 * ranks have joined; one rank per GPU — RCCL refuses two ranks on one device).  Joining changes nothing by itself: a call
 * whose options carry allreduce == NULL stays rank-local.  A collective solve / calibration is asked for explicitly with
 * options.allreduce = ecal_comm_allreduce and options.allreduce_user = the context (ecal_lm_options.rank / world_size =
 * ecal_comm_rank / ecal_comm_size of that context; anything else is ECAL_ERR_INVALID).  ecal_comm_allreduce_sum_dev: d_buf[0 ..
 * n) summed over the ranks in place, enqueued on `stream` (no host synchronisation); a no-op without a communicator. */
#define ECAL_COMM_ID_BYTES 128
int ecal_comm_unique_id(void *id_out /*[ECAL_COMM_ID_BYTES]*/);
int ecal_comm_init(ecal_ctx *ctx, const void *unique_id, int rank, int world_size);
int ecal_comm_destroy(ecal_ctx *ctx);
int ecal_comm_size(const ecal_ctx *ctx);   /* 1 without a communicator */
int ecal_comm_rank(const ecal_ctx *ctx);
int ecal_comm_allreduce_sum_dev(ecal_ctx *ctx, double *d_buf, size_t n_doubles, void *stream);

/* all-reduce of the solver / calibration: any transport with this signature, or ecal_comm_allreduce (user = the ecal_ctx that
 * joined the communicator; ECAL_ERR_COMM if it never did) */
typedef int (*ecal_allreduce_fn)(void *user, double *d_buf, size_t n_doubles, void *stream);
int ecal_comm_allreduce(void *user /*ecal_ctx* */, double *d_buf, size_t n_doubles, void *stream);
typedef struct ecal_lm_options {
    int max_num_iterations;
    double function_tolerance, gradient_tolerance, parameter_tolerance;
    double initial_trust_region_radius, max_trust_region_radius, min_relative_decrease;
    double min_lm_diagonal, max_lm_diagonal;
    int jacobi_scaling;
    ecal_allreduce_fn allreduce; /* NULL: this rank alone.  ecal_comm_allreduce (+ allreduce_user = ctx): the context's RCCL communicator */
    void *allreduce_user;
    /* distributed == 1 (needs allreduce): every rank owns its OWN spline segments in its own ecal_solver (its residuals,
     * its control points) and only the 9 intrinsics are shared.  Exchanged per evaluation: the 91-double head (cost,
     * intrinsics gradient and block); per linear solve: the 10 x 10 Schur sums of the rank's banded factorisation + a
     * failure flag + one slot per rank (gradient max-norm); per step: four scalars.  Every rank factorises only its own
     * band and returns its own control points; all ranks return the same intrinsics, cost and summary.
     * distributed == 0 with allreduce set: every rank holds the whole parameter vector and the whole normal-equation
     * buffer is summed (residuals of one segment may then be spread over ranks).
     * distributed == 2 (needs allreduce): TIME SHARDS OF ONE SPLINE — every rank builds its ecal_solver with
     * the whole spline layout (one segment) and the residuals of ITS time range, cut at ecal_solver_time_shard_cuts' times, and
     * passes the same start vector.  The control points fall into world_size interiors separated by 3-control-point
     * separators; exchanged per Jacobian evaluation: the 91-double head + the separators' records (612 doubles per cut); per
     * linear solve: the interiors' 46 x 46 Gram blocks (1082 doubles per rank); per step: four scalars; once at the end: the
     * parameter vector.  Every rank factorises its own interior and all return the same, complete solution. */
    int distributed, rank, world_size;
} ecal_lm_options;
/* cut_time [world_size - 1]: rank r owns the residuals with cut_time[r - 1] <= t < cut_time[r] (distributed == 2) */
int ecal_solver_time_shard_cuts(const double *knots /*[n_cp + 4]*/, uint32_t n_cp, int world_size, double *cut_time);
typedef struct ecal_lm_summary {
    int iterations, successful_steps, unsuccessful_steps, jacobian_evaluations, cost_evaluations;
    int termination; /* 0 = converged (a tolerance fired), 1 = max_num_iterations reached */
    double initial_cost, final_cost, seconds;
    double seconds_evaluate;     /* H2D parameters + kernels + all-reduce + D2H buffer, summed */
    double seconds_linear_solve; /* banded-arrow Cholesky + model change on the host, summed */
} ecal_lm_summary;
int ecal_solver_create(ecal_ctx *ctx, const ecal_spline_problem *problem, ecal_solver **out);
/* The problem built in place (EventCalibSpline.cpp:181-235 adds the residual blocks where the association finds them):
 * problem->obs / time / lm_id / seg_id are DEVICE pointers — the outputs of ecal_associate_ranges_dev, consumed when the call
 * returns —, problem->n_res their capacity and *d_n_res (device; NULL: n_res itself) the number of residuals; the other members
 * (seg_cp_off, knots, landmarks) stay host pointers.  Records and chunk table are made by kernels on `stream`; one 12-byte
 * read-back.  Same solver object, same checks (ECAL_ERR_INVALID) as ecal_solver_create. */
int ecal_solver_create_dev(ecal_ctx *ctx, const ecal_spline_problem *problem, const uint32_t *d_n_res, void *stream,
                           ecal_solver **out);
uint64_t ecal_solver_num_residuals(const ecal_solver *s);
/* Host-pointer convenience of the two (what host/event_calib_spline.hpp calls): the association of every spline segment over a
 * resident ecal_stream + the solver built on the result, the residual arrays never leaving HBM.  kf_time / kf_circles / ranges
 * [n_ranges][2] are host tables (ecal_associate_ranges_dev's arguments); layout = the problem without its residual arrays (obs,
 * time, lm_id, seg_id, n_res ignored); n_ranges must equal layout->n_segments.  ecal_solver_num_residuals says how many were found. */
int ecal_solver_create_from_stream(ecal_ctx *ctx, const ecal_stream *es, const double *kf_time, const double *kf_circles,
                                   uint32_t n_keyframes, uint32_t n_circles, const double *ranges, uint32_t n_ranges, double max_dt,
                                   double edge_tol, const ecal_spline_problem *layout, ecal_solver **out);
void ecal_solver_destroy(ecal_solver *s);
size_t ecal_solver_param_size(const ecal_solver *s);
size_t ecal_solver_normal_size(const ecal_solver *s);
uint32_t ecal_solver_num_chunks(const ecal_solver *s);
int ecal_solver_evaluate_dev(ecal_solver *s, const double *d_params, int with_jacobian, double *d_accum, void *stream);
int ecal_solver_evaluate(ecal_solver *s, const double *params, int with_jacobian, double *accum);
/* The Ceres CostFunction::Evaluate seam (CalibReprojectionError{,_SO3}::Create, EventCalibSpline.hpp:137-146,231-240:
 * AutoDiffCostFunction<..., 1, 9, 4, 4, 4, 4, 3, 3, 3, 3> + EigenQuaternionParameterization / LocalParameterizationSO3) for
 * every residual of the problem at `params`: r[k] = the raw residual (no loss function); J[k][33] (optional) = the raw row
 * of the tangent-space Jacobian, columns [ intrinsics 9 | rotation tangent of control points cp0[k] .. cp0[k]+3, 3 each |
 * translation of the same four, 3 each ]; cp0[k] (optional) = the first of the four control points residual k touches. */
int ecal_residuals_dev(ecal_solver *s, const double *d_params, double *d_r /*[n_res]*/, double *d_J /*[n_res][33] or NULL*/,
                       uint32_t *d_cp0 /*[n_res] or NULL*/, void *stream);
int ecal_residuals(ecal_solver *s, const double *params, double *r, double *J, uint32_t *cp0);
void ecal_lm_default_options(ecal_lm_options *opt);
int ecal_solver_solve(ecal_solver *s, double *params /*in: start, out: solution*/, const ecal_lm_options *opt,
                      ecal_lm_summary *summary);
/* PinholeCamera::inverseRadialDistortion (core/sensor/src/PinholeCamera.cpp:69-95): (k1,k2,k3,k4) -> the
 * five inverse-polynomial coefficients that initialise k1..k5 (EventCalibSpline.cpp:101-105). */
void ecal_inverse_radial_distortion(const double *k4, double *b5);

/* ---- init calibration on calibration views ---------------------------------------------------------
 * Replaces the OpenCV calls of EventCalibIni::cvCalibration (event_camera_calib/src/EventCalibIni.cpp:149-347):
 *   cv::calibrateCamera(objectPoints, imagePoints, imageSize, K, dist(8), rvecs, tvecs, flag | CALIB_USE_LU)  :198-199
 *   cv::fisheye::calibrate(objectPoints, imagePoints, imageSize, K, dist(4), rvecs, tvecs, flag)              :188
 *   cv::solvePnPRansac(objectPoints[0], imageP, K, dist, rvec, tvec, false, 50, 4.0, 0.99, inliers, IPPE)     :258-259
 * with the flag word assembled as in parameters.hpp:48-69.  OpenCV is third party (not in the reference tree):
 * the algorithms are restated (oracle/calib_oracle.py lists them), parity with OpenCV itself is unpinned.
 *
 * Intrinsics slots (12 doubles): model 0 = fx fy cx cy k1 k2 p1 p2 k3 k4 k5 k6 (cv::calibrateCamera's K and its
 * 8 distortion coefficients in OpenCV's order), model 1 = fx fy cx cy alpha k1 k2 k3 k4 (cv::fisheye).
 * A view = the n_pts circle centres of one keyframe in grid order (CirclesEventFrame::features()); the board
 * points obj [n_pts][3] are calcBoardCornerPositions (EventCalibIni.cpp:99-115) and must have z == 0.
 * Pose of a view: rvec (cv::Rodrigues vector) and tvec, camera <- board.
 *
 * ecal_calib_view_blocks_dev: the data-parallel piece — per view v the normal-equation blocks of its
 *   2 n_pts reprojection residuals r = projected - measured with the analytic Jacobian J (columns of fixed
 *   intrinsics zeroed; with FIX_ASPECT_RATIO fx is tied to aspect_ratio * fy and its column folded into fy's):
 *   d_blocks[272 v ..] = Hii [12][12] | Hiv [12][6] | Hvv [6][6] | gi = (J^T r)_intr [12] | gv [6] | cost = r^T r | pad.
 *   with_jacobian = 0 writes only cost (slot 270).  Blocks of different views are independent: views shard across
 *   GPUs and only Schur-reduced 12 x 12 records are summed (ecal_calibrate_views below).
 * ecal_pnp_batch_dev: planar pose of F frames at once — undistort, homography, IPPE (both solutions, the one with
 *   the smaller reprojection error), `rounds` consensus rounds with inliers = reprojection error <= reproj_thresh
 *   px (0 = none; a deterministic stand-in for the RANSAC loop), then refine_iters Levenberg-Marquardt iterations
 *   on the pose (0 = none, as solvePnPRansac's final SOLVEPNP_IPPE call; 20 = cvFindExtrinsicCameraParams2).
 *   d_valid [F][n_pts] (or NULL) masks missing circles.  d_pose [F][6] = rvec, tvec; d_inlier [F][n_pts];
 *   d_err [F] = sum of squared reprojection errors over the inliers; d_ok [F] = 0 if no pose was found.
 * ecal_calibrate_views: the whole calibration for this process's views (host pointers).  With options.allreduce
 *   set, every rank passes its own shard of the views (n_views may differ, 0 allowed) and all ranks return the
 *   same intrinsics; rvecs / tvecs / per_view_err are those of the local views.
 */
#define ECAL_CALIB_FIX_ASPECT_RATIO    (1u << 0)  /* cv::CALIB_FIX_ASPECT_RATIO: fx = aspect_ratio * fy */
#define ECAL_CALIB_FIX_PRINCIPAL_POINT (1u << 1)  /* cv::CALIB_FIX_PRINCIPAL_POINT / fisheye::CALIB_FIX_PRINCIPAL_POINT */
#define ECAL_CALIB_ZERO_TANGENT_DIST   (1u << 2)  /* cv::CALIB_ZERO_TANGENT_DIST */
#define ECAL_CALIB_FIX_K1              (1u << 3)
#define ECAL_CALIB_FIX_K2              (1u << 4)
#define ECAL_CALIB_FIX_K3              (1u << 5)
#define ECAL_CALIB_FIX_K4              (1u << 6)
#define ECAL_CALIB_FIX_K5              (1u << 7)
#define ECAL_CALIB_FIX_K6              (1u << 8)
#define ECAL_CALIB_FIX_SKEW            (1u << 9)  /* cv::fisheye::CALIB_FIX_SKEW */
#define ECAL_CALIB_RECOMPUTE_EXTRINSIC (1u << 10) /* cv::fisheye::CALIB_RECOMPUTE_EXTRINSIC */
#define ECAL_CALIB_USE_INTRINSIC_GUESS  (1u << 11) /* cv::CALIB_USE_INTRINSIC_GUESS / cv::fisheye::CALIB_USE_INTRINSIC_GUESS: res->intr
                                                     (all 12 slots) is the START of the iteration instead of the models' own
                                                     initialisation (homographies / max(w, h) / pi) */
#define ECAL_CALIB_BLOCK_DOUBLES 272
typedef struct ecal_calib_options {
    int model;            /* 0 = pinhole (Calibrate_UseFisheyeModel: 0), 1 = fisheye */
    uint32_t flags;       /* ECAL_CALIB_* */
    double aspect_ratio;  /* Calibrate_FixAspectRatio (used with ECAL_CALIB_FIX_ASPECT_RATIO) */
    int max_iter;         /* 0 = OpenCV's default TermCriteria: 30 (calibrateCamera) / 100 (fisheye) */
    double eps;           /* 0 = DBL_EPSILON */
    ecal_allreduce_fn allreduce; /* NULL: this rank alone.  ecal_comm_allreduce (+ allreduce_user = ctx): the context's RCCL communicator */
    void *allreduce_user;
} ecal_calib_options;
typedef struct ecal_calib_result {
    double intr[12];
    double rms;           /* sqrt(sum of squared reprojection errors / number of points), all ranks' views */
    int iterations, jacobian_evaluations, error_evaluations;
    double seconds;
} ecal_calib_result;
void ecal_calib_default_options(ecal_calib_options *opt);
int ecal_calib_view_blocks_dev(ecal_ctx *ctx, const double *d_obj /*[n_pts][3]*/, uint32_t n_pts, const double *d_img /*[V][n_pts][2]*/,
                               uint32_t n_views, int model, uint32_t flags, double aspect_ratio, const double *d_intr /*[12]*/,
                               const double *d_view_params /*[V][6]*/, int with_jacobian, double *d_blocks /*[V][272]*/, void *stream);
int ecal_pnp_batch_dev(ecal_ctx *ctx, const double *d_obj, uint32_t n_pts, const double *d_img /*[F][n_pts][2]*/,
                       const uint32_t *d_valid /*[F][n_pts] or NULL*/, uint32_t n_frames, int model, const double *d_intr /*[12]*/,
                       double reproj_thresh, int rounds, int refine_iters, double *d_pose /*[F][6]*/, uint32_t *d_inlier /*or NULL*/,
                       double *d_err /*or NULL*/, uint32_t *d_ok /*or NULL*/, void *stream);
/* host-buffer form of ecal_pnp_batch_dev */
int ecal_pnp_batch(ecal_ctx *ctx, const double *obj, uint32_t n_pts, const double *img, const uint32_t *valid /*or NULL*/,
                   uint32_t n_frames, int model, const double *intr /*[12]*/, double reproj_thresh, int rounds, int refine_iters,
                   double *pose /*[F][6]*/, uint32_t *inlier /*or NULL*/, double *err /*or NULL*/, uint32_t *ok /*or NULL*/);
/* The sequential keyframe gates of EventCalibIni::cvCalibration (event_camera_calib/src/EventCalibIni.cpp:281-302) on results that
 * are known for all keyframes at once (ecal_pnp_batch, ecal_rectify_keyframes): frame f, in time order, is discarded when its PnP
 * failed or EventCalibIni::checkPose (:327-347: translational speed |twb_f - twb_last| / dt below (0.25 / step) * 2 and angular
 * speed |acos((trace(Rsw_f Rsw_last^T) - 1) / 2)| / dt below (5e-4 pi) * 2 / step) fails against the LAST ACCEPTED frame (the first
 * accepted frame has none to be checked against), then when its rectification failed; otherwise it is accepted.  Host code, no
 * context: the loop carries a dependence from frame to frame.  accepted[0 .. *n_accepted) = the accepted frames' indices. */
int ecal_pose_gates(uint32_t n_frames, const double *Rsw /*[F][9] row-major*/, const double *twb /*[F][3]*/, const double *time /*[F]*/,
                    const uint8_t *pnp_ok /*[F]*/, const uint8_t *rect_ok /*[F]*/, double motion_time_step,
                    uint32_t *accepted /*[F]*/, uint32_t *n_accepted, uint32_t *n_discarded_by_check_pose,
                    uint32_t *n_discarded_by_rectify);
int ecal_calibrate_views(ecal_ctx *ctx, const double *obj /*[n_pts][3]*/, uint32_t n_pts, const double *img /*[V][n_pts][2]*/,
                         uint32_t n_views, double width, double height, const ecal_calib_options *opt, ecal_calib_result *res,
                         double *rvecs /*[V][3] or NULL*/, double *tvecs /*[V][3] or NULL*/, double *per_view_err /*[V] or NULL*/);

/* The fisheye init calibration (cv::fisheye::calibrate at EventCalibIni.cpp:186-190) with ONE start procedure for every
 * caller: the reference's own start first (principal point at the centre, f = max(w, h) / pi, no guess); only when that fails —
 * ECAL_ERR_INVALID or a non-finite result — the radial model is calibrated on the same views with
 * ECAL_CALIB_FISHEYE_PRECALIB_FLAGS and its fx, fy, cx, cy start the fisheye model (ECAL_CALIB_USE_INTRINSIC_GUESS).
 * *start_used: 0 the reference's start, 1 the radial guess was needed, 2 the caller's own guess (opt->flags carries
 * ECAL_CALIB_USE_INTRINSIC_GUESS, res->intr the guess).  opt->model is ignored (fisheye). */
#define ECAL_CALIB_FISHEYE_PRECALIB_FLAGS                                                                                  \
    (ECAL_CALIB_FIX_PRINCIPAL_POINT | ECAL_CALIB_ZERO_TANGENT_DIST | ECAL_CALIB_FIX_ASPECT_RATIO | ECAL_CALIB_FIX_K3 | ECAL_CALIB_FIX_K4 | \
     ECAL_CALIB_FIX_K5 | ECAL_CALIB_FIX_K6)
int ecal_calibrate_fisheye_views(ecal_ctx *ctx, const double *obj, uint32_t n_pts, const double *img, uint32_t n_views, double width,
                                 double height, const ecal_calib_options *opt, ecal_calib_result *res, double *rvecs, double *tvecs,
                                 double *per_view_err, int *start_used);

/* ---- initial spline fit + evaluation (host; no GPU involved) -------------------------------------------
 * ecal_spline_fit: BsplineReal<dim>(3, Q, controlPointsNum, u) (core/spline/include/opengv2/spline/
 *   BsplineReal.hpp:17-100,329-449) as EventCalibSpline builds twbSplines_ / QwbSplines_ from the keyframe poses
 *   (event_camera_calib/src/EventCalibSpline.cpp:61-91), and BsplineSO3's knotSpacing + initialGuess
 *   (core/spline/src/BsplineSO3.cpp:60-72,198-279; its Ceres refinement optimizeCP: ecal_spline_so3_refine below).  u [m] ascending sample parameters (timestamps, first and
 *   last already widened by 3 steps as :63-66), data [m][dim]; outputs the clamped knot vector [n_cp + 4]
 *   (NURBS book 9.68) — the layout ecal_spline_problem.knots takes — and the control points [n_cp][dim]: first and
 *   last interpolate, the interior ones minimise the squared distance at the interior samples.
 *   ECAL_ERR_INVALID if n_cp < 4, m < 2 or a control point has no supporting sample.
 * ecal_spline_eval: BsplineReal::evaluate(u, 0, ..) (:454-470) at m parameters (updateMap, EventCalibSpline.cpp:
 *   253-317); ECAL_ERR_RANGE outside the knot range. */
/* ecal_spline_so3_refine: BsplineSO3::optimizeCP (core/spline/src/BsplineSO3.cpp:285-341) — the control points of the
 *   cumulative cubic SO3 spline (unit quaternions x y z w, [n_cp][4], in: the initial guess of ecal_spline_fit on the
 *   quaternion coefficients, out: refined) fitted on the group to the sample rotations [m][4] at parameters u [m]:
 *   minimises sum_i |log(S_i^-1 X(u_i))|^2 / 2 (P3ApproximationError, BsplineSO3.hpp:121-153) over steps cp <- cp exp(delta)
 *   (LocalParameterizationSO3, :190-222), first and last control point constant, Ceres' trust-region loop restated,
 *   function / gradient tolerance 1e-10, max_iterations <= 0 = Ceres' default 50.  Host only. */
int ecal_spline_so3_refine(const double *knots /*[n_cp+4]*/, uint32_t n_cp, double *cp_quat /*[n_cp][4]*/,
                           const double *sample_quat /*[m][4]*/, const double *u /*[m]*/, uint32_t m, int max_iterations,
                           double *initial_cost /*or NULL*/, double *final_cost /*or NULL*/, int *iterations /*or NULL*/);
int ecal_spline_fit(const double *u, const double *data, uint32_t m, uint32_t dim, uint32_t n_cp, double *knots /*[n_cp+4]*/,
                    double *cp /*[n_cp][dim]*/);
int ecal_spline_eval(const double *knots, const double *cp, uint32_t n_cp, uint32_t dim, const double *u, uint32_t m,
                     double *out /*[m][dim]*/);

#ifdef __cplusplus
}
#endif
#endif /* ECAL_H_ */
