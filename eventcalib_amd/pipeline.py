"""Device-resident detection pipeline: packed event stream -> windows -> unique pixel sets ->
DBSCAN labels (-> circle candidates), every stage a HIP kernel behind the C ABI.

torch is plumbing only here: it owns the HBM buffers and the stream; all compute goes through
libecal.so (eventcalib_amd.capi).  Mirrors the per-piece loop of the reference driver
(event_camera_calib/test/eventCameraCalib.cpp:49-62) for a whole batch of windows at once.
"""
import numpy as np
import torch

from .capi import Context, PackedPoints

RECORD = 25


class DetectPipeline:
    """Buffers are sized once (grow-only) so a timed loop performs no allocation."""

    def __init__(self, ctx: Context, device=None, packed=True, want_event_point=True):
        """packed: between the stages the points of integer-pixel windows travel as 4-byte words (ecal_packed_points) instead
        of 16-byte doubles; `xy` (positiveEvents_ / negativeEvents_ as doubles) is then written on first access.  Results are
        bit for bit the same either way."""
        self.ctx = ctx
        self.dev = torch.device("cuda", ctx.device) if device is None else device
        self._cap_windows = 0
        self._cap_slots = 0
        self.packed = bool(packed)
        self.want_event_point = bool(want_event_point)   # False: ecal_slice_events_*_dev gets d_event_point = NULL
        self._xy_stale = False

    @property
    def xy(self):
        """[slots, 2] f64: the windows' points as doubles (unpacked on demand after a packed run)."""
        if self._xy_stale:
            st = torch.cuda.current_stream(self.dev).cuda_stream
            self.ctx.unpack_points_dev(self._pk(), self.seg_off.data_ptr(), self.seg_cnt.data_ptr(), 2 * self.S, self._xy.data_ptr(), st)
            self._xy_stale = False
        return self._xy

    def _pk(self):
        return PackedPoints(self.xy16.data_ptr(), self.seg_fmt.data_ptr())

    def _ensure(self, S, slots):
        dev = self.dev
        if S > self._cap_windows:
            self.win_lo = torch.empty(S, dtype=torch.int32, device=dev)
            self.win_hi = torch.empty(S, dtype=torch.int32, device=dev)
            self.win_base = torch.empty(S + 1, dtype=torch.int32, device=dev)
            self.seg_off = torch.empty(2 * S, dtype=torch.int32, device=dev)
            self.seg_cnt = torch.empty(2 * S, dtype=torch.int32, device=dev)
            self.n_clusters = torch.empty(2 * S, dtype=torch.int32, device=dev)
            self.flags = torch.zeros(4, dtype=torch.int32, device=dev)
            self.win_info = torch.empty(S, 4, dtype=torch.int32, device=dev)
            self.grid_order = torch.empty(S, 128, dtype=torch.int32, device=dev)
            self.grid_found = torch.empty(S, dtype=torch.int32, device=dev)
            self.seg_fmt = torch.zeros(2 * S, dtype=torch.int32, device=dev)
            self._cap_windows = S
        if slots > self._cap_slots:
            self._xy = torch.empty(slots, 2, dtype=torch.float64, device=dev)
            self.xy16 = torch.empty(slots, dtype=torch.int32, device=dev)
            self.event_point = torch.empty(slots, dtype=torch.int32, device=dev)
            self.labels = torch.empty(slots, dtype=torch.int32, device=dev)
            self.kept_labels = torch.empty(slots, dtype=torch.int32, device=dev)
            self.rep = torch.empty(slots, dtype=torch.int32, device=dev)
            self.cand_pair = torch.empty(slots, 2, dtype=torch.int32, device=dev)
            self.cand_xyr = torch.empty(slots, 3, dtype=torch.float64, device=dev)
            self._cap_slots = slots

    def set_windows(self, t0, t1):
        """Inclusive windows [t0[s], t1[s]] (numpy float64 or torch)."""
        both = torch.as_tensor(np.stack([np.asarray(t0, dtype=np.float64), np.asarray(t1, dtype=np.float64)])).to(self.dev)
        self.t0, self.t1 = both[0], both[1]          # one upload for both bounds
        self.S = int(self.t0.numel())

    def set_detect_params(self, cluster_min=5, need_clusters=36, radius_threshold=15.511363636363637, fit_circle=False,
                          knn_num=3):
        """clusterMinSample, rows*cols, circleRadiusThreshold_, fitCircle, knn_num (defaults: example.yaml, 346x260)."""
        self.det = (int(cluster_min), int(need_clusters), float(radius_threshold), bool(fit_circle), int(knn_num))

    def run(self, events, eps=4.0, minpts=2, slots=None, max_win_events=0, max_seg_points=0, detect=True, slice_only=False,
            fused=False, exact_ties=True):
        """events: uint8 CUDA tensor holding n*25 bytes.  Enqueues bounds -> slice -> DBSCAN -> candidate
        extraction on the current torch stream; results stay in HBM (self.xy / seg_off / seg_cnt /
        labels / n_clusters / win_info / cand_pair / cand_xyr / kept_labels / rep)."""
        assert events.is_cuda and events.dtype == torch.uint8
        n = events.numel() // RECORD
        S = self.S
        slots = n if slots is None else slots
        self._ensure(S, slots)
        st = torch.cuda.current_stream(self.dev).cuda_stream
        c = self.ctx
        ep_ptr = self.event_point.data_ptr() if self.want_event_point else 0
        c.window_bounds_dev(events.data_ptr(), n, self.t0.data_ptr(), self.t1.data_ptr(), S, self.win_lo.data_ptr(),
                            self.win_hi.data_ptr(), self.win_base.data_ptr(), st)
        self._xy_stale = False
        if fused:               # the three stages below as one call (ecal_detect_fused_dev): same arrays, same results
            if not hasattr(self, "det"):
                self.set_detect_params()
            c.detect_fused_dev(events.data_ptr(), n, self.win_lo.data_ptr(), self.win_hi.data_ptr(), self.win_base.data_ptr(), S,
                               max_win_events, max_seg_points, slots, eps, minpts, self.det[0], self.det[1], self.det[2],
                               self._xy.data_ptr(), self.seg_off.data_ptr(), self.seg_cnt.data_ptr(), self.event_point.data_ptr(),
                               self.flags.data_ptr(), self.labels.data_ptr(), self.n_clusters.data_ptr(), self.win_info.data_ptr(),
                               self.cand_pair.data_ptr(), self.cand_xyr.data_ptr(), self.kept_labels.data_ptr(), self.rep.data_ptr(),
                               st, fit_circle=self.det[3], knn_num=self.det[4])
            return self
        if self.packed:
            pk = self._pk()
            self._xy_stale = True
            c.slice_events_packed_dev(events.data_ptr(), n, self.win_lo.data_ptr(), self.win_hi.data_ptr(), self.win_base.data_ptr(), S,
                                      max_win_events, slots, self._xy.data_ptr(), self.seg_off.data_ptr(), self.seg_cnt.data_ptr(),
                                      ep_ptr, self.flags.data_ptr(), pk, st)
            if slice_only:
                return self
            c.dbscan_batch_packed_dev(self._xy.data_ptr(), self.seg_off.data_ptr(), self.seg_cnt.data_ptr(), 2 * S, slots, max_seg_points,
                                      eps, minpts, self.labels.data_ptr(), self.n_clusters.data_ptr(), pk, st)
            if detect:
                if not hasattr(self, "det"):
                    self.set_detect_params()
                want = 0 if exact_ties else 1          # ECAL_TIES_REFERENCE / ECAL_TIES_SMALLER_PID
                was = c.get_median_ties()
                if was != want:
                    c.set_median_ties(want)
                try:
                    c.extract_batch_packed_dev(self._xy.data_ptr(), self.seg_off.data_ptr(), self.seg_cnt.data_ptr(), self.labels.data_ptr(),
                                               self.n_clusters.data_ptr(), S, slots, eps, self.det[0], self.det[1], self.det[2],
                                               self.win_info.data_ptr(), self.cand_pair.data_ptr(), self.cand_xyr.data_ptr(),
                                               self.kept_labels.data_ptr(), self.rep.data_ptr(), pk, st, fit_circle=self.det[3],
                                               knn_num=self.det[4])
                finally:
                    if was != want:
                        c.set_median_ties(was)
            return self
        c.slice_events_dev(events.data_ptr(), n, self.win_lo.data_ptr(), self.win_hi.data_ptr(),
                           self.win_base.data_ptr(), S, max_win_events, slots, self._xy.data_ptr(),
                           self.seg_off.data_ptr(), self.seg_cnt.data_ptr(), ep_ptr,
                           self.flags.data_ptr(), st)
        if slice_only:          # profiling aid: bounds + slicing only
            return self
        c.dbscan_batch_dev(self._xy.data_ptr(), self.seg_off.data_ptr(), self.seg_cnt.data_ptr(), 2 * S, slots,
                           max_seg_points, eps, minpts, self.labels.data_ptr(), self.n_clusters.data_ptr(), st)
        if detect and exact_ties:
            # the reference's own representative where a cluster's median rank is tied in norm: the members' order inside
            # Clusters[c] + libstdc++'s nth_element on it, for the windows that have such a cluster (ecal_extract_batch_exact_dev)
            if not hasattr(self, "det"):
                self.set_detect_params()
            c.extract_batch_exact_dev(self._xy.data_ptr(), self.seg_off.data_ptr(), self.seg_cnt.data_ptr(), self.labels.data_ptr(),
                                      self.n_clusters.data_ptr(), S, slots, eps, self.det[0], self.det[1], self.det[2],
                                      self.win_info.data_ptr(), self.cand_pair.data_ptr(), self.cand_xyr.data_ptr(),
                                      self.kept_labels.data_ptr(), self.rep.data_ptr(), st, fit_circle=self.det[3], knn_num=self.det[4])
        elif detect:
            if not hasattr(self, "det"):
                self.set_detect_params()
            c.extract_batch_dev(self._xy.data_ptr(), self.seg_off.data_ptr(), self.seg_cnt.data_ptr(),
                                self.labels.data_ptr(), self.n_clusters.data_ptr(), S, slots, self.det[0], self.det[1],
                                self.det[2], self.win_info.data_ptr(), self.cand_pair.data_ptr(),
                                self.cand_xyr.data_ptr(), self.kept_labels.data_ptr(), self.rep.data_ptr(), st,
                                fit_circle=self.det[3], knn_num=self.det[4])
        return self

    def order_grid(self, rows=9, cols=4):
        """cv::findCirclesGrid's job on the candidates of the last run(): self.grid_found[s], self.grid_order[s, :rows*cols]."""
        st = torch.cuda.current_stream(self.dev).cuda_stream
        order = self.grid_order[: self.S].view(-1)[: self.S * rows * cols]
        self.ctx.grid_order_dev(self.win_info.data_ptr(), self.seg_off.data_ptr(), self.cand_xyr.data_ptr(), self.S, rows,
                                cols, order.data_ptr(), self.grid_found.data_ptr(), st)
        return order.view(self.S, rows * cols), self.grid_found[: self.S]

    def gather_features(self, rows=9, cols=4):
        """Ordered circles [S, rows*cols, 3] of the last run() + order_grid() (NaN rows where no grid was found), on device."""
        import ctypes
        st = torch.cuda.current_stream(self.dev).cuda_stream
        M = rows * cols
        if getattr(self, "_feat_cap", 0) < self.S * M:
            self.feat = torch.empty(self.S * M * 3, dtype=torch.float64, device=self.dev)
            self._feat_cap = self.S * M
        L = self.ctx._L
        vp, u32 = ctypes.c_void_p, ctypes.c_uint32
        L.ecal_gather_features_dev.argtypes = [vp, vp, vp, vp, vp, vp, u32, u32, vp, vp]
        L.ecal_gather_features_dev.restype = ctypes.c_int
        self.ctx._check(L.ecal_gather_features_dev(self.ctx._h, self.win_info.data_ptr(), self.seg_off.data_ptr(),
                                                   self.cand_xyr.data_ptr(), self.grid_order.data_ptr(), self.grid_found.data_ptr(),
                                                   self.S, M, self.feat.data_ptr(), st))
        return self.feat[: self.S * M * 3].view(self.S, M, 3)

    def overflowed(self):
        return bool(self.flags[0].item())
