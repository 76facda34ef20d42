"""CPU suite: known-answer tests for the ingest/slicing oracle (oracle/event_oracle.cpp)."""
import numpy as np

import oracle_lib as O


def test_window_bounds_inclusive_both_ends():
    t = np.array([0.0, 1.0, 1.0, 2.0, 3.0, 3.0, 4.0])
    rec = O.pack_events(t, np.arange(7.0), np.zeros(7), np.ones(7))
    assert O.window_bounds(rec, 1.0, 3.0) == (1, 6)        # lower_bound(1) .. upper_bound(3): EventFrame.cpp:14-15
    assert O.window_bounds(rec, 0.5, 0.9) == (1, 1)
    assert O.window_bounds(rec, -1.0, 10.0) == (0, 7)
    assert O.window_bounds(rec, 4.0, 4.0) == (6, 7)
    assert O.check_sorted(rec) == 0
    assert O.check_sorted(O.pack_events(t[::-1].copy(), np.zeros(7), np.zeros(7), np.ones(7))) == -5


def test_event_frame_dedupe_cancel_and_order():
    #            k: 0      1      2      3      4      5      6      7
    x = np.array([5.0,   7.0,   5.0,   9.0,   7.0,   1.0,   9.0,   -0.0])
    y = np.array([5.0,   7.0,   5.0,   9.0,   7.0,   1.0,   9.0,   0.0])
    p = np.array([1,     0,     1,     1,     1,     0,     1,     0])
    # pixel (5,5): + twice -> one positive point (first occurrence k=0)
    # pixel (7,7): - at k=1 and + at k=4 -> erased from both sets (EventFrame.cpp:24-32)
    # pixel (9,9): + twice -> positive, first occurrence k=3 ; (1,1) and (0,0): negative
    rec = O.pack_events(np.arange(8) * 1e-4, x, y, p)
    pos, neg, ep = O.event_frame(rec, 0, 8)
    assert pos.tolist() == [[5.0, 5.0], [9.0, 9.0]]
    assert neg.tolist() == [[1.0, 1.0], [0.0, 0.0]]
    assert ep.tolist() == [0, -1, 0, 1, -1, 0, 1, 1]
    # a sub-window sees only its own events
    pos, neg, ep = O.event_frame(rec, 1, 4)
    assert pos.tolist() == [[5.0, 5.0], [9.0, 9.0]] and neg.tolist() == [[7.0, 7.0]]
    assert ep.tolist() == [0, 0, 1]


def test_negative_zero_is_the_same_pixel():
    rec = O.pack_events([0.0, 1e-4], [0.0, -0.0], [-0.0, 0.0], [1, 0])
    pos, neg, ep = O.event_frame(rec, 0, 2)
    assert pos.shape[0] == 0 and neg.shape[0] == 0 and ep.tolist() == [-1, -1]


def test_nonzero_polarity_byte_is_positive():
    rec = O.pack_events([0.0, 1e-4], [3.0, 4.0], [3.0, 4.0], [2, 255])   # read into a bool, Event.hpp:45
    pos, neg, ep = O.event_frame(rec, 0, 2)
    assert pos.shape[0] == 2 and neg.shape[0] == 0


# ---- the reference's element order (EventFrame.cpp:12-13,34-35 + utility.hpp:38-51 on libstdc++) --------------------
def _random_window(rng, trial):
    n = int(rng.integers(1, 3000))
    W = int(rng.choice([8, 40, 346]))
    H = int(rng.choice([8, 30, 260]))
    x = rng.integers(0, W, n).astype(float)
    y = rng.integers(0, H, n).astype(float)
    if trial % 5 == 0:
        x = x * 0.5 - 3          # non-integer and negative coordinates
    if trial % 7 == 0:
        x[rng.random(n) < 0.05] = -0.0
    p = rng.integers(0, 2, n)
    return O.pack_events(np.arange(n) * 1e-6, x, y, p), n


def test_reference_order_is_a_permutation_of_the_canonical_sets():
    rng = np.random.default_rng(11)
    for trial in range(40):
        rec, n = _random_window(rng, trial)
        rp, rn, rep = O.event_frame(rec, 0, n, "reference")
        cp, cn, cep = O.event_frame(rec, 0, n, "canonical")
        assert sorted(map(tuple, rp)) == sorted(map(tuple, cp)) and sorted(map(tuple, rn)) == sorted(map(tuple, cn))
        assert np.array_equal(rep < 0, cep < 0)
        # event_point points at the event's own pixel
        t, xy, pol = rec.reshape(-1, 25)[:, :8], rec.reshape(-1, 25)[:, 8:24].copy().view(np.float64), rec.reshape(-1, 25)[:, 24]
        for k in range(0, n, 37):
            if rep[k] >= 0:
                assert np.array_equal((rp if pol[k] else rn)[rep[k]], xy[k])


def test_restated_list_rules_equal_the_real_unordered_set():
    """oracle_event_frame_model (the rules of eventcalib_amd/csrc/slice_order.hpp, which the HIP slicer follows: bucket
    counts 13, 29, 59, ...; insert at the front of the bucket's run or of the list; rehash walk) against the real
    std::unordered_set with the restated EigenMatrixHash."""
    rng = np.random.default_rng(1)
    for trial in range(120):
        rec, n = _random_window(rng, trial)
        a = O.event_frame(rec, 0, n, "reference")
        b = O.event_frame(rec, 0, n, "model")
        for u, v in zip(a, b):
            assert np.array_equal(u, v), (trial, n)


def test_known_answer_small_set():
    """Hand-checkable case: three positive pixels land in buckets h % 13; each new bucket goes to the FRONT of the list."""
    rec = O.pack_events(np.arange(3) * 1e-6, [1.0, 2.0, 3.0], [0.0, 0.0, 0.0], [1, 1, 1])
    pos, neg, ep = O.event_frame(rec, 0, 3, "reference")
    L = O.lib()
    b = [L.oracle_pixel_hash(float(v), 0.0) % 13 for v in (1, 2, 3)]
    assert len(set(b)) == 3                      # three different buckets -> pure reverse insertion order
    assert pos.tolist() == [[3.0, 0.0], [2.0, 0.0], [1.0, 0.0]] and ep.tolist() == [2, 1, 0]


def test_hash_restatement_and_bucket_steps_match_libstdcxx():
    import ctypes
    import os
    L = O.lib()
    O._declare_events(L)
    rng = np.random.default_rng(2)
    vals = [0.0, -0.0, 1.0, 345.0, 259.0, 0.5, -3.25, 1e300, 5e-324, np.inf] + list(rng.normal(size=50)) + list(rng.integers(0, 2048, 200).astype(float))
    for x in vals:
        for y in vals[:12]:
            assert L.oracle_pixel_hash(x, y) == L.oracle_pixel_hash_restated(x, y), (x, y)
    # growth of a real unordered_set == the policy object's steps
    bc = np.zeros(12000, np.uint64)
    L.oracle_bucket_counts(12000, O._p(bc, O._u64p))
    steps = O.bucket_steps(28)
    seen = sorted(set(int(v) for v in bc))
    assert seen == [int(v) for v in steps[:len(seen)]]
    for k in range(12000):                       # key number k+1 lives in the first step >= k+1
        assert bc[k] == steps[np.searchsorted(steps, k + 1)]
    # the product's tables (libecal.so exports them; loading needs no GPU)
    so = os.path.join(O.ROOT, "eventcalib_amd", "libecal.so")
    E = ctypes.CDLL(so)
    E.ecal_ref_bucket_step.argtypes = [ctypes.c_int]
    E.ecal_ref_bucket_step.restype = ctypes.c_uint64
    E.ecal_ref_pixel_hash.argtypes = [ctypes.c_double, ctypes.c_double]
    E.ecal_ref_pixel_hash.restype = ctypes.c_uint64
    assert [E.ecal_ref_bucket_step(e) for e in range(28)] == [int(v) for v in steps]
    assert E.ecal_ref_bucket_step(28) == 0
    for x in vals:
        for y in vals[:12]:
            assert E.ecal_ref_pixel_hash(x, y) == L.oracle_pixel_hash(x, y), (x, y)


def test_golden_eventframe_order_fixtures():
    """tests/golden/eventframe_order_*.npz (made by make_eventframe_golden.py on g++ 11.4's libstdc++ with the reference's
    kd-tree): the oracle on this box — real container and restated rules — reproduces points, event map and labels."""
    import glob
    import os
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eventframe_order_*.npz")))
    assert len(files) >= 4
    for f in files:
        g = np.load(f)
        rec, eps, minpts = g["records"], float(g["eps"]), int(g["minpts"])
        po, eo, seg = 0, 0, 0
        for lo, hi in g["bounds"]:
            for order in ("reference", "model"):
                pos, neg, ep = O.event_frame(rec, int(lo), int(hi), order)
                assert (pos.shape[0], neg.shape[0]) == tuple(g["seg_cnt"][seg:seg + 2]), f
                assert np.array_equal(np.concatenate([pos, neg]).view(np.uint64), g["xy"][po:po + len(pos) + len(neg)].view(np.uint64)), f
                assert np.array_equal(ep, g["event_point"][eo:eo + hi - lo]), f
            for pts in (pos, neg):
                if pts.shape[0]:
                    rc, lab, nc = O.dbscan(pts, eps, minpts)
                    assert np.array_equal(lab, g["labels"][po:po + pts.shape[0]]) and nc == g["n_clusters"][seg], f
                po += pts.shape[0]
                seg += 1
            eo += hi - lo
