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
