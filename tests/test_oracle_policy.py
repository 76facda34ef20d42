"""CPU suite: known-answer tests of the policy oracle (oracle/policy_oracle.cpp): MultiProcess::process
(eventCameraCalib.cpp:34-97,168-179) + TrackingBase::process + EventCalibIni::track (EventCalibIni.cpp:23-97)."""
import math

import numpy as np

import oracle_lib as O

STEP = 5e-4
ROWS, COLS = 9, 4


def grid(angle=0.0, shift=(100.0, 80.0), pitch=20.0):
    """ordered circles of an asymmetric 9 x 4 grid in the image, rotated by `angle`: [36, 3]"""
    pts = np.array([[(2 * j + i % 2) * pitch, i * pitch] for i in range(ROWS) for j in range(COLS)], float)
    c, s = math.cos(angle), math.sin(angle)
    xy = pts @ np.array([[c, s], [-s, c]]) + np.array(shift)
    return np.concatenate([xy, np.full((ROWS * COLS, 1), 6.0)], axis=1)


def test_gate_threshold_known_answer():
    # the pattern turns by `a` between two frames dt apart: every row's direction turns by a, the median angle is a;
    # accepted iff a / dt < (5e-4 pi) / step = pi rad/s for step = 5e-4 (EventCalibIni.cpp:81-83)
    for a, dt, want in ((0.010, 4e-3, True), (0.0130, 4e-3, False), (0.0120, 4e-3, True), (0.05, 4e-3, False),
                        (0.05, 0.02, True), (-0.0130, 4e-3, False)):
        assert O.track_gate(grid(0.3), 1.0, grid(0.3 + a), 1.0 + dt, ROWS, COLS, STEP) == want, (a, dt)
        assert O.track_gate(grid(0.3), 1.0 + dt, grid(0.3 + a), 1.0, ROWS, COLS, STEP) == want      # |time distance|
    # a translation or a change of scale does not turn the rows
    assert O.track_gate(grid(0.2), 1.0, grid(0.2 + 1e-6, shift=(150.0, 60.0), pitch=23.0), 1.004, ROWS, COLS, STEP)


def test_window_arithmetic_when_nothing_is_found():
    # extractFeatures() always fails, few events: the window grows by one step until it is LONGER than 3 len (3, 4, ... 10
    # steps), then slides by one step and starts again at 3 steps (eventCameraCalib.cpp:75-79); one piece of 40 steps.
    # With a step that is a power of two every sum is exact:
    step = 2.0 ** -11
    seen = []

    def detect(t0, t1):
        seen.append((t0, t1))
        return False, 100, None
    r = O.policy_run(detect, 2.0, 2.0 + 40 * step, 1, step, 4000)
    assert len(r["time"]) == 0 and r["windows"] == len(seen)
    lens = [round((b - a) / step) for a, b in seen]
    starts = [round((a - 2.0) / step) for a, b in seen]
    assert lens[:9] == [3, 4, 5, 6, 7, 8, 9, 10, 3] and starts[:9] == [0] * 8 + [1]
    assert all(b < 2.0 + 40 * step for a, b in seen)                     # :50, strict
    # many events: the window never grows, it slides (:67-69)
    seen.clear()
    r = O.policy_run(lambda t0, t1: (seen.append((t0, t1)) or (False, 5000, None)), 2.0, 2.0 + 40 * step, 1, step, 4000)
    assert [round((b - a) / step) for a, b in seen] == [3] * len(seen)
    assert [round((a - 2.0) / step) for a, b in seen] == list(range(len(seen))) and len(seen) == 37


def test_window_arithmetic_in_floating_point():
    # MotionTimeStep = 5e-4 (example.yaml:14) is not a binary fraction: whether a 9-step window counts as "longer than
    # 3 len" depends on the rounding of `duration.second += motionTimeStep` — the windows must be the ones these exact
    # double operations give (the same three statements in Python floats)
    seen = []
    t_start, n_steps = 5.0, 60
    r = O.policy_run(lambda t0, t1: (seen.append((t0, t1)) or (False, 100, None)), t_start, t_start + n_steps * STEP, 1, STEP, 4000)
    ln = 3 * STEP
    step_piece = (t_start + n_steps * STEP - t_start) / 1
    lo, hi = (t_start + n_steps * STEP) - step_piece * 1, (t_start + n_steps * STEP) - step_piece * 0
    first, second, want = lo, lo + ln, []
    while second < hi:
        want.append((first, second))
        if (second - first) > 3 * ln:
            first += STEP
            second = first + ln
        else:
            second += STEP
    assert seen == want and r["windows"] == len(want)
    assert {round((b - a) / STEP) for a, b in seen} >= {3, 4, 5, 6, 7, 8, 9}


def test_accepted_windows_jump_by_the_frame_gap_and_pieces_are_taken_from_the_back():
    order = []

    def detect(t0, t1):
        order.append(t0)
        return True, 1500, grid(0.1 + 0.2 * (t0 - 2.0))                  # slow rotation: 0.2 rad/s < pi rad/s
    r = O.policy_run(detect, 2.0, 2.0 + 64 * STEP, 2, STEP, 4000)
    # pieces: [2.016, 2.032) is piece 0, [2.0, 2.016) piece 1; the worker pops the back -> the earlier piece first (:40-41)
    assert order[0] == 2.0 and order[4] == 2.016
    # every window succeeds: first + 3 steps, then a gap of 5 -> one keyframe per 8 steps: starts 0, 8, 16, 24 per piece
    # (the window starting at step 24 ends at 27 < 32)
    assert [round((a - 2.0) / STEP) for a in order] == [0, 8, 16, 24, 32, 40, 48, 56]
    assert np.allclose(r["time"], 2.0 + (np.array([0, 8, 16, 24, 32, 40, 48, 56]) + 1.5) * STEP)
    assert np.array_equal(r["events_num"], [1500] * 8) and r["features"].shape == (8, 36, 3)
    assert np.array_equal(r["duration"][:, 1] - r["duration"][:, 0] > 0, [True] * 8)


def test_own_piece_gate_and_single_worker_map_differ_as_documented():
    # the pattern turns fast (4 rad/s > pi rad/s): against the PREVIOUS keyframe (8 steps = 4 ms earlier) a window fails the
    # gate; own-piece mode therefore keeps only each piece's first frame ... (rejected windows with few events grow, so the
    # count of evaluations differs too)
    def detect(t0, t1):
        return True, 1500, grid(4.0 * ((t0 + t1) / 2 - 2.0))
    own = O.policy_run(detect, 2.0, 2.0 + 128 * STEP, 4, STEP, 4000, mode=0)
    assert len(own["time"]) == 4                                         # one per piece: TrackingBase's initialisation rule
    # ... while the reference with one worker has ONE map: only the very first frame (of the piece taken first, the earliest
    # in time) is accepted unconditionally; later pieces are checked against keyframes().lower_bound() = the next keyframe
    # in time if one exists, else the last one (EventCalibIni.cpp:26-36)
    one = O.policy_run(detect, 2.0, 2.0 + 128 * STEP, 4, STEP, 4000, mode=1)
    assert 1 <= len(one["time"]) < 4 and one["time"][0] == own["time"][0]
    # slow motion: both accept everything
    slow = lambda t0, t1: (True, 1500, grid(0.2 * ((t0 + t1) / 2 - 2.0)))
    a, b = O.policy_run(slow, 2.0, 2.0 + 128 * STEP, 4, STEP, 4000, mode=0), O.policy_run(slow, 2.0, 2.0 + 128 * STEP, 4, STEP, 4000, mode=1)
    assert np.array_equal(a["time"], b["time"]) and len(a["time"]) == 16


def _both_nth(a, nth):
    """(libstdc++'s std::nth_element, the product's restatement of it) on copies of a."""
    import ctypes
    import eventcalib_amd
    L, P = O.lib(), eventcalib_amd.load_library()
    for f in (L.oracle_nth_element_f64, P.ecal_ref_nth_element_f64):
        f.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
        f.restype = None
    x, y = np.array(a, np.float64), np.array(a, np.float64)
    L.oracle_nth_element_f64(x.ctypes.data, len(x), nth)
    P.ecal_ref_nth_element_f64(y.ctypes.data, len(y), nth)
    return x, y


def test_restated_nth_element_is_the_library_s_move_for_move():
    """ref_nth_element.hpp (the gate's median on the device, the clusters' median representative) against the real
    std::nth_element: the WHOLE array afterwards, bit for bit — ties, NaNs (every comparison false: EventCalibIni.cpp:75-78
    takes acos of a cosine that may round above 1), and inputs that exhaust introselect's depth limit (heap-select branch)."""
    rng = np.random.default_rng(3)
    for trial in range(3000):
        n = int(rng.integers(1, 70))
        a = rng.integers(0, max(2, n // 3), n).astype(np.float64) if trial % 2 else rng.normal(size=n)
        if trial % 3 == 0:
            a[rng.random(n) < 0.25] = np.nan
        nth = int(rng.integers(0, n))
        x, y = _both_nth(a, nth)
        assert np.array_equal(x, y, equal_nan=True), (trial, a, nth)
    # inputs built by McIlroy's adversary against the library itself: the depth limit 2 lg n runs out, the heap-select branch
    # decides (oracle_nth_killer; the compare count of a plain introselect round is ~n, these take several times that)
    import ctypes
    L = O.lib()
    L.oracle_nth_killer.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    L.oracle_nth_killer.restype = None
    for n in (9, 32, 64, 200, 1000, 4096):
        for nth in (n // 2, n - 2, 1):
            a = np.empty(n)
            L.oracle_nth_killer(n, nth, a.ctypes.data)
            assert np.array_equal(np.sort(a), np.arange(n))          # a permutation
            x, y = _both_nth(a, nth)
            assert np.array_equal(x, y), (n, nth)
            assert x[nth] == nth
            b = a.copy()
            b[::5] = np.nan
            x, y = _both_nth(b, nth)
            assert np.array_equal(x, y, equal_nan=True), (n, nth)


def test_gate_with_a_nan_angle_is_the_library_s_nth_element():
    """Two frames whose rows are parallel to rounding: some cosines round above 1, acos gives NaN (no clamp in the reference,
    EventCalibIni.cpp:75).  The verdict is whatever std::nth_element leaves at rows / 2 — the oracle runs the real one."""
    a = grid(0.3)
    seen_nan = accepted = 0
    for k in range(400):
        b = grid(0.3 + 1e-9 * k, shift=(100.0 + 0.37 * k, 80.0), pitch=20.0 + 0.01 * k)
        ok = O.track_gate(a, 1.0, b, 1.004, ROWS, COLS, STEP)
        accepted += ok
    assert accepted >= 300     # nearly identical orientation: accepted unless NaNs land on the median position
