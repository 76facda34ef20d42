"""GPU: bounded, fixed-seed sweeps of the three fuzzers (tests/fuzz_*.py — the scripts that found the ordering races of
earlier rounds), collected so that every driver run repeats them:
  * fuzz_pipeline: random streams at 0.5 - 3.5 Mev/s, 5 - 30 % noise through slicing, DBSCAN and the exact extraction,
    every window == the CPU oracle (EventFrame.cpp:10-36, dbscan.h:115-265, CirclesEventFrame.cpp:89-312);
  * fuzz_policy: the device policy's look-ahead == one window per piece and pass (eventCameraCalib.cpp:49-95);
  * fuzz_shared_map: both keyframe gates == oracle/policy_oracle.cpp (TrackingBase.cpp:16-46, EventCalibIni.cpp:23-97).
Sized for about a minute in all on the GPU box; `python tests/fuzz_*.py N` runs the long form."""
import time

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import eventcalib_amd
    c = eventcalib_amd.Context(0)
    yield c
    c.close()


def test_fuzz_pipeline_bounded(ctx):
    import fuzz_pipeline as F
    tic = time.time()
    r = F.run(range(7000, 7003), ctx=ctx, verbose=False)
    print("\n[fuzz] pipeline: %d streams, %d windows == oracle (%d paired, %d with tied medians), %.1f s"
          % (r["streams"], r["windows"], r["exact"], r["tied"], time.time() - tic))
    assert r["streams"] == 36 and r["exact"] > 0


def test_fuzz_pipeline_fused_entry_bounded(ctx):
    import fuzz_pipeline as F
    r = F.run([7100], rates=(1.0e6, 2.7e6), ctx=ctx, verbose=False, fused=True)
    assert r["streams"] == 4


def test_fuzz_policy_lookahead_bounded(ctx):
    import fuzz_policy as F
    tic = time.time()
    r = F.run([7200], ctx=ctx, verbose=False)
    print("\n[fuzz] policy look-ahead: %d runs, %d keyframes, %.1f s" % (r["runs"], r["keyframes"], time.time() - tic))
    assert r["runs"] == 12 and r["keyframes"] > 0


def test_fuzz_shared_map_bounded(ctx):
    import fuzz_shared_map as F
    tic = time.time()
    r = F.run([7300], ctx=ctx, verbose=False)
    print("\n[fuzz] keyframe gates vs the policy oracle: %d runs, %d keyframes, %.1f s" % (r["runs"], r["keyframes"], time.time() - tic))
    assert r["runs"] == 16 and r["keyframes"] > 0
