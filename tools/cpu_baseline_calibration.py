"""How close is the CPU "port" baseline to the genuine reference?  Times the oracle's DBSCAN::Run driver (oracle/dbscan_oracle.cpp)
on its restated k-d tree and on the REFERENCE's compiled kdtree.cpp (oracle/_ref/libkdtree_ref.so) over the same synthetic
polarity slices (36 half-arcs + noise, ~1300 points: SURVEY Appendix E's probe of the genuine DBSCAN::Run measured 1538 us per
call at 1292 points in this container).  Run where /root/reference exists (this container); numbers go to BASELINE.md / DESIGN.md."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
import synth

rng = np.random.default_rng(1)
xy, off = synth.arc_slices(rng, n_slices=400, noise_frac=0.1)
cnt = np.diff(off)
print("slices %d, points per slice %.0f (min %d, max %d)" % (len(cnt), cnt.mean(), cnt.min(), cnt.max()))
for ref in (False, True):
    if ref and not O.have_ref_kdtree():
        print("oracle/_ref absent: reference kd-tree not timed")
        continue
    O.set_kd_backend(ref)
    for rep in range(2):
        t = time.perf_counter()
        labels, ncl = O.dbscan_batch(xy, off[:-1], cnt.astype(np.uint32), 4.0, 2)
        el = time.perf_counter() - t
    print("%-44s %.0f us per Run() call, %.3f Mpoints/s, %d clusters" % (O.kd_backend(), el / len(cnt) * 1e6, cnt.sum() / el / 1e6, int(ncl.sum())))
O.set_kd_backend(False)
