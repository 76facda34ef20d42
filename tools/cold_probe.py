"""Debug tool (GPU box): the C++ driver as a cold process, several times in a row on the same stream file, stage seconds of every
run — what the FIRST process on a box pays that later ones do not.  python tools/cold_probe.py [events] [runs] [env=VALUE ...]"""
import os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, synth_stream as SS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
extra = dict(a.split("=", 1) for a in sys.argv[3:])
tmp = tempfile.mkdtemp(prefix="ecal_cold_", dir="/dev/shm")
try:
    ev = SS.make_stream(n, rate=1e6, t_start=5.0, seed=12345, device="cuda")
    ev.cpu().numpy().tofile(os.path.join(tmp, "events.bin"))
    del ev
    open(os.path.join(tmp, "settings.yaml"), "w").write(bench.CHAIN_YAML % dict(start=5.0, pieces=1270))
    exe = os.path.join(ROOT, "eventcalib_amd", "unit_test_eventCameraCalib")
    for r in range(runs):
        t0 = time.perf_counter()
        out = subprocess.run([exe, os.path.join(tmp, "settings.yaml"), os.path.join(tmp, "events.bin"), tmp, "batch"], capture_output=True, text=True,
                             env=dict(os.environ, **extra))
        wall = time.perf_counter() - t0
        st = " ".join("%s %.3f" % (l.split()[1], float(l.split()[2])) for l in out.stdout.splitlines() if l.startswith("stage "))
        print("run %d: wall %.3f | %s" % (r, wall, st), flush=True)
        if out.returncode:
            print(out.stdout[-300:], out.stderr[-300:])
        err = [l for l in out.stderr.splitlines() if "ecal" in l or "finish_stream" in l or "EventContainer" in l]
        for l in err[:24]:
            print("   ", l)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
