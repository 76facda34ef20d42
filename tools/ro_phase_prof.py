"""Debug tool: shader-clock cycles between the barriers of the reference-order slicer (slice_hash_ref_kernel), thread 0 of
every workgroup, from a -DECAL_PHASE_PROF build made under /tmp.  Run on the GPU box:  python tools/ro_phase_prof.py [events]
Never used by tests or bench."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
src = os.path.join(ROOT, "eventcalib_amd", "csrc")
out = "/tmp/libecal_prof.so"
objs = []
for f in sorted(os.listdir(src)):
    if f.endswith(".hip"):
        o = "/tmp/prof_" + f[:-4] + ".o"
        if f == "ecal_events.hip" or not os.path.exists(os.path.join(src, "build", f[:-4] + ".o")):
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                                   "-DECAL_PHASE_PROF", "-c", os.path.join(src, f), "-o", o])
        else:
            o = os.path.join(src, "build", f[:-4] + ".o")
        objs.append(o)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-o", out] + objs + ["-L/opt/rocm/lib", "-lrccl"])
import torch
import eventcalib_amd.capi as capi
capi.lib_path = lambda: out
import eventcalib_amd
from eventcalib_amd.pipeline import DetectPipeline
import synth_stream as SS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
ctx = eventcalib_amd.Context(0); pipe = DetectPipeline(ctx)
ev = SS.make_stream(n, device="cuda")
t0, t1 = SS.tiled_windows(5.0, 5.0 + (n - 1) / 1e6)
pipe.set_windows(t0, t1)
L = capi.load_library()
buf = (ctypes.c_ulonglong * 16)()
pipe.run(ev, slice_only=True); torch.cuda.synchronize()
L.ecal_debug_ro_cycles(buf, 1)
pipe.run(ev, slice_only=True); torch.cuda.synchronize()
L.ecal_debug_ro_cycles(buf, 1)
v = list(buf); wg = len(t0)
names = {1: "decode + tables + lookups", 2: "key ballots", 3: "rank scan", 4: "bucket gather + words", 5: "words -> registers",
         6: "early epochs (13..127 buckets)", 7: "table clear", 8: "epoch 257", 9: "epoch 541", 10: "epoch 1109", 11: "kept bitmap",
         12: "bitmap prefix", 13: "final index", 14: "event -> index table", 15: "outputs"}
tot = sum(v[1:16])
for i in range(1, 16):
    print("%-32s %9.0f cycles/WG  %5.1f %%" % (names[i], v[i] / wg, 100.0 * v[i] / max(tot, 1)))
print("total %.0f cycles/WG (s_memtime ticks: 100 MHz constant clock on gfx9 => x ~24 for shader cycles at 2.4 GHz)" % (tot / wg))
