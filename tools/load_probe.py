"""Debug: where ecal_stream_create_from_file spends its time (ECAL_TRACE=load) on a 1.25 GB file in /dev/shm."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["ECAL_TRACE"] = "load"
import numpy as np, torch
import eventcalib_amd, synth_stream as SS
n = 50_000_000
ev = SS.make_stream(n, device="cuda")
path = "/dev/shm/ecal_load_probe.bin"
ev.cpu().numpy().tofile(path)
del ev
ctx = eventcalib_amd.Context(0)
L = ctx._L
L.ecal_stream_create_from_file.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_void_p)]
L.ecal_stream_destroy.argtypes = [ctypes.c_void_p]
for rep in range(3):
    h = ctypes.c_void_p()
    t = time.perf_counter()
    rc = L.ecal_stream_create_from_file(ctx._h, path.encode(), 0.0, 0, 0.0, ctypes.byref(h))
    print("rep", rep, "rc", rc, "%.4f s" % (time.perf_counter() - t), flush=True)
    L.ecal_stream_destroy(h)
os.remove(path)
