"""hipMalloc of large blocks from a Python process, with and without torch imported first (torch brings the HIP runtime of its
wheel; the first libamdhip64 loaded serves every later caller in the process).  Usage: python scripts/micro/malloc_py.py [torch] [lib]"""
import ctypes
import os
import sys
import time

if "torch" in sys.argv:
    import torch
    torch.cuda.init()
    x = torch.zeros(1, device="cuda")
    print("torch", torch.__version__, torch.version.hip)
if "lib" in sys.argv:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from wfmash_amd import capi
    h = capi.Handle(0)
hip = ctypes.CDLL("libamdhip64.so")
for line in open("/proc/self/maps"):
    if "libamdhip64" in line and "r-xp" in line:
        print("runtime:", line.split()[-1])
p = ctypes.c_void_p()
for rep in range(2):
    for gb in (1, 4, 8, 16):
        t = time.time()
        r = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(gb << 30))
        t1 = time.time()
        hip.hipMemset(p, 0, ctypes.c_size_t(gb << 30))
        hip.hipDeviceSynchronize()
        t2 = time.time()
        hip.hipFree(p)
        print(f"hipMalloc {gb:2d} GB: rc {r}, {(t1 - t) * 1e3:7.1f} ms; memset {(t2 - t1) * 1e3:6.1f} ms; free {(time.time() - t2) * 1e3:6.1f} ms")
