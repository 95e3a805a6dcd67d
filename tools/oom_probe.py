#!/usr/bin/env python3
"""tools/oom_probe.py <GB held> [world] -- a proof split over `world` rank-threads (L5-mult) with <GB held> taken from the device by
this process first: running out of memory must end in VPIN_ENOMEM / VPIN_ECOMM on every rank, never in a fault (round 6: the full
GPU suite once died in test_threads_l5_mult_by_two_ranks when an earlier test's children were still giving their memory back).
VPIN_SEGV_TRACE=1 prints the native call stack of a fault."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
hold, world = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 2
hip = C.CDLL("libamdhip64.so")
hip.hipSetDevice(0)
held = []
for _ in range(hold // 4):
    p = C.c_void_p()
    rc = hip.hipMalloc(C.byref(p), C.c_size_t(4 << 30))
    assert rc == 0, rc
    held.append(p)
import test_gpu_dist as T  # noqa: E402
try:
    single, out, st = T._prove_threads(world, "L5", "mult", None)
    T._same(single, out)
    print(f"held {hold} GB: proof by {world} ranks fine")
except BaseException as e:  # noqa: BLE001
    print(f"held {hold} GB: clean failure: {e!r}"[:600])
