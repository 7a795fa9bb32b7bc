"""Training step (tools/bench_train.run, graphed and eager) under debug-library switches, same process and box:
    XMLHIP_LIB=$PWD/tvretrieval_amd/csrc/libxmlhip_dbg.so python tools/ab_train_step.py 300 0 300 0
(300: the 128 x 128 weight-gradient kernel everywhere, 0: the product dispatch)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_train  # noqa: E402
from tvretrieval_amd import _lib  # noqa: E402

lib = _lib.load()
for abl in [int(a) for a in sys.argv[1:]] or [300, 0]:
    lib.xml_debug_set_q2c_ablation(ctypes.c_int(abl))
    for graph in (True, False):
        r = bench_train.run(steps=20, warmup=5, graph=graph)
        print("abl %d graph %d : %.3f ms/step" % (abl, graph, r["ms_per_step"]), flush=True)
