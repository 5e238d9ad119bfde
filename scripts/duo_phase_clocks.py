"""Phase clocks of the two-sided solve's wavefronts in the headline's grouped dispatch.  Needs the experiment build
  python -m mrs_uav_trajectory_generation_amd.build --variant stamps -DMRS_TG_DUO_STAMPS=1
and MRS_TG_LIB_PATH pointing at libmrs_tg_stamps.so; runs bench.py's headline in this process, then reads the stamps of the last
dispatches (mrs_tg_debug_duo_stamps) and prints the mean / median clocks between the stamps (s_memtime: the shader clock).
  MRS_TG_LIB_PATH=$PWD/mrs_uav_trajectory_generation_amd/libmrs_tg_stamps.so python scripts/duo_phase_clocks.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--gpus", "1", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-extras"] + sys.argv[1:]
import bench
try:
    bench.main()
except SystemExit:
    pass
from mrs_uav_trajectory_generation_amd import api
L = api.load_library()
buf = (C.c_ulonglong * (2048 * 16))()
rc = L.mrs_tg_debug_duo_stamps(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 16).astype(np.int64)
a = a[a[:, 0] != 0]
print("rc", rc, "wavefronts with stamps", len(a))
order = [0, 1, 13, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]
names = {1: "entry -> path index known (kernel arguments)", 13: "-> loads of the first trip issued", 2: "-> loads arrived, LDS written, end checks",
         3: "-> ballots, LDS fence", 4: "-> constants, longest side: before the forward loop", 5: "-> forward step 0", 6: "-> forward step 1",
         7: "-> forward steps 2..", 8: "-> join", 9: "-> first backward step", 10: "-> other backward steps", 11: "-> cost / status issued",
         12: "-> stores acknowledged"}
tot = a[:, 12] - a[:, 0]
print("wavefront life: mean %.0f  median %.0f  min %d  max %d clocks" % (tot.mean(), np.median(tot), tot.min(), tot.max()))
for k0, k1 in zip(order[:-1], order[1:]):
    d = a[:, k1] - a[:, k0]
    print("  %-55s mean %8.0f  median %8.0f  (%.1f %%)" % (names[k1], d.mean(), np.median(d), 100 * d.mean() / tot.mean()))
