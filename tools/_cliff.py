import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import numpy as np
import small_shape_timing as T
for pol, lt in (("IQUV", 25), ("IQUV", 27), ("IQUV", 29)):
    try:
        T.run("%s l_trunc %d" % (pol, lt), pol, lt, 4096, 8, np.float64, reps=3)
    except Exception as e:
        print("l_trunc", lt, "failed:", e)
