import sys, os
sys.path.insert(0, os.getcwd())
import bench
print(bench.golden_parity(0)["rms_dANI"])
