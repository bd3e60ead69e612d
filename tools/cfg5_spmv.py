"""cfg 5's share: the SpMV entry of bench.py alone (solver kernel + the general kernels on the same matrix)."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from feddlib_amd import capi
out = bench.extra_cfg5_share(capi, 0, None, only=("q1",))
print(json.dumps({k: out["q1"][k] for k in ("ms_per_step", "spmv", "spmv_general_kernel_back_to_back")}, indent=1))
