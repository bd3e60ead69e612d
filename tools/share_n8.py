"""The per-GPU share of the 8-GPU run on one GPU (bench.py's per_gpu_share_n8 entry) for a list of option sets.
usage: share_n8.py "gmres_s=8,gmres_spec=0" "gmres_s=8" ...   (development aid)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from feddlib_amd import capi  # noqa: E402


class A:
    target, restart = 64, 100


for opts in sys.argv[1:] or [""]:
    os.environ["FEDD_OPTIONS"] = opts
    r = bench.extra_per_gpu_share(capi, 0, A, 145)
    print(opts or "(defaults)", json.dumps({k: r[k] for k in ("ms_per_step", "device_ms_per_step", "wall_minus_device_ms",
                                                               "device_us_per_iteration", "gmres")}), flush=True)
    print("   ", r["phases_device_ms_per_step"], flush=True)
