"""P2 assembly on a structured cube (bench.py's p2_cube entry alone).  usage: p2_cube.py [cells]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from feddlib_amd import capi  # noqa: E402

print(json.dumps(bench.extra_p2_cube(capi, 0, int(sys.argv[1]) if len(sys.argv) > 1 else 64), indent=1))
