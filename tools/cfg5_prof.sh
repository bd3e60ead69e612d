#!/bin/bash
# kernel trace of one cfg-5-share step per coarse space (tools/cfg5_rotations.py): usage tools/cfg5_prof.sh KIND [CELLS]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=${1:-rgdsw}
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cfg5_$K -o t -- python3 $R/tools/cfg5_rotations.py 94 $K ${2:-0} > $R/gpurun_out/cfg5_prof_$K.log 2>&1
f=$(find $R/gpurun_out/prof_cfg5_$K -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp $f $R/gpurun_out/cfg5_${K}_kernel_stats.csv; else
  d=$(find $R/gpurun_out/prof_cfg5_$K -name '*.db' | head -1); python3 $R/tools/rocpd_stats.py $d > $R/gpurun_out/cfg5_${K}_kernel_stats.csv; fi
rm -rf $R/gpurun_out/prof_cfg5_$K
