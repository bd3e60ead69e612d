#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/first_step.py 214 > $R/gpurun_out/first_step.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_first -o first -- python3 $R/tools/first_step.py 214 > $R/gpurun_out/first_step_prof.log 2>&1
