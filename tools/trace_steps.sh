#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_steps -o steps -- python3 $R/tools/steps214.py 214 3 > $R/gpurun_out/trace_steps.log 2>&1
