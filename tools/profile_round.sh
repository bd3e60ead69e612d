# Round profile: kernel statistics of the bench command, PMC traffic passes (FETCH_SIZE / WRITE_SIZE in their own runs),
# then the plain bench line.  usage (on the GPU box): bash tools/profile_round.sh <tag> <git sha of the tree>   e.g. r03_a 4c2c77d
set -e
TAG=${1:-r03_a}
SHA=${2:-unknown}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
P=$R/gpurun_out/prof_$TAG
mkdir -p $P $R/gpurun_out/pmc_${TAG}_FETCH_SIZE $R/gpurun_out/pmc_${TAG}_WRITE_SIZE
# headline steps only: the per-kernel averages of the statistics are then averages over the launches bench.py's own HIP
# events average over (the two-level steps, with their shorter Krylov bases, get their own pass)
rocprofv3 --kernel-trace --stats --output-format csv -d $P -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cfg2 --no-two-level --no-extras > $P/bench.json 2> $P/err.log
echo stats done
mkdir -p ${P}_two
rocprofv3 --kernel-trace --stats --output-format csv -d ${P}_two -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cfg2 --no-extras > ${P}_two/bench.json 2> ${P}_two/err.log
find ${P}_two -name "*kernel_trace.csv" -size +20M -delete
echo stats with the two-level steps done
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$C -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cfg2 --no-two-level --no-extras > $R/gpurun_out/pmc_${TAG}_$C/bench.json 2> $R/gpurun_out/pmc_${TAG}_$C/err.log
  echo pmc $C done
done
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE gpurun_out/pmc_traffic_$TAG.json 214,214,214 $TAG $SHA
find gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE -name "*counter_collection.csv" -size +20M -delete
find $P -name "*kernel_trace.csv" -size +20M -delete
python3 bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo bench done
