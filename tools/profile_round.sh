set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof_m $R/gpurun_out/pmc_m_FETCH_SIZE $R/gpurun_out/pmc_m_WRITE_SIZE
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_m -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cfg2 > $R/gpurun_out/prof_m/bench.json 2> $R/gpurun_out/prof_m/err.log
echo stats done
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_m_$C -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cfg2 --no-two-level > $R/gpurun_out/pmc_m_$C/bench.json 2> $R/gpurun_out/pmc_m_$C/err.log
  echo pmc $C done
done
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_m_FETCH_SIZE gpurun_out/pmc_m_WRITE_SIZE gpurun_out/pmc_traffic_m.json 214,214,214
find gpurun_out/pmc_m_FETCH_SIZE gpurun_out/pmc_m_WRITE_SIZE -name "*counter_collection.csv" -size +20M -delete
python3 bench.py > gpurun_out/bench_m.json 2> gpurun_out/bench_m.err
echo bench done
