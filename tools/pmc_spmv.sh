# SQ / cache counters of the solver's SpMV kernel (development aid): bash tools/pmc_spmv.sh [cells]
R=$GRAFT_REPO_ROOT
M=${1:-214}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_spmv
rm -rf $OUT && mkdir -p $OUT
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/ab_spmv.py $M spmv_classes=1 > $OUT/p$i.log 2>&1 || echo "set $i failed"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_spmv/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_spmv" in k:
            out[k.replace("(anonymous namespace)::", "").split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/pmc_spmv/summary.txt", "w") as fh:
    for k, d in out.items():
        fh.write(k + "\n")
        for name, v in sorted(d.items()):
            fh.write("   %-32s %14.1f  (n=%d)\n" % (name, sum(v) / len(v), len(v)))
print(open("gpurun_out/pmc_spmv/summary.txt").read())
PY
