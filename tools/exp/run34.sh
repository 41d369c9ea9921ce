# SQ / LDS / VMEM counters of the K = 800 forward (gemm128g<true,false>) and of 4096^3, one counter group per pass (unknown names are skipped)
export TMPDIR=/tmp
root=$PWD
out=$PWD/gpurun_out/r06_run34_gemm_sq_counters.log
: > $out
for shape in "24000 800 2400 fwd" "4096 4096 4096 fwd" "24000 2400 800 dx"; do
for ctr in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_IFETCH" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  d=/tmp/pmc_$$_$RANDOM; mkdir -p $d
  (cd /tmp && timeout 120 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -o p -- python $root/tools/gemm_one.py $shape) > $d/run.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" "$shape" >> $out <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gemm1" in r["Kernel_Name"]]
agg = collections.defaultdict(float); disp = set()
for r in rows:
    agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
n = max(1, len(disp))
print("[%s] %d launches: " % (sys.argv[2], n) + ", ".join("%s %.4g" % (k, v / n) for k, v in sorted(agg.items())), flush=True)
PY
  else echo "[$shape] $ctr: not collected ($(grep -i -m1 -E 'error|invalid|not found|unknown' $d/run.log | cut -c1-160))" >> $out; fi
  rm -rf $d
done
done
