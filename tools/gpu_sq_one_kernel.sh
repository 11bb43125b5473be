#!/bin/bash
# SQ / TA / TCP counters of the dispatches of ONE kernel (name prefix $1) in a command ($2 ...): per-dispatch rows, medians of the heavy ones
set -u
export TMPDIR=/tmp
K=$1; shift
O=gpurun_out/sq_one; rm -rf $O; mkdir -p $O
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $O/$name -- $CMD > $O/$name.log 2>&1; }
CMD="$*"
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run b SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_BRANCH
run c SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_FLAT
run d TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum
run e GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python - $O "$K" <<'PY'
import csv,glob,sys,os,statistics
O,K=sys.argv[1],sys.argv[2]
vals={}
for f in glob.glob(os.path.join(O,"**","*counter_collection.csv"),recursive=True):
    for row in csv.DictReader(open(f,newline="")):
        n=row["Kernel_Name"]
        n=n[5:] if n.startswith("void ") else n
        if not n.startswith(K): continue
        vals.setdefault(row["Counter_Name"],[]).append(float(row["Counter_Value"]))
for c,v in sorted(vals.items()):
    v=sorted(v); top=v[len(v)//2:]      # the heavier half of the dispatches
    print(f"{c:34s} n={len(v):3d} median_of_heavy_half={statistics.median(top):.4g} max={v[-1]:.4g} min={v[0]:.4g}")
PY
find $O -name "*.csv" -size +1M -delete
