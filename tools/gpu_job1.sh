#!/bin/bash
# round-2 GPU job 1: baseline of the round-1 kernels (tests, bench, launcher error path, SQ counters of the shipped tendency kernels)
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_job1; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log ) 
timeout 600 python bench.py > $O/bench_base.json 2> $O/bench_base.err
timeout 300 python bench.py --gpus 2 --size 64 --steps 1 --warmup 0 --launch-timeout 200 --collective-timeout 60 > $O/bench_2gpu_on_1gpu_box.json 2> $O/bench_2gpu.err
timeout 300 python bench.py --slab --no-cpu-baseline > $O/bench_slab1.json 2> $O/bench_slab1.err
rocprofv3 -L > $O/counters_full.txt 2>&1
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-compressible"
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $O/pmc_sq1 -- $B > $O/pmc_sq1.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM -d $O/pmc_sq2 -- $B > $O/pmc_sq2.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/pmc_tcc -- $B > $O/pmc_tcc.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE -d $O/pmc_tcp -- $B > $O/pmc_tcp.log 2>&1
python tools/pmc_summary.py $O/pmc_summary.json $O/pmc_sq1 $O/pmc_sq2 $O/pmc_tcc $O/pmc_tcp > $O/pmc_summary.log 2>&1
# keep the merge small: raw csvs are large
find $O -name "*counter_collection.csv" -size +20M -delete
tail -3 $O/pytest_gpu.log; head -c 600 $O/bench_base.json; echo; cat $O/bench_2gpu_on_1gpu_box.json
