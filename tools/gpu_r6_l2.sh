#!/bin/bash
# Round 6: does k_ac_forward2 lose its neighbour lines from the L2?  Variants: barrier per level, non-temporal loads / stores of the unshared words.
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_l2; mkdir -p $O
L=$PWD/breeze.jl_amd/lib
line() {
python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d.get('kernels_ms_per_step',{})
print('$1', round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items() if 'forward' in a or 'backward' in a})"
}
for rep in 1 2; do
for v in "X=0" "BREEZE_HIP_LIB=$L/libbreeze_hip_var_bar.so" "BREEZE_HIP_LIB=$L/libbreeze_hip_var_nt1.so" "BREEZE_HIP_LIB=$L/libbreeze_hip_var_nt3.so" "BREEZE_HIP_LIB=$L/libbreeze_hip_var_barnt3.so" "BREEZE_HIP_LIB=$L/libbreeze_hip_var_barnt3.so BZ_AC_MW=2" "BZ_AC_FWD2=0"; do
env $v timeout 300 python tools/bench_compressible.py --steps 4 --warmup 2 2>$O/err.log | tail -1 | line "[${v##*/}]" || tail -5 $O/err.log
done; done
run() { name=$1; shift; env "$@" timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/$name -- python tools/bench_compressible.py --steps 1 > $O/$name.log 2>&1; }
run f_def X=0
run f_bar BREEZE_HIP_LIB=$L/libbreeze_hip_var_bar.so
run f_nt3 BREEZE_HIP_LIB=$L/libbreeze_hip_var_nt3.so
run f_barnt3 BREEZE_HIP_LIB=$L/libbreeze_hip_var_barnt3.so
run f_old BZ_AC_FWD2=0
for n in f_def f_bar f_nt3 f_barnt3 f_old; do
python tools/pmc_summary.py $O/$n.json $O/$n > /dev/null 2>&1
python - $O/$n.json $n <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if 'forward' in k and 'FETCH_SIZE' in v: print(sys.argv[2], k[:50], round(v['FETCH_SIZE']*2*1024/1e9,2), 'GB fetched')
PY
done
find $O -name "*.csv" -size +4M -delete; find $O -name "*.db" -delete
