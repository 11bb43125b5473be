export TMPDIR=/tmp
for v in "" "BZ_NO_FUSE_LEVEL_SUMS=1"; do
for f in "" "--float32"; do
for sc in "" "--single-calls"; do
echo "== $v $f $sc"
env $v python tools/bench_bomex.py --steps 30 $f $sc | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items() if 'subs' in k or 'project' in k or 'forcing' in k})"
done; done; done
