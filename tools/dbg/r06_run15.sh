cd $GRAFT_REPO_ROOT
for n in default sa_rs2.0 sa_rs4.0 sa_rs1e9; do
  lib=mici_amd/lib/ab_$n.so; [ $n = default ] && lib=mici_amd/lib/libmici_amd.so
  for c in c3b_d128 c3b_d256; do
    MICI_AMD_LIB=$lib timeout 600 python bench.py --config $c --no-extra-configs --no-cpu-baseline --steps 2 --warmup 1 2>&1 | tail -1 | python -c "
import sys,json
l=sys.stdin.read().strip()
try:
  d=json.loads(l); e=d['roofline'].get('executed',{}); print('$n $c', d['value'], d['ms_per_step'], 'sweeps/step', e.get('jacobi_sweeps_per_chain_step'), 'products', e.get('mfma_products_per_chain_step'))
except Exception as ex: print('$n $c ERR', l[-300:])
"
  done
done
