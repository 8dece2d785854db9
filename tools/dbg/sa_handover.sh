# how often the SoftAbs refinement hands over to the Jacobi sweeps on c3b_dense: warm (first pass above the start threshold,
# or a split cluster) against restarts from the identity (passes that stopped contracting)
for m in 1 2 3; do MICI_AMD_RTC_FLAGS="-DMM_SA_DBG_COUNT=$m" python bench.py --config c3b_dense --no-extra-configs --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); e=d['roofline']['executed']; s=e['jacobi_sweeps_per_chain_step']
print('mode $m (1 warm hand-overs, 2 restarts, 3 split clusters among the warm ones): per step %.3f (raw %.1f) refined %.1f' % ((s - 2.65) / 1e3, s, e['refined_decompositions_per_chain_step']))"; done
