#!/bin/bash
cd $GRAFT_REPO_ROOT
for e in 0 1 2 3; do
cp mici_amd/lib/exp$e.so mici_amd/lib/libmici_amd.so
timeout 120 python bench.py --config c3b --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs 2>&1 | grep -h "softabs prof:" | head -1 | sed "s/^/exp$e: /"
done
