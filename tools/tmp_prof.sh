set -e
cd $GRAFT_REPO_ROOT
MICI_AMD_HIPCC_FLAGS="-DMM_SOFTABS_PROF" python -m mici_amd.build --force > /tmp/build.log 2>&1 || { tail -20 /tmp/build.log; exit 1; }
python bench.py --config c3b --steps 1 --warmup 0 --chains-per-gpu 256 --no-cpu-baseline 2>&1 | grep "softabs prof"
