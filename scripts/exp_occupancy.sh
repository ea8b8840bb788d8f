set -e
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --steps 3 --warmup 1 --parity-sample 200 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e6,3),'Mreads/s', round(d['ms_per_step'],1),'ms')"; }
BBMSA_CXXFLAGS="-DBBMSA_MIN_WAVES=2" python -m bbmap_amd.build --force >/dev/null 2>&1
run "waves2 G32R5"
BBMSA_LANES_PER_JOB=16 run "waves2 G16R10"
BBMSA_LANES_PER_JOB=64 run "waves2 G64R3"
BBMSA_CXXFLAGS="-DBBMSA_MIN_WAVES=3" python -m bbmap_amd.build --force >/dev/null 2>&1
run "waves3 G32R5"
BBMSA_LANES_PER_JOB=64 run "waves3 G64R3"
BBMSA_CXXFLAGS="-DBBMSA_MIN_WAVES=4" python -m bbmap_amd.build --force >/dev/null 2>&1
run "waves4 G32R5"
BBMSA_LANES_PER_JOB=64 run "waves4 G64R3"
