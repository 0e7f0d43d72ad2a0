cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c24; rm -rf $O; mkdir -p $O
for k in 24 32 48 64 96 128; do MI_DEGENSAC_FAN=$k timeout 300 python bench.py --config c5 --steps 5 --warmup 1 --no-secondary --no-cpu-baseline --parity-pairs 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 fan$k ms', d['ms_per_step'])"; done
