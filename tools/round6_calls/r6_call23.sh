cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c23; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round6.py tests/test_gpu_stated_sizes.py -x -q > $O/t.log 2>&1; echo "rc $?" >> $O/t.log
for i in 1 2; do timeout 300 python bench.py --config c5 --steps 5 --warmup 1 --no-secondary --no-cpu-baseline --parity-pairs 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 ms', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])"; done
MI_DEGENSAC_FAN=16 timeout 300 python bench.py --config c5 --steps 5 --warmup 1 --no-secondary --no-cpu-baseline --parity-pairs 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 fan16 ms', d['ms_per_step'])"
tail -3 $O/t.log
