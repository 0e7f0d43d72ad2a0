cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c63; rm -rf $O; mkdir -p $O
for i in 1 2 3; do python bench.py --config c5 --steps 3 --warmup 1 --no-secondary --parity-pairs 1 --no-cpu-baseline 2>/dev/null | cut -c150-260; done
timeout 900 python tools/gpu_fuzz.py large 60 1201 > $O/fuzz_large.log 2>&1; tail -1 $O/fuzz_large.log
timeout 300 python tools/gpu_phases.py 1 50000 0.1 200000 2>&1 | grep -v amdgpu | grep "^mean" | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_golden.py tests/test_gpu_round6.py tests/test_gpu_stated_sizes.py -x -q 2>&1 | tail -2
