cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c50; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity_f.py tests/test_gpu_parity_h.py tests/test_gpu_variants.py tests/test_gpu_golden.py tests/test_gpu_round5.py -x -q > $O/t1.log 2>&1; tail -2 $O/t1.log
timeout 600 python tools/gpu_fuzz.py 600 951 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
timeout 600 python tools/gpu_fuzz_h.py 300 952 > $O/fuzz_h.log 2>&1; tail -1 $O/fuzz_h.log
timeout 300 python tools/gpu_phases.py 256 2>&1 | grep -v amdgpu | cut -c1-400 | head -8
