cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c25; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/t_gpu.log 2>&1; echo "gpu rc $?" >> $O/t_gpu.log
timeout 900 python tools/gpu_fuzz.py large 80 812 > $O/fuzz_large.log 2>&1; tail -2 $O/fuzz_large.log
timeout 300 python tools/gpu_phases.py 1 50000 0.1 200000 > $O/ph_c5.log 2>&1; grep -v amdgpu $O/ph_c5.log | head -2 | cut -c1-420
tail -3 $O/t_gpu.log
