cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c39; rm -rf $O; mkdir -p $O
AB_H_PAIRS=256,257,300,384,512,768,1024 timeout 900 python tools/gpu_ab_h.py 0 > $O/ab_h.log 2>&1; grep -v amdgpu $O/ab_h.log | tail -40
timeout 1800 python -m pytest tests -m gpu -x -q > $O/t_gpu.log 2>&1; echo "gpu rc $?" >> $O/t_gpu.log; tail -3 $O/t_gpu.log
timeout 900 python tools/gpu_fuzz_h.py 400 77 > $O/fuzz_h.log 2>&1; tail -1 $O/fuzz_h.log
