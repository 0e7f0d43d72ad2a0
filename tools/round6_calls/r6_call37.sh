cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c37; rm -rf $O; mkdir -p $O
timeout 600 python tools/gpu_ab_h.py 0 6 > $O/ab_h.log 2>&1; grep -v amdgpu $O/ab_h.log | tail -30
timeout 900 python -m pytest tests/test_gpu_hjob.py -x -q > $O/t_hjob.log 2>&1; tail -3 $O/t_hjob.log
