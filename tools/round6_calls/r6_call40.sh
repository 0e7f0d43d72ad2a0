cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c40; rm -rf $O; mkdir -p $O
timeout 1500 python tools/gpu_ab5.py 300,384,512,640,767 base=0:0 t256=0:2 t128=0:3 t256s=8:2 > $O/ab_f.log 2>&1; grep -v amdgpu $O/ab_f.log | cut -c1-150 | tail -50
timeout 600 python -m pytest tests/test_gpu_round6.py -x -q > $O/t6.log 2>&1; tail -2 $O/t6.log
