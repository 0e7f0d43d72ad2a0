cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c41; rm -rf $O; mkdir -p $O
timeout 900 python tools/gpu_fan_small.py 40 0 0x1705 0x0f05 > $O/fan_small.log 2>&1; grep -v amdgpu $O/fan_small.log | cut -c1-220 | tail -50
timeout 600 python tools/gpu_ab5.py 520,640 base=0:0 > $O/ab_f.log 2>&1; grep -v amdgpu $O/ab_f.log | cut -c1-150 | tail
