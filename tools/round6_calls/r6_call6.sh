cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c6; rm -rf $O; mkdir -p $O
timeout 900 python tools/gpu_ab5.py 4096 nomix=32:0:-1:0:0 mix=0:0:-1:0:0 w384=0:0:-1:0:0:0:384:256 w320=0:0:-1:0:0:0:320:384 w384a=0:0:-1:0:0:0x1040a0:384:256 > $O/ab_mix.log 2>&1
grep -v amdgpu $O/ab_mix.log
