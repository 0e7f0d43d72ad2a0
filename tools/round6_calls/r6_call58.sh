cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c58; rm -rf $O; mkdir -p $O
MI_DEGENSAC_TUNING=2 timeout 300 python tools/gpu_lo.py 1024 2>&1 | grep -v amdgpu > $O/lo.log; cat $O/lo.log
