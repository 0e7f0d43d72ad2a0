cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c36; rm -rf $O; mkdir -p $O
timeout 600 python tools/gpu_ab_h.py 0 6 5 10 > $O/ab_h.log 2>&1; grep -v amdgpu $O/ab_h.log | tail -30
