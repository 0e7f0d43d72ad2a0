cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c34; rm -rf $O; mkdir -p $O
timeout 1500 python tools/gpu_soak_mem.py 120 > $O/soak.log 2>&1; grep -v amdgpu $O/soak.log | tail -3
