cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c7; rm -rf $O; mkdir -p $O
timeout 600 python tools/gpu_mixtl.py 4096 1 2 > $O/tl_1_2.log 2>&1
timeout 600 python tools/gpu_mixtl.py 4096 384 256 > $O/tl_384.log 2>&1
timeout 600 python tools/gpu_mixtl.py 4096 1 2 0 32 > $O/tl_nomix.log 2>&1
grep -v amdgpu $O/tl_1_2.log; grep -v amdgpu $O/tl_384.log; grep -v amdgpu $O/tl_nomix.log
