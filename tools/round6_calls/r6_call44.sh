cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c44; rm -rf $O; mkdir -p $O
timeout 600 python tools/gpu_eig2.py > $O/eig2.log 2>&1; grep -v amdgpu $O/eig2.log | tail
