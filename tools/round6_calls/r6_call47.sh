cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c47; rm -rf $O; mkdir -p $O
timeout 900 python tools/gpu_ab5.py 512,384,256 base=0:0 ask=0x208:0 on=8:0 > $O/ab.log 2>&1; grep -v amdgpu $O/ab.log | cut -c1-200
