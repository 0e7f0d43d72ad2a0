cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c26; rm -rf $O; mkdir -p $O
timeout 600 python tools/gpu_ab5.py 512,1024 base=0:0 t256=0:2 t256s=8:2 2>&1 | grep -v amdgpu > $O/ab.log; cat $O/ab.log
