cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c5; rm -rf $O; mkdir -p $O
timeout 600 python tools/gpu_mix.py 4096 0.3333 > $O/mix.log 2>&1
grep -v amdgpu $O/mix.log
