cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c49; rm -rf $O; mkdir -p $O
timeout 300 python tools/gpu_phases.py 256 2>&1 | grep -v amdgpu | cut -c1-700 > $O/ph256.log; cat $O/ph256.log
