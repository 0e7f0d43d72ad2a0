cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c4; rm -rf $O; mkdir -p $O
timeout 600 python tools/gpu_phases.py 512 > $O/phases_512.log 2>&1
grep -v amdgpu $O/phases_512.log | cut -c1-900
