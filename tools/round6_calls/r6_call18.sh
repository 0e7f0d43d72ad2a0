cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c18; rm -rf $O; mkdir -p $O
timeout 300 python tools/gpu_phases.py 1024 > $O/phases_c2_1024.log 2>&1
grep -v amdgpu $O/phases_c2_1024.log | head -3 | cut -c1-500
