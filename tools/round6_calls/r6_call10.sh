cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c10; rm -rf $O; mkdir -p $O
for k in 0 32 64 128; do MI_DEGENSAC_FAN=$k timeout 120 python tools/gpu_phases.py 1 50000 0.1 200000 > $O/ph_fan$k.log 2>&1; echo "== fan $k"; grep -v amdgpu $O/ph_fan$k.log | head -2 | cut -c1-420; done
