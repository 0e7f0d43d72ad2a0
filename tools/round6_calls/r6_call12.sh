cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c12; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_units.py tests/test_gpu_stated_sizes.py tests/test_gpu_golden.py -x -q > $O/t.log 2>&1; echo "rc $?" >> $O/t.log
for k in 0 64; do MI_DEGENSAC_FAN=$k timeout 120 python tools/gpu_phases.py 1 50000 0.1 200000 > $O/ph_fan$k.log 2>&1; echo "== fan $k"; grep -v amdgpu $O/ph_fan$k.log | head -2 | cut -c1-420; done
timeout 600 python tools/gpu_ab5.py 4096,512 base=0:0 > $O/ab.log 2>&1; grep -v amdgpu $O/ab.log
tail -4 $O/t.log
