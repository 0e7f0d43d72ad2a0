cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c9; rm -rf $O; mkdir -p $O
C5_ORACLE=0 MI_DEGENSAC_FAN=0 timeout 120 python tools/gpu_c5.py > $O/c5_nofan.log 2>&1; echo "rc $?" >> $O/c5_nofan.log
C5_ORACLE=0 timeout 120 python tools/gpu_c5.py > $O/c5_fan.log 2>&1; echo "rc $?" >> $O/c5_fan.log
C5_ORACLE=0 timeout 120 python tools/gpu_c5.py 20000 50000 > $O/c5_fan_small.log 2>&1; echo "rc $?" >> $O/c5_fan_small.log
timeout 600 python -m pytest tests/test_gpu_stated_sizes.py -x -q -k c5 > $O/t_c5.log 2>&1; echo "rc $?" >> $O/t_c5.log
grep -v amdgpu $O/c5_nofan.log; grep -v amdgpu $O/c5_fan.log; grep -v amdgpu $O/c5_fan_small.log; tail -5 $O/t_c5.log
