cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c29; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q > $O/t6.log 2>&1; echo "rc $?" >> $O/t6.log; tail -5 $O/t6.log
MI_DEGENSAC_FAN=8 timeout 600 python -m pytest tests/test_gpu_round6.py tests/test_gpu_stated_sizes.py -x -q -k "fan or c5" > $O/t6_fan8.log 2>&1; echo "rc $?" >> $O/t6_fan8.log; tail -3 $O/t6_fan8.log
