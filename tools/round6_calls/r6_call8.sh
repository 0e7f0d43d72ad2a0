cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c8; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q > $O/t_round6.log 2>&1; echo "rc $?" >> $O/t_round6.log
tail -15 $O/t_round6.log
