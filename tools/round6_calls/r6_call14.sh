cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c14; rm -rf $O; mkdir -p $O
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_round6.py -x -q > $O/t6_$i.log 2>&1; echo "rc $?" >> $O/t6_$i.log; tail -3 $O/t6_$i.log; done
