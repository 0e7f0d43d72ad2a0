cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c13; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q > $O/t6.log 2>&1; echo "rc $?" >> $O/t6.log
timeout 1800 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_round6.py > $O/t_gpu.log 2>&1; echo "gpu rc $?" >> $O/t_gpu.log
timeout 600 python bench.py --config c5 --steps 3 --warmup 1 --no-secondary --parity-pairs 1 > $O/bench_c5.json 2> $O/bench_c5.err; echo "rc $?" >> $O/bench_c5.err
tail -6 $O/t6.log; tail -4 $O/t_gpu.log; cut -c1-300 $O/bench_c5.json; tail -2 $O/bench_c5.err
