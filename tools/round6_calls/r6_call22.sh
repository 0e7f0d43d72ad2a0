cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c22; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/t_gpu.log 2>&1; echo "gpu rc $?" >> $O/t_gpu.log
timeout 900 python tools/gpu_fuzz.py large 60 712 > $O/fuzz_large.log 2>&1; tail -2 $O/fuzz_large.log
timeout 600 python bench.py --config c5 --steps 3 --warmup 1 --no-secondary --parity-pairs 1 > $O/bench_c5.json 2> $O/bench_c5.err; echo "rc $?" >> $O/bench_c5.err
tail -3 $O/t_gpu.log; cut -c1-260 $O/bench_c5.json; tail -1 $O/bench_c5.err
