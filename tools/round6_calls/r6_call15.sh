cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c15; rm -rf $O; mkdir -p $O
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $?" >> $O/bench_line.err
timeout 1500 python tools/gpu_fuzz.py 1500 611 > $O/fuzz_main.log 2>&1; tail -3 $O/fuzz_main.log
timeout 900 python tools/gpu_fuzz.py large 30 612 > $O/fuzz_large.log 2>&1; tail -3 $O/fuzz_large.log
timeout 900 python tools/gpu_fuzz.py batches 40 613 > $O/fuzz_batches.log 2>&1; tail -3 $O/fuzz_batches.log
tail -2 $O/bench_line.err
