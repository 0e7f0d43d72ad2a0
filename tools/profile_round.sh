# Round profile: kernel-trace stats, the two PMC passes (separate runs, --kernel-trace only), the bench lines.  Run on the GPU box:
#   gpurun -- bash tools/profile_round.sh   -> gpurun_out/r2c/ (copy the summaries into profiles/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_profiled.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --parity-pairs 0 > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --parity-pairs 0 > $O/pmc_write.log 2>&1
python tools/pmc_summary.py c2 4096 $O/pmc_fetch $O/pmc_write $O > $O/pmc_summary.log 2>&1
cp $O/r2_pmc_c2.json profiles/r2_pmc_c2.json     # on the box: the bench lines below read the counters of THIS source revision
python bench.py --steps 3 --warmup 1 > $O/bench_line.json 2> $O/bench_line.err
python bench.py --steps 3 --warmup 1 --pairs-per-gpu 512 --no-cpu-baseline > $O/bench_line_512.json 2>/dev/null
python bench.py --config c3 --steps 3 --warmup 1 > $O/bench_line_c3.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -- python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c3_profiled.log 2>&1
python tools/gpu_host_batch.py 4096 > $O/host_batch.log 2>&1
python tools/gpu_set_aside_stats.py 4096 255 0 > $O/set_aside.log 2>&1
C5_ORACLE=0 python tools/gpu_c5.py > $O/c5.log 2>&1
find $O -name "*.csv" | head -20; tail -2 $O/pmc_summary.log; cut -c1-400 $O/bench_line.json
