# Round profile: kernel-trace stats, the two PMC passes (separate runs, --kernel-trace only), the bench lines.  Run on the GPU box:
#   gpurun -- bash tools/profile_round.sh   -> gpurun_out/r3p/ (tools/collect_profiles.sh copies the summaries into profiles/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p; rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 3 --warmup 1 $B > $O/bench_profiled.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --steps 1 --warmup 0 $B --parity-pairs 0 > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py --steps 1 --warmup 0 $B --parity-pairs 0 > $O/pmc_write.log 2>&1
python tools/pmc_summary.py c2 4096 $O/pmc_fetch $O/pmc_write $O > $O/pmc_summary.log 2>&1
cp $O/r3_pmc_c2.json profiles/r3_pmc_c2.json     # on the box: the bench lines below read the counters of THIS source revision
python bench.py --steps 3 --warmup 1 > $O/bench_line.json 2> $O/bench_line.err
python bench.py --steps 3 --warmup 1 --pairs-per-gpu 512 $B > $O/bench_line_512.json 2>/dev/null
python bench.py --config c3 --steps 3 --warmup 1 --no-secondary > $O/bench_line_c3.json 2>/dev/null
python bench.py --config c5 --steps 2 --warmup 1 --no-secondary --parity-pairs 1 > $O/bench_line_c5.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -- python bench.py --config c3 --steps 3 --warmup 1 $B > $O/bench_c3_profiled.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -- python bench.py --config c5 --steps 2 --warmup 1 $B --parity-pairs 0 > $O/bench_c5_profiled.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_matcher -- python tools/gpu_matcher.py > $O/matcher.log 2>&1
for p in 8192 16384; do python bench.py --steps 2 --warmup 1 --pairs-per-gpu $p $B --parity-pairs 4 > $O/bench_line_$p.json 2>/dev/null; done
python tools/gpu_host_batch.py 4096 > $O/host_batch.log 2>&1
find $O -name "*.csv" | head -30; tail -2 $O/pmc_summary.log; cut -c1-400 $O/bench_line.json
