# Round profile: kernel-trace stats, the PMC passes (separate runs, --kernel-trace only) of C2 / C3 / C5, the bench lines.
# Run on the GPU box:   gpurun -- bash tools/profile_round.sh   -> gpurun_out/r6p/ ; then here: bash tools/collect_profiles.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=r6; O=gpurun_out/${R}p; rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 3 --warmup 1 $B > $O/bench_profiled.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --steps 1 --warmup 0 $B --parity-pairs 0 > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py --steps 1 --warmup 0 $B --parity-pairs 0 > $O/pmc_write.log 2>&1
python tools/pmc_summary.py c2 4096 $O/pmc_fetch $O/pmc_write $O > $O/pmc_summary.log 2>&1
for cfg in c3 c5; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${cfg}_fetch -- python bench.py --config $cfg --steps 1 --warmup 0 $B --parity-pairs 0 > $O/${cfg}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${cfg}_write -- python bench.py --config $cfg --steps 1 --warmup 0 $B --parity-pairs 0 > $O/${cfg}_write.log 2>&1
done
python tools/pmc_summary.py c3 1024 $O/c3_fetch $O/c3_write $O > $O/sum_c3.log 2>&1
python tools/pmc_summary.py c5 1 $O/c5_fetch $O/c5_write $O > $O/sum_c5.log 2>&1
cp $O/${R}_pmc_c2.json $O/${R}_pmc_c3.json $O/${R}_pmc_c5.json profiles/     # on the box: the bench lines below read the counters of THIS source revision
python bench.py --steps 5 --warmup 2 > $O/bench_line.json 2> $O/bench_line.err
python bench.py --steps 3 --warmup 1 --pairs-per-gpu 512 $B > $O/bench_line_512.json 2>/dev/null
python bench.py --config c3 --steps 3 --warmup 1 --no-secondary > $O/bench_line_c3.json 2>/dev/null
python bench.py --config c5 --steps 2 --warmup 1 --no-secondary --parity-pairs 1 > $O/bench_line_c5.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -- python bench.py --config c3 --steps 3 --warmup 1 $B > $O/bench_c3_profiled.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -- python bench.py --config c5 --steps 2 --warmup 1 $B --parity-pairs 0 > $O/bench_c5_profiled.log 2>&1
for p in 8192 16384; do python bench.py --steps 2 --warmup 1 --pairs-per-gpu $p $B --parity-pairs 4 > $O/bench_line_$p.json 2>/dev/null; done
python tools/gpu_host_batch.py 4096 > $O/host_batch.log 2>&1
python tools/gpu_ab.py base nostream serial > $O/ab_f.log 2>&1
python tools/gpu_ab_h.py 0 > $O/ab_h.log 2>&1
# phase tables (development build: tools/libmi_degensac_dev.so = `make -C pydegensac_amd/csrc dev` of the same sources)
python tools/gpu_phases.py 1 50000 0.1 200000 > $O/phases_c5.log 2>&1
python tools/gpu_phases_h.py 1024 > $O/phases_c3.log 2>&1
python tools/gpu_phases.py 1024 > $O/phases_c2_1024.log 2>&1
find $O -name "*.csv" | head -30; tail -1 $O/pmc_summary.log; tail -1 $O/sum_c3.log; tail -1 $O/sum_c5.log; cut -c1-400 $O/bench_line.json
