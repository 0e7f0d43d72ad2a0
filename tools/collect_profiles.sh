# Copies the newest summaries of gpurun_out/r2c (merged back from `gpurun -- bash tools/profile_round.sh`) into profiles/.
O=gpurun_out/r2c
new() { ls -t $1 | head -1; }
cp $(new "$O/stats/*/*_kernel_stats.csv") profiles/r2_bench_kernel_stats.csv
cp $(new "$O/stats_c3/*/*_kernel_stats.csv") profiles/r2_bench_c3_kernel_stats.csv
f=$(new "$O/stats/*/*_kernel_trace.csv");    (head -1 $f; grep 'dg_find_fundamental_kernel<256' $f | head -2) > profiles/r2_bench_kernel_trace_head.csv
f=$(new "$O/stats_c3/*/*_kernel_trace.csv"); (head -1 $f; grep 'dg_find_homography_kernel<128' $f | head -2) > profiles/r2_bench_c3_kernel_trace_head.csv
for k in fetch write; do f=$(new "$O/pmc_$k/*/*_counter_collection.csv"); (head -1 $f; grep 'dg_find_fundamental_kernel<256' $f) > profiles/r2_bench_pmc_${k}_size.csv; done
cp $O/r2_pmc_c2.json profiles/
tail -1 $O/bench_line.json > profiles/r2_bench_line.json
tail -1 $O/bench_line_512.json > profiles/r2_bench_line_512_pairs.json
tail -1 $O/bench_line_c3.json > profiles/r2_bench_line_c3.json
python -c "import bench; print('sources', bench.source_id(), 'traffic', bench.pmc_traffic('c2', 4096))"
