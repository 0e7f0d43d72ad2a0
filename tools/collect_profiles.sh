# Copies the newest summaries of gpurun_out/r3p (merged back from `gpurun -- bash tools/profile_round.sh`) into profiles/.
O=gpurun_out/r3p
new() { ls -t $1 | head -1; }
cp $(new "$O/stats/*/*_kernel_stats.csv") profiles/r3_bench_kernel_stats.csv
cp $(new "$O/stats_c3/*/*_kernel_stats.csv") profiles/r3_bench_c3_kernel_stats.csv
cp $(new "$O/stats_c5/*/*_kernel_stats.csv") profiles/r3_bench_c5_kernel_stats.csv
cp $(new "$O/stats_matcher/*/*_kernel_stats.csv") profiles/r3_matcher_kernel_stats.csv
f=$(new "$O/stats/*/*_kernel_trace.csv");    (head -1 $f; grep 'dg_find_fundamental_kernel<256' $f | head -2) > profiles/r3_bench_kernel_trace_head.csv
f=$(new "$O/stats_c3/*/*_kernel_trace.csv"); (head -1 $f; grep 'dg_find_homography_kernel<128' $f | head -2) > profiles/r3_bench_c3_kernel_trace_head.csv
f=$(new "$O/stats_c5/*/*_kernel_trace.csv"); (head -1 $f; grep 'dg_find_fundamental_kernel' $f | head -2) > profiles/r3_bench_c5_kernel_trace_head.csv
for k in fetch write; do f=$(new "$O/pmc_$k/*/*_counter_collection.csv"); (head -1 $f; grep 'dg_find_fundamental_kernel<256' $f) > profiles/r3_bench_pmc_${k}_size.csv; done
cp $O/r3_pmc_c2.json profiles/
tail -1 $O/bench_line.json > profiles/r3_bench_line.json
tail -1 $O/bench_line_512.json > profiles/r3_bench_line_512_pairs.json
tail -1 $O/bench_line_c3.json > profiles/r3_bench_line_c3.json
tail -1 $O/bench_line_c5.json > profiles/r3_bench_line_c5.json
python - <<'PY'
import json
out = {}
for p in (8192, 16384):
    try:
        j = json.loads(open(f"gpurun_out/r3p/bench_line_{p}.json").read().strip().splitlines()[-1])
        out[str(p)] = {k: j[k] for k in ("value", "ms_per_step", "pairs_set_aside")} | {"frac": j["roofline"]["frac"], "kernel_ms": j["roofline"]["kernel_ms"]}
    except Exception as e:
        out[str(p)] = {"error": str(e)}
json.dump(out, open("profiles/r3_bench_batch_sizes.json", "w"), indent=1)
PY
grep -v amdgpu.ids $O/matcher.log > profiles/r3_matcher.log
python -c "import bench; print('sources', bench.source_id(), 'traffic', bench.pmc_traffic('c2', 4096))"
