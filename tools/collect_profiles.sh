# Copies the newest summaries of gpurun_out/r6p (merged back from `gpurun -- bash tools/profile_round.sh`) into profiles/.
R=r6; O=gpurun_out/${R}p
new() { ls -t $1 | head -1; }
cp $(new "$O/stats/*/*_kernel_stats.csv") profiles/${R}_bench_kernel_stats.csv
cp $(new "$O/stats_c3/*/*_kernel_stats.csv") profiles/${R}_bench_c3_kernel_stats.csv
cp $(new "$O/stats_c5/*/*_kernel_stats.csv") profiles/${R}_bench_c5_kernel_stats.csv
f=$(new "$O/stats/*/*_kernel_trace.csv");    (head -1 $f; grep 'dg_find_fundamental_kernel<256' $f | head -2) > profiles/${R}_bench_kernel_trace_head.csv
f=$(new "$O/stats_c3/*/*_kernel_trace.csv"); (head -1 $f; grep 'dg_find_homography_kernel<128' $f | head -2) > profiles/${R}_bench_c3_kernel_trace_head.csv
f=$(new "$O/stats_c5/*/*_kernel_trace.csv"); (head -1 $f; grep 'dg_find_fundamental_kernel' $f | head -2) > profiles/${R}_bench_c5_kernel_trace_head.csv
for k in fetch write; do f=$(new "$O/pmc_$k/*/*_counter_collection.csv"); (head -1 $f; grep 'dg_find_fundamental_kernel<256' $f) > profiles/${R}_bench_pmc_${k}_size.csv; done
for c in c3 c5; do for k in fetch write; do f=$(new "$O/${c}_$k/*/*_counter_collection.csv"); (head -1 $f; grep 'dg_find_' $f | head -40) > profiles/${R}_bench_${c}_pmc_${k}_size.csv; done; done
cp $O/${R}_pmc_c2.json $O/${R}_pmc_c3.json $O/${R}_pmc_c5.json profiles/
tail -1 $O/bench_line.json > profiles/${R}_bench_line.json
tail -1 $O/bench_line_512.json > profiles/${R}_bench_line_512_pairs.json
tail -1 $O/bench_line_c3.json > profiles/${R}_bench_line_c3.json
tail -1 $O/bench_line_c5.json > profiles/${R}_bench_line_c5.json
grep -v amdgpu.ids $O/ab_f.log > profiles/${R}_ab_fundamental.log
grep -v amdgpu.ids $O/ab_h.log > profiles/${R}_ab_homography.log
grep -v amdgpu.ids $O/host_batch.log > profiles/${R}_host_batch.log
grep -v amdgpu.ids $O/phases_c5.log | cut -c1-1200 > profiles/${R}_phases_c5.log
grep -v amdgpu.ids $O/phases_c3.log | cut -c1-1200 > profiles/${R}_phases_c3.log
grep -v amdgpu.ids $O/phases_c2_1024.log | grep -v '^pair ' | cut -c1-1200 > profiles/${R}_phases_c2_1024.log
python - <<'PY'
import json
out = {}
for p in (8192, 16384):
    try:
        j = json.loads(open(f"gpurun_out/r6p/bench_line_{p}.json").read().strip().splitlines()[-1])
        out[str(p)] = {k: j[k] for k in ("value", "ms_per_step", "pairs_set_aside")} | {"frac": j["roofline"]["frac"], "kernel_ms": j["roofline"]["kernel_ms"]}
    except Exception as e:
        out[str(p)] = {"error": str(e)}
json.dump(out, open("profiles/r6_bench_batch_sizes.json", "w"), indent=1)
PY
python -c "import bench; print('sources', bench.source_id(), 'traffic', bench.pmc_traffic('c2', 4096))"
