cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c78; rm -rf $O; mkdir -p $O
python -c "import bench; print('sources', bench.source_id())" > $O/sources.log 2>&1
timeout 1500 python tools/gpu_fuzz.py 10000 2001 > $O/fuzz_main.log 2>&1; tail -1 $O/fuzz_main.log
timeout 900 python tools/gpu_fuzz_h.py 3000 2002 > $O/fuzz_h.log 2>&1; tail -1 $O/fuzz_h.log
timeout 900 python tools/gpu_fuzz.py large 250 2003 > $O/fuzz_large.log 2>&1; tail -1 $O/fuzz_large.log
timeout 900 python tools/gpu_fuzz.py batches 150 2004 > $O/fuzz_batches.log 2>&1; tail -1 $O/fuzz_batches.log
timeout 600 python tools/gpu_fuzz.py edges 4000 2005 > $O/fuzz_edges.log 2>&1; tail -1 $O/fuzz_edges.log
timeout 600 python tools/gpu_fuzz.py legacy 1000 2006 > $O/fuzz_legacy.log 2>&1; tail -1 $O/fuzz_legacy.log
timeout 600 python tools/gpu_fuzz_h2el.py 60 2007 > $O/fuzz_h2el.log 2>&1; tail -1 $O/fuzz_h2el.log
timeout 900 python tools/gpu_soak_mem.py 60 > $O/soak_mem.log 2>&1; tail -2 $O/soak_mem.log
timeout 600 python bench.py > $O/bench_default.log 2> $O/bench_default.err; cut -c1-300 $O/bench_default.log
