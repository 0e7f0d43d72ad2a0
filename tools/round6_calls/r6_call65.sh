cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c65; rm -rf $O; mkdir -p $O
timeout 1500 python tools/gpu_fuzz.py 10000 1401 > $O/fuzz_main.log 2>&1; tail -1 $O/fuzz_main.log
timeout 900 python tools/gpu_fuzz_h.py 3000 1402 > $O/fuzz_h.log 2>&1; tail -1 $O/fuzz_h.log
timeout 1200 python tools/gpu_fuzz.py large 250 1403 > $O/fuzz_large.log 2>&1; tail -1 $O/fuzz_large.log
timeout 1200 python tools/gpu_fuzz.py batches 150 1404 > $O/fuzz_batches.log 2>&1; tail -1 $O/fuzz_batches.log
timeout 900 python tools/gpu_fuzz.py edges 4000 1405 > $O/fuzz_edges.log 2>&1; tail -1 $O/fuzz_edges.log
timeout 600 python tools/gpu_fuzz.py legacy 1000 1406 > $O/fuzz_legacy.log 2>&1; tail -1 $O/fuzz_legacy.log
timeout 600 python tools/gpu_fuzz_h2el.py 60 1407 > $O/fuzz_h2el.log 2>&1; tail -1 $O/fuzz_h2el.log
