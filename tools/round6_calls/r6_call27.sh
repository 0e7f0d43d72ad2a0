cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c27; rm -rf $O; mkdir -p $O
timeout 1200 python tools/gpu_fuzz.py 4000 621 > $O/fuzz_main.log 2>&1; tail -1 $O/fuzz_main.log
timeout 900 python tools/gpu_fuzz.py edges 2000 622 > $O/fuzz_edges.log 2>&1; tail -1 $O/fuzz_edges.log
timeout 900 python tools/gpu_fuzz_h.py 1000 623 > $O/fuzz_h.log 2>&1; tail -1 $O/fuzz_h.log
timeout 600 python tools/gpu_fuzz_h2el.py 30 624 > $O/fuzz_h2el.log 2>&1; tail -1 $O/fuzz_h2el.log
timeout 600 python tools/gpu_fuzz.py legacy 600 625 > $O/fuzz_legacy.log 2>&1; tail -1 $O/fuzz_legacy.log
timeout 900 python tools/gpu_fuzz.py set-aside 30 626 > $O/fuzz_aside.log 2>&1; tail -1 $O/fuzz_aside.log
timeout 900 python tools/gpu_fuzz.py batches 80 627 > $O/fuzz_batches.log 2>&1; tail -1 $O/fuzz_batches.log
