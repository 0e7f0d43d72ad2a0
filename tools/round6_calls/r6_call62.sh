cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c62; rm -rf $O; mkdir -p $O
timeout 1200 python tools/gpu_fuzz.py 3000 1101 > $O/fuzz_main.log 2>&1; tail -1 $O/fuzz_main.log
timeout 900 python tools/gpu_fuzz.py edges 1500 1102 > $O/fuzz_edges.log 2>&1; tail -1 $O/fuzz_edges.log
timeout 900 python tools/gpu_fuzz_h.py 800 1103 > $O/fuzz_h.log 2>&1; tail -1 $O/fuzz_h.log
timeout 600 python tools/gpu_fuzz_h2el.py 30 1104 > $O/fuzz_h2el.log 2>&1; tail -1 $O/fuzz_h2el.log
timeout 600 python tools/gpu_fuzz.py legacy 400 1105 > $O/fuzz_legacy.log 2>&1; tail -1 $O/fuzz_legacy.log
timeout 900 python tools/gpu_fuzz.py set-aside 20 1106 > $O/fuzz_aside.log 2>&1; tail -1 $O/fuzz_aside.log
timeout 900 python tools/gpu_fuzz.py batches 60 1107 > $O/fuzz_batches.log 2>&1; tail -1 $O/fuzz_batches.log
timeout 900 python tools/gpu_fuzz.py large 80 1108 > $O/fuzz_large.log 2>&1; tail -1 $O/fuzz_large.log
timeout 900 python tools/gpu_soak_mem.py 60 > $O/soak.log 2>&1; tail -2 $O/soak.log
