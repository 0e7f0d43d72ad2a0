cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c77; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/t_gpu.log 2>&1; rc=$?; echo "gpu rc $rc" >> $O/t_gpu.log; tail -3 $O/t_gpu.log
[ $rc -ne 0 ] && exit 1
timeout 600 python tools/gpu_fuzz.py 1500 1951 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
grep -q "1500/1500 identical" $O/fuzz.log || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log | cut -c1-80
timeout 1200 python tools/gpu_fuzz.py 4000 1961 > $O/fuzz_main.log 2>&1; tail -1 $O/fuzz_main.log
timeout 900 python tools/gpu_fuzz.py edges 1500 1962 > $O/fuzz_edges.log 2>&1; tail -1 $O/fuzz_edges.log
timeout 900 python tools/gpu_fuzz_h.py 1000 1963 > $O/fuzz_h.log 2>&1; tail -1 $O/fuzz_h.log
timeout 600 python tools/gpu_fuzz.py legacy 400 1965 > $O/fuzz_legacy.log 2>&1; tail -1 $O/fuzz_legacy.log
timeout 900 python tools/gpu_fuzz.py batches 60 1967 > $O/fuzz_batches.log 2>&1; tail -1 $O/fuzz_batches.log
timeout 900 python tools/gpu_fuzz.py set-aside 30 1966 > $O/fuzz_aside.log 2>&1; tail -1 $O/fuzz_aside.log
timeout 900 python tools/gpu_fuzz.py large 100 1968 > $O/fuzz_large.log 2>&1; tail -1 $O/fuzz_large.log
timeout 600 python tools/gpu_fuzz_h2el.py 30 1969 > $O/fuzz_h2el.log 2>&1; tail -1 $O/fuzz_h2el.log
bash tools/profile_round.sh > $O/profile.log 2>&1; tail -1 $O/profile.log | cut -c1-300
