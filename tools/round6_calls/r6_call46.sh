cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c46; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/t_gpu.log 2>&1; echo "gpu rc $?" >> $O/t_gpu.log; tail -3 $O/t_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log | cut -c1-100
timeout 900 python tools/gpu_fuzz.py 1500 941 > $O/fuzz_main.log 2>&1; tail -1 $O/fuzz_main.log
timeout 900 python tools/gpu_fuzz.py edges 800 942 > $O/fuzz_edges.log 2>&1; tail -1 $O/fuzz_edges.log
timeout 600 python tools/gpu_fuzz.py legacy 300 943 > $O/fuzz_legacy.log 2>&1; tail -1 $O/fuzz_legacy.log
timeout 900 python tools/gpu_fuzz.py batches 40 944 > $O/fuzz_batches.log 2>&1; tail -1 $O/fuzz_batches.log
