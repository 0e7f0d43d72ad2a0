cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c69; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/t_gpu.log 2>&1; echo "gpu rc $?" >> $O/t_gpu.log; tail -3 $O/t_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log | cut -c1-80
timeout 900 python tools/gpu_fuzz.py large 120 1702 > $O/fuzz_large.log 2>&1; tail -1 $O/fuzz_large.log
timeout 900 python bench.py > $O/bench.log 2>$O/bench.err; cut -c1-400 $O/bench.log
