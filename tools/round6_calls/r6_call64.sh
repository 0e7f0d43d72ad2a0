cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c64; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/t_gpu.log 2>&1; echo "gpu rc $?" >> $O/t_gpu.log; tail -3 $O/t_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log | cut -c1-80
timeout 900 python tools/gpu_fuzz.py 1000 1301 > $O/fuzz_main.log 2>&1; tail -1 $O/fuzz_main.log
bash tools/profile_round.sh > $O/profile.log 2>&1; tail -1 $O/profile.log | cut -c1-300
