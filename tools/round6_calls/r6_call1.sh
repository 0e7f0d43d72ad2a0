cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c1; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q > $O/t_round6.log 2>&1; echo "round6 rc $?" >> $O/t_round6.log
timeout 300 python tools/gpu_simd.py 2048 0 > $O/simd_t256.log 2>&1
timeout 300 python tools/gpu_simd.py 2048 3 > $O/simd_t128.log 2>&1
timeout 300 python tools/gpu_simd.py 2048 1 > $O/simd_t512.log 2>&1
timeout 900 python tools/gpu_ab5.py 4096 base=0:0:-1:0:0 t128=0:3:-1:0:0 t128x2=0:3:-1:0:2 t128x1=0:3:-1:0:1 t256x1=0:2:-1:0:1 > $O/ab_percu.log 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $?" >> $O/bench_line.err
timeout 1800 python -m pytest tests -m gpu -x -q > $O/t_gpu.log 2>&1; echo "gpu rc $?" >> $O/t_gpu.log
tail -3 $O/t_round6.log; cat $O/simd_t256.log $O/simd_t128.log | head -40; cat $O/ab_percu.log; tail -3 $O/t_gpu.log; cut -c1-600 $O/bench_line.json
