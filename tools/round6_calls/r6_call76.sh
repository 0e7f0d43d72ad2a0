cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c76; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/t_gpu.log 2>&1; echo "gpu rc $?" >> $O/t_gpu.log; tail -3 $O/t_gpu.log
timeout 600 python tools/gpu_fuzz.py 1500 1941 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
timeout 600 python tools/gpu_fuzz_h.py 600 1942 > $O/fuzz_h.log 2>&1; tail -1 $O/fuzz_h.log
timeout 600 python tools/gpu_fuzz.py large 60 1943 > $O/fuzz_large.log 2>&1; tail -1 $O/fuzz_large.log
for i in 1 2; do
  MI_DEGENSAC_LIB=tools/libmi_degensac_prev.so timeout 400 python tools/gpu_ab5.py 4096,512 prev=0:0 2>&1 | grep -v amdgpu | cut -c1-210
  timeout 400 python tools/gpu_ab5.py 4096,512 cur=0:0 2>&1 | grep -v amdgpu | cut -c1-210
  MI_DEGENSAC_LIB=tools/libmi_degensac_prev.so timeout 400 python tools/gpu_ab_h.py 0 2>&1 | grep "C3 x 1024 helpers 1\|one C3 pair per call, helpers 1" | sed 's/^/prev /' | cut -c1-210
  timeout 400 python tools/gpu_ab_h.py 0 2>&1 | grep "C3 x 1024 helpers 1\|one C3 pair per call, helpers 1" | sed 's/^/cur  /' | cut -c1-210
done
for i in 1 2; do
  MI_DEGENSAC_LIB=tools/libmi_degensac_prev.so python bench.py --config c5 --steps 3 --warmup 1 --no-secondary --parity-pairs 1 --no-cpu-baseline 2>/dev/null | cut -c178-230
  python bench.py --config c5 --steps 3 --warmup 1 --no-secondary --parity-pairs 1 --no-cpu-baseline 2>/dev/null | cut -c178-230
done
