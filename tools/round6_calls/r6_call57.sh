cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c57; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity_f.py tests/test_gpu_parity_h.py tests/test_gpu_variants.py tests/test_gpu_golden.py tests/test_gpu_round5.py tests/test_gpu_hjob.py tests/test_gpu_round6.py -x -q > $O/t1.log 2>&1; tail -2 $O/t1.log
timeout 600 python tools/gpu_fuzz.py 600 991 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
timeout 600 python tools/gpu_fuzz_h.py 300 992 > $O/fuzz_h.log 2>&1; tail -1 $O/fuzz_h.log
timeout 600 python tools/gpu_fuzz.py large 30 993 > $O/fuzz_large.log 2>&1; tail -1 $O/fuzz_large.log
timeout 300 python tools/gpu_phases.py 256 2>&1 | grep -v amdgpu | cut -c1-330 | grep "^mean\|^pair 161"
for i in 1 2; do
  MI_DEGENSAC_LIB=tools/libmi_degensac_prev.so timeout 400 python tools/gpu_ab5.py 4096 prev=0:0 2>&1 | grep -v amdgpu | cut -c1-210 >> $O/ab.log
  timeout 400 python tools/gpu_ab5.py 4096 cur=0:0 2>&1 | grep -v amdgpu | cut -c1-210 >> $O/ab.log
done
cat $O/ab.log
MI_DEGENSAC_LIB=tools/libmi_degensac_prev.so timeout 300 python tools/gpu_ab_h.py 0 2>&1 | grep "C3 x 1024 helpers 1\|one C3 pair per call, helpers 1"
timeout 300 python tools/gpu_ab_h.py 0 2>&1 | grep "C3 x 1024 helpers 1\|one C3 pair per call, helpers 1"
