cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c66; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity_f.py tests/test_gpu_variants.py tests/test_gpu_units.py -x -q 2>&1 | tail -2
timeout 600 python tools/gpu_fuzz.py 1000 1501 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
timeout 600 python tools/gpu_fuzz.py large 60 1502 > $O/fuzz_large.log 2>&1; tail -1 $O/fuzz_large.log
for i in 1 2; do
  MI_DEGENSAC_LIB=tools/libmi_degensac_prev.so timeout 400 python tools/gpu_ab5.py 4096,512 prev=0:0 2>&1 | grep -v amdgpu | cut -c1-210 >> $O/ab.log
  timeout 400 python tools/gpu_ab5.py 4096,512 cur=0:0 2>&1 | grep -v amdgpu | cut -c1-210 >> $O/ab.log
done
cat $O/ab.log
