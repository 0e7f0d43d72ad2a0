cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c60; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_golden.py -q 2>&1 | tail -3
timeout 600 python tools/gpu_fuzz.py large 60 1003 > $O/fuzz_large.log 2>&1; tail -1 $O/fuzz_large.log
MI_DEGENSAC_LIB=tools/libmi_degensac_prev.so timeout 600 python tools/gpu_fuzz.py large 60 1003 > $O/fuzz_large_prev.log 2>&1; tail -1 $O/fuzz_large_prev.log
