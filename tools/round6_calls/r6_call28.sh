cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do
for L in "" tools/libmi_degensac_ch1.so; do env ${L:+MI_DEGENSAC_LIB=$L} timeout 300 python bench.py --config c5 --steps 5 --warmup 1 --no-secondary --no-cpu-baseline --parity-pairs 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 lib=[$L] ms', d['ms_per_step'])"; done
done
env MI_DEGENSAC_LIB=tools/libmi_degensac_ch1.so timeout 600 python -m pytest tests/test_gpu_units.py tests/test_gpu_stated_sizes.py -x -q 2>&1 | tail -2
for L in "" tools/libmi_degensac_ch1.so; do env ${L:+MI_DEGENSAC_LIB=$L} timeout 400 python tools/gpu_ab5.py 4096,512 "x=0:0" 2>&1 | grep -v amdgpu; done
