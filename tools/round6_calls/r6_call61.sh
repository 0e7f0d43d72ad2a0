cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in $MASKS; do
  echo "mask $m"; MI_DEGENSAC_LIB=tools/libmi_degensac_m$m.so timeout 600 python -m pytest tests/test_gpu_golden.py -q 2>&1 | tail -2 | head -1
  MI_DEGENSAC_LIB=tools/libmi_degensac_m$m.so timeout 600 python tools/gpu_fuzz.py large 40 1003 2>&1 | tail -1
done
