cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do
MI_DEGENSAC_PER_CU=1 timeout 400 python tools/gpu_ab5.py 2048 "base_1perCU=0:0" 2>&1 | grep -v amdgpu
MI_DEGENSAC_LIB=tools/libmi_degensac_nospill.so timeout 400 python tools/gpu_ab5.py 2048 "nospill_1perCU=0:0" 2>&1 | grep -v amdgpu
done
