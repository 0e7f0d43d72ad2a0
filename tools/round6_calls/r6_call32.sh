cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout 400 python tools/gpu_ab5.py 4096 "base=0:0" 2>&1 | grep -v amdgpu
MI_DEGENSAC_LIB=tools/libmi_degensac_e2.so timeout 400 python tools/gpu_ab5.py 4096 "e2=0:0" 2>&1 | grep -v amdgpu
done
