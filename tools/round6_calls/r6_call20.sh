cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c20; rm -rf $O; mkdir -p $O
for i in 1 2 3; do
  MI_DEGENSAC_LIB=tools/libmi_degensac_r5.so timeout 400 python tools/gpu_ab5.py 4096,512 r5=0:0 2>&1 | grep -v amdgpu >> $O/ab.log
  timeout 400 python tools/gpu_ab5.py 4096,512 cur=0:0 2>&1 | grep -v amdgpu >> $O/ab.log
done
cat $O/ab.log
