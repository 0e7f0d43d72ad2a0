cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c17; rm -rf $O; mkdir -p $O
for i in 1 2; do
  for v in r5 vB HEAD; do
    if [ $v = HEAD ]; then timeout 400 python tools/gpu_ab5.py 4096 HEAD=0:0 2>&1 | grep -v amdgpu >> $O/ab.log; else MI_DEGENSAC_LIB=tools/libmi_degensac_$v.so timeout 400 python tools/gpu_ab5.py 4096 $v=0:0 2>&1 | grep -v amdgpu >> $O/ab.log; fi
  done
done
cat $O/ab.log
