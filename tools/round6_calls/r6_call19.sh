cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c19; rm -rf $O; mkdir -p $O
for i in 1 2; do
  for v in r5 vB HEAD; do
    if [ $v = HEAD ]; then L=""; else L="tools/libmi_degensac_$v.so"; fi
    echo "== $v" >> $O/ab.log
    env ${L:+MI_DEGENSAC_LIB=$L} timeout 400 python tools/gpu_ab5.py 512,64 $v=0:0 2>&1 | grep -v amdgpu >> $O/ab.log
    env ${L:+MI_DEGENSAC_LIB=$L} timeout 400 python tools/gpu_ab.py base 2>&1 | grep "single call" >> $O/ab.log
  done
done
cat $O/ab.log
