cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c42; rm -rf $O; mkdir -p $O
for f in 33 2 11 6; do
  for t in 0 0x1705; do
    echo "=== pair $f tune $t" >> $O/ph.log
    PH_FIRST=$f PH_TUNE=$t timeout 300 python tools/gpu_phases.py 1 2000 0.4 100000 2>&1 | grep -v amdgpu | grep "batch ms\|^pair" | cut -c1-600 >> $O/ph.log
  done
done
cat $O/ph.log
