cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c2; rm -rf $O; mkdir -p $O
MI_DEGENSAC_TUNING=2 timeout 300 python tools/gpu_lo.py 1024 > $O/lo_prof.log 2>&1
timeout 400 python tools/gpu_ab5.py 4096,512 base=0:0 > $O/ab_base.log 2>&1
MI_DEGENSAC_LIB=tools/libmi_degensac_exp1.so timeout 400 python tools/gpu_ab5.py 4096,512 exp1=0:0 > $O/ab_exp1.log 2>&1
timeout 400 python tools/gpu_ab5.py 4096,512 base=0:0 > $O/ab_base2.log 2>&1
timeout 300 python tools/gpu_phases.py 1024 > $O/phases_c2_1024.log 2>&1
cat $O/lo_prof.log | grep -v amdgpu.ids; cat $O/ab_base.log $O/ab_exp1.log $O/ab_base2.log | grep -v amdgpu; head -3 $O/phases_c2_1024.log | cut -c1-600
