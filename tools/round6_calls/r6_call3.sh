cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c3; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_round5.py tests/test_gpu_variants.py -x -q > $O/t_stream.log 2>&1; echo "rc $?" >> $O/t_stream.log
timeout 600 python tools/gpu_ab5.py 512,64,4096 base=0:0 > $O/ab_base.log 2>&1
timeout 600 python tools/gpu_ab.py base nostream > $O/ab_f.log 2>&1
tail -3 $O/t_stream.log; grep -v amdgpu $O/ab_base.log; grep -v amdgpu $O/ab_f.log | tail -30
