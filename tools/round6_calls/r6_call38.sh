cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c38; rm -rf $O; mkdir -p $O
AB_H_PAIRS=384,512,640,768,1536,2048 timeout 900 python tools/gpu_ab_h.py 0 5 6 7 > $O/ab_h.log 2>&1; grep -v amdgpu $O/ab_h.log | tail -40
