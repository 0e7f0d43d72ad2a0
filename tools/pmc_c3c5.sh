cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p35; rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-secondary --parity-pairs 0"
for cfg in c3 c5; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${cfg}_fetch -- python bench.py --config $cfg --steps 1 --warmup 0 $B > $O/${cfg}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${cfg}_write -- python bench.py --config $cfg --steps 1 --warmup 0 $B > $O/${cfg}_write.log 2>&1
done
python tools/pmc_summary.py c3 1024 $O/c3_fetch $O/c3_write $O > $O/sum_c3.log 2>&1
python tools/pmc_summary.py c5 1 $O/c5_fetch $O/c5_write $O > $O/sum_c5.log 2>&1
tail -1 $O/sum_c3.log; tail -1 $O/sum_c5.log
