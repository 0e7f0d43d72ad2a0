cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c30; rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-secondary --steps 1 --warmup 0 --parity-pairs 0"
MI_DEGENSAC_LIB=tools/libmi_degensac_nospill.so rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w_ns -- python bench.py $B > $O/w_ns.log 2>&1
MI_DEGENSAC_LIB=tools/libmi_degensac_nospill.so rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f_ns -- python bench.py $B > $O/f_ns.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w_base -- python bench.py $B > $O/w_base.log 2>&1
for d in w_ns f_ns w_base; do f=$(ls $O/$d/*/*_counter_collection.csv | head -1); echo "== $d"; python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(float)
for r in rows:
    if 'dg_find_fundamental_kernel<256' in r['Kernel_Name']: acc[r['Counter_Name']] += float(r['Counter_Value'])
print({k: round(v / 1048576, 2) for k, v in acc.items()}, "GiB (counter is in KiB)")
PY
done
grep -h "ms_per_step" $O/w_ns.log | cut -c1-200
