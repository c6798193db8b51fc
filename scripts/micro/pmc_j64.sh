cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for b in jacobi64_bench_local jacobi64_bench; do
  $R/scripts/micro/$b 4 | tail -1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pj_$b -o pmc -- $R/scripts/micro/$b 4 > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob('/tmp/pj_$b/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(float); n=collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        if 'k64' in r['Kernel_Name']:
            acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
print('$b', {k: '%.3e'%(v/max(1,n[k])*1) for k,v in acc.items()}, dict(n))
PY
done
