# LDS counters of the 16 x 16 solver micro-benchmarks (run on the GPU box): bash scripts/micro/pmc_j16.sh <binary> <args...>
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; b=$1; shift
$R/scripts/micro/$b "$@" | tail -1
rm -rf /tmp/pj16; rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pj16 -o pmc -- $R/scripts/micro/$b "$@" > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(float); n=collections.Counter()
for fn in glob.glob('/tmp/pj16/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        if 'k_eigh' in r['Kernel_Name']:
            acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
print('$b', {k: '%.3e'%(v/max(1,n[k])) for k,v in acc.items()}, n['SQ_INSTS_LDS'])
PY
