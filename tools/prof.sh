set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -- python3 bench.py --no-cpu > $O/stats_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -- python3 bench.py --no-cpu --config c3 > $O/stats_c3.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -- python3 bench.py --no-cpu --steps 250 --warmup 10 > $O/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --no-cpu --steps 250 --warmup 10 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --no-cpu --steps 250 --warmup 10 > $O/pmc_write.log 2>&1
find $O -name "*.csv" | head -40
python3 - <<'PY'
import csv,glob,json,collections
O='gpurun_out/prof'
out={}
for tag in ('pmc_sq','pmc_fetch','pmc_write'):
    for f in glob.glob(O+'/'+tag+'/**/*counter_collection.csv',recursive=True):
        acc=collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if 'k_step_hybrid' in r['Kernel_Name']:
                acc[r['Counter_Name']]+=float(r['Counter_Value'])
        out.update(acc)
print(json.dumps(out,indent=1))
json.dump(out,open(O+'/pmc_sums.json','w'),indent=1)
PY
