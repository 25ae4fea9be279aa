# rocprofv3 kernel stats for the C4 / C5 shapes and SQ counters for the 7v7 kernel (C3).
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_more; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4 -- python3 bench.py --no-cpu --config c4 > $O/stats_c4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -- python3 bench.py --no-cpu --config c5 > $O/stats_c5.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_c3 -- python3 bench.py --no-cpu --config c3 --steps 250 --warmup 10 > $O/pmc_c3.log 2>&1
python3 - <<'PY'
import csv,glob,json,collections
O='gpurun_out/prof_more'
acc=collections.defaultdict(float)
for f in glob.glob(O+'/pmc_c3/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_step_hybrid2' in r['Kernel_Name']: acc[r['Counter_Name']]+=float(r['Counter_Value'])
steps=260; B=65536
acc['_valu_insts_per_arena_step']=acc['SQ_INSTS_VALU']/(steps*B) if acc else 0
acc['_note']="k_step_hybrid2<7,3>, bench.py --no-cpu --config c3 --steps 250 --warmup 10 (260 steps x 65536 arenas, 32 arenas per wave); wave-instructions per arena-step = SQ_INSTS_VALU / arena-steps"
json.dump(acc,open(O+'/hybrid2_7v7_pmc.json','w'),indent=1); print(json.dumps(acc,indent=1))
for c in ('c4','c5'):
    for f in glob.glob(O+'/stats_%s/**/*kernel_stats.csv'%c,recursive=True): print(c, open(f).read().splitlines()[1][:160])
PY
