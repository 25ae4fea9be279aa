cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3e4; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > $O/stats.log 2>&1
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200 | head -12
f=$(find $O/stats -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[-12:]:
    print(r['Kernel_Name'][:60], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, 'us', 'start', (int(r['Start_Timestamp'])-int(rows[-12]['Start_Timestamp']))/1e3)
PY
