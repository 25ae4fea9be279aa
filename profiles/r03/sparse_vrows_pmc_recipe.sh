# WRITE_SIZE / FETCH_SIZE of the C3 launch with and without sparse velocity rows (librgs_hip_vr.so = make SUFFIX=_vr EXTRA=-DGSIM_SPARSE_VROWS=1)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3pmc; rm -rf $O; mkdir -p $O
for s in base _vr; do
  sfx=$s; [ "$s" = "base" ] && sfx=""
  export RGS_LIB_SUFFIX=$sfx
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w_$s -- python3 bench.py --config c3 --no-cpu > $O/w_$s.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f_$s -- python3 bench.py --config c3 --no-cpu > $O/f_$s.log 2>&1
done
python3 - <<'PY'
import csv,glob
for d in sorted(glob.glob('gpurun_out/r3pmc/*_*')):
    if not d.endswith(('base','_vr')): continue
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        rows=list(csv.DictReader(open(f)))
        ids=[int(r['Dispatch_Id']) for r in rows if 'k_step_' in r['Kernel_Name']]
        last=max(ids); tot=sum(float(r['Counter_Value']) for r in rows if int(r['Dispatch_Id'])==last)
        name=[r['Counter_Name'] for r in rows if int(r['Dispatch_Id'])==last][0]
        print(d, name, 'KB', tot, 'GB', tot*1024/1e9)
PY
