cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e7; rm -rf $O; mkdir -p $O
RGS_LIB_SUFFIX=_stamps timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --stamps --no-cpu > $O/st.json 2> $O/st.err
cp gpurun_out/wave_log.npy $O/wave_log_drv.npy
RGS_LIB_SUFFIX=_stamps timeout -k 10 200 python bench.py --config c2 --stamps --no-cpu > $O/st_c2.json 2>> $O/st.err
cp gpurun_out/wave_log.npy $O/wave_log_c2.npy
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3e7/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); e=d['events']
    print(f, '%.4g'%d['value'], json.dumps(e.get('wave_log')), {k:e['stamps'][k] for k in ('rounds','g0_fill','g0_scan0','g0_events','g0_arenas')})
PY
