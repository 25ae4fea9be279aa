cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e5; rm -rf $O; mkdir -p $O
for rep in 1 2; do
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > $O/drv_$rep.json 2> $O/drv.err
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rebalance > $O/drv_noreb_$rep.json 2>> $O/drv.err
done
RGS_LIB_SUFFIX=_stamps timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --stamps --no-cpu > $O/st.json 2>> $O/drv.err
RGS_LIB_SUFFIX=_stamps timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --stamps --no-cpu --no-rebalance > $O/st_noreb.json 2>> $O/drv.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3e5/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); e=d['events']; st=e.get('stamps')
        print(f, '%.4g'%d['value'], d['roofline']['kernel_ms_per_launch'], 'dealt', e.get('hot_arenas_dealt_to_other_workgroups'), 'flagged', e['flagged'])
        if st: print('    avg total', st['total_cycles']/1024, 'max', st['max_total_cycles'], 'ratio', st['max_total_cycles']/(st['total_cycles']/1024), 'lane', st['lane_cycles']/1024, 'wave', st['wave_cycles']/1024, 'rounds', st['rounds'], 'max lane passes', st['max_rounds'])
    except Exception as ex: print(f, 'ERR', ex)
PY
