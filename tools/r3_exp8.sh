cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e8; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "dealt_evenly or pooled_kernels" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python tools/heat_probe2.py 2>/dev/null | tail -7
for rep in 1 2 3; do
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > $O/drv_$rep.json 2> $O/drv.err
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rebalance > $O/drv_noreb_$rep.json 2>> $O/drv.err
done
RGS_LIB_SUFFIX=_stamps timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --stamps --no-cpu > $O/st.json 2>> $O/drv.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3e8/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); e=d['events']; st=e.get('stamps')
        print(f, '%.4g'%d['value'], d['roofline']['kernel_ms_per_launch'], 'dealt', e.get('hot_arenas_dealt_to_other_workgroups'), 'flagged', e['flagged'])
        if st: print('    avg total', st['total_cycles']/1024, 'max', st['max_total_cycles'], 'ratio', st['max_total_cycles']/(st['total_cycles']/1024), json.dumps(e.get('wave_log')))
    except Exception as ex: print(f, 'ERR', ex)
PY
