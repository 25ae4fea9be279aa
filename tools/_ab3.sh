set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab3
rm -f gpurun_out/ab3/*.json
timeout -k 10 300 python tools/pool_smoke.py > gpurun_out/ab3/smoke.log 2>&1 || { tail -30 gpurun_out/ab3/smoke.log; exit 1; }
cat gpurun_out/ab3/smoke.log
for s in _sync "" _b2 _b8 _b99 _b8a2; do
  RGS_LIB_SUFFIX=$s timeout -k 10 300 python bench.py --variant 4 --cpu-sample-arenas 2048 > gpurun_out/ab3/c2${s}.json 2> gpurun_out/ab3/c2${s}.err || true
  RGS_LIB_SUFFIX=$s timeout -k 10 300 python bench.py --variant 4 --steps 20 --warmup 5 --cpu-sample-arenas 2048 > gpurun_out/ab3/drv${s}.json 2> gpurun_out/ab3/drv${s}.err || true
  RGS_LIB_SUFFIX=$s timeout -k 10 300 python bench.py --config c4 --variant 4 --arenas 65536 --cpu-sample-arenas 2048 > gpurun_out/ab3/c4${s}.json 2> gpurun_out/ab3/c4${s}.err || true
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab3/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, '%.4g'%d['value'], d['parity']['bit_identical'], d['events']['flagged'])
    except Exception as e: print(f, 'ERR', e)
PY
