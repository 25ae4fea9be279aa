set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab4
rm -f gpurun_out/ab4/*.json
RGS_LIB_SUFFIX=_duo timeout -k 10 300 python tools/pool_smoke.py > gpurun_out/ab4/smoke.log 2>&1 || { tail -30 gpurun_out/ab4/smoke.log; exit 1; }
cat gpurun_out/ab4/smoke.log
for s in $SUFFIXES; do
  [ "$s" = "base" ] && s=""
  RGS_LIB_SUFFIX=$s timeout -k 10 300 python bench.py --variant 4 --cpu-sample-arenas 2048 > gpurun_out/ab4/c2${s}.json 2> gpurun_out/ab4/c2${s}.err || true
  RGS_LIB_SUFFIX=$s timeout -k 10 300 python bench.py --variant 4 --steps 20 --warmup 5 --cpu-sample-arenas 2048 > gpurun_out/ab4/drv${s}.json 2> gpurun_out/ab4/drv${s}.err || true
  RGS_LIB_SUFFIX=$s timeout -k 10 300 python bench.py --config c4 --variant 4 --cpu-sample-arenas 2048 > gpurun_out/ab4/c4${s}.json 2> gpurun_out/ab4/c4${s}.err || true
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab4/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, '%.4g'%d['value'], d['parity']['bit_identical'], d['events']['flagged'])
    except Exception as e: print(f, 'ERR', e)
PY
