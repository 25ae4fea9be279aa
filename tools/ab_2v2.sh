# A/B of library builds on the 2v2 shapes (driver command, C2, C4), alternating three times
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab2; rm -rf $O; mkdir -p $O
for rep in 1 2 3; do
  for s in "$@"; do
    sfx=$s; [ "$s" = "base" ] && sfx=""
    RGS_LIB_SUFFIX=$sfx timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu > $O/drv_${s}_$rep.json 2>> $O/err.log || true
    RGS_LIB_SUFFIX=$sfx timeout -k 10 300 python bench.py --no-cpu > $O/c2_${s}_$rep.json 2> $O/err.log || true
    RGS_LIB_SUFFIX=$sfx timeout -k 10 300 python bench.py --config c4 --no-cpu > $O/c4_${s}_$rep.json 2>> $O/err.log || true
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab2/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, '%.4g'%d['value'])
    except Exception as e: print(f, 'ERR', e)
PY
