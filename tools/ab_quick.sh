# quick A/B of library builds: driver command, C2, C3 (no CPU legs), alternating twice
cd $GRAFT_REPO_ROOT
O=gpurun_out/abq; rm -rf $O; mkdir -p $O
for rep in 1 2; do
  for s in "$@"; do
    sfx=$s; [ "$s" = "base" ] && sfx=""
    RGS_LIB_SUFFIX=$sfx timeout -k 10 300 python bench.py --no-cpu > $O/c2_${s}_$rep.json 2> $O/err.log || true
    RGS_LIB_SUFFIX=$sfx timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu > $O/drv_${s}_$rep.json 2>> $O/err.log || true
    RGS_LIB_SUFFIX=$sfx timeout -k 10 300 python bench.py --config c3 --no-cpu > $O/c3_${s}_$rep.json 2>> $O/err.log || true
    RGS_LIB_SUFFIX=$sfx timeout -k 10 300 python bench.py --config c5 --no-streamed --no-cpu > $O/c5_${s}_$rep.json 2>> $O/err.log || true
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/abq/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, '%.4g'%d['value'])
    except Exception as e: print(f, 'ERR', e)
PY
