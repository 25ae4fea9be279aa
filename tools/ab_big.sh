# A/B of library builds on the large-team shapes: C3 (7v7), C5 (4v4), 6v6 and 5v5 at 65 536 arenas, alternating twice
cd $GRAFT_REPO_ROOT
O=gpurun_out/abb; rm -rf $O; mkdir -p $O
for rep in 1 2; do
  for s in "$@"; do
    sfx=$s; [ "$s" = "base" ] && sfx=""
    RGS_LIB_SUFFIX=$sfx timeout -k 10 300 python bench.py --config c3 --no-cpu > $O/c3_${s}_$rep.json 2>> $O/err.log || true
    RGS_LIB_SUFFIX=$sfx timeout -k 10 300 python bench.py --config c5 --no-streamed --no-cpu > $O/c5_${s}_$rep.json 2>> $O/err.log || true
    RGS_LIB_SUFFIX=$sfx timeout -k 10 300 python bench.py --config c3 --n-per-team 6 --steps 500 --no-cpu > $O/n6_${s}_$rep.json 2>> $O/err.log || true
    RGS_LIB_SUFFIX=$sfx timeout -k 10 300 python bench.py --config c3 --n-per-team 5 --steps 500 --no-cpu > $O/n5_${s}_$rep.json 2>> $O/err.log || true
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/abb/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, '%.4g'%d['value'], "")
    except Exception as e: print(f, 'ERR', e)
PY
