# A/B of two library builds on every bench configuration, same box, alternating (run through gpurun):
#   tools/ab_libs.sh <suffixA> <suffixB>      ("base" = librgs_hip.so; others = librgs_hip<suffix>.so from make SUFFIX=...)
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/abl; rm -rf $O; mkdir -p $O
for rep in 1 2; do
  for s in "$@"; do
    sfx=$s; [ "$s" = "base" ] && sfx=""
    RGS_LIB_SUFFIX=$sfx timeout -k 10 300 python bench.py --cpu-sample-arenas 1024 --cpu-seconds 0.5 > $O/c2_${s}_$rep.json 2> $O/err.log || true
    RGS_LIB_SUFFIX=$sfx timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-sample-arenas 1024 --cpu-seconds 0.5 > $O/drv_${s}_$rep.json 2>> $O/err.log || true
    RGS_LIB_SUFFIX=$sfx timeout -k 10 300 python bench.py --config c3 --cpu-sample-arenas 256 --cpu-seconds 0.5 > $O/c3_${s}_$rep.json 2>> $O/err.log || true
    RGS_LIB_SUFFIX=$sfx timeout -k 10 300 python bench.py --config c4 --cpu-sample-arenas 1024 --cpu-seconds 0.5 > $O/c4_${s}_$rep.json 2>> $O/err.log || true
    RGS_LIB_SUFFIX=$sfx timeout -k 10 300 python bench.py --config c5 --no-streamed --cpu-sample-arenas 512 --cpu-seconds 0.5 > $O/c5_${s}_$rep.json 2>> $O/err.log || true
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/abl/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, '%.4g'%d['value'], d['parity']['bit_identical'], "")
    except Exception as e: print(f, 'ERR', e)
PY
