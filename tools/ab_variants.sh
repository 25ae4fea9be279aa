# A/B of stepper variants on one bench.py configuration, same box (run through gpurun):
#   tools/ab_variants.sh "<bench args>" <variant> [<variant> ...]     e.g.  tools/ab_variants.sh "--config c4" 0 2 4 5
set -e
cd $GRAFT_REPO_ROOT
ARGS=$1; shift
O=gpurun_out/abv; mkdir -p $O
for v in "$@"; do
  for rep in 1 2; do
    timeout -k 10 300 python bench.py $ARGS --variant $v --cpu-sample-arenas 1024 > $O/v${v}_$rep.json 2> $O/v${v}_$rep.err || true
    python - "$O/v${v}_$rep.json" "$v" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('variant', sys.argv[2], '%.4g'%d['value'], d['parity']['bit_identical'], d['events']['flagged'])
except Exception as e: print('variant', sys.argv[2], 'ERR', e)
PY
  done
done
