# rocprofv3 passes of ONE bench.py command line (run on the GPU box through gpurun):
#   tools/prof.sh <tag> [bench.py arguments]        e.g.  tools/prof.sh c2_driver --steps 20 --warmup 5
# Pass 1: kernel trace + stats (per-kernel durations).  Passes 2-4: PMC counters, each in its own run with
# --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md "rocprofv3 PMC slots").
# The program itself follows `--` (no env/bash hop).  Output: gpurun_out/prof/<tag>/, summarised by
# tools/prof_summarize.py into the entry bench.py looks up in profiles/r03/pmc.json.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; shift
O=gpurun_out/prof/$TAG; rm -rf $O; mkdir -p $O
ARGS="--no-cpu --no-streamed $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py $ARGS > $O/stats.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -- python3 bench.py $ARGS > $O/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py $ARGS > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py $ARGS > $O/pmc_write.log 2>&1
python3 tools/prof_summarize.py $TAG $ARGS
