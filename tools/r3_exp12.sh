cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e12; rm -rf $O; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
bash tools/ab_libs.sh base _r3a 2>&1 | tail -24
