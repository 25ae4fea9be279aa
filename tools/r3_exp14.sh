cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e14; rm -rf $O; mkdir -p $O
RGS_LIB_SUFFIX=_kn timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -k "not noise and not rng" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
bash tools/ab_2v2.sh base _kn 2>&1 | tail -19
