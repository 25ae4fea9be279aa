cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e11; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "device_noise or noise" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
timeout -k 10 100 python tools/noise_rate.py 4096 20 2 exact 2>&1 | tail -1
timeout -k 10 100 python tools/noise_rate.py 4096 20 2 device 2>&1 | tail -1
timeout -k 10 100 python tools/noise_rate.py 65536 100 2 device 0.01 2>&1 | tail -1
timeout -k 10 100 python tools/noise_rate.py 65536 100 2 device 0.05 2>&1 | tail -1
