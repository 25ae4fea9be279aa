cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e15; rm -rf $O; mkdir -p $O
for k in 1000 200; do
RGS_LIB_SUFFIX=_stamps timeout -k 10 200 python bench.py --config c2 --steps $k --stamps --no-cpu > $O/st_c2_$k.json 2>> $O/err
cp gpurun_out/wave_log.npy $O/wave_log_c2_$k.npy
echo "== C2 $k steps"; python tools/wave_log_xcd.py $O/wave_log_c2_$k.npy
done
RGS_LIB_SUFFIX=_stamps timeout -k 10 200 python bench.py --steps 20 --warmup 5 --stamps --no-cpu > $O/st_drv.json 2>> $O/err
cp gpurun_out/wave_log.npy $O/wave_log_drv.npy
echo "== driver window"; python tools/wave_log_xcd.py $O/wave_log_drv.npy
