set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/rw; rm -rf $O; mkdir -p $O
for rep in 1 2; do for n in 2 3; do
  timeout -k 10 120 python tools/noise_rate.py 65536 20 $n device | sed 's/^/3 waves: /' | tee -a $O/rates.txt
  RGS_LIB_SUFFIX=_rw2 timeout -k 10 120 python tools/noise_rate.py 65536 20 $n device | sed 's/^/2 waves: /' | tee -a $O/rates.txt
  RGS_LIB_SUFFIX=_rw4 timeout -k 10 120 python tools/noise_rate.py 65536 20 $n device | sed 's/^/4 waves: /' | tee -a $O/rates.txt
done; done
