set -e
mkdir -p gpurun_out
L=gpurun_out/hs_lean.log
: > $L
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_optimize.py -x -q --timeout 280 --timeout-method thread >> $L 2>&1
for nh in 4 8 16 32; do
  ACCELGPU_HS_LEAN_FROM=3 timeout -k 10 120 python scripts/hs_sweep.py $nh 200 >> $L 2>&1
  ACCELGPU_HS_LEAN_FROM=9 timeout -k 10 120 python scripts/hs_sweep.py $nh 200 >> $L 2>&1
done
ACCELGPU_HS_LEAN_FROM=3 timeout -k 10 120 python scripts/hs_sweep.py 32 600 >> $L 2>&1
ACCELGPU_HS_LEAN_FROM=9 timeout -k 10 120 python scripts/hs_sweep.py 32 600 >> $L 2>&1
tail -30 $L
