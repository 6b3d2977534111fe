set -e
for nh in 1 2 4 8; do
 for j in 0 1 2; do
  ACCELGPU_HS_J=$j timeout -k 10 120 python scripts/hs_sweep.py $nh 200
 done
 ACCELGPU_HS_LEAN_FROM=3 ACCELGPU_HS_J=0 timeout -k 10 120 python scripts/hs_sweep.py $nh 200
done
