set -e
for lds in 0 20000 26000 40000 53000; do
 for j in 0 1 2; do
  echo "HS_LDS=$lds J=$j"
  ACCELGPU_HS_LDS=$lds ACCELGPU_HS_J=$j timeout -k 10 120 python scripts/hs_sweep.py 16 200
 done
done
