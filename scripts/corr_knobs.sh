# Sweep of the plane-generation launch knobs on the GPU box (one line per setting).
set -e
for zl in 1 0; do for chunk in 8 16 32; do for tile in "4 16" "8 8" "16 4"; do
  set -- $tile
  echo "ZL=$zl CHUNK=$chunk TZ=$1 TC=$2: $(ACCELGPU_CORR_ZL=$zl ACCELGPU_CORR_CHUNK=$chunk ACCELGPU_CORR_TZ=$1 ACCELGPU_CORR_TC=$2 timeout -k 10 100 python scripts/corr_sweep.py 200 | cut -c1-60)"
done; done; done
