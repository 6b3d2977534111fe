set -e
for corr in 1 0; do for ser in 0 16 32; do for chunk in 8 16 32; do
  echo "CORR=$corr SER=$ser CHUNK=$chunk: $(ACCELGPU_CORR=$corr ACCELGPU_CORR_SER=$ser ACCELGPU_CORR_CHUNK=$chunk timeout -k 10 100 python scripts/corr_sweep.py 200 | cut -c1-60)"
  [ $corr = 0 ] && [ $ser != 0 ] && break
done; [ $corr = 0 ] && break; done; done
