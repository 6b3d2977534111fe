set -e
mkdir -p gpurun_out/r03f
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03f/pytest_gpu.log 2>&1
tail -3 gpurun_out/r03f/pytest_gpu.log
timeout -k 10 400 python bench.py > gpurun_out/r03f/bench.json 2> gpurun_out/r03f/bench.err
cut -c1-1500 gpurun_out/r03f/bench.json
timeout -k 10 300 python scripts/pf_dense_sweep.py > gpurun_out/r03f/dense_sweep.txt 2>&1
tail -25 gpurun_out/r03f/dense_sweep.txt
timeout -k 10 300 python scripts/time_pipeline.py 128 > gpurun_out/r03f/pipeline_rate.txt 2>&1
tail -12 gpurun_out/r03f/pipeline_rate.txt
bash scripts/profile_round.sh r03f
timeout -k 10 600 python scripts/range_split_measure.py cfg4 > gpurun_out/r03f/range_split.txt 2>&1
tail -4 gpurun_out/r03f/range_split.txt
