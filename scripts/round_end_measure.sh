set -e
TAG=${1:-r03f}
mkdir -p gpurun_out/${TAG}
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/${TAG}/pytest_gpu.log 2>&1
tail -3 gpurun_out/${TAG}/pytest_gpu.log
timeout -k 10 400 python bench.py > gpurun_out/${TAG}/bench.json 2> gpurun_out/${TAG}/bench.err
cut -c1-1500 gpurun_out/${TAG}/bench.json
timeout -k 10 300 python scripts/pf_dense_sweep.py > gpurun_out/${TAG}/dense_sweep.txt 2>&1
tail -25 gpurun_out/${TAG}/dense_sweep.txt
timeout -k 10 300 python scripts/time_pipeline.py 128 > gpurun_out/${TAG}/pipeline_rate.txt 2>&1
tail -12 gpurun_out/${TAG}/pipeline_rate.txt
bash scripts/profile_round.sh ${TAG}
timeout -k 10 600 python scripts/range_split_measure.py cfg4 > gpurun_out/${TAG}/range_split.txt 2>&1
tail -4 gpurun_out/${TAG}/range_split.txt
