cd $GRAFT_REPO_ROOT
python tools/bench_generate.py --batches 1 8 16 --steps 2 8 > gpurun_out/r03h_generate.txt 2>&1
python tools/bench_e2e.py > gpurun_out/r03h_e2e.json 2> gpurun_out/r03h_e2e.err
python -m pytest tests -m gpu -x -q > gpurun_out/r03h_gpu_tests.txt 2>&1; tail -3 gpurun_out/r03h_gpu_tests.txt
grep "^{'schedule" gpurun_out/r03h_generate.txt | cut -c1-150; tail -1 gpurun_out/r03h_e2e.json | cut -c1-600
