python -m pytest tests/test_gpu_chain_fullsize.py tests/test_gpu_shims.py tests/test_gpu_init_chain.py -q -m gpu -s 2>&1 | grep -v "^$" | tail -15
python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err; tail -c 1500 gpurun_out/r06_bench_default.json | head -c 400; echo
bash tools/profile_round.sh r06 2>&1 | tail -3
PMC_ARGS="--steps 1 --warmup 0 --cpu-sample 0 --solver-cpu-sample 0 --p2-pieces 0 --no-h2d --ingest-events 0 --calib-views 0 --e2e-events 0" timeout 1200 bash tools/pmc_sq.sh r06ne > /dev/null 2>&1; ls gpurun_out/r06ne
