python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err; tail -c 300 gpurun_out/r06_bench_default.json; echo
bash tools/profile_round.sh r06 2>&1 | tail -2
PMC_ARGS="--steps 1 --warmup 0 --cpu-sample 0 --solver-cpu-sample 0 --p2-pieces 0 --no-h2d --ingest-events 0 --calib-views 0 --e2e-events 0" timeout 1200 bash tools/pmc_sq.sh r06ne > /dev/null 2>&1; ls gpurun_out/r06ne
timeout 900 bash tools/pmc_sq.sh r06sq > /dev/null 2>&1; ls gpurun_out/r06sq
