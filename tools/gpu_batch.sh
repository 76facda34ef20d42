python -m pytest tests/test_gpu_oracle_chain.py -q -m gpu -s 2>&1 | grep "\[chain\] grid\|passed\|failed"
python -m pytest tests/test_gpu_grid.py -q -m gpu 2>&1 | tail -3
python bench.py --steps 3 --warmup 1 --cpu-sample 0 --solver-iters 0 --no-h2d --calib-views 0 --ingest-events 0 --e2e-events 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d['policy_p2'], indent=0)[:3000])"
