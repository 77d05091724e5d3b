cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_tests.log 2>&1; grep -n "passed\|failed" gpurun_out/r02_gpu_tests.log | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_final.json 2> gpurun_out/r02_bench_final.err; cat gpurun_out/r02_bench_final.json | cut -c1-1800
