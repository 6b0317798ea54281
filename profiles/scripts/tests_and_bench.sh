# end-of-session check on one MI355X: smoke, the whole GPU suite, the default bench line
mkdir -p gpurun_out/final
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/final/pytest_gpu.txt 2>&1; grep "passed\|failed\|^FAILED" gpurun_out/final/pytest_gpu.txt | tail -8
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; tail -c 300 gpurun_out/final/bench.json
