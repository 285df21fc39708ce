# round-5 checks on the GPU box: the suites touched last + randomised sweeps (summaries under gpurun_out/r5b/)
mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_frontend.py tests/test_gpu_graded.py tests/test_gpu_quad.py tests/test_gpu_regression.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3
FUZZ_OLD_GUARD=1 timeout 300 python tools/fuzz_parity.py 70 501 2>&1 | grep -a "FAIL" | cut -c1-400
timeout 600 python tools/fuzz_parity.py 300 501 > gpurun_out/r5b/fuzz_parity_full.txt 2>&1; tail -1 gpurun_out/r5b/fuzz_parity_full.txt; grep -n "FAIL" gpurun_out/r5b/fuzz_parity_full.txt | head -20 | cut -c1-400
FUZZ_WIDE=1 timeout 600 python tools/fuzz_parity.py 60 502 2>&1 | tail -1
timeout 600 python tools/fuzz_parity.py 300 505 2>&1 | grep -a "FAIL\|failures" | cut -c1-400
python bench.py --config c4split --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | cut -c1-200
python bench.py --config c5 --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | cut -c1-200
