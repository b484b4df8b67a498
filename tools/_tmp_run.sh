timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_workloads.py tests/test_gpu_training.py -m gpu -q -k "wide_input or MEGNet or megnet or cfg4 or cfg5 or linear" 2>&1 | grep -E "^E  .*|passed|failed|Error|^FAILED" | head -12
timeout 900 python bench.py --model megnet --no-cpu-baseline --no-extras 2>&1 | tail -1 | cut -c1-210
