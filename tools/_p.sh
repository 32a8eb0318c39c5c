timeout 1500 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_modules.py tests/test_gpu_train_mlp.py tests/test_gpu_round2.py -x -q -m gpu 2>&1 | tail -3
