timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -k "row_run" 2>&1 | tail -5
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -5
