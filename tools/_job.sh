python -m pytest tests/test_gpu_parity.py -q -x -k "knn_batch or window or exchange" 2>&1 | tail -5
