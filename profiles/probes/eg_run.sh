timeout 400 python -m pytest tests/test_gpu_encoder_general.py tests/test_gpu_encoder.py -x -q 2>&1 | tail -3
timeout 300 python profiles/encoder_general_microbench.py 2>&1 | grep shape
