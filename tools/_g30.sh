timeout 1200 python -m pytest tests/test_cpp_host.py tests/test_gpu_host_path.py -x -q -m gpu 2>&1 | tail -3
