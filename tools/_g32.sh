for i in 1 2 3 4 5 6; do
  timeout 600 python -X faulthandler -m pytest tests/test_cpp_host.py tests/test_gpu_host_path.py -x -q -m gpu > gpurun_out/t_crash_$i.txt 2>&1; echo "run $i rc=$?"
done
grep -l "Fatal\|Segmentation\|dumped\|Aborted" gpurun_out/t_crash_*.txt
