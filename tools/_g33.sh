for i in $(seq 1 16); do
  timeout 600 python -m pytest tests/test_cpp_host.py tests/test_gpu_host_path.py -x -q -m gpu > gpurun_out/t_rep_$i.txt 2>&1; rc=$?; echo "run $i rc=$rc"
  if [ $rc -ne 0 ]; then cp gpurun_out/t_rep_$i.txt gpurun_out/t_rep_FAILED_$i.txt; fi
done
ls gpurun_out/t_rep_FAILED_* 2>/dev/null
