timeout 3400 python -m pytest tests -m gpu -x -q > gpurun_out/t_gpu_all.txt 2>&1; tail -3 gpurun_out/t_gpu_all.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
