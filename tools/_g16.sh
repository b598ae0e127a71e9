bash tools/pcie_inclusive.sh r4 > /dev/null 2>&1; cat gpurun_out/r4_pcie_inclusive.txt
python tools/msm_dev_finish.py 2>&1 | grep -v amdgpu
