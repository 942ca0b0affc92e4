#!/bin/bash
set -x
mkdir -p gpurun_out/r04
python scripts/r04_rank_step_lab.py 8 40 2 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/lab_plain.txt; cat gpurun_out/r04/lab_plain.txt
LAB_MAIN_FIRST=1 python scripts/r04_rank_step_lab.py 8 40 2 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/lab_main_first.txt; cat gpurun_out/r04/lab_main_first.txt
LAB_MAIN_FIRST=close python scripts/r04_rank_step_lab.py 8 40 2 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/lab_main_first_closed.txt; cat gpurun_out/r04/lab_main_first_closed.txt
