#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
timeout 300 python tools/diag_config2.py > gpurun_out/r2d_diag.log 2>&1; echo "diag rc=$?"; grep "^\[" gpurun_out/r2d_diag.log | cut -c1-1500
timeout 300 python tools/conv_rs.py > gpurun_out/r2d_conv_rs.log 2>&1; echo "conv rc=$?"; grep -v amdgpu gpurun_out/r2d_conv_rs.log | tail -20
timeout 200 python tools/conv_rs.py --small > gpurun_out/r2d_conv_small.log 2>&1; grep "^small" gpurun_out/r2d_conv_small.log | head -8
