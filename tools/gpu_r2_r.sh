#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/fcc_fold.py 300 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2r_fcc_fold.log
