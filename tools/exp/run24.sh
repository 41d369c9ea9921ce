for rep in 1 2; do
  for m in bf16 f32; do
    echo "fused $m: $(python tools/c3_step.py 5 $m 2>&1 | grep '\[c3\]' | cut -c1-330)"
    echo "plain $m: $(W2L_NET_NOFUSE=1 python tools/c3_step.py 5 $m 2>&1 | grep '\[c3\]' | cut -c1-330)"
  done
done > gpurun_out/r06_run24_c3_fuse_ab.log 2>&1
