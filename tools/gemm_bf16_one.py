"""one bf16-operand GEMM shape through w2l_gemm_bf16, timed with events (TF/s); used under rocprofv3 --pmc by tools/pmc.sh
usage: gemm_bf16_one.py M N K [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import ops, _lib
if os.environ.get("W2L_GEMM_H256") is not None or os.environ.get("W2L_GEMM_KSPLIT") is not None: _lib.use_probe().__enter__()   # the probe build honours the variant switch
M, N, K = [int(v) for v in sys.argv[1:4]]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
a = torch.randn(M, K, device="cuda"); b = torch.randn(N, K, device="cuda")
A, _ = ops.bf16_convert(a); B, _ = ops.bf16_convert(b)
out = torch.empty(M, N, device="cuda")
for _ in range(3): ops.gemm_bf16(A, B, K, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): ops.gemm_bf16(A, B, K, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print("gemm_bf16 [H256=%s] M=%d N=%d K=%d: %.1f us  %.1f TF/s" % (os.environ.get("W2L_GEMM_H256", "auto"), M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9))
