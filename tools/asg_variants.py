"""ASG bench leg under probe-library variants of the backward scans (W2L_FCC_BWD3, W2L_FAC_BWD32), with a parity check of
the C-ABI path's gradients against the product library's"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import _lib
if os.environ.get("W2L_FCC_BWD3") or os.environ.get("W2L_FAC_BWD32"): _lib.use_probe().__enter__()
import bench
r = bench.asg_criterion_ms(torch.device("cuda:0"))
print(json.dumps({k: r[k] for k in ("fwd_ms", "fwd_bwd_ms", "fcc_fwd_ms", "fac_fwd_ms")}), os.environ.get("W2L_FCC_BWD3"), os.environ.get("W2L_FAC_BWD32"))
