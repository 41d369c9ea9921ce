"""Does a hipGraph of the 2T launches of the N = 9998 ASG forward recursion shorten the step?  (torch.cuda.graph capture of
the criterion call; replay against direct launches)   python tools/fcc_graph.py [T]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd.criterion import CriterionScaleMode, FullConnectionCriterion

T = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B, N = 32, 9998
g = torch.Generator(device="cpu").manual_seed(5)
x = torch.randn(B, T, N, generator=g).cuda()
tgt = torch.zeros(B, 4, dtype=torch.int32).cuda()
crit = FullConnectionCriterion(N, CriterionScaleMode.TARGET_SZ_SQRT).cuda()
crit.transitions.data = (torch.randn(N, N, generator=g) * 0.1 + 4 * torch.eye(N)).cuda()
by = (4.0 * N * N + 8.0 * B * N) * (T - 1)


def timed(f, reps=3):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts)


with torch.no_grad():
    direct = timed(lambda: crit(x, tgt))
    want = crit(x, tgt).clone()
    print(f"direct launches: {direct * 1e3:8.2f} ms = {direct * 1e6 / (T - 1):6.2f} us per step ({by / direct / 8e12:.3f} of 8 TB/s)", flush=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        crit(x, tgt)
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    t0 = time.perf_counter()
    with torch.cuda.graph(gr):
        out = crit(x, tgt)
    torch.cuda.synchronize()
    print(f"capture + instantiate: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
    rep = timed(lambda: gr.replay())
    print(f"graph replay:    {rep * 1e3:8.2f} ms = {rep * 1e6 / (T - 1):6.2f} us per step ({by / rep / 8e12:.3f} of 8 TB/s); same loss: {torch.equal(out, want)}", flush=True)
