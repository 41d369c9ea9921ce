"""ASG at the config-4 criterion shape through w2l_asg_forward / w2l_asg_backward (mode `one`) or through the composed calls
(mode `comp`): a few forward + backward iterations for a rocprofv3 kernel trace.
   rocprofv3 --kernel-trace -d DIR -o NAME -- python tools/asg_onecall_timeline.py one ; python tools/asg_onecall_timeline.py dump DB"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "dump":
    import sqlite3
    db = sqlite3.connect(sys.argv[2])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = list(db.execute(f"select name, start, end, {q} from kernels order by start"))
    # the last iteration: from the last fac_rows_k on
    i0 = max(i for i, r in enumerate(rows) if "fac_rows_k" in r[0])
    i0 = max(0, i0 - 2)
    t0 = rows[i0][1]
    for n, s, e, qid in rows[i0:]:
        print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{qid}  {n[:70]}")
    sys.exit(0)
import torch
import bench
from wav2letter_amd import _lib, CriterionScaleMode
mode_sel = sys.argv[1]
dev = torch.device("cuda:0")
B, T, N, L = 64, 2000, 30, 300
g = torch.Generator(device="cpu").manual_seed(4)
x = torch.randn(B, T, N, generator=g).to(dev)
tgt = torch.full((B, L), -1, dtype=torch.int32)
for b in range(B):
    l = int(torch.randint(60, L + 1, (1,), generator=g))
    y = torch.randint(0, 28, (l,), generator=g, dtype=torch.int32)
    for i in range(1, l):
        if y[i] == y[i - 1]:
            y[i] = (y[i] + 1) % 28
    tgt[b, :l] = y
tgt = tgt.to(dev)
Lb = _lib.lib()
trans = (torch.eye(N) * 4.0 + 0.1 * torch.randn(N, N, generator=g)).to(dev).contiguous()
mode = int(CriterionScaleMode.TARGET_SZ_SQRT)
loss = torch.empty(B, device=dev); loss2 = torch.empty(B, device=dev); gl = torch.ones(B, device=dev)
dx = torch.empty_like(x); dx2 = torch.empty_like(x); dt = torch.empty(N, N, device=dev); dt2 = torch.empty(N, N, device=dev)
ts = torch.empty(B, dtype=torch.int32, device=dev)
wf = torch.empty(Lb.w2l_fcc_workspace_size(B, T, N), dtype=torch.uint8, device=dev)
wa = torch.empty(Lb.w2l_fac_workspace_size(B, T, N, L), dtype=torch.uint8, device=dev)
wasg = torch.empty(Lb.w2l_asg_workspace_size(B, T, N, L), dtype=torch.uint8, device=dev)
side = torch.cuda.Stream(device=dev)
ck = _lib.check
def one():
    s = torch.cuda.current_stream(dev).cuda_stream
    ck(Lb.w2l_asg_forward(B, T, N, L, mode, x.data_ptr(), tgt.data_ptr(), trans.data_ptr(), loss.data_ptr(), wasg.data_ptr(), s))
    ck(Lb.w2l_asg_backward(B, T, N, L, tgt.data_ptr(), trans.data_ptr(), gl.data_ptr(), dx.data_ptr(), dt.data_ptr(), wasg.data_ptr(), s))
def comp():
    cur = torch.cuda.current_stream(dev)
    ck(Lb.w2l_batch_target_size(B, L, T, tgt.data_ptr(), ts.data_ptr(), cur.cuda_stream))
    side.wait_stream(cur)
    ck(Lb.w2l_fcc_forward(B, T, N, mode, x.data_ptr(), ts.data_ptr(), trans.data_ptr(), loss.data_ptr(), wf.data_ptr(), side.cuda_stream))
    ck(Lb.w2l_fac_forward(B, T, N, L, mode, x.data_ptr(), tgt.data_ptr(), ts.data_ptr(), trans.data_ptr(), loss2.data_ptr(), wa.data_ptr(), cur.cuda_stream))
    cur.wait_stream(side)
    ck(Lb.w2l_axpy(loss.data_ptr(), loss2.data_ptr(), B, -1.0, cur.cuda_stream))
    side.wait_stream(cur)
    ck(Lb.w2l_fcc_backward(B, T, N, trans.data_ptr(), gl.data_ptr(), dx.data_ptr(), dt.data_ptr(), wf.data_ptr(), side.cuda_stream))
    ck(Lb.w2l_fac_backward(B, T, N, L, tgt.data_ptr(), ts.data_ptr(), gl.data_ptr(), dx2.data_ptr(), dt2.data_ptr(), wa.data_ptr(), cur.cuda_stream))
    cur.wait_stream(side)
    ck(Lb.w2l_axpy(dx.data_ptr(), dx2.data_ptr(), B * T * N, -1.0, cur.cuda_stream))
    ck(Lb.w2l_axpy(dt.data_ptr(), dt2.data_ptr(), N * N, -1.0, cur.cuda_stream))
f = one if mode_sel == "one" else comp
for _ in range(6):
    f()
torch.cuda.synchronize()
