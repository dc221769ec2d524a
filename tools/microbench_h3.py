"""EXPERIMENTAL two-term fp16 (h3) GEMM vs the split-bf16 (x6) and fp32-MFMA GEMMs: error against float64, TFLOP/s."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from anyloc_amd import ops

dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(0)
SHAPES = [(300, 200, 48), (2051, 1536, 1536), (32330, 4608, 1536), (32330, 1536, 1536), (32330, 8192, 1536), (32330, 1536, 4096)]
if os.environ.get("H3_BIG"):
    SHAPES = SHAPES[2:]
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for (M, N, K) in SHAPES:
    a = torch.randn(M, K, generator=g, device=dev) * (0.5 + torch.rand(M, 1, generator=g, device=dev))
    a[:, ::97] *= 50.0
    w = torch.randn(N, K, generator=g, device=dev) * 0.02
    bias = torch.randn(N, generator=g, device=dev)
    a2, w2 = ops.split_h2(a), ops.split_h2(w)
    c = ops.gemm_nt_h3(a2, w2, M, N, K, bias)
    rows = slice(0, min(M, 512))
    ref = a[rows].double() @ w.double().t() + bias.double()
    mag = a[rows].double().abs() @ w.double().abs().t()
    eh = float(((c[rows].double() - ref).abs() / mag).max())
    et = float(((c[-64:].double() - (a[-64:].double() @ w.double().t() + bias.double())).abs() / (a[-64:].double().abs() @ w.double().abs().t())).max())
    a3, w3 = ops.split_x3(a), ops.split_x3(w)
    e6 = float(((ops.gemm_nt_x6(a3, w3, M, N, K, bias)[rows].double() - ref).abs() / mag).max())
    th = timeit(lambda: ops.gemm_nt_h3(a2, w2, M, N, K, bias))
    t6 = timeit(lambda: ops.gemm_nt_x6(a3, w3, M, N, K, bias))
    ts = timeit(lambda: ops.split_h2(a))
    fl = 2.0 * M * N * K
    print(json.dumps(dict(M=M, N=N, K=K, err_h3=eh, err_h3_tail=et, err_x6=e6, ms_h3=round(th, 4), tf_h3=round(fl / th / 1e9, 1),
                          ms_x6=round(t6, 4), tf_x6=round(fl / t6 / 1e9, 1), ms_split=round(ts, 4))), flush=True)
