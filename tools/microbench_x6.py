"""Split-bf16 (x6) GEMM vs the fp32-MFMA GEMM: accuracy against float64 and throughput (run on the GPU box)."""
import sys, os, json, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from anyloc_amd import ops

dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(0)
res = []
SHAPES = [(300, 200, 48), (1000, 384, 384), (2048, 1536, 1536), (32330, 4608, 1536), (32330, 1536, 1536),
          (32330, 8192, 1536), (32330, 1536, 4096)]
if os.environ.get("X6_BIG"):
    SHAPES = SHAPES[3:]
for (M, N, K) in SHAPES:
    a = torch.randn(M, K, generator=g, device=dev) * (0.5 + torch.rand(M, 1, generator=g, device=dev))
    w = torch.randn(N, K, generator=g, device=dev) * 0.02
    bias = torch.randn(N, generator=g, device=dev)
    a3, w3 = ops.split_x3(a), ops.split_x3(w)
    c6 = ops.gemm_nt_x6(a3, w3, M, N, K, bias)
    c32 = ops.gemm_nt(a, w, bias) if K % 4 == 0 else None
    rows = slice(0, min(M, 512))
    ref = a[rows].double() @ w.double().t() + bias.double()
    mag = (a[rows].double().abs() @ w.double().abs().t())
    e6 = float(((c6[rows].double() - ref).abs() / mag).max())
    e32 = float(((c32[rows].double() - ref).abs() / mag).max()) if c32 is not None else None
    # tail rows too
    reft = a[-64:].double() @ w.double().t() + bias.double()
    et = float(((c6[-64:].double() - reft).abs() / (a[-64:].double().abs() @ w.double().abs().t())).max())
    def timeit(fn, n=10):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n
    t6 = timeit(lambda: ops.gemm_nt_x6(a3, w3, M, N, K, bias))
    t32 = timeit(lambda: ops.gemm_nt(a, w, bias)) if c32 is not None else None
    ts = timeit(lambda: ops.split_x3(a))
    fl = 2.0 * M * N * K
    r = dict(M=M, N=N, K=K, err_x6=e6, err_x6_tail=et, err_f32=e32, ms_x6=round(t6, 4), tf_x6=round(fl / t6 / 1e9, 1),
             ms_f32=t32 and round(t32, 4), tf_f32=t32 and round(fl / t32 / 1e9, 1), ms_split_a=round(ts, 4),
             split_gbs=round(10.0 * M * K / ts / 1e6, 1))
    print(json.dumps(r), flush=True)
    res.append(r)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/microbench_x6.json", "w"), indent=1)
