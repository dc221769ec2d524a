"""Run by tests/test_gpu_vlad_topk.py::test_vlad_fused_parts in a subprocess (a fresh process,
configured through ANYLOC_OPTIONS): hard VLAD through the fused kernel with the options vlad_parts / vlad_two_pass /
vlad_fused_v against the CPU oracle, the same workspace reused across calls with different inputs (the reducing
workgroup of one call has the previous call's partial sums in its L1: the agent-scope acquire has to drop them); and one
k-means step (option kmeans_fused_v) on inputs chosen to hit the close-call logic of the fp16-screening kernel:
isotropic rows (top-2 gaps of ~1e-3), duplicated centres (exact ties), zero / tiny / huge rows."""
import sys

import torch

from anyloc_amd import ops, synth
from oracle import vlad_ref

DEV = "cuda"


def l2rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def main():
    worst = 0.0
    for K, D, N, n_img in ((32, 1536, 529, 5), (8, 384, 256, 3), (17, 768, 100, 7)):
        centers = 0.7 * synth.clustered_tokens(1, K, D, n_modes=K, seed=K)[0]
        first = None
        for rep in range(4):
            x = synth.clustered_tokens(n_img, N, D, n_modes=K, seed=100 * rep + N)
            imgs = [x[i, :N - 13 * i] for i in range(n_img)]                  # ragged, same total every repetition
            imgs[1] = imgs[1][:0] if rep == 2 else imgs[1]
            out, lab = ops.vlad([t.to(DEV) for t in imgs], centers.to(DEV), return_labels=True)
            off = 0
            for i, t in enumerate(imgs):
                lab_i = lab[off:off + len(t)].cpu()
                off += len(t)
                ref = vlad_ref.vlad_hard(t, centers, labels=lab_i)[0] if len(t) else torch.zeros(K * D)
                if len(t):
                    agree = float((lab_i == vlad_ref.hard_labels(t, centers)).float().mean())
                    assert agree > 0.995, (K, D, N, rep, i, agree)
                    err = l2rel(out[i], ref)
                    worst = max(worst, err)
                    assert err < 1e-5, (K, D, N, rep, i, err)
                else:
                    assert float(out[i].abs().max()) == 0.0
            if rep == 0:
                first = (imgs, out.clone())
        again = ops.vlad([t.to(DEV) for t in first[0]], centers.to(DEV))
        assert torch.equal(again, first[1]), "not reproducible run to run"
    kmeans_close_calls()
    print(f"ok worst={worst:.2e}")


def kmeans_close_calls():
    for D, K, n in ((1536, 32, 4111), (768, 20, 1500), (384, 16, 2050), (1024, 32, 999)):
        g = torch.Generator().manual_seed(D + K)
        x = torch.randn(n, D, generator=g) * (0.2 + 3.0 * torch.rand(n, 1, generator=g))
        c = torch.randn(K, D, generator=g)
        c[K - 1] = c[1]                                   # an exact tie: the lower index has to win
        x[5] = 0.0                                        # every score equal
        x[6] *= 1e-7                                      # below the fp16 range of the screening pass
        x[7] *= 3e4                                       # beyond it
        x[8] = c[3] + 1e-4 * torch.randn(D, generator=g)  # a clear winner
        x[9] *= 1e-25 / float(x[9].abs().max())           # non-zero, but every square underflows: NOT an all-zero row
        for mode in ("cosine", "euclidean"):
            sums, counts, lab = ops.kmeans_step(x.to(DEV), c.to(DEV), mode, True)
            lab = lab.cpu()
            xd, cd = x.double(), c.double()
            if mode == "cosine":
                sc = torch.nn.functional.normalize(xd) @ torch.nn.functional.normalize(cd).T
                scale = torch.ones(n, dtype=torch.float64)
            else:
                sc = 2 * xd @ cd.T - (cd * cd).sum(1)[None]
                scale = xd.norm(dim=1) * cd.norm(dim=1).max() + (cd * cd).sum(1).max()
            sc[:, K - 1] = sc[:, 1]                      # (the GEMM may round the two identical columns differently)
            ref = sc.max(dim=1)[1]
            bad = (lab != ref).nonzero().flatten()
            for i in bad.tolist():
                gap = float(sc[i, ref[i]] - sc[i, lab[i]])
                assert gap <= 2e-6 * float(scale[i]) + 1e-300, (D, K, mode, i, gap, float(scale[i]))
                assert not (gap == 0.0 and int(lab[i]) > int(ref[i])), (D, K, mode, i, "tie must go to the lower index")
            assert int(lab[5]) == 0 or mode == "euclidean"
            assert int(lab[8]) == 3
            if mode == "cosine":                          # the tiny row still orders its (tiny) scores: exact resolution, not the zero-row rule
                assert int(lab[9]) == int(ref[9]), (D, K, int(lab[9]), int(ref[9]))
            assert len(bad) <= 8, (D, K, mode, len(bad))
            onehot = (lab[None] == torch.arange(K)[:, None]).double()
            assert torch.equal(counts.cpu(), onehot.sum(-1).float())
            ref_sums = onehot @ xd
            err = float((sums.cpu().double() - ref_sums).norm() / ref_sums.norm())
            assert err < 1e-6, (D, K, mode, err)


if __name__ == "__main__":
    sys.exit(main())
