"""Run by tests/test_gpu_vlad_topk.py::test_vlad_fused_parts in a subprocess (the path switches are read once per
process): hard VLAD through the fused kernel with the environment's ANYLOC_VLAD_PARTS / ANYLOC_VLAD_TWO_PASS against the
CPU oracle, the same workspace reused across calls with different inputs (the reducing workgroup of one call has the
previous call's partial sums in its L1: the agent-scope acquire has to drop them)."""
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
    print(f"ok worst={worst:.2e}")


if __name__ == "__main__":
    sys.exit(main())
