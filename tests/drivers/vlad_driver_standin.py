"""TEST INFRASTRUCTURE -- a stand-in for the reference's ``scripts/dino_v2_vlad.py`` that can travel to a box without the
reference tree.  It is NOT a copy of that script: it restates, with this repository's own tiny dataset reader, the sequence
of calls the reference driver makes on the ``utilities`` surface (reference scripts/dino_v2_vlad.py):

    :157-161   VLAD(num_clusters, None, vlad_mode=..., cache_dir=...), DinoV2ExtractFeatures(model, layer, facet, device=...)
    :164-188   per image: ToTensor / Normalize (torchvision), ``.to(device)``, CenterCrop to multiples of 14, ``dino(img[None])``,
               ``.cpu()``, ``torch.cat``
    :195-215   vocabulary: ``vlad.fit(normalise(all database descriptors))`` unless the cache holds one
    :225-260   ``vlad.generate_multi(descriptors, image names)`` for database and queries (cache ids -> ``_r/_l`` files)
    :372-376   ``get_top_k_recall(top_k_vals, db_vlads, qu_vlads, positives)``; prints ``R@k`` lines

so that ``python -m anyloc_amd.run tests/drivers/vlad_driver_standin.py ...`` exercises, in a fresh interpreter on the GPU
box, what running the reference script exercises: the module shims (tyro, torchvision), ``from utilities import ...``, CPU
tensors in and out of every call, the cache protocol.  Images come from the datasets-vg folder layout written by
tools/make_synth_dataset.py (database / queries folders, ``@east@north@id@.jpg`` names; positives within 25 m)."""
import glob
import os
import time
from dataclasses import dataclass, field
from typing import List, Literal, Optional

import numpy as np
import torch
import tyro
from PIL import Image
from torch.nn import functional as F
from torchvision import transforms as T

from utilities import VLAD, DinoV2ExtractFeatures, get_top_k_recall, seed_everything


@dataclass
class Args:
    data_dir: str = "./data"
    dataset: str = "st_lucia"
    cache_dir: Optional[str] = None
    model_type: Literal["dinov2_vits14", "dinov2_vitb14", "dinov2_vitl14", "dinov2_vitg14"] = "dinov2_vits14"
    desc_layer: int = 9
    desc_facet: Literal["query", "key", "value", "token"] = "value"
    num_clusters: int = 8
    vlad_assignment: Literal["hard", "soft"] = "hard"
    top_k_vals: List[int] = field(default_factory=lambda: [1, 2, 3])
    use_gpu: bool = True


def read_split(root, split):
    paths = sorted(glob.glob(os.path.join(root, "images", "test", split, "*.jpg")))
    utm = np.array([[float(v) for v in os.path.basename(p).split("@")[1:3]] for p in paths])
    return paths, utm


def main(a: Args):
    seed_everything(42)
    device = torch.device("cuda" if a.use_gpu and torch.cuda.is_available() else "cpu")
    root = os.path.join(a.data_dir, a.dataset)
    db_paths, db_utm = read_split(root, "database")
    qu_paths, qu_utm = read_split(root, "queries")
    assert db_paths and qu_paths, "empty dataset"
    positives = np.empty(len(qu_paths), dtype=object)
    for i, q in enumerate(qu_utm):
        positives[i] = np.nonzero(np.linalg.norm(db_utm - q[None], axis=1) <= 25.0)[0]
    to_input = T.Compose([T.ToTensor(), T.Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])])
    vlad = VLAD(a.num_clusters, None, vlad_mode=a.vlad_assignment, cache_dir=a.cache_dir)
    dino = DinoV2ExtractFeatures(a.model_type, a.desc_layer, a.desc_facet, device=device)

    def extract(paths):
        descs = []
        for p in paths:
            img = to_input(Image.open(p).convert("RGB")).to(device)
            c, h, w = img.shape
            img_in = T.CenterCrop(((h // 14) * 14, (w // 14) * 14))(img)[None, ...]
            descs.append(dino(img_in).cpu())
        return torch.cat(descs, dim=0)

    names = lambda paths: [os.path.join(a.dataset, os.path.splitext(os.path.basename(p))[0]) for p in paths]
    t0 = time.time()
    full_db = extract(db_paths)
    print(f"Full database descriptor shape: {full_db.shape}")
    if vlad.can_use_cache_vlad():
        vlad.fit(None)
    else:
        vlad.fit(F.normalize(full_db.reshape(-1, full_db.shape[-1]), dim=1))
    db_vlads = vlad.generate_multi(full_db, names(db_paths) if a.cache_dir else None)
    print(f"Database VLADs shape: {db_vlads.shape}")
    full_qu = extract(qu_paths)
    qu_vlads = vlad.generate_multi(full_qu, names(qu_paths) if a.cache_dir else None)
    print(f"Query VLADs shape: {qu_vlads.shape}")
    dists, indices, recalls = get_top_k_recall(a.top_k_vals, db_vlads, qu_vlads, positives)
    for k in a.top_k_vals:
        print(f"R@{k}: {recalls[k]:.4f}")
    print(f"top-1 indices: {indices[:, 0].tolist()}")
    print(f"device of the results: {db_vlads.device} {indices.device}; {time.time() - t0:.2f} s")


if __name__ == "__main__":
    main(tyro.cli(Args))
