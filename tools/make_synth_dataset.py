"""Write a tiny synthetic place-recognition set in the datasets-vg folder layout the reference's
``BaseDataset`` reads (reference ``dvgl_benchmark/datasets_ws.py:85-105,188-198``):

    <root>/<name>/images/test/database/@<utm_e>@<utm_n>@<id>@.jpg
    <root>/<name>/images/test/queries/@<utm_e>@<utm_n>@<id>@.jpg

Query q depicts place (q mod n_db); places are 100 m apart so the 25 m positive radius
(reference ``configs.py:161``) makes place q mod n_db the only positive.  ``name`` must be one of
the dataset names the reference's CLI accepts that fall through to the generic loader
(e.g. ``st_lucia``, reference ``scripts/dino_v2_vlad.py:345-347``)."""
import argparse
import os
import sys

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import synth  # noqa: E402


def write(root, name="st_lucia", n_db=24, n_qu=8, h=224, w=224, seed=42):
    db, qu, _ = synth.synthetic_places(n_db, n_qu, h, w, seed=seed)
    mean = np.array(synth.IMAGENET_MEAN, np.float32).reshape(3, 1, 1)
    std = np.array(synth.IMAGENET_STD, np.float32).reshape(3, 1, 1)
    base = os.path.join(root, name, "images", "test")
    for sub, imgs in (("database", db), ("queries", qu)):
        d = os.path.join(base, sub)
        os.makedirs(d, exist_ok=True)
        for i, im in enumerate(imgs):
            place = i if sub == "database" else i % n_db
            rgb = np.clip((im.numpy() * std + mean) * 255.0 + 0.5, 0, 255).astype(np.uint8).transpose(1, 2, 0)
            east, north = 1000.0 + 100.0 * place, 5000.0 + (3.0 if sub == "queries" else 0.0)
            Image.fromarray(rgb).save(os.path.join(d, f"@{east:.2f}@{north:.2f}@{i:05d}@.jpg"), quality=95)
    return os.path.join(root, name)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--name", default="st_lucia")
    ap.add_argument("--n-db", type=int, default=24)
    ap.add_argument("--n-qu", type=int, default=8)
    ap.add_argument("--size", type=int, nargs=2, default=[224, 224])
    a = ap.parse_args()
    print(write(a.root, a.name, a.n_db, a.n_qu, a.size[0], a.size[1]))
