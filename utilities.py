"""Drop-in ``utilities`` module: the reference's class surface for the
AnyLoc-VLAD-DINOv2 hot path (reference ``utilities.py`` and its distilled copy
``demo/utilities.py``), backed by hand-written HIP kernels for MI355X.

Put this directory first on ``sys.path`` (or use ``python -m anyloc_amd.run
<reference script> ...``) and ``scripts/dino_v2_vlad.py`` /
``demo/anyloc_vlad_generate.py`` import it in place of their own copy:

    DinoV2ExtractFeatures   anyloc_amd/extractor.py   (reference utilities.py:216-288)
    VLAD                    anyloc_amd/vlad.py        (reference utilities.py:624-1008)
    get_top_k_recall        anyloc_amd/retrieval.py   (reference utilities.py:390-469)

The remaining names are the thin host-side helpers the two target scripts and
the dataset loaders import from ``utilities`` (``seed_everything``,
``reduce_pca``, ``to_np``, ``CustomDataset``, ``od_down_links``).  As in the
reference, importing this module seeds every RNG with 42 and prints one line
(reference ``utilities.py:1011``).
"""
import os
import random
from typing import List, Tuple, Union

import numpy as np
import torch

from anyloc_amd.extractor import DinoV2ExtractFeatures, _DINO_FACETS, _DINO_V2_MODELS  # noqa: F401
from anyloc_amd.kmeans import KMeans  # noqa: F401
from anyloc_amd.retrieval import get_top_k_recall  # noqa: F401
from anyloc_amd.vlad import VLAD  # noqa: F401

# download links the demo imports (reference demo/utilities.py:17-24)
od_down_links = {
    "cache": "https://iiitaphyd-my.sharepoint.com/:u:/g/personal/avneesh_mishra_research_iiit_ac_in/"
             "EW-ZqUeWWexNhbLEQvsCk2wBeucxNlhEpsfeUHHOreyLag",
    "test_imgs": "https://www.robots.ox.ac.uk/~mobile/IJRR_2008_Dataset/Data/CityCentre/Images.zip",
    "test_imgs_od": "https://iiitaphyd-my.sharepoint.com/:u:/g/personal/avneesh_mishra_research_iiit_ac_in/"
                    "EUnym1SWsrNIuOvwAdwMLgMBBxt3rgoy9zi98LanjA8wmA?e=4bNLUo",
}


class CustomDataset:
    """Parent of the reference's custom dataset loaders (reference ``utilities.py:25-74``):
    subclasses set ``database_num``, ``queries_num``, ``soft_positives_per_query`` and
    ``images_paths``."""

    def __init__(self) -> None:
        self.database_num = None
        self.queries_num = None
        self.soft_positives_per_query = None

    def get_image_paths(self):
        if hasattr(self, "images_paths"):
            return self.images_paths
        raise NotImplementedError("Not handled!")

    def get_positives(self):
        if hasattr(self, "soft_positives_per_query"):
            return self.soft_positives_per_query
        raise NotImplementedError("Not handled!")

    def get_image_relpaths(self, i: Union[int, List[int]]) -> Union[List[str], str]:
        """Last ``_imgs_level`` (default 2) path components of image(s) ``i``."""
        single = type(i) == int
        wanted = [i] if single else i
        paths = self.get_image_paths()
        depth = getattr(self, "_imgs_level", 2)
        rel = ["/".join(paths[k].split("/")[-depth:]) for k in wanted]
        return rel[0] if single else rel

    def __getitem__(self, index):
        raise NotImplementedError("Not created!")

    def __len__(self):
        if hasattr(self, "images_paths"):
            return len(self.get_image_paths())
        raise NotImplementedError("Not handled!")


def to_np(x, ret_type=float) -> np.ndarray:
    """``x`` (tensor or array-like) as a NumPy array of dtype ``ret_type``
    (reference ``utilities.py:79-95``)."""
    if type(x) == torch.Tensor:
        arr = x.detach().cpu().numpy()
    else:
        arr = np.array(x)
    return arr.astype(ret_type)


def seed_everything(seed=42):
    """Seed python / NumPy / torch RNGs and request deterministic cuDNN-MIOpen
    behaviour (reference ``utilities.py:505-519``)."""
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False
    print(f"Seed set to: {seed} (type: {type(seed)})")


def reduce_pca(train_descs: np.ndarray, test_descs: np.ndarray, lower_dim: int,
               low_factor: float = 0.0, fallback: int = 256, svd_solver: str = 'full',
               whitening: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """PCA dimensionality reduction fitted on ``train_descs`` and applied to both sets
    (reference ``utilities.py:522-586``, reached with ``--pca-dim-reduce``).  ``low_factor`` > 0
    mixes in that fraction of the lowest-eigenvalue basis vectors; with too few samples both sets
    are first projected jointly to ``fallback`` dimensions.  Runs on the device
    (``anyloc_amd.pca``: Gram / scatter matrix and projections on the fp32 MFMA GEMM)."""
    from anyloc_amd import pca
    return pca.reduce_pca(train_descs, test_descs, lower_dim, low_factor, fallback, svd_solver, whitening)


seed_everything()
