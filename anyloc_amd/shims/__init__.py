"""Minimal stand-ins for third-party modules that the reference's *scripts* import but that
are absent from this image (no network: nothing can be installed).  They are NOT product code
and are never used when the real package is importable: ``anyloc_amd.run`` registers a shim
in ``sys.modules`` only if ``importlib.util.find_spec(name)`` finds nothing.

    tyro                 dataclass CLI parser (the subset the two target scripts use)
    torchvision          transforms.{Compose,ToTensor,Normalize,CenterCrop,Resize,Lambda,...}
    natsort              natsorted
    faiss, cv2, wandb, onedrivedownloader   import-only (never called on the hot path)
"""
