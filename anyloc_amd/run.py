"""Run one of the reference's scripts UNMODIFIED on top of this implementation:

    python -m anyloc_amd.run /path/to/AnyLoc/scripts/dino_v2_vlad.py --prog.data-vg-dir ... [args]
    python -m anyloc_amd.run /path/to/AnyLoc/demo/anyloc_vlad_generate.py --in-dir ... --no-use-example

What it does, and nothing more:
  1. puts this repository first on ``sys.path`` and pre-imports its ``utilities`` module, so the
     script's ``from utilities import ...`` binds to the HIP-backed surface (a script's own directory
     precedes PYTHONPATH, so ``demo/utilities.py`` would otherwise shadow it);
  2. registers stand-ins (``anyloc_amd/shims``) ONLY for third-party modules that are not
     importable in this image (tyro, torchvision, natsort, faiss, cv2, wandb,
     onedrivedownloader) -- a real installation always wins;
  3. ``runpy.run_path``s the script as ``__main__`` with the remaining argv.
"""
import importlib
import importlib.util
import os
import re
import runpy
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _missing(name):
    try:
        return importlib.util.find_spec(name) is None
    except (ImportError, ValueError):
        return True


def _natsorted(seq, key=None):
    def nk(s):
        s = key(s) if key else s
        return [int(t) if t.isdigit() else t.lower() for t in re.split(r"(\d+)", str(s))]
    return sorted(seq, key=nk)


def _import_only(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)

    def __getattr__(attr):
        raise AttributeError(f"module {name!r} is an import-only stand-in (the real package is not installed); "
                             f"{name}.{attr} is not available")
    m.__getattr__ = __getattr__
    return m


def install_shims(verbose=True):
    installed = []
    if _missing("tyro"):
        from .shims import tyro_shim
        m = types.ModuleType("tyro")
        m.cli = tyro_shim.cli
        sys.modules["tyro"] = m
        installed.append("tyro")
    if _missing("torchvision"):
        from .shims import torchvision_shim
        sys.modules.update(torchvision_shim.build_modules())
        installed.append("torchvision")
    if _missing("natsort"):
        sys.modules["natsort"] = _import_only("natsort", natsorted=_natsorted)
        installed.append("natsort")
    if _missing("faiss"):
        faiss = _import_only("faiss")
        contrib = _import_only("faiss.contrib")
        tu = _import_only("faiss.contrib.torch_utils")
        faiss.contrib, contrib.torch_utils = contrib, tu
        sys.modules.update({"faiss": faiss, "faiss.contrib": contrib, "faiss.contrib.torch_utils": tu})
        installed.append("faiss")
    for name in ("cv2", "wandb"):
        if _missing(name):
            sys.modules[name] = _import_only(name)
            installed.append(name)
    if _missing("onedrivedownloader"):
        def download(*a, **k):
            raise RuntimeError("onedrivedownloader is not installed (no network): place the files by hand")
        sys.modules["onedrivedownloader"] = _import_only("onedrivedownloader", download=download)
        installed.append("onedrivedownloader")
    if verbose and installed:
        print(f"[anyloc_amd.run] stand-ins registered for missing modules: {', '.join(installed)}")
    return installed


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit(__doc__)
    script = os.path.abspath(argv[0])
    if not os.path.isfile(script):
        raise SystemExit(f"no such script: {script}")
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    install_shims()
    import utilities  # noqa: F401  (ours: binds `from utilities import ...` in the script)
    import torch.hub
    from . import extractor
    torch.hub.load = extractor.hub_load      # scripts that call the hub model directly (dino_v2_global_vpr.py:115)
    assert os.path.dirname(os.path.abspath(utilities.__file__)) == ROOT
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
