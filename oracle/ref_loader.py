"""TEST INFRASTRUCTURE ONLY -- loader for the real reference (container-only).

Loads the four RDKit-free modules of the reference by file path from
/root/reference (SURVEY.md section 8c recipe): an empty ``rdkit`` stub and empty
shell packages ``polygon``, ``polygon.utils``, ``polygon.vae`` are placed in
``sys.modules`` so the reference's eager ``__init__`` chains (which import
RDKit) are skipped.  Nothing of the reference is copied; this file only exists
so that ``oracle/make_golden.py`` can generate golden vectors here.  It is never
imported on the GPU box (/root/reference does not exist there).
"""
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("POLYGON_REFERENCE", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REF_ROOT, "polygon", "vae", "vae_model.py"))


def _load(dotted, relpath):
    path = os.path.join(REF_ROOT, relpath)
    spec = importlib.util.spec_from_file_location(dotted, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[dotted] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    """Returns a namespace with the reference's VAE, vocab, annealers, trainer."""
    if not reference_available():
        raise RuntimeError("reference not mounted at %s" % REF_ROOT)
    saved = {k: sys.modules.get(k) for k in list(sys.modules)
             if k == "rdkit" or k == "polygon" or k.startswith("polygon.")}
    for k in saved:
        del sys.modules[k]
    try:
        rdkit = types.ModuleType("rdkit")
        rdkit.rdBase = types.ModuleType("rdkit.rdBase")
        sys.modules["rdkit"] = rdkit
        sys.modules["rdkit.rdBase"] = rdkit.rdBase
        for pkg in ("polygon", "polygon.utils", "polygon.vae"):
            m = types.ModuleType(pkg)
            m.__path__ = []
            sys.modules[pkg] = m
        ns = types.SimpleNamespace()
        ns.moses_utils = _load("polygon.utils.moses_utils", "polygon/utils/moses_utils.py")
        ns.smiles_char_dict = _load("polygon.utils.smiles_char_dict", "polygon/utils/smiles_char_dict.py")
        ns.vae_model = _load("polygon.vae.vae_model", "polygon/vae/vae_model.py")
        ns.vae_misc = _load("polygon.vae.vae_misc", "polygon/vae/vae_misc.py")
        # vae_trainer needs polygon.utils.utils.save_model; the real module imports RDKit at
        # top level, so provide a stub module exposing only a torch.save based save_model.
        uu = types.ModuleType("polygon.utils.utils")
        import torch

        def save_model(model, output_path, device=None):
            dev = model.device if device is None else device
            model.to("cpu")
            torch.save(model.state_dict(), output_path)
            model.to(dev)
        uu.save_model = save_model
        sys.modules["polygon.utils.utils"] = uu
        ns.vae_trainer = _load("polygon.vae.vae_trainer", "polygon/vae/vae_trainer.py")
        ns.VAE = ns.vae_model.VAE
        ns.VAETrainer = ns.vae_trainer.VAETrainer
        return ns
    finally:
        for k in [k for k in list(sys.modules)
                  if k == "rdkit" or k.startswith("rdkit.") or k == "polygon" or k.startswith("polygon.")]:
            del sys.modules[k]
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
