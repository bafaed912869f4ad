"""Host helpers on the VAE path: checkpoint I/O, seeding, device-string validation
(polygon/utils/utils.py:63-110,500-527).  RDKit-based scoring utilities of the reference stay in the reference
(host CPU, out of scope); ``canonicalize_list`` is bound lazily so that this module imports without RDKit."""
import re

import numpy as np
import torch


def load_model(model_class, model_definition, device, model_params=None, copy_to_cpu=True):
    """utils.py:63-88: build the model (default hparams unless model_params), strict load_state_dict, eval()."""
    model = model_class(**model_params).to(device) if model_params else model_class().to(device)
    if "cpu" in str(device):
        state = torch.load(model_definition, map_location="cpu", weights_only=True)
    else:
        state = torch.load(model_definition, map_location=device, weights_only=True)
    model.load_state_dict(state)
    model.eval()
    return model


def save_model(model, output_path, device=None):
    """utils.py:90-110: the 74-key state_dict, tensors on the CPU."""
    if device is None:
        try:
            device = model.device
        except Exception:
            raise RuntimeError("Model must have a `device` parametere if none is specified.")
    state = {k: v.detach().to('cpu') for k, v in model.state_dict().items()}
    torch.save(state, output_path)


def set_random_seed(seed, device):
    """utils.py:500-510."""
    np.random.seed(seed)
    torch.manual_seed(seed)
    if 'cuda' in str(device):
        torch.cuda.manual_seed(seed)


def torch_device(arg):
    """argparse type of --device (utils.py:512-527): 'cpu' or 'cuda[:N]'; TypeError otherwise or if unavailable."""
    if re.match('^(cuda(:[0-9]+)?|cpu)$', arg) is None:
        raise TypeError('Wrong device format: {}'.format(arg))
    if arg != 'cpu':
        parts = arg.split(':')
        if (not torch.cuda.is_available()) or (len(parts) > 1 and int(parts[1]) > torch.cuda.device_count()):
            raise TypeError('Wrong device: {} is not available'.format(arg))
    return arg


def canonicalize(smiles, include_stereocenters=True):
    """utils.py:447-466 (host RDKit): canonical SMILES or None for an invalid molecule."""
    try:
        from rdkit import Chem
    except ImportError as e:
        raise ImportError("canonicalize needs RDKit on the host (scoring stays on the CPU, unchanged)") from e
    mol = Chem.MolFromSmiles(smiles)
    return None if mol is None else Chem.MolToSmiles(mol, isomericSmiles=include_stereocenters)


def canonicalize_list(smiles_list, include_stereocenters=True):
    """utils.py:469-486: canonical forms of the valid molecules, order kept (duplicates are NOT removed there either)."""
    out = (canonicalize(s, include_stereocenters) for s in smiles_list)
    return [s for s in out if s is not None]
