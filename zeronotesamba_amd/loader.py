"""Drop-in for /root/reference/zeroNoteSamba/loader.py::load_models on MI355X.

Same arguments and return triple `(criterion, optimizer, model)`, same checkpoint names, freezing rule
(loader.py:29-34: `requires_grad = False` on `*.pretrained`) and learning-rate rules (loader.py:36-43, 55-62, 66):
frozen -> lr, fine-tune of the pretext weights -> 0.5*lr*10e-2, CLMR fine-tune -> 0.5*lr, vanilla -> lr.
The model is the HIP-backed Down_CNN / DS_CNN, the criterion the fused BCE kernel, the optimizer the fused Adam.
`state_dict` lets a caller pass already-loaded weights (the .pth files are not shipped with this repository).
"""
from typing import Dict, Optional, Tuple

import torch

from .epochs import BCELoss
from .models.models import DS_CNN, Down_CNN
from .optim import Adam

SHIFT_CKPT = "models/saved/shift_pret_cnn_16.pth"   # loader.py:25
CLMR_CKPT = "models/saved/clmr_pret_cnn_16.pth"     # loader.py:48


def load_models(_status: str, _pre: str, _lr: float, state_dict: Optional[Dict[str, torch.Tensor]] = None,
                device: str = "cuda") -> Tuple[BCELoss, torch.optim.Optimizer, torch.nn.Module]:
    """
    Function for loading loss, optimizer, and model.
    -- _status: pretrained, vanilla, clmr, or samplecnn?
    -- _pre: frozen weights
    -- _lr: learning rate
    """
    criterion = BCELoss()
    model: torch.nn.Module

    print("\n{} learning mode...".format(_status))

    if _status == "pretrained":
        model = Down_CNN().to(device)
        if state_dict is None:
            state_dict = torch.load(SHIFT_CKPT, map_location=torch.device(device))
        model.pretext.load_state_dict(state_dict)

        if _pre == "frozen":
            for param in model.pretext.anchor.pretrained.parameters():
                param.requires_grad = False
            for param in model.pretext.postve.pretrained.parameters():
                param.requires_grad = False
            optimizer = Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=_lr, betas=(0.9, 0.999))
        else:
            optimizer = Adam(model.parameters(), lr=0.5 * _lr * 10e-2, betas=(0.9, 0.999))

    elif _status == "clmr":
        model = DS_CNN().to(device)
        if state_dict is None:
            state_dict = torch.load(CLMR_CKPT, map_location=torch.device(device))
        model.load_state_dict(state_dict)

        if _pre == "frozen":
            for param in model.pretrained.parameters():
                param.requires_grad = False
            optimizer = Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=_lr, betas=(0.9, 0.999))
        else:
            optimizer = Adam(model.parameters(), lr=0.5 * _lr, betas=(0.9, 0.999))

    else:
        model = DS_CNN().to(device)
        optimizer = Adam(model.parameters(), lr=_lr, betas=(0.9, 0.999))

    return criterion, optimizer, model
