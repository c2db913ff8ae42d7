"""`load_models(_status, _pre, _lr)` for the MI355X kernels: the downstream model, its loss and its optimizer.

Mirrors what /root/reference/zeroNoteSamba/loader.py:8-69 decides -- which network, which checkpoint, which sub-modules
are frozen, which learning rate -- as a table (`RULES`) instead of a chain of branches; the objects it returns are this
package's own: the HIP-backed `Down_CNN` / `DS_CNN`, the fused BCE kernel (`epochs.BCELoss`) and the fused multi-tensor
`optim.Adam`.  The reference loads the checkpoints from fixed paths relative to the working directory; the .pth files do not
ship with either repository, so a caller may hand the weights in (`state_dict=`).
"""
from typing import Callable, Dict, NamedTuple, Optional, Sequence, Tuple

import torch

from .epochs import BCELoss
from .models.models import DS_CNN, Down_CNN
from .optim import Adam

SHIFT_CKPT = "models/saved/shift_pret_cnn_16.pth"   # loader.py:25
CLMR_CKPT = "models/saved/clmr_pret_cnn_16.pth"     # loader.py:48
BETAS = (0.9, 0.999)                                # loader.py:39,43,58,62,66


class _Rule(NamedTuple):
    build: Callable[[], torch.nn.Module]            # the network
    checkpoint: Optional[str]                       # file the reference reads its weights from (None: random init)
    load_into: Callable[[torch.nn.Module], torch.nn.Module]     # the sub-module that takes the checkpoint
    backbones: Callable[[torch.nn.Module], Sequence[torch.nn.Module]]   # what `_pre == "frozen"` freezes
    finetune_lr: Callable[[float], float]           # learning rate when the backbone trains too


# _status -> rule.  Anything that is neither "pretrained" nor "clmr" ("vanilla", "samplecnn", ...) is a randomly
# initialised DS_CNN trained at `_lr` (loader.py:64-66).
RULES: Dict[str, _Rule] = {
    # zeroNoteSamba's own pretext weights under Down_CNN (loader.py:22-43): both branches' conv stacks freeze; fine-tuning
    # runs at 0.5 * lr * 10e-2 (the literal of loader.py:43: a twentieth of lr)
    "pretrained": _Rule(Down_CNN, SHIFT_CKPT, lambda m: m.pretext,
                        lambda m: (m.pretext.anchor.pretrained, m.pretext.postve.pretrained), lambda lr: 0.5 * lr * 10e-2),
    # the CLMR baseline: one DS_CNN (loader.py:45-62); fine-tuning at half of lr
    "clmr": _Rule(DS_CNN, CLMR_CKPT, lambda m: m, lambda m: (m.pretrained,), lambda lr: 0.5 * lr),
}
_FROM_SCRATCH = _Rule(DS_CNN, None, lambda m: m, lambda m: (), lambda lr: lr)


def load_models(_status: str, _pre: str, _lr: float, state_dict: Optional[Dict[str, torch.Tensor]] = None,
                device: str = "cuda") -> Tuple[BCELoss, torch.optim.Optimizer, torch.nn.Module]:
    """Returns `(criterion, optimizer, model)` like the reference.
    _status: "pretrained" | "clmr" | anything else (a DS_CNN from scratch);  _pre: "frozen" freezes the rule's backbones and
    trains the rest at `_lr`, any other value fine-tunes everything at the rule's reduced rate;  _lr: base learning rate.
    state_dict: weights to use instead of reading the rule's checkpoint file;  device: where the model lives."""
    rule = RULES.get(_status, _FROM_SCRATCH)
    print("\n{} learning mode...".format(_status))          # the reference announces the mode (loader.py:19)
    model = rule.build().to(device)
    if rule.checkpoint is not None:
        weights = state_dict if state_dict is not None else torch.load(rule.checkpoint, map_location=torch.device(device))
        rule.load_into(model).load_state_dict(weights)
    if rule.checkpoint is not None and _pre == "frozen":
        for backbone in rule.backbones(model):
            backbone.requires_grad_(False)
        trainable = [p for p in model.parameters() if p.requires_grad]
        optimizer = Adam(trainable, lr=_lr, betas=BETAS)
    else:
        optimizer = Adam(model.parameters(), lr=rule.finetune_lr(_lr), betas=BETAS)
    return BCELoss(), optimizer, model
