"""zeronotesamba_amd -- MI355X (gfx950) implementation of zeroNoteSamba's pretext hot path.

Module layout mirrors the reference package so call sites only change the package name:

    zeroNoteSamba.processing.input_rep.generate_XQT -> zeronotesamba_amd.processing.input_rep.generate_XQT
    zeroNoteSamba.models.models.{_CNN,DS_CNN,Pretext_CNN,Down_CNN} -> zeronotesamba_amd.models.models.*
    zeroNoteSamba.models.loss_functions.NTXent -> zeronotesamba_amd.models.loss_functions.NTXent
    zeroNoteSamba.pretext.{train_epoch,val_epoch} -> zeronotesamba_amd.pretext.*
    zeroNoteSamba.epochs (model/loss/step part)    -> zeronotesamba_amd.epochs.*

All arithmetic runs in hand-written HIP kernels behind the C ABI of include/zns.h
(libzns_hip.so, loaded with ctypes).  There is no CPU or stock-PyTorch fallback.
"""
__version__ = "0.1.0"


def frozen_weights():
    """Context manager for inference / validation loops that call the model many times with unchanged weights
    (sample_script.py:46-48 over many files): the kernel-order weight operands are then built once, see _ops.frozen_weights."""
    from ._ops import frozen_weights as _fw
    return _fw()
