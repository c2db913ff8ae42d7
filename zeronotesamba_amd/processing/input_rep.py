"""Drop-in for /root/reference/zeroNoteSamba/processing/input_rep.py::generate_XQT on MI355X.

`generate_XQT(signal, sample_rate, mode)` keeps the reference signature and result: float32
`log(|XQT| + 1e-9)` of shape (96, 1 + len(signal)//256), mode "vqt" or "cqt", anything else raises
`Exception("Mode can only be vqt or cqt!")` (input_rep.py:56-57).  The transform itself (8-octave
decimation pyramid + folded per-octave FIR bank + abs/log epilogue) runs in zns_xqt_f32; this function
only moves the waveform to the device and the spectrogram back, as the reference's callers expect a
numpy array.  `generate_XQT_batch` is the device-resident form for pipelines that keep going on the GPU.
`plot_XQT` (matplotlib) is out of scope.
"""
from typing import Union

import numpy as np
import numpy.typing as npt
import torch

from .. import _lib, _ops

_MODES = {"vqt": _lib.ZNS_MODE_VQT, "cqt": _lib.ZNS_MODE_CQT}


def generate_XQT_batch(signals: torch.Tensor, sample_rate: int, mode: str, out: torch.Tensor = None,
                       stages: int = 3, ws: torch.Tensor = None, exact: bool = False) -> torch.Tensor:
    """signals: float32 [n_clips, n_samples] on a ROCm device -> float32 [n_clips, 96, 1 + n_samples//256].
    out: optional preallocated result (steady-state pipelines reuse it).  stages: 3 = the whole transform (vqt: ONE fused
    launch, the pyramid never leaves LDS; cqt: per-stage launches); 1 / 2 run the decimation pyramid / the filter bank alone on
    the shared workspace (bench.py times them separately); 3 | _lib.ZNS_XQT_SEPARATE forces the per-stage launches (same bits);
    | _lib.ZNS_XQT_EXACT selects the validation pyramid that emulates resampy's per-tap rounding bit for bit (slower).
    exact=True is the same switch for callers that never see the flag constants: the default pyramid runs resampy's taps in
    resampy's order with one float32 fused multiply-add per tap (bit for bit its numpy model; 1e-5 of the loudest bin and
    `1e-4 |V| + 1e-6 max|V|` per bin from the restated reference, i.e. up to a few 1e-4 in log() in bins a thousand times
    below the loudest -- tests/test_vqt.py::test_default_arithmetic_log_deviation_by_bin_level records the figures); the exact
    pyramid rounds like resampy after every tap and keeps 1e-4 in log() in EVERY bin, at about twice the time.  A
    one-time dataset or memory-bank pass that wants the reference's weak bins to the last digit asks for it.
    ws: optional caller-owned uint8 workspace of zns_xqt_workspace_bytes (a pipeline that overlaps two batches on two
    streams gives each its own; the default is one grow-only buffer per device)."""
    if mode not in _MODES:
        raise Exception("Mode can only be vqt or cqt!")
    _ops.require_device(signals, "generate_XQT input")
    if signals.dim() != 2:
        raise ValueError(f"generate_XQT_batch expects [n_clips, n_samples], got {tuple(signals.shape)}")
    x = signals.contiguous()
    n_clips, n = x.shape
    L = _lib.lib()
    T = L.zns_xqt_frames(n)
    if out is None:
        out = torch.empty((n_clips, 96, T), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (n_clips, 96, T) or out.dtype != torch.float32 or out.device != x.device or not out.is_contiguous():
        raise ValueError(f"generate_XQT_batch: out must be a contiguous float32 [{n_clips}, 96, {T}] tensor on {x.device}")
    if exact:
        stages = int(stages) | _lib.ZNS_XQT_EXACT
    if ws is None:
        ws = _ops.workspace(L.zns_xqt_workspace_bytes(n_clips, n, _MODES[mode]), x.device, slot=1)
    with _ops.on_device(x):       # the FIR bank is cached per device
        rc = L.zns_xqt_stages_f32(x.data_ptr(), n_clips, n, int(sample_rate), _MODES[mode], out.data_ptr(),
                                  ws.data_ptr(), ws.numel(), int(stages), _ops.stream_ptr(x.device))
    if rc == _lib.ZNS_ERR_MODE:
        raise Exception("Mode can only be vqt or cqt!")
    _lib.check(rc, "zns_xqt_f32")
    return out


def generate_XQT(signal: Union[npt.NDArray[np.float32], torch.Tensor], sample_rate: int,
                 mode: str, exact: bool = False) -> npt.NDArray[np.float32]:
    """
    Generates a high-resolution XQT spectrogram.
    -- signal: signal to compute XQT on (1-D float32 waveform)
    -- sample_rate: self-explanatory (16000, the reference's working rate)
    -- mode: can be either vqt or cqt
    -- exact (not in the reference's signature): the pyramid that rounds like resampy tap by tap, see generate_XQT_batch
    """
    if mode not in _MODES:
        raise Exception("Mode can only be vqt or cqt!")
    if not torch.cuda.is_available():
        raise RuntimeError("zeronotesamba_amd: generate_XQT needs a ROCm device; there is no CPU path")
    sig = torch.as_tensor(np.ascontiguousarray(signal, dtype=np.float32)) if not torch.is_tensor(signal) else signal
    if sig.dim() != 1:
        raise ValueError(f"generate_XQT expects a 1-D waveform, got shape {tuple(sig.shape)}")
    dev = sig.device if sig.is_cuda else torch.device("cuda", torch.cuda.current_device())
    out = generate_XQT_batch(sig.to(dev, dtype=torch.float32).reshape(1, -1), sample_rate, mode, exact=exact)
    return out[0].cpu().numpy()
