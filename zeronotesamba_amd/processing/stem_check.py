"""Drop-in for /root/reference/zeroNoteSamba/processing/stem_check.py::{compute_rms, check_CL_clips} on MI355X.

The RMS gate decides whether a random 10 s crop has enough energy in both the drum stem and the rest of the mix
(frame 2048 / hop 512 RMS of both, share of frames with anchor in (positive/2, positive*4)).  `zns_rms_gate_f32`
evaluates any number of candidate crops of two device-resident signals in one launch; the thresholds
`lower_p < share <= upper_p` are applied on the host exactly as the reference does (stem_check.py:47-51).
`check_drum_stem` (yaml-driven dataset curation) is out of scope.
"""
from typing import Optional, Sequence, Tuple, Union

import numpy as np
import numpy.typing as npt
import torch

from .. import _lib, _ops

ArrayLike = Union[npt.NDArray[np.float32], torch.Tensor]


def to_device_signal(x: ArrayLike, what: str) -> torch.Tensor:
    if not torch.cuda.is_available():
        raise RuntimeError("zeronotesamba_amd: the RMS gate needs a ROCm device; there is no CPU path")
    t = x if torch.is_tensor(x) else torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32))
    if t.dim() != 1:
        raise ValueError(f"{what} must be a 1-D waveform, got shape {tuple(t.shape)}")
    if not t.is_cuda:
        t = t.to(torch.device("cuda", torch.cuda.current_device()))
    return t.to(torch.float32).contiguous()


def gate_counts(anchor: torch.Tensor, positive: torch.Tensor, starts: Optional[Sequence[int]], n_samples: int,
                want_rms: bool = False):
    """Frames passing the ratio test for each candidate crop [start, start + n_samples) of two device signals.
    Returns (counts int32 [n_cand] on the device, n_frames, rms_a, rms_p) -- no host sync."""
    _ops.require_device(anchor, "anchor signal")
    _ops.require_device(positive, "positive signal")
    if anchor.numel() != positive.numel():
        raise ValueError("anchor and positive must have the same length")
    L = _lib.lib()
    dev = anchor.device
    if starts is None:
        starts = [0]
    n_cand = len(starts)
    if min(starts) < 0 or max(starts) + n_samples > anchor.numel():
        raise IndexError("candidate crop outside the signal")
    st = torch.tensor(list(starts), dtype=torch.int64).to(dev, non_blocking=True)
    n_frames = L.zns_rms_frames(n_samples)
    counts = torch.empty(n_cand, dtype=torch.int32, device=dev)
    ra = torch.empty((n_cand, n_frames), dtype=torch.float32, device=dev) if want_rms else None
    rp = torch.empty((n_cand, n_frames), dtype=torch.float32, device=dev) if want_rms else None
    _ops.same_device(anchor, (positive,), "RMS gate")
    with _ops.on_device(anchor):
        rc = L.zns_rms_gate_f32(anchor.data_ptr(), positive.data_ptr(), st.data_ptr(), n_cand, int(n_samples),
                                ra.data_ptr() if want_rms else None, rp.data_ptr() if want_rms else None,
                                counts.data_ptr(), _ops.stream_ptr(dev))
    _lib.check(rc, "zns_rms_gate_f32")
    counts._zns_keep = st          # the start table must outlive the enqueued kernel
    return counts, n_frames, ra, rp


def compute_rms(signal: ArrayLike) -> Tuple[npt.NDArray[np.float32], float, float]:
    """
    Function for combining a signal's Root Mean Square (RMS) value.
    -- signal: input waveform
    """
    x = to_device_signal(signal, "signal")
    _, _, ra, _ = gate_counts(x, x, [0], x.numel(), want_rms=True)
    rms = ra.cpu().numpy()                      # shape (1, n_frames), like librosa.feature.rms
    return rms, np.mean(rms), np.std(rms)


def check_CL_clips(anchor: ArrayLike, positive: ArrayLike, lower_p: float, upper_p: float) -> bool:
    """
    Function for thresholding anchor vs positive. Goal is to make sure drum clip has enough energy.
    -- anchor: selected stem combination
    -- positive: other stem combination
    -- lower_p: lower RMS percentage threshold
    -- upper_p: upper RMS percentage threshold
    """
    a = to_device_signal(anchor, "anchor")
    p = to_device_signal(positive, "positive")
    counts, n_frames, _, _ = gate_counts(a, p, [0], a.numel())
    rms_perc = int(counts.item()) / n_frames
    return bool(lower_p < rms_perc <= upper_p)
