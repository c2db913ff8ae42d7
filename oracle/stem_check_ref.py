"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's clip selection for the online pretext path.

Follows /root/reference/zeroNoteSamba/processing/stem_check.py:8-51 (compute_rms, check_CL_clips) and
/root/reference/zeroNoteSamba/pretext.py:30-86 (drum_anchor_positive).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module.

PARITY UNPINNED for the one third-party call: `librosa.feature.rms` (librosa 0.8.1, pyproject.toml:34) is not
installed and cannot be imported here.  Its published algorithm is restated in `feature_rms`:
    y = np.pad(y, frame_length // 2, mode="reflect"); x = util.frame(y, frame_length, hop_length)
    rms = sqrt(mean(abs(x) ** 2, axis=0, keepdims=True))
Known answers that pin it without librosa (tests/test_oracle_online.py): constant signal -> |c|, sinusoid ->
A/sqrt(2) on interior frames, frame count 1 + n // hop, scaling law, reflect-edge frames by brute force.
Everything else in this file is plain numpy / `random` and follows the reference line by line in behaviour.
"""
import random
from typing import Callable, Dict, Tuple

import numpy as np

FRAME, HOP = 2048, 512   # stem_check.py:13


def feature_rms(y: np.ndarray, frame_length: int = FRAME, hop_length: int = HOP) -> np.ndarray:
    """librosa.feature.rms(y=..., center=True, pad_mode="reflect") -> shape (1, 1 + len(y) // hop)."""
    y = np.asarray(y)
    ypad = np.pad(y, frame_length // 2, mode="reflect")
    n_frames = 1 + (len(ypad) - frame_length) // hop_length
    frames = np.lib.stride_tricks.as_strided(ypad, shape=(frame_length, n_frames),
                                             strides=(ypad.strides[0], hop_length * ypad.strides[0]))
    power = np.mean(np.abs(frames) ** 2, axis=0, keepdims=True)
    return np.sqrt(power)


def compute_rms(signal: np.ndarray) -> Tuple[np.ndarray, float, float]:
    """stem_check.py:8-19."""
    rms = feature_rms(signal)
    return rms, float(np.mean(rms)), float(np.std(rms))


def gate_fraction(anchor: np.ndarray, positive: np.ndarray) -> float:
    """stem_check.py:34-45: share of frames where the anchor's RMS lies in (positive/2, positive*4)."""
    a = feature_rms(anchor.T)[0]
    p = feature_rms(positive.T)[0]
    ok = (a > p / 2) & (a < p * 4)
    return int(np.sum(ok.astype(int))) / len(ok)


def check_CL_clips(anchor: np.ndarray, positive: np.ndarray, lower_p: float, upper_p: float) -> bool:
    """stem_check.py:22-51."""
    return bool(lower_p < gate_fraction(anchor, positive) <= upper_p)


def mix_stems(stems: Dict[str, np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
    """pretext.py:46-64: positive = the drums stem; anchor = float32 sum of every other stem, in dict order."""
    anchor = None
    pos = None
    for name, sig in stems.items():
        if name == "drums":
            pos = np.array(sig, dtype=np.float32)
        elif anchor is None:
            anchor = np.array(sig, dtype=np.float32)
        else:
            anchor += sig
    if anchor is None:
        raise Exception("Anchor is still None.")
    return anchor, pos


def drum_anchor_positive(stems: Dict[str, np.ndarray], ymldict: dict, generate_XQT: Callable,
                         rng=random) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray, int]:
    """pretext.py:30-86.  Returns the reference's 4-tuple plus the accepted start sample (for the tests)."""
    length = int(float(ymldict.get("clip_len", -1)))
    mode = str(ymldict.get("input_mode", ""))
    lower_p = float(ymldict.get("lower_p", -1.0))
    upper_p = float(ymldict.get("upper_p", -1.0))
    anchor, pos = mix_stems(stems)
    n = length * 16000
    tries = 0
    while True:
        start = rng.randint(0, len(anchor) - n - 1)
        a, p = anchor[start:start + n], pos[start:start + n]
        ok = check_CL_clips(a, p, lower_p, upper_p)
        tries += 1
        if tries > 9:
            lower_p = lower_p / 2
        if ok:
            break
    return anchor, pos, generate_XQT(a, 16000, mode), generate_XQT(p, 16000, mode), start
