"""The device-side part of the reference's preprocessing step between features and training
(preprocess_data/preprocess_adversary_data.py:20-83, 131, 356-423): per-speaker statistics, windows of
200 frames every 50 (one zero-padded window for short clips), z-norm or min-max normalisation with the
statistics of each clip's speaker, and the Gaussian class-balance augmentation.  Pickle / fold / file
handling of that script stays out of scope."""
import torch

from . import ops
from ._lib import lib, check, current_stream_ptr, require_cuda

WIN_LEN, SHIFT_LEN = 200, 50
MODES = {"znorm": 0, "min_max": 1}


def speaker_stats(mel_btf: torch.Tensor, spk: torch.Tensor = None, n_speakers: int = 1, population: str = "windows",
                  lengths: torch.Tensor = None, whole_clip: torch.Tensor = None, win=WIN_LEN, shift=SHIFT_LEN) -> torch.Tensor:
    """mel (B, T, F) fp32, spk (B) int32 speaker index per clip -> stats (S, 4, F) = mean, std, min, max
    (np.nanmean / nanstd / nanmin / nanmax, :360-367).

    population='windows' (default) is the reference's: its statistics run over the rows of the SAVED items
    (:26-27), so a frame counts once per 200/50 window that contains it, frames behind a clip's last window never,
    and the clips of test-split speakers (`whole_clip[b]` true: saved whole, :55-59) or clips shorter than a window
    count each frame once.  `lengths` (B) gives the valid frames of zero-padded ragged clips.
    population='frames' weights every frame of every clip once (what an online pipeline without the window
    bookkeeping would compute; not the reference's numbers)."""
    require_cuda(mel_btf)
    B, T, F = mel_btf.shape
    mel_btf = mel_btf.float().contiguous()
    if spk is not None:
        spk = spk.to(device=mel_btf.device, dtype=torch.int32).contiguous()
    ws = torch.empty(lib.sept_speaker_stats_workspace_doubles(B, F), dtype=torch.float64, device=mel_btf.device)
    stats = torch.empty((n_speakers, 4, F), dtype=torch.float32, device=mel_btf.device)
    if population == "windows":
        if lengths is not None:
            lengths = lengths.to(device=mel_btf.device, dtype=torch.int32).contiguous()
        if whole_clip is not None:
            whole_clip = whole_clip.to(device=mel_btf.device, dtype=torch.uint8).contiguous()
        check(lib.sept_speaker_stats_windows(mel_btf.data_ptr(), spk.data_ptr() if spk is not None else None,
                                             lengths.data_ptr() if lengths is not None else None,
                                             whole_clip.data_ptr() if whole_clip is not None else None, B, T, F, n_speakers,
                                             int(win), int(shift), ws.data_ptr(), stats.data_ptr(),
                                             current_stream_ptr(mel_btf.device)), "sept_speaker_stats_windows")
        return stats
    if population != "frames":
        raise ValueError(f"population must be 'windows' or 'frames', got {population!r}")
    if lengths is not None or whole_clip is not None:
        raise ValueError("lengths / whole_clip belong to population='windows'")
    check(lib.sept_speaker_stats(mel_btf.data_ptr(), spk.data_ptr() if spk is not None else None, B, T, F, n_speakers,
                                 ws.data_ptr(), stats.data_ptr(), current_stream_ptr(mel_btf.device)), "sept_speaker_stats")
    return stats


def window_normalize(mel_btf, stats, spk=None, norm="znorm", win=WIN_LEN, shift=SHIFT_LEN):
    """(B, T, F) -> (B * nwin, win, F): windows [shift*i, shift*i + win) of every clip, normalised with the
    statistics of the clip's speaker (:377-381); nwin = int((T - win) / shift) + 1, or one zero-padded
    window when T < win (:43-45, :30-35)."""
    require_cuda(mel_btf, stats)
    B, T, F = mel_btf.shape
    nwin = 1 if T < win else (T - win) // shift + 1
    mel_btf = mel_btf.float().contiguous()
    if spk is not None:
        spk = spk.to(device=mel_btf.device, dtype=torch.int32).contiguous()
    out = torch.empty((B * nwin, win, F), dtype=torch.float32, device=mel_btf.device)
    check(lib.sept_window_norm_spk(mel_btf.data_ptr(), stats.contiguous().data_ptr(),
                                   spk.data_ptr() if spk is not None else None, MODES[norm], out.data_ptr(), B, T, F, win,
                                   shift, nwin, current_stream_ptr(mel_btf.device)), "sept_window_norm_spk")
    return out


def add_gaussian(x: torch.Tensor, std: float = 0.05) -> torch.Tensor:
    """x + Normal(0, std) from the device's Philox stream (the augmentation noise of :416-417)."""
    require_cuda(x)
    x = x.float().contiguous()
    out = torch.empty_like(x)
    r = ops.rng(x.device, "augment")
    check(lib.sept_add_normal(x.data_ptr(), out.data_ptr(), x.numel(), float(std), r.seed, r.counter.data_ptr(), r._next(),
                              current_stream_ptr(x.device)), "sept_add_normal")
    return out


def balance_by_augmentation(windows: torch.Tensor, labels: torch.Tensor, std: float = 0.05, generator=None):
    """Class-balance augmentation (:392-423): every class smaller than the largest one receives
    `max - count` extra samples, each a randomly chosen member of that class plus Normal(0, std) noise.
    Returns (windows', labels') with the new samples appended."""
    lab = labels.view(-1).cpu()
    classes, counts = torch.unique(lab, return_counts=True)
    top = int(counts.max())
    picks = []
    for c, n in zip(classes.tolist(), counts.tolist()):
        if n == top:
            continue
        members = torch.nonzero(lab == c).view(-1)
        picks.append(members[torch.randint(0, len(members), (top - n,), generator=generator)])
    if not picks:
        return windows, labels
    idx = torch.cat(picks).to(windows.device)
    extra = add_gaussian(windows.index_select(0, idx), std)
    return torch.cat((windows, extra), 0), torch.cat((labels.view(-1), labels.view(-1).index_select(0, idx.to(labels.device))), 0)
