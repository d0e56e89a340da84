"""CPU oracle for the step between the two halves of the hot path (TEST INFRASTRUCTURE ONLY): windowing,
per-speaker statistics and normalisation of preprocess_data/preprocess_adversary_data.py, restated in numpy
line by line.  Only tests/ may import this file; the product (sept_amd/preprocess.py over libsept_hip) never does.

What the reference does (win_len W = 200, shift_len = W / 4 = 50, `--shift 1`):

  save_data_dict (:41-83)       a clip of L frames gives save_len = 1 if L < W else int((L - W) / shift) + 1 items.
                                Train / validation / adversary splits save the window rows
                                save_data[i*shift : i*shift + W]; a TEST-split speaker's clip is saved ONCE, whole
                                (`break` after the first item, :57-59).
  write_data_dict (:20-38)      EVERY ROW of every saved item is appended to training_norm_dict[speaker] (:26-27) --
                                so the statistics population is the rows of the saved WINDOWS: a frame counts once per
                                window that contains it (0-4 times at 200 / 50; frames behind the last window never),
                                and a test-split clip's frames count once each.  Items shorter than W are padded with
                                NaN, then fillna(0) (:29-35): zeros BEFORE the normalisation; the NaN rows are never
                                part of the statistics (only the len(data) real rows are appended).
  statistics (:356-367)         np.nanmean / nanstd (population, ddof 0) / nanmin / nanmax over that row list, per mel bin.
  normalisation (:371-381)      znorm: (x - mean) / (std + 1e-5);  min_max: (x - min) / (max - min) * 2 - 1.

Parity status: PINNED to the reference's own code.  The script as a whole cannot run here (hard-coded corpus root,
IEMOCAP label files, pickles), but tools/make_goldens_preprocess.py executes its write_data_dict / save_data_dict
(:20-83) and its statistics + normalisation block (:357-390), read from the reference file in the build container, on
the synthetic clips of tests/preprocess_synth.py and records the results in tests/golden/preprocess_golden.npz;
tests/test_oracle_preprocess.py holds this restatement to them at 1e-10 (statistics 1e-12).
"""
import numpy as np


def saved_items(save_data, win_len=200, shift_len=50, test_split=False):
    """The (rows appended to the speaker's norm list, stored 'data' array) pairs of one clip (:41-83, :20-38)."""
    L = len(save_data)
    padding = L < win_len
    save_len = 1 if L < win_len else int((L - win_len) / shift_len) + 1
    out = []
    for i in range(save_len):
        data = save_data if test_split else save_data[i * shift_len:i * shift_len + win_len]
        rows = [data[k, :] for k in range(len(data))]                   # :26-27
        if padding:                                                      # :29-35
            tmp = np.empty([win_len, data.shape[1]])
            tmp[:, :] = np.nan
            tmp[:len(data), :] = data
            stored = np.nan_to_num(tmp, nan=0.0)
        else:
            stored = data
        out.append((rows, stored))
        if test_split:
            break
    return out


def speaker_statistics(clips, speakers, test_speakers=(), win_len=200, shift_len=50):
    """clips: list of (L_i, F) arrays; speakers: list of ids.  -> {speaker: {'mean','std','min','max'}} (:356-367)."""
    norm_dict = {}
    for clip, spk in zip(clips, speakers):
        norm_dict.setdefault(spk, [])
        for rows, _ in saved_items(np.asarray(clip, dtype=np.float64), win_len, shift_len, spk in test_speakers):
            norm_dict[spk].extend(rows)
    F = np.asarray(clips[0]).shape[1]
    stats = {}
    for spk, rows in norm_dict.items():
        a = np.array(rows).reshape(-1, F)
        stats[spk] = {"mean": np.nanmean(a, axis=0), "std": np.nanstd(a, axis=0),
                      "min": np.nanmin(a, axis=0), "max": np.nanmax(a, axis=0)}
    return stats


def normalise(stored, st, norm="znorm"):
    """:377-381 on one stored item."""
    if norm == "znorm":
        return (stored - st["mean"]) / (st["std"] + 1e-5)
    return (stored - st["min"]) / (st["max"] - st["min"]) * 2 - 1


def frame_multiplicity(L, win_len=200, shift_len=50, test_split=False):
    """How many saved rows each frame of an L-frame clip contributes (the weights behind speaker_statistics)."""
    m = np.zeros(L, dtype=np.int64)
    if test_split or L < win_len:
        m[:] = 1
        return m
    for i in range(int((L - win_len) / shift_len) + 1):
        m[i * shift_len:i * shift_len + win_len] += 1
    return m
