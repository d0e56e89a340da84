#!/usr/bin/env python3
"""Writes tests/golden/mel_golden.npz from the float64 feature oracle (oracle/mel_oracle.py).

The reference's feature function cannot be imported here (torchaudio / opensmile /
python_speech_features / moviepy are absent -- SURVEY.md section 8c), so these vectors pin
the ORACLE and the HIP kernel against each other and against analytic known answers; they
are not outputs of the reference ("parity unpinned" at the torchaudio boundary).

Inputs are regenerated from the seed by the tests; only small output slices + checksums
are stored:  torch.manual_seed(8); randn(4, 80000) * 0.1  (SURVEY.md section 8c (ii)).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from oracle.mel_oracle import mel_spectrogram_f64, melscale_fbanks_htk  # noqa: E402

FRAMES = (0, 250, 500)  # first (reflect edge), middle, last (reflect edge)


def main():
    torch.manual_seed(8)
    x = (torch.randn(4, 80000) * 0.1).numpy()
    out = {"frames": np.array(FRAMES)}
    for n_fft in (800, 1600, 1024):
        for F in (80, 128):
            db = mel_spectrogram_f64(x, n_fft, F)  # (4, F, 501)
            key = f"n{n_fft}_f{F}"
            out[key + "_slices"] = db[:, :, list(FRAMES)].astype(np.float64)
            out[key + "_sum"] = np.array(db.sum())
            out[key + "_abs_sum"] = np.array(np.abs(db).sum())
            fb = melscale_fbanks_htk(n_fft // 2 + 1, F).numpy()
            out[key + "_fb_nnz"] = np.array((fb > 0).sum())
            out[key + "_fb_colsum"] = fb.sum(0).astype(np.float32)
            out[key + "_fb_first"] = np.array([(fb[:, m] > 0).argmax() for m in range(F)])
    path = os.path.join(ROOT, "tests", "golden", "mel_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
