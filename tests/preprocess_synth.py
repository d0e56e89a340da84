"""Synthetic mel clips shared by tools/make_goldens_preprocess.py (which runs the REFERENCE's preprocessing code on them)
and the tests that hold the oracle and the HIP kernels to the recorded results."""
import numpy as np

F, WIN = 8, 200
# (length in frames, speaker, split) -- short clips (< 200 frames: zero padded), exact fit, ragged tails, test speakers
# (their clips are stored whole and once)
CLIPS = [(420, "Ses01F", "training"), (301, "Ses01M", "training"), (200, "Ses01F", "training"), (249, "Ses02F", "validation"),
         (120, "Ses01M", "training"), (333, "Ses02F", "validation"), (199, "Ses01F", "training"), (250, "Ses01M", "training"),
         (501, "Ses03M", "adv_training"), (180, "Ses03M", "adv_training"), (260, "Ses04F", "adv_validation"),
         (377, "Ses05M", "test"), (150, "Ses05M", "test"), (501, "Ses05F", "test")]


def synthetic_clips(seed=5):
    rng = np.random.default_rng(seed)
    return [(rng.standard_normal((L, F)) * 9.0 - 30.0 + 3.0 * k) for k, (L, _, _) in enumerate(CLIPS)]
