"""oracle/preprocess_oracle.py against (a) the vectors the REFERENCE's own code produced on synthetic clips
(tests/golden/preprocess_golden.npz, written by tools/make_goldens_preprocess.py, which executes
preprocess_adversary_data.py:20-83 and :357-390) and (b) hand-computable cases."""
import os

import numpy as np

from oracle import preprocess_oracle as po
from tests.preprocess_synth import CLIPS, F, WIN, synthetic_clips


def reference_items(G, norm):
    """{key: (split, length, checksums, edge rows)} of one normalisation mode of the golden file"""
    return {str(k): (str(s), int(n), c, e) for k, s, n, c, e in
            zip(G[f"{norm}_keys"], G[f"{norm}_splits"], G[f"{norm}_len"], G[f"{norm}_sums"], G[f"{norm}_edges"])}


def oracle_items(clips, stats, norm):
    """the oracle's stored + normalised items, keyed like the reference's dictionaries (sentence_file + '_' + i)"""
    out = {}
    for k, (clip, (L, spk, split)) in enumerate(zip(clips, CLIPS)):
        for i, (_, stored) in enumerate(po.saved_items(clip, WIN, WIN // 4, split == "test")):
            out[f"clip{k:02d}_{i}"] = (split, po.normalise(stored, stats[spk], norm))
    return out


def test_oracle_reproduces_the_reference_run(golden_dir):
    """The restatement against the reference's own functions: same stored items (count, keys, lengths -- short clips
    padded to 200 rows, test-split clips whole and once), same per-speaker statistics population, same normalised data."""
    G = np.load(os.path.join(golden_dir, "preprocess_golden.npz"))
    clips = synthetic_clips()
    np.testing.assert_allclose([c.sum() for c in clips], G["clip_sums"], rtol=1e-12)       # same synthetic inputs
    assert int(G["F"]) == F and int(G["win_len"]) == WIN
    speakers = [c[1] for c in CLIPS]
    test_speakers = {c[1] for c in CLIPS if c[2] == "test"}
    stats = po.speaker_statistics(clips, speakers, test_speakers, WIN, WIN // 4)
    assert sorted(stats) == [str(s) for s in G["stat_speakers"]]
    for j, spk in enumerate(G["stat_speakers"]):
        for w, which in enumerate(("mean", "std", "min", "max")):
            np.testing.assert_allclose(stats[str(spk)][which], G["stats"][j, w], rtol=1e-12, atol=1e-12)
    for norm in ("znorm", "min_max"):
        want = reference_items(G, norm)
        got = oracle_items(clips, stats, norm)
        assert sorted(got) == sorted(want)
        for key, (split, d) in got.items():
            w_split, w_len, w_sums, w_edges = want[key]
            assert split == w_split and d.shape == (w_len, F), key
            np.testing.assert_allclose([d.sum(), np.abs(d).sum(), (d * d).sum()], w_sums, rtol=1e-10)
            np.testing.assert_allclose(np.concatenate([d[:4], d[-4:]]), w_edges, rtol=1e-10, atol=1e-12)
            if f"{norm}_{key}" in G.files:
                np.testing.assert_allclose(d, G[f"{norm}_{key}"][0], rtol=1e-5, atol=1e-5)       # stored as float32


def test_saved_items_counts_and_padding():
    F = 3
    clip = np.arange(501 * F, dtype=np.float64).reshape(501, F)
    items = po.saved_items(clip)                       # int((501 - 200) / 50) + 1 = 7 windows
    assert len(items) == 7 and all(len(rows) == 200 for rows, _ in items)
    assert np.array_equal(items[3][1], clip[150:350])
    assert len(po.saved_items(clip, test_split=True)) == 1 and len(po.saved_items(clip, test_split=True)[0][0]) == 501
    short = clip[:120]
    (rows, stored), = po.saved_items(short)
    assert len(rows) == 120 and stored.shape == (200, F) and np.array_equal(stored[:120], short) and (stored[120:] == 0).all()


def test_statistics_population_is_the_saved_rows():
    rng = np.random.default_rng(0)
    F = 4
    a, b = rng.normal(size=(320, F)), rng.normal(size=(150, F))
    st = po.speaker_statistics([a, b], ["s", "s"])
    mult = po.frame_multiplicity(320)                  # windows [0,200) [50,250) [100,300): frames 300.. never saved
    assert mult[:50].tolist() == [1] * 50 and mult[100:200].tolist() == [3] * 100 and mult[300:].sum() == 0
    rows = np.concatenate([np.repeat(a, mult, axis=0), b])
    np.testing.assert_allclose(st["s"]["mean"], rows.mean(0), rtol=1e-12)
    np.testing.assert_allclose(st["s"]["std"], rows.std(0), rtol=1e-12)
    np.testing.assert_allclose(st["s"]["min"], np.minimum(a[:300].min(0), b.min(0)))
    st_test = po.speaker_statistics([a, b], ["s", "s"], test_speakers={"s"})
    np.testing.assert_allclose(st_test["s"]["mean"], np.concatenate([a, b]).mean(0), rtol=1e-12)
    z = po.normalise(a[:200], st["s"], "znorm")
    np.testing.assert_allclose(z, (a[:200] - st["s"]["mean"]) / (st["s"]["std"] + 1e-5))
    mm = po.normalise(a[:200], st["s"], "min_max")
    assert mm.min() >= -1 - 1e-12 and mm.max() <= 1 + 1e-12
