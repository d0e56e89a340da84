"""oracle/preprocess_oracle.py against hand-computable cases of preprocess_adversary_data.py:20-83, 356-381."""
import numpy as np

from oracle import preprocess_oracle as po


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
