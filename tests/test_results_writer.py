"""The on-disk hand-over written after training (src/main.py:134-153): file names, shapes and the
1/sqrt(n) scaling of the embeddings."""
import numpy as np
import pytest


def test_save_results_writes_the_reference_files(tmp_path):
    pytest.importorskip("torch")
    try:
        from sparc_amd.SpectralNet import save_results
    except Exception as e:  # the HIP library is mandatory for the package
        pytest.skip(f"sparc_amd not importable here: {e}")
    rng = np.random.default_rng(0)
    n, k, din = 50, 4, 7
    Y = rng.standard_normal((n, k)).astype(np.float32)
    X = rng.standard_normal((n, din)).astype(np.float32)
    labels = rng.integers(0, 3, n)
    masks = [rng.random(n) < 0.3 for _ in range(3)]
    out = tmp_path / "sparc_results" / "golden"
    save_results(str(out), Y, X, labels, masks[0], masks[1], masks[2], X)
    names = sorted(p.name for p in out.iterdir())
    assert names == sorted(["embeddings.npy", "features.npy", "labels.npy", "test_mask.npy", "val_mask.npy",
                            "train_mask.npy", "X_embedded.npy"])
    emb = np.load(out / "embeddings.npy")
    assert emb.dtype == np.float32 or emb.dtype == np.float64
    assert np.allclose(emb, Y / np.sqrt(n), rtol=0, atol=1e-7)
    assert np.array_equal(np.load(out / "features.npy"), X)
    assert np.array_equal(np.load(out / "test_mask.npy"), masks[0])
    assert np.array_equal(np.load(out / "train_mask.npy"), masks[2])
