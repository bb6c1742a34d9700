"""SpectralNet facade with the reference's interface (src/SpectralNet.py:10-113) for the
SpectralNet stage.  The autoencoder and Siamese pre-training stages of the reference are
separate stages that are not part of this package: configs that ask for them are refused
loudly rather than silently skipped."""
from __future__ import annotations

import numpy as np
import torch

from .SpectralTrainer import SpectralTrainer


class SpectralNet:
    def __init__(self, n_clusters: int, config: dict):
        self.n_clusters = n_clusters
        self.config = config
        self.embeddings_ = None
        if not torch.cuda.is_available():
            raise RuntimeError("sparc_amd needs an MI355X (ROCm device); it has no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device())
        print(f"Using {self.device} device")

    def fit(self, train, val):
        """train = (parts, features_batches, support_batches, y_train_batches, train_mask_batches),
        val likewise (src/SpectralNet.py:25-58)."""
        if self.config.get("should_use_ae", False):
            raise NotImplementedError("the autoencoder pre-embedding stage (src/AETrainer.py) is not part of this package: "
                                      "embed the features first and set should_use_ae to false")
        if self.config.get("should_use_siamese", False):
            raise NotImplementedError("the Siamese stage is not part of this package (the reference's SpectralTrainer "
                                      "ignores its result, src/SpectralTrainer.py:144-167)")
        self.siamese_net = None
        spectral_trainer = SpectralTrainer(self.config, self.device, is_sparse=False, siamese_net=None)
        self.spec_net = spectral_trainer.train(train, val)

    def embed(self, X: torch.Tensor) -> np.ndarray:
        return np.array(X.detach().cpu())

    def predict(self, X: torch.Tensor, chunk_rows: int = 1 << 18) -> np.ndarray:
        """Embeddings of every row of X with the orthonormalisation weights left by the last
        training step (src/SpectralNet.py:80-98).  The rows are independent, so a whole graph
        (2.4 M nodes for Amazon2M) is pushed through in row chunks: X may stay on the host."""
        n = int(X.shape[0])
        k = int(self.spec_net.orthonorm_weights.shape[0])
        out = np.empty((n, k), dtype=np.float32)
        self.spec_net.eval()
        with torch.no_grad():
            for a in range(0, n, chunk_rows):
                x = X[a:a + chunk_rows].to(self.device, dtype=torch.float32)
                out[a:a + chunk_rows] = self.spec_net(x, should_update_orth_weights=False).cpu().numpy()
        self.embeddings_ = out
        return self.embeddings_


def save_results(directory: str, Y: np.ndarray, X, labels, test_mask, val_mask, train_mask, X_embedded) -> None:
    """The on-disk hand-over to the downstream SPARC models (src/main.py:134-153): seven .npy files,
    `embeddings.npy` holding Y / sqrt(n)."""
    import os
    os.makedirs(directory, exist_ok=True)
    Y = np.asarray(Y)
    np.save(os.path.join(directory, "embeddings.npy"), Y / np.sqrt(Y.shape[0]))
    np.save(os.path.join(directory, "features.npy"), X.cpu().numpy() if torch.is_tensor(X) else np.asarray(X))
    np.save(os.path.join(directory, "labels.npy"), np.asarray(labels))
    np.save(os.path.join(directory, "test_mask.npy"), np.asarray(test_mask))
    np.save(os.path.join(directory, "val_mask.npy"), np.asarray(val_mask))
    np.save(os.path.join(directory, "train_mask.npy"), np.asarray(train_mask))
    np.save(os.path.join(directory, "X_embedded.npy"), np.asarray(X_embedded))
