"""SpectralNet facade with the reference's interface (src/SpectralNet.py:10-113) for the
SpectralNet stage and the autoencoder pre-embedding stage.  The Siamese stage (whose result the
reference's SpectralTrainer never uses) is not part of this package: configs that ask for it are
refused loudly rather than silently skipped."""
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
        parts, features_batches, support_batches, y_train_batches, train_mask_batches = train
        idx_parts, val_features_batches, val_support_batches, y_val_batches, val_mask_batches = val
        if self.config.get("should_use_ae", False):
            from .AETrainer import AETrainer
            ae_trainer = AETrainer(self.config, self.device)
            self.ae_net = ae_trainer.train(features_batches, val_features_batches)
            features_batches, val_features_batches = ae_trainer.embed(features_batches, val_features_batches)
            train = parts, features_batches, support_batches, y_train_batches, train_mask_batches
            val = idx_parts, val_features_batches, val_support_batches, y_val_batches, val_mask_batches
        if self.config.get("should_use_siamese", False):
            raise NotImplementedError("the Siamese stage is not part of this package (the reference's SpectralTrainer "
                                      "ignores its result, src/SpectralTrainer.py:144-167)")
        self.siamese_net = None
        spectral_trainer = SpectralTrainer(self.config, self.device, is_sparse=False, siamese_net=None)
        self.spec_net = spectral_trainer.train(train, val)

    def embed(self, X: torch.Tensor) -> np.ndarray:
        """The SpectralNet input of every row of X (src/SpectralNet.py:61-77): the autoencoder code
        when the autoencoder stage is on, X itself otherwise."""
        if self.config.get("should_use_ae", False):
            X = self._encode_chunks(X)
        return X.detach().cpu().numpy().copy()

    def _encode_chunks(self, X: torch.Tensor, chunk_rows: int = 1 << 18) -> torch.Tensor:
        self.ae_net.eval()
        with torch.no_grad():
            return torch.cat([self.ae_net.encode(X[a:a + chunk_rows].to(self.device, dtype=torch.float32)).cpu()
                              for a in range(0, int(X.shape[0]), chunk_rows)], dim=0)

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
                if self.config.get("should_use_ae", False):
                    x = self.ae_net.encode(x)
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
