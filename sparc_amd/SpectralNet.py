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

    def predict(self, X: torch.Tensor) -> np.ndarray:
        """Embeddings with the orthonormalisation weights left by the last training step
        (src/SpectralNet.py:80-98)."""
        X = X.to(self.device)
        with torch.no_grad():
            self.embeddings_ = self.spec_net(X, should_update_orth_weights=False).detach().cpu().numpy()
        return self.embeddings_
