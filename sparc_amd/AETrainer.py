"""Autoencoder pre-embedding stage with the reference's interface (src/AETrainer.py:7-185): the
stage that turns raw node features into the SpectralNet input when config["should_use_ae"] is set.
Same module layout (state_dict keys encoder.{0,2,4,6}.*, decoder.{0,2,4,6}.*), config keys, stdout
lines and checkpoint path; the layers run through ops.LinearAct (library GEMMs with bias+ReLU in the
epilogue, activation derivative + bias gradient in libsparc_hip.so), features stay on the device."""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.optim as optim

from .graph import require_gpu
from .ops import ACT_NONE, ACT_RELU, LinearAct


def _run(seq: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    mods = list(seq)
    i = 0
    while i < len(mods):
        lin = mods[i]
        relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
        x = LinearAct.apply(x, lin.weight, lin.bias, ACT_RELU if relu else ACT_NONE)
        i += 2 if relu else 1
    return x


class AE(nn.Module):
    def __init__(self, architecture: dict, input_dim: int):
        super().__init__()
        self.architecture = architecture
        a = architecture
        self.encoder = nn.Sequential(
            nn.Linear(input_dim, a["hidden_dim1"]), nn.ReLU(),
            nn.Linear(a["hidden_dim1"], a["hidden_dim2"]), nn.ReLU(),
            nn.Linear(a["hidden_dim2"], a["hidden_dim3"]), nn.ReLU(),
            nn.Linear(a["hidden_dim3"], a["output_dim"]),
        )
        self.decoder = nn.Sequential(
            nn.Linear(a["output_dim"], a["hidden_dim3"]), nn.ReLU(),
            nn.Linear(a["hidden_dim3"], a["hidden_dim2"]), nn.ReLU(),
            nn.Linear(a["hidden_dim2"], a["hidden_dim1"]), nn.ReLU(),
            nn.Linear(a["hidden_dim1"], input_dim),
        )

    def encode(self, x: torch.Tensor) -> torch.Tensor:
        return _run(self.encoder, x)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _run(self.decoder, _run(self.encoder, x))

    def embed_single(self, X: torch.Tensor) -> torch.Tensor:
        return self.encode(X)


class AETrainer:
    def __init__(self, config: dict, device: torch.device):
        self.device = device
        self.dataset = config["dataset"]
        self.ae_config = config["ae"]
        self.lr = self.ae_config["lr"]
        self.epochs = self.ae_config["epochs"]
        self.lr_decay = self.ae_config["lr_decay"]
        self.patience = self.ae_config["patience"]
        self.n_samples = self.ae_config["n_samples"]
        self.batch_size = self.ae_config["batch_size"]
        self.architecture = self.ae_config["architecture"]
        self.output_dim = self.architecture["output_dim"]
        self.weights_path = f"./weights/{self.dataset}/ae_weights_{self.output_dim}.pth"

    def _dev(self, x: torch.Tensor) -> torch.Tensor:
        x = x.to(self.device, dtype=torch.float32)
        return x.view(x.size(0), -1)

    def train(self, features_batches, val_features_batches) -> AE:
        require_gpu(self.device)
        self.features_batches = features_batches
        self.val_features_batches = val_features_batches
        self.criterion = nn.MSELoss()
        self.ae_net = AE(self.architecture, input_dim=self.features_batches[0].shape[1]).to(self.device)
        self.optimizer = optim.Adam(self.ae_net.parameters(), lr=self.lr, fused=True)
        self.scheduler = optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode="min", factor=self.lr_decay,
                                                              patience=self.patience)
        if os.path.exists(self.weights_path):
            self.ae_net.load_state_dict(torch.load(self.weights_path, weights_only=True))
            return self.ae_net

        feats = [self._dev(f) for f in self.features_batches]      # one-time: features to the device
        self._val_dev = [self._dev(f) for f in self.val_features_batches]
        batches = len(feats)
        print("Training Autoencoder:")
        for epoch in range(self.epochs):
            train_loss = torch.zeros((), dtype=torch.float64, device=self.device)
            self.ae_net.train()
            for pid in range(batches):
                self.optimizer.zero_grad()
                loss = self.criterion(self.ae_net(feats[pid]), feats[pid])
                loss.backward()
                self.optimizer.step()
                train_loss += loss.detach()
            train_loss = float(train_loss) / batches                  # the epoch's host sync
            valid_loss = self.validate(self.val_features_batches)
            self.scheduler.step(valid_loss)
            current_lr = self.optimizer.param_groups[0]["lr"]
            if current_lr <= self.ae_config["min_lr"]:
                break
            print("Epoch: {}/{}, Train Loss: {:.4f}, Valid Loss: {:.4f}, LR: {:.6f}".format(
                epoch + 1, self.epochs, train_loss, valid_loss, current_lr))

        os.makedirs(os.path.dirname(self.weights_path), exist_ok=True)
        torch.save(self.ae_net.state_dict(), self.weights_path)
        return self.ae_net

    def validate(self, valid_loader) -> float:
        self.ae_net.eval()
        total = torch.zeros((), dtype=torch.float64, device=self.device)
        cached = getattr(self, "_val_dev", None)
        if cached is not None and valid_loader is not self.val_features_batches:
            cached = None   # the device copies belong to the batches given to train(), not to this argument
        with torch.no_grad():
            for pid in range(len(valid_loader)):
                x = cached[pid] if cached is not None else self._dev(valid_loader[pid])
                total += self.criterion(self.ae_net(x), x)
        return float(total) / len(valid_loader)

    def embed(self, features_batches, val_features_batches):
        print("Embedding data ...")
        self.ae_net.eval()
        with torch.no_grad():
            enc = [self.ae_net.encode(self._dev(f)) for f in features_batches]
            enc_val = [self.ae_net.encode(self._dev(f)) for f in val_features_batches]
        return enc, enc_val

    def embed_single(self, X: torch.Tensor) -> torch.Tensor:
        self.ae_net.eval()
        with torch.no_grad():
            return self.ae_net.encode(X.to(self.device))
