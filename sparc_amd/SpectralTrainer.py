"""SpectralNet model, loss and trainer with the reference's interface
(src/SpectralTrainer.py) on top of the MI355X kernels.

Same class names, constructor arguments, config keys, state_dict layout, stdout
line and checkpoint path as the reference; what differs is where the work runs:

  * the support tuple of every batch is turned into a device-resident ClusterBatch
    once (the reference densifies it and rebuilds the m x m affinity on every step,
    src/SpectralTrainer.py:210, :233);
  * the row permutation of a batch (:211-213, :218-219) is drawn from numpy's RNG in
    the same order but never applied: loss and gradients do not depend on it, and
    the one quantity that does (the sign pattern of the orthonormalisation weights)
    only needs the first k entries of the permutation.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

from .graph import ClusterBatch, require_gpu
from .ops import ACT_RELU, ACT_TANH, ORTH_CHOLESKY, ORTH_REFERENCE, LapLoss, LinearAct, OrthApply, OrthWeights, SpectralMLP, SplitRows


class SpectralNetModel(nn.Module):
    """MLP + orthonormalisation layer (src/SpectralTrainer.py:16-108)."""

    def __init__(self, architecture: dict, input_dim: int, device: torch.device, orth_mode: str = "reference"):
        super().__init__()
        self.architecture = architecture
        self.num_of_layers = self.architecture["n_layers"]
        self.layers = nn.ModuleList()
        self.input_dim = input_dim
        self.device = device
        self.orthonorm_weights = torch.eye(self.input_dim).to(self.device)  # as the reference (:24)
        self.rotation_matrix = None
        self.orth_mode = {"reference": ORTH_REFERENCE, "cholesky": ORTH_CHOLESKY}[orth_mode]

        current_dim = self.input_dim
        for layer, dim in self.architecture.items():
            if layer == "n_layers":
                continue
            if layer == "output_dim":
                self.layers.append(nn.Sequential(nn.Linear(current_dim, dim), nn.Tanh()))
            else:
                self.layers.append(nn.Sequential(nn.Linear(current_dim, dim), nn.ReLU()))
                current_dim = dim

    def _make_orthonorm_weights(self, Y: torch.Tensor, top_idx=None) -> torch.Tensor:
        """sqrt(m) * inverse(R @ diag(sign(diag R))) of the reference (:62-68), from the Gram
        matrix.  top_idx: the rows that lead the reference's (permuted) batch; default 0..k-1."""
        return OrthWeights.apply(Y, top_idx, self.orth_mode)

    def embed_pre_orth(self, x: torch.Tensor) -> torch.Tensor:
        kinds = [isinstance(layer[1], nn.Tanh) for layer in self.layers]
        if kinds[-1] and not any(kinds[:-1]) and x.dtype == torch.float32:   # ReLU ... ReLU, Tanh: the fused MLP
            params = [p for layer in self.layers for p in (layer[0].weight, layer[0].bias)]
            return SpectralMLP.apply(x.contiguous(), *params)
        for layer, is_tanh in zip(self.layers, kinds):
            x = LinearAct.apply(x, layer[0].weight, layer[0].bias, ACT_TANH if is_tanh else ACT_RELU)
        return x

    def forward(self, x: torch.Tensor, should_update_orth_weights: bool = True, top_idx=None) -> torch.Tensor:
        Y_tilde = self.embed_pre_orth(x)
        if should_update_orth_weights:
            self.orthonorm_weights = self._make_orthonorm_weights(Y_tilde, top_idx)
        return OrthApply.apply(Y_tilde, self.orthonorm_weights)

    def forward_pair(self, x_orth: torch.Tensor, x_grad: torch.Tensor, top_idx=None) -> torch.Tensor:
        """The two forwards of one training step (src/SpectralTrainer.py:222-223 then :232) as ONE pass
        of the MLP over both batches: the parameters do not change between them, so the rows are the
        same; the orthonormalisation weights come from x_orth's rows and are applied to x_grad's."""
        m2 = x_orth.shape[0]
        Yt_orth, Yt_grad = SplitRows.apply(self.embed_pre_orth(torch.cat([x_orth, x_grad], dim=0)), m2)
        self.orthonorm_weights = self._make_orthonorm_weights(Yt_orth, top_idx)
        return OrthApply.apply(Yt_grad, self.orthonorm_weights)

    def set_rotation_matrix(self, rotation_matrix):
        self.rotation_matrix = rotation_matrix


class SpectralNetLoss(nn.Module):
    """Rayleigh quotient of the batch Laplacian (src/SpectralTrainer.py:111-139).

    W is the ClusterBatch returned by SpectralTrainer._get_valid_affinity_matrix (the
    reference passes the dense m x m matrix it describes).
    """

    def forward(self, W, Y: torch.Tensor, is_normalized: bool = False) -> torch.Tensor:
        if not isinstance(W, ClusterBatch):
            raise TypeError("W must be a sparc_amd.graph.ClusterBatch (build one with ClusterBatch.from_support)")
        return LapLoss.apply(Y, W, bool(is_normalized))


class SpectralTrainer:
    def __init__(self, config: dict, device: torch.device, is_sparse: bool = False, siamese_net: nn.Module = None):
        self.device = device
        self.is_sparse = is_sparse
        self.spectral_config = config["spectral"]
        self.dataset = config["dataset"]
        self.lr = self.spectral_config["lr"]
        self.epochs = self.spectral_config["epochs"]
        self.lr_decay = self.spectral_config["lr_decay"]
        self.patience = self.spectral_config["patience"]
        self.batch_size = self.spectral_config["batch_size"]
        self.architecture = self.spectral_config["architecture"]
        self.n_clusters = self.spectral_config["n_clusters"]
        self.weights_path = f"./weights/{self.dataset}/spectralnet_weights.pth"
        self.orth_mode = self.spectral_config.get("orth_mode", "reference")
        self._batches = {}

    # -- the reference's two helpers, returning device-resident operators -------------------
    def _get_support_matrix(self, support, cluster_sizes=None) -> ClusterBatch:
        key = id(support)
        hit = self._batches.get(key)
        if hit is None or hit[0] is not support:
            hit = (support, ClusterBatch.from_support(support, self.device, cluster_sizes=cluster_sizes))
            self._batches[key] = hit
        return hit[1]

    def _cluster_sizes(self):
        """Row counts of the METIS clusters inside every training batch, when `parts` (the shuffled
        cluster list) and spectral.bsize describe the batches (src/utils.py:76-80); a hint for the
        LDS tiling only."""
        bsize = self.spectral_config.get("bsize")
        try:
            sizes = [len(p) for p in self.parts]
            out = [sizes[st:st + bsize] for st in range(0, len(sizes), bsize)]
            ok = len(out) == len(self.features_batches) and all(
                sum(o) == int(f.shape[0]) for o, f in zip(out, self.features_batches))
            return out if ok else None
        except Exception:
            return None

    def _get_valid_affinity_matrix(self, W: ClusterBatch, X=None) -> ClusterBatch:
        return W  # the affinity vectors were prepared together with the CSR

    def train(self, train, val) -> SpectralNetModel:
        require_gpu(self.device)
        self.parts, self.features_batches, self.support_batches, self.y_train_batches, self.train_mask_batches = train
        self.idx_parts, self.val_features_batches, self.val_support_batches, self.y_val_batches, self.val_mask_batches = val
        self.idx_parts = list(range(len(self.parts)))
        self.counter = 0
        OrthWeights.check_and_reset(self.device)   # failures of earlier, unrelated calls do not belong to this run
        self.criterion = SpectralNetLoss()
        self.spectral_net = SpectralNetModel(self.architecture, input_dim=self.features_batches[0].shape[1],
                                             device=self.device, orth_mode=self.orth_mode).to(self.device)
        # spectral.hip_graph (not a reference key; default off): whole steps replayed as HIP graphs, one launch per step
        hip_graph = bool(self.spectral_config.get("hip_graph", False))
        if hip_graph:
            lr_t = torch.tensor(float(self.lr), device=self.device)
            self.optimizer = optim.Adam(self.spectral_net.parameters(), lr=lr_t, fused=True, capturable=True)
        else:
            self.optimizer = optim.Adam(self.spectral_net.parameters(), lr=self.lr, fused=True)  # one launch per step
        self.scheduler = optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode="min", factor=self.lr_decay,
                                                              patience=self.patience)
        k = self.architecture["output_dim"]

        # one-time: features and supports to the device
        feats = [f.to(self.device, dtype=torch.float32) for f in self.features_batches]
        sizes = self._cluster_sizes()
        sup = [self._get_support_matrix(s, sizes[i] if sizes else None).warm(k, measure=True)
               for i, s in enumerate(self.support_batches)]
        vfeats = [f.to(self.device, dtype=torch.float32) for f in self.val_features_batches]
        vsup = [self._get_support_matrix(s).warm(k, training=False, measure=True) for s in self.val_support_batches]

        graphed = None
        if hip_graph:
            from .stepgraph import GraphedSteps
            graphed = GraphedSteps(self.spectral_net, self.criterion, self.optimizer, feats, sup, k, self.device)

        min_val_loss = 1000000000
        batches = len(feats)
        batches_val = len(vfeats)
        print("Training SpectralNet:")
        for epoch in range(self.epochs):
            train_loss = torch.zeros((), dtype=torch.float64, device=self.device)
            for pid in range(batches):
                pid_2 = np.random.choice(batches)
                np.random.permutation(len(feats[pid]))           # drawn as the reference does (:211), not applied
                perm_2 = np.random.permutation(len(feats[pid_2]))  # (:218) only its head matters

                # Orthogonality step (autograd stays on, as in the reference :222-223) and the
                # forward of the gradient step (:232), one MLP pass over both batches
                if graphed is not None:
                    train_loss += graphed.step(pid, pid_2, perm_2[:k])
                    continue
                self.spectral_net.train()
                self.optimizer.zero_grad()
                Y = self.spectral_net.forward_pair(feats[pid_2], feats[pid], top_idx=perm_2[:k])
                W = self._get_valid_affinity_matrix(sup[pid], feats[pid])
                loss = self.criterion(W, Y, is_normalized=True)
                loss.backward()
                self.optimizer.step()
                train_loss += loss.detach()

            valid_loss = torch.zeros((), dtype=torch.float64, device=self.device)
            self.spectral_net.eval()
            with torch.no_grad():
                for pid in range(batches_val):
                    Y = self.spectral_net(vfeats[pid], should_update_orth_weights=False)
                    W = self._get_valid_affinity_matrix(vsup[pid], vfeats[pid])
                    valid_loss += self.criterion(W, Y, is_normalized=True)
            train_loss, valid_loss = float(train_loss), float(valid_loss)  # the epoch's only host sync
            if OrthWeights.check_and_reset(self.device) != 0:   # any step of the epoch, not only the last
                raise torch.linalg.LinAlgError("orthonormalisation: the batch's Gram matrix is not positive definite "
                                               "(the reference's torch.inverse(R) fails on the same input)")

            self.scheduler.step(valid_loss)
            current_lr = float(self.optimizer.param_groups[0]["lr"])
            if hip_graph and self.optimizer.param_groups[0]["lr"] is not lr_t:   # the graphs read the rate from lr_t
                lr_t.fill_(current_lr)
                self.optimizer.param_groups[0]["lr"] = lr_t
            if current_lr <= self.spectral_config["min_lr"]:
                break
            print(f"Epoch: {epoch + 1}/{self.epochs}, Train Loss: {train_loss / batches:.7f}, "
                  f"Valid Loss: {valid_loss / batches_val:.7f}, LR: {current_lr:.6f}")

            if valid_loss <= min_val_loss:
                min_val_loss = valid_loss
                self.save_model()

        return self.spectral_net

    def save_model(self):
        os.makedirs(os.path.dirname(self.weights_path), exist_ok=True)  # the reference never creates weights/<dataset>/
        torch.save(self.spectral_net.state_dict(), self.weights_path)
