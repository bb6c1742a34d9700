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
from .ops import (ACT_RELU, ACT_TANH, ORTH_CHOLESKY, ORTH_REFERENCE, LapLoss, LinearAct, OrthApply, OrthPair, OrthWeights,
                  SpectralMLP, SpectralMLPGram, SplitRows)


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
        self.grad_ready_hook = None   # hook(layer, dW, db) -> (dW, db), called during the backward (ops._layer_done)
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

    def embed_pre_orth(self, x: torch.Tensor, gram_rows: int = 0):
        """The MLP's output for the rows of x.  gram_rows > 0: (output, G) with G = output[:gram_rows]^T output[:gram_rows]
        in float64 when the output layer's kernel took the product along, None otherwise."""
        kinds = [isinstance(layer[1], nn.Tanh) for layer in self.layers]
        if kinds[-1] and not any(kinds[:-1]) and x.dtype == torch.float32:   # ReLU ... ReLU, Tanh: the fused MLP
            params = [p for layer in self.layers for p in (layer[0].weight, layer[0].bias)]
            if gram_rows > 0:
                y, G = SpectralMLPGram.apply(x.contiguous(), self.grad_ready_hook, gram_rows, *params)
                return y, (G if G.numel() else None)
            return SpectralMLP.apply(x.contiguous(), self.grad_ready_hook, *params)
        for layer, is_tanh in zip(self.layers, kinds):
            x = LinearAct.apply(x, layer[0].weight, layer[0].bias, ACT_TANH if is_tanh else ACT_RELU)
        return (x, None) if gram_rows > 0 else x

    def forward(self, x: torch.Tensor, should_update_orth_weights: bool = True, top_idx=None) -> torch.Tensor:
        Y_tilde = self.embed_pre_orth(x)
        if should_update_orth_weights:
            self.orthonorm_weights = self._make_orthonorm_weights(Y_tilde, top_idx)
        return OrthApply.apply(Y_tilde, self.orthonorm_weights)

    def forward_pair(self, x_orth: torch.Tensor, x_grad: torch.Tensor, top_idx=None) -> torch.Tensor:
        """The two forwards of one training step (src/SpectralTrainer.py:222-223 then :232) as ONE pass
        of the MLP over both batches: the parameters do not change between them, so the rows are the
        same; the orthonormalisation weights come from x_orth's rows and are applied to x_grad's."""
        m2, m1 = x_orth.shape[0], x_grad.shape[0]
        if m2 == 0 or m1 == 0:
            Yt = self.embed_pre_orth(torch.cat([x_orth, x_grad], dim=0))
            Yt_orth, Yt_grad = SplitRows.apply(Yt, m2)
            self.orthonorm_weights = self._make_orthonorm_weights(Yt_orth, top_idx)
            return OrthApply.apply(Yt_grad, self.orthonorm_weights)
        # the Gram matrix of the orthogonality batch's rows comes out of the output layer's launch
        Yt, G = self.embed_pre_orth(torch.cat([x_orth, x_grad], dim=0), gram_rows=m2)
        Y, self.orthonorm_weights = OrthPair.apply(Yt, m2, top_idx, self.orth_mode, G)   # one graph node for both parts
        return Y

    def set_rotation_matrix(self, rotation_matrix):
        self.rotation_matrix = rotation_matrix


class SpectralNetLoss(nn.Module):
    """Rayleigh quotient of the batch Laplacian (src/SpectralTrainer.py:111-139).

    W is the ClusterBatch returned by SpectralTrainer._get_valid_affinity_matrix, or the dense
    m x m affinity itself, as the reference passes it (:116): a dense matrix has no sparsity for
    the kernels to use, so it takes the same quadratic form with one library product.
    """

    def forward(self, W, Y: torch.Tensor, is_normalized: bool = False) -> torch.Tensor:
        if isinstance(W, ClusterBatch):
            return LapLoss.apply(Y, W, bool(is_normalized))
        if torch.is_tensor(W) and W.dim() == 2 and W.shape[0] == W.shape[1] == Y.shape[0] and W.is_floating_point():
            return self._dense(W, Y, bool(is_normalized))
        raise TypeError("W must be a sparc_amd.graph.ClusterBatch (build one with ClusterBatch.from_support) or the "
                        "dense m x m affinity matrix of the batch")

    @staticmethod
    def _dense(W: torch.Tensor, Y: torch.Tensor, is_normalized: bool) -> torch.Tensor:
        """sum_ij W_ij |y_i - y_j|^2 / (2 m)  =  [ 1/2 sum_i (d_i + c_i) |y_i|^2 - sum_i y_i . (W y)_i ] / m  with the row
        sums d and column sums c of W (src/SpectralTrainer.py:130-137, where it is written with torch.cdist)."""
        require_gpu(W.device)
        m = Y.shape[0]
        W = W.detach().to(Y.dtype)
        d = W.sum(dim=1)
        if is_normalized:
            Y = Y / torch.sqrt(d[:, None])
        half = 0.5 * (d + W.sum(dim=0))
        return (Y * (half[:, None] * Y - W @ Y)).sum() / m    # tr(Y^T L Y) / m, L = 1/2 (D_r + D_c) - W


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
        self.orth_weights_path = f"./weights/{self.dataset}/spectralnet_orthonorm_weights.pth"
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

    def _get_valid_affinity_matrix(self, W, X=None):
        """A ClusterBatch comes back as it is: its affinity vectors were prepared together with the CSR.  A dense
        support matrix (what the reference's _get_support_matrix returns) is taken through the reference's own
        dense sequence (src/SpectralTrainer.py:288-291) with library products; SpectralNetLoss accepts the result."""
        if torch.is_tensor(W):
            require_gpu(W.device)
            W0 = W + 1e-6
            W2 = W0 @ W0
            Wp = W0 + W2 + W2 @ W0
            Wp.fill_diagonal_(0)
            return torch.nn.functional.normalize(Wp, p=2, dim=1)
        return W

    # -- set-up steps of train(), separate so that the epoch loop can be driven on stand-ins ---
    def _setup_model(self, input_dim: int):
        self.criterion = SpectralNetLoss()
        self.spectral_net = SpectralNetModel(self.architecture, input_dim=input_dim, device=self.device,
                                             orth_mode=self.orth_mode).to(self.device)

    def _setup_optimizer(self, hip_graph: bool):
        self._lr_tensor = None
        if hip_graph:   # the graphs read the learning rate from a device tensor
            self._lr_tensor = torch.tensor(float(self.lr), device=self.device)
            self.optimizer = optim.Adam(self.spectral_net.parameters(), lr=self._lr_tensor, fused=True, capturable=True)
        else:
            if torch.device(self.device).type == "cuda":   # all parameter tensors in one launch that fills the chip
                from .optim import Adam
                self.optimizer = Adam(self.spectral_net.parameters(), lr=self.lr)
            else:
                self.optimizer = optim.Adam(self.spectral_net.parameters(), lr=self.lr)
        self.scheduler = optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode="min", factor=self.lr_decay,
                                                              patience=self.patience)

    def _prepare_batches(self, k: int, mine, mine_val):
        """One-time: this rank's features and supports to the device (train batches `mine`, validation batches
        `mine_val`, both lists of indices into the tuples train() received)."""
        measure = self.spectral_config.get("kernel_choice", "rule") == "measure"
        sizes = self._cluster_sizes()
        feats = [self.features_batches[b].to(self.device, dtype=torch.float32) for b in mine]
        sup = [self._get_support_matrix(self.support_batches[b], sizes[b] if sizes else None).warm(k, measure=measure)
               for b in mine]
        vfeats = [self.val_features_batches[b].to(self.device, dtype=torch.float32) for b in mine_val]
        vsup = [self._get_support_matrix(self.val_support_batches[b]).warm(k, training=False, measure=measure)
                for b in mine_val]
        return feats, sup, vfeats, vsup

    def train(self, train, val) -> SpectralNetModel:
        """The reference's loop (src/SpectralTrainer.py:170-272).  Inside an initialised torch.distributed process
        group of more than one rank (one process per GPU) the loop is data-parallel: the training and validation
        batches are dealt round-robin to the ranks (each rank prepares and keeps only its own), every rank takes the
        same number of steps per epoch on its own batches, the MLP gradient is averaged over the ranks layer by
        layer during the backward (the only collective of a step), the epoch's loss sums and the validation loss are
        summed over the ranks before the scheduler sees them, and rank 0 alone prints and writes the checkpoint.
        Without a process group (or with one rank) it is the reference's sequential loop, draw for draw.

        Config keys beyond the reference's, all optional: spectral.orth_mode ("reference" | "cholesky"),
        spectral.hip_graph (whole steps replayed as HIP graphs; one GPU), spectral.kernel_choice ("rule": the kernel
        of a batch follows from its structure, so runs repeat bit for bit; "measure": the candidates are timed on
        the batch and the fastest kept)."""
        require_gpu(self.device)
        self.parts, self.features_batches, self.support_batches, self.y_train_batches, self.train_mask_batches = train
        self.idx_parts, self.val_features_batches, self.val_support_batches, self.y_val_batches, self.val_mask_batches = val
        self.idx_parts = list(range(len(self.parts)))
        self.counter = 0
        OrthWeights.check_and_reset(self.device)   # failures of earlier, unrelated calls do not belong to this run
        self._setup_model(self.features_batches[0].shape[1])
        world, rank = self._world()
        # spectral.hip_graph (not a reference key; default off): whole steps replayed as HIP graphs, one launch per step
        hip_graph = bool(self.spectral_config.get("hip_graph", False)) and world == 1
        self._setup_optimizer(hip_graph)
        k = self.architecture["output_dim"]
        mine, mine_val = self._my_share(len(self.features_batches), len(self.val_features_batches), world, rank)
        feats, sup, vfeats, vsup = self._prepare_batches(k, mine, mine_val)
        graphed = None
        if hip_graph:
            from .stepgraph import GraphedSteps
            graphed = GraphedSteps(self.spectral_net, self.criterion, self.optimizer, feats, sup, k, self.device)
        return self._run_epochs(feats, sup, vfeats, vsup, k, graphed, n_val_total=len(self.val_features_batches))

    @staticmethod
    def _world():
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(), dist.get_rank()
        return 1, 0

    @staticmethod
    def _my_share(n_train: int, n_val: int, world: int, rank: int):
        """Indices of the training and validation batches this rank owns (round-robin).  A rank that would be left
        without a training batch (more ranks than batches) trains on one of the others' so that every rank takes
        part in every collective; validation batches are never doubled (their losses are summed over the ranks)."""
        from .parallel import shard_batches
        mine = shard_batches(n_train, world, rank) or [rank % max(n_train, 1)]
        return mine, shard_batches(n_val, world, rank)

    def _run_epochs(self, feats, sup, vfeats, vsup, k: int, graphed=None, n_val_total=None):
        world, rank = self._world()
        reducer = None
        if world > 1:
            from .parallel import LayerwiseReducer, all_reduce_sum, broadcast
            params = list(self.spectral_net.parameters())
            for p in params:                      # every replica starts from rank 0's weights
                broadcast(p.data, 0)
            reducer = LayerwiseReducer(params)
            self.spectral_net.grad_ready_hook = reducer   # (called as hook(layer, dW, db); its destination() lets the backward write into the buckets)
        min_val_loss = 1000000000
        batches = len(feats)
        batches_val = len(vfeats) if n_val_total is None else n_val_total
        steps = batches
        if world > 1:                             # the ranks take the same number of steps: the largest share's
            t = torch.tensor([batches], dtype=torch.int64)
            t = t.to(self.device) if torch.distributed.get_backend() == "nccl" else t
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            steps = int(t)
        if rank == 0:
            print("Training SpectralNet:")
        for epoch in range(self.epochs):
            train_loss = torch.zeros((), dtype=torch.float64, device=self.device)
            for step in range(steps):
                pid = step % batches
                pid_2 = np.random.choice(batches)
                np.random.permutation(len(feats[pid]))           # drawn as the reference does (:211), not applied
                perm_2 = np.random.permutation(len(feats[pid_2]))  # (:218) only its head matters

                # Orthogonality step (autograd stays on, as in the reference :222-223) and the
                # forward of the gradient step (:232), one MLP pass over both batches
                if graphed is not None:
                    train_loss += graphed.step(pid, pid_2, perm_2[:k])
                    continue
                self.spectral_net.train()
                self.optimizer.zero_grad(set_to_none=True)
                Y = self.spectral_net.forward_pair(feats[pid_2], feats[pid], top_idx=perm_2[:k])
                W = self._get_valid_affinity_matrix(sup[pid], feats[pid])
                loss = self.criterion(W, Y, is_normalized=True)
                loss.backward(gradient=LapLoss.unit_gradient(loss))
                if reducer is not None and reducer.finish() == 0:
                    from .parallel import all_reduce_mean_gradients
                    all_reduce_mean_gradients(list(self.spectral_net.parameters()))
                self.optimizer.step()
                train_loss += loss.detach()

            valid_loss = torch.zeros((), dtype=torch.float64, device=self.device)
            self.spectral_net.eval()
            with torch.no_grad():
                for pid in range(len(vfeats)):
                    Y = self.spectral_net(vfeats[pid], should_update_orth_weights=False)
                    W = self._get_valid_affinity_matrix(vsup[pid], vfeats[pid])
                    valid_loss += self.criterion(W, Y, is_normalized=True)
            failed = OrthWeights.worst(self.device)
            n_steps = steps
            if world > 1:     # one small collective per epoch: both loss sums and the failure flag of every rank
                packed = torch.stack([train_loss, valid_loss, (failed.double().reshape(()) if failed is not None
                                                                else torch.zeros_like(train_loss))])
                all_reduce_sum(packed)
                train_loss, valid_loss = packed[0], packed[1]
                if failed is not None:
                    OrthWeights.set_worst(self.device, int(packed[2] != 0))
                n_steps = steps * world
            train_loss, valid_loss = float(train_loss), float(valid_loss)  # the epoch's only host sync
            if OrthWeights.check_and_reset(self.device) != 0:   # any step of the epoch, not only the last
                raise torch.linalg.LinAlgError("orthonormalisation: the batch's Gram matrix is not positive definite "
                                               "(the reference's torch.inverse(R) fails on the same input)")

            self.scheduler.step(valid_loss)
            current_lr = float(self.optimizer.param_groups[0]["lr"])
            lr_t = self._lr_tensor
            if lr_t is not None and self.optimizer.param_groups[0]["lr"] is not lr_t:   # the graphs read the rate from lr_t
                lr_t.fill_(current_lr)
                self.optimizer.param_groups[0]["lr"] = lr_t
            if current_lr <= self.spectral_config["min_lr"]:
                break
            if rank == 0:
                print(f"Epoch: {epoch + 1}/{self.epochs}, Train Loss: {train_loss / n_steps:.7f}, "
                      f"Valid Loss: {valid_loss / batches_val:.7f}, LR: {current_lr:.6f}")

            if valid_loss <= min_val_loss:
                min_val_loss = valid_loss
                if rank == 0:
                    self.save_model()

        return self.spectral_net

    def save_model(self):
        """The reference's checkpoint (state_dict of the MLP, src/SpectralTrainer.py:339-343) and, in a file next to
        it, the orthonormalisation weights of the last training step: they are an attribute of the reference's module,
        not a registered buffer (:24), so a reloaded state_dict alone cannot predict."""
        os.makedirs(os.path.dirname(self.weights_path), exist_ok=True)  # the reference never creates weights/<dataset>/
        torch.save(self.spectral_net.state_dict(), self.weights_path)
        torch.save({"orthonorm_weights": self.spectral_net.orthonorm_weights.detach().cpu()}, self.orth_weights_path)

    def load_model(self, input_dim: int) -> SpectralNetModel:
        """A SpectralNetModel with the checkpoint's MLP weights and, when the file next to it exists, the
        orthonormalisation weights that were current when the checkpoint was written."""
        self._setup_model(input_dim)
        self.spectral_net.load_state_dict(torch.load(self.weights_path, map_location=self.device, weights_only=True))
        if os.path.exists(self.orth_weights_path):
            extra = torch.load(self.orth_weights_path, map_location=self.device, weights_only=True)
            self.spectral_net.orthonorm_weights = extra["orthonorm_weights"].to(self.device)
        return self.spectral_net
