#!/usr/bin/env python3
"""Generate tests/golden/g7_cora.npz: the reference's own data path on the shipped Cora graph
(data/cora/cora-{G,id_map,class_map}.json through src/utils.py:140-255 load_graphsage_data,
partition_graph with one cluster, preprocess_multicluster / preprocess), followed by the reference's
affinity, orthonormalisation, loss and backward on the m = 2437 training batch at k = 64.

cora-feats.npy is not shipped with the reference: the loader is given synthetic features
(RandomState(123).rand(2708, 1433) < 0.012), so only the GRAPH side is real Cora.
Runs only where /root/reference exists; imports the reference read-only (no bytecode, scratch cwd);
the .npz holds inputs and reference outputs only.  Large outputs are stored as row norms and a fixed
sample of entries to keep the file small.
"""
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

import numpy as np
import torch

REF = os.environ.get("SPARC_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    if not os.path.isdir(os.path.join(REF, "src")):
        print("reference not present; nothing to do")
        return 0
    metis = types.ModuleType("metis")   # absent third-party package; never called with one cluster
    metis.part_graph = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("METIS is not used with num_clusters = 1"))
    sys.modules["metis"] = metis
    sys.path.insert(0, os.path.join(REF, "src"))
    scratch = tempfile.mkdtemp(prefix="sparc_golden_cora_")
    os.chdir(scratch)
    os.makedirs("data/cora")
    for f in ("cora-G.json", "cora-id_map.json", "cora-class_map.json"):
        os.symlink(os.path.join(REF, "data", "cora", f), os.path.join("data", "cora", f))
    feats = (np.random.RandomState(123).rand(2708, 1433) < 0.012).astype(np.float32)
    np.save("data/cora/cora-feats.npy", feats)
    import warnings
    warnings.filterwarnings("ignore")
    import utils as ref_utils
    import partition_utils as ref_part
    import SpectralTrainer as ref_st
    torch.set_num_threads(8)

    (num_data, train_adj, full_adj, feats_s, _, _, labels, train_data, val_data, test_data) = \
        ref_utils.load_graphsage_data("/data", "cora")
    g = {"num_data": num_data, "n_train": len(train_data), "n_val": len(val_data), "n_test": len(test_data),
         "train_adj_nnz": train_adj.nnz, "full_adj_nnz": full_adj.nnz,
         "train_data": train_data, "val_data": val_data, "test_data": test_data}
    fa = full_adj.tocsr()
    off = fa - __import__("scipy.sparse", fromlist=["x"]).diags(fa.diagonal())
    g["offdiag_values"] = np.unique(off.data)
    g["diag_values"] = np.unique(fa.diagonal())
    g["full_indptr"], g["full_indices"], g["full_data"] = fa.indptr, fa.indices, fa.data
    ta = train_adj.tocsr()
    g["train_indptr"], g["train_indices"], g["train_data_values"] = ta.indptr, ta.indices, ta.data

    # src/main.py:103-116 with config/cora.json: one cluster, bsize 100000, diag_lambda -1
    y_train = np.zeros(labels.shape)
    y_train[train_data, :] = labels[train_data, :]
    train_mask = ref_utils.sample_mask(train_data, labels.shape[0])
    _, parts = ref_part.partition_graph(train_adj, train_data, 1)
    parts = [np.array(pt) for pt in parts]
    np.random.seed(0)
    fb, sb, yb, mb = ref_utils.preprocess_multicluster(train_adj, parts, feats_s, y_train, train_mask, 1, 100000, -1)
    coords, values, shape = sb[0]
    g["train_part"] = parts[0]
    g["sup_coords"], g["sup_values"], g["sup_shape"] = coords, values, np.asarray(shape)
    val_mask = ref_utils.sample_mask(val_data, labels.shape[0])
    _, vfb, vsb, _, _ = ref_utils.preprocess(full_adj, feats_s, np.zeros(labels.shape), val_mask, np.arange(num_data), 1, -1)
    vc, vv, vs = vsb[0]
    g["val_coords"], g["val_values"], g["val_shape"] = vc, vv, np.asarray(vs)
    g["val_rowsums"] = np.bincount(vc[:, 0], weights=vv, minlength=vs[0])

    # the reference's affinity, orthonormalisation, loss and backward on this batch, k = 64
    cfg = {"dataset": "cora", "spectral": {"architecture": {"n_layers": 4, "hidden_dim1": 1024, "hidden_dim2": 1024,
                                                            "hidden_dim3": 512, "output_dim": 64},
                                           "epochs": 1, "lr": 1e-3, "lr_decay": 0.1, "min_lr": 1e-7, "batch_size": 100000,
                                           "n_clusters": 7, "patience": 10}}
    trainer = ref_st.SpectralTrainer(cfg, torch.device("cpu"))
    crit = ref_st.SpectralNetLoss()
    m, k = shape[0], 64
    rng = np.random.default_rng(7)
    A = trainer._get_support_matrix((coords, values, shape))
    W = trainer._get_valid_affinity_matrix(A, None)
    g["W_rowsum"], g["W_colsum"] = W.sum(1).numpy(), W.sum(0).numpy()
    idx = rng.integers(0, m, size=(256, 2))
    g["W_sample_idx"], g["W_sample"] = idx, W[idx[:, 0], idx[:, 1]].numpy()
    # a well-conditioned pre-orthonormalisation output (the tanh of a random MLP is close to rank-deficient at k = 64)
    Yt = torch.tensor(rng.standard_normal((m, k)), dtype=torch.float32)
    model = ref_st.SpectralNetModel(cfg["spectral"]["architecture"], input_dim=32, device=torch.device("cpu"))
    Wo = model._make_orthonorm_weights(Yt)
    Y = (Yt @ Wo).detach().requires_grad_(True)
    loss = crit(W, Y, is_normalized=True)
    loss.backward()
    # Yt is not stored (624 KB): the test draws it again from the same generator state; its checksum is
    g["Yt_sum"], g["Yt_abs_sum"] = Yt.double().sum().numpy(), Yt.double().abs().sum().numpy()
    g["orth"] = Wo.numpy()
    g["loss"] = loss.detach().numpy()
    g["loss_plain"] = crit(W, Y.detach(), is_normalized=False).numpy()
    g["Y_rownorm"] = Y.detach().norm(dim=1).numpy()
    g["gradY_rownorm"] = Y.grad.norm(dim=1).numpy()
    rows = rng.integers(0, m, size=64)
    g["grad_rows"], g["gradY_rows"], g["Y_rows"] = rows, Y.grad[rows].numpy(), Y.detach()[rows].numpy()
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "g7_cora.npz"), **g)
    print({k_: (v.shape if hasattr(v, "shape") and getattr(v, "ndim", 0) else v) for k_, v in g.items() if k_ in
           ("num_data", "n_train", "n_val", "n_test", "train_adj_nnz", "full_adj_nnz", "offdiag_values", "diag_values", "sup_shape", "loss")})
    print("support nnz", len(values), "val support nnz", len(vv), "file", os.path.getsize(os.path.join(OUT, "g7_cora.npz")))
    return 0


if __name__ == "__main__":
    sys.exit(main())
