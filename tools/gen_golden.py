#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the imported reference on the CPU.

Runs only where /root/reference exists (the development container); it is a
no-op elsewhere.  The reference's modules are imported read-only with bytecode
writing disabled and a scratch working directory, because the reference writes
./weights relative to the cwd.  The absent third-party `metis` package is
replaced by a stub whose part_graph() returns a planted assignment: the METIS
cut is an *input* of the path under test, not part of it.

The .npz files hold inputs and the reference's outputs only (no source text).
"""
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

import numpy as np
import scipy.sparse as sp
import torch

REF = os.environ.get("SPARC_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

PLANTED = {"groups": None}


def _stub_metis():
    mod = types.ModuleType("metis")

    def part_graph(adj_lists, nparts, seed=None):
        g = PLANTED["groups"]
        assert g is not None and len(g) == len(adj_lists)
        return 0, list(g)

    mod.part_graph = part_graph
    sys.modules["metis"] = mod


def sbm_adjacency(n, blocks, p_in, p_out, rng, weight2_frac=0.3):
    """Symmetric adjacency with unit diagonal and off-diagonal weights in {1,2}."""
    lab = np.sort(rng.integers(0, blocks, size=n))
    P = np.where(lab[:, None] == lab[None, :], p_in, p_out)
    up = np.triu(rng.random((n, n)) < P, 1)
    w = np.where(rng.random((n, n)) < weight2_frac, 2.0, 1.0)
    A = np.triu(up * w, 1)
    A = A + A.T + np.eye(n)
    return sp.csr_matrix(A.astype(np.float32)), lab


def tiny_config(k, dataset="golden"):
    return {
        "dataset": dataset, "n_clusters": k, "should_use_ae": False, "should_use_siamese": False,
        "spectral": {
            "architecture": {"n_layers": 4, "hidden_dim1": 32, "hidden_dim2": 32, "hidden_dim3": 16, "output_dim": k},
            "epochs": 2, "lr": 1e-3, "lr_decay": 0.1, "min_lr": 1e-7, "batch_size": 1024,
            "n_neighbors": 8, "scale_k": 2, "is_local_scale": False, "n_clusters": k, "patience": 10,
            "bsize": 2, "num_clusters": 4, "num_clusters_val": 2, "num_clusters_test": 1, "diag_lambda": -1,
        },
    }


def main():
    if not os.path.isdir(os.path.join(REF, "src")):
        print("reference not present; nothing to do")
        return 0
    _stub_metis()
    sys.path.insert(0, os.path.join(REF, "src"))
    os.makedirs(OUT, exist_ok=True)
    scratch = tempfile.mkdtemp(prefix="sparc_golden_")
    os.chdir(scratch)
    os.makedirs("weights/golden", exist_ok=True)
    import warnings
    warnings.filterwarnings("ignore")
    import utils as ref_utils
    import partition_utils as ref_part
    import SpectralTrainer as ref_st
    torch.set_num_threads(4)

    # ---------------- G1: support construction --------------------------------
    rng = np.random.default_rng(11)
    n = 240
    adj, lab = sbm_adjacency(n, 6, 0.15, 0.01, rng)
    feats = rng.standard_normal((n, 5))
    y = np.eye(6)[lab]
    mask = rng.random(n) < 0.8
    idx_nodes = np.nonzero(mask)[0].astype(np.int32)
    train_adj = sp.csr_matrix(adj.toarray() * np.outer(mask, mask))
    g1 = {"adj_indptr": adj.indptr, "adj_indices": adj.indices, "adj_data": adj.data, "n": n,
          "mask": mask, "idx_nodes": idx_nodes, "labels": lab}
    num_clusters = 6
    groups = (lab[idx_nodes]).tolist()
    PLANTED["groups"] = groups
    part_adj, parts = ref_part.partition_graph(train_adj, idx_nodes, num_clusters)
    g1["train_indptr"], g1["train_indices"], g1["train_data"] = train_adj.indptr, train_adj.indices, train_adj.data
    g1["groups"] = np.asarray(groups)
    g1["part_adj_indptr"], g1["part_adj_indices"], g1["part_adj_data"] = part_adj.indptr, part_adj.indices, part_adj.data
    for li, lam in enumerate([-1, 0.01]):
        np.random.seed(5)
        ps = [np.array(p) for p in parts]
        fb, sb, yb, mb = ref_utils.preprocess_multicluster(train_adj, ps, feats, y, mask, num_clusters, 2, lam)
        g1[f"mc{li}_nbatch"] = len(sb)
        for b, (coords, values, shape) in enumerate(sb):
            g1[f"mc{li}_coords{b}"] = coords
            g1[f"mc{li}_values{b}"] = values
            g1[f"mc{li}_shape{b}"] = np.asarray(shape)
            g1[f"mc{li}_feat{b}"] = fb[b].numpy()
            g1[f"mc{li}_mask{b}"] = mb[b]
        g1[f"mc{li}_parts_after_shuffle"] = np.concatenate(ps)
        g1[f"mc{li}_parts_sizes"] = np.asarray([len(p) for p in ps])
    # validation flavour: partition the FULL graph, normalise intra-part adjacency
    for li, lam in enumerate([-1, 0.01]):
        PLANTED["groups"] = (lab % 3).tolist()
        parts_v, fbv, sbv, ybv, mbv = ref_utils.preprocess(adj, feats, y, mask, np.arange(n), 3, lam)
        for b, (coords, values, shape) in enumerate(sbv):
            g1[f"val{li}_coords{b}"] = coords
            g1[f"val{li}_values{b}"] = values
            g1[f"val{li}_shape{b}"] = np.asarray(shape)
            g1[f"val{li}_part{b}"] = np.asarray(parts_v[b])
    # degree arrays as the reference computes them (src/utils.py:23, :33)
    g1["rowsum_raw"] = np.array(train_adj.sum(1)).flatten()
    g1["rowsum_enh"] = np.array((train_adj + sp.eye(n)).sum(1)).flatten()
    np.savez_compressed(os.path.join(OUT, "g1_support.npz"), **g1)

    # ---------------- G2/G4: affinity + loss ----------------------------------
    cfg = tiny_config(8)
    trainer = ref_st.SpectralTrainer(cfg, torch.device("cpu"))
    crit = ref_st.SpectralNetLoss()
    g2 = {}
    m = 150
    rng = np.random.default_rng(3)
    sub, _ = sbm_adjacency(m, 3, 0.12, 0.01, rng)
    # make a few isolated nodes (only the unit diagonal survives)
    dense = sub.toarray()
    for iso in (7, 91):
        dense[iso, :] = 0
        dense[:, iso] = 0
        dense[iso, iso] = 1
    sub = sp.csr_matrix(dense)
    flavours = {
        "raw": ref_utils.sparse_to_tuple(sub),
        "rownorm": ref_utils.sparse_to_tuple(ref_utils.normalize_adj(sp.csr_matrix((sub.toarray() > 0) * 1))),
        "enh": ref_utils.sparse_to_tuple(ref_utils.normalize_adj_diag_enhance(sub, 0.01)),
    }
    Yk = torch.tensor(rng.standard_normal((m, 8)), dtype=torch.float32)
    for name, sup in flavours.items():
        A = trainer._get_support_matrix(sup)
        W = trainer._get_valid_affinity_matrix(A, None)
        g2[f"{name}_coords"], g2[f"{name}_values"], g2[f"{name}_shape"] = sup[0], sup[1], np.asarray(sup[2])
        g2[f"{name}_W"] = W.numpy()
        g2[f"{name}_d"] = W.sum(1).numpy()
        g2[f"{name}_colsum"] = W.sum(0).numpy()
        g2[f"{name}_loss_norm"] = crit(W, Yk, is_normalized=True).numpy()
        g2[f"{name}_loss_plain"] = crit(W, Yk, is_normalized=False).numpy()
    g2["Y"] = Yk.numpy()
    np.savez_compressed(os.path.join(OUT, "g2_affinity_loss.npz"), **g2)

    # ---------------- G3: orthonormalisation ----------------------------------
    g3 = {}
    torch.manual_seed(4)
    model = ref_st.SpectralNetModel(cfg["spectral"]["architecture"], input_dim=12, device=torch.device("cpu"))
    cases = {
        "gauss": torch.randn(200, 8),
        "mlp": None,
        "wide": torch.randn(90, 16),
    }
    with torch.no_grad():
        cases["mlp"] = torch.nn.Sequential(*[l for l in model.layers])(torch.randn(300, 12))
    for name, Yt in cases.items():
        Wo = model._make_orthonorm_weights(Yt)
        _, R = torch.linalg.qr(Yt)
        Yo = Yt @ Wo
        g3[f"{name}_Yt"] = Yt.numpy()
        g3[f"{name}_orth"] = Wo.numpy()
        g3[f"{name}_signs"] = torch.sign(torch.diag(R)).numpy()
        g3[f"{name}_ortho_err"] = (Yo.t() @ Yo / Yt.shape[0] - torch.eye(Yt.shape[1])).abs().max().numpy()
        g3[f"{name}_cond"] = np.linalg.cond(Yt.numpy().astype(np.float64))
    np.savez_compressed(os.path.join(OUT, "g3_orth.npz"), **g3)

    # ---------------- G5/G6: one optimisation step + validation ----------------
    g5 = {}
    rng = np.random.default_rng(21)
    m1, m2, din, k = 180, 140, 12, 8
    sup1 = ref_utils.sparse_to_tuple(sbm_adjacency(m1, 3, 0.1, 0.01, rng)[0])
    supv = ref_utils.sparse_to_tuple(ref_utils.normalize_adj(sp.csr_matrix((sbm_adjacency(m2, 2, 0.1, 0.01, rng)[0].toarray() > 0) * 1)))
    x1 = torch.tensor(rng.standard_normal((m1, din)), dtype=torch.float32)
    x2 = torch.tensor(rng.standard_normal((m2, din)), dtype=torch.float32)
    perm1 = rng.permutation(m1)
    perm2 = rng.permutation(m2)
    torch.manual_seed(9)
    net = ref_st.SpectralNetModel(cfg["spectral"]["architecture"], input_dim=din, device=torch.device("cpu"))
    for name, p in net.state_dict().items():
        g5["init." + name] = p.numpy().copy()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    # loop body of SpectralTrainer.train, src/SpectralTrainer.py:207-238, driven on reference objects
    fb = x1
    W = trainer._get_support_matrix(sup1)
    fb = fb[perm1]
    W = W[perm1][:, perm1]
    fb2 = x2[perm2]
    net.eval()
    net(fb2, should_update_orth_weights=True)
    net.train()
    opt.zero_grad()
    Y = net(fb, should_update_orth_weights=False)
    Wv = trainer._get_valid_affinity_matrix(W, fb)
    loss = crit(Wv, Y, is_normalized=True)
    loss.backward()
    for name, p in net.named_parameters():
        g5["grad." + name] = p.grad.numpy().copy()
    opt.step()
    for name, p in net.state_dict().items():
        g5["after." + name] = p.numpy().copy()
    g5["loss"] = loss.detach().numpy()
    g5["Y"] = Y.detach().numpy()
    g5["orth"] = net.orthonorm_weights.detach().numpy()
    g5["orth_requires_grad"] = np.asarray(net.orthonorm_weights.requires_grad)
    g5["x1"], g5["x2"], g5["perm1"], g5["perm2"] = x1.numpy(), x2.numpy(), perm1, perm2
    g5["sup1_coords"], g5["sup1_values"], g5["sup1_shape"] = sup1[0], sup1[1], np.asarray(sup1[2])
    # G6: validation loss with the carried orth weights, after the Adam step
    net.eval()
    with torch.no_grad():
        Wd = trainer._get_support_matrix(supv)
        Yv = net(x2, should_update_orth_weights=False)
        Wd = trainer._get_valid_affinity_matrix(Wd, x2)
        g5["val_loss"] = crit(Wd, Yv, is_normalized=True).numpy()
        g5["val_Y"] = Yv.numpy()
    g5["supv_coords"], g5["supv_values"], g5["supv_shape"] = supv[0], supv[1], np.asarray(supv[2])
    np.savez_compressed(os.path.join(OUT, "g5_step.npz"), **g5)

    # ---------------- G8: the reference trainer end to end (2 epochs) ----------
    g8 = {}
    rng = np.random.default_rng(33)
    nb, din = 3, 10
    feats_b, sups = [], []
    for b in range(nb):
        mb = 90 + 10 * b
        sups.append(ref_utils.sparse_to_tuple(sbm_adjacency(mb, 3, 0.12, 0.01, rng)[0]))
        feats_b.append(torch.tensor(rng.standard_normal((mb, din)), dtype=torch.float32))
    vf, vs = [], []
    for b in range(2):
        mb = 70 + 15 * b
        vs.append(ref_utils.sparse_to_tuple(ref_utils.normalize_adj(sp.csr_matrix((sbm_adjacency(mb, 2, 0.1, 0.01, rng)[0].toarray() > 0) * 1))))
        vf.append(torch.tensor(rng.standard_normal((mb, din)), dtype=torch.float32))
    for b in range(nb):
        g8[f"feat{b}"], g8[f"coords{b}"], g8[f"values{b}"], g8[f"shape{b}"] = feats_b[b].numpy(), sups[b][0], sups[b][1], np.asarray(sups[b][2])
    for b in range(2):
        g8[f"vfeat{b}"], g8[f"vcoords{b}"], g8[f"vvalues{b}"], g8[f"vshape{b}"] = vf[b].numpy(), vs[b][0], vs[b][1], np.asarray(vs[b][2])
    parts = [np.arange(f.shape[0]) for f in feats_b]
    train = (parts, feats_b, sups, [None] * nb, [None] * nb)
    val = (list(range(2)), vf, vs, [None] * 2, [None] * 2)
    import io
    import contextlib
    torch.manual_seed(0)
    np.random.seed(0)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        tr = ref_st.SpectralTrainer(tiny_config(6), torch.device("cpu"))
        net = tr.train(train, val)
    for name, p in net.state_dict().items():
        g8["final." + name] = p.numpy().copy()
    g8["final_orth"] = net.orthonorm_weights.detach().numpy()
    g8["stdout"] = np.asarray(buf.getvalue())
    torch.manual_seed(0)
    init = ref_st.SpectralNetModel(tiny_config(6)["spectral"]["architecture"], input_dim=din, device=torch.device("cpu"))
    for name, p in init.state_dict().items():
        g8["init." + name] = p.numpy().copy()
    # the facade's predict on all nodes with the trained model (src/SpectralNet.py:80-98)
    import SpectralNet as ref_sn
    with contextlib.redirect_stdout(io.StringIO()):
        facade = ref_sn.SpectralNet(n_clusters=6, config=tiny_config(6))
    facade.device = torch.device("cpu")
    facade.spec_net = net
    Xall = torch.cat(feats_b + vf, dim=0)
    g8["predict_X"] = Xall.numpy()
    g8["predict_Y"] = facade.predict(Xall).copy()
    np.savez_compressed(os.path.join(OUT, "g8_trainer.npz"), **g8)
    print(buf.getvalue())

    # ---------------- G9: the autoencoder stage (src/AETrainer.py), 3 epochs ----------
    import AETrainer as ref_ae
    g9 = {}
    rng = np.random.default_rng(44)
    din9 = 20
    fb = [torch.tensor(rng.standard_normal((70 + 9 * b, din9)), dtype=torch.float32) for b in range(3)]
    vb = [torch.tensor(rng.standard_normal((40 + 5 * b, din9)), dtype=torch.float32) for b in range(2)]
    cfg9 = {"dataset": "golden_ae", "ae": {"architecture": {"hidden_dim1": 24, "hidden_dim2": 24, "hidden_dim3": 12, "output_dim": 5},
                                           "epochs": 3, "n_samples": 1000, "lr": 1e-3, "lr_decay": 0.1, "min_lr": 1e-7,
                                           "patience": 10, "batch_size": 256}}
    os.makedirs("weights/golden_ae", exist_ok=True)
    torch.manual_seed(1)
    buf9 = io.StringIO()
    with contextlib.redirect_stdout(buf9):
        tr9 = ref_ae.AETrainer(cfg9, torch.device("cpu"))
        ae = tr9.train(fb, vb)
        enc, enc_val = tr9.embed(fb, vb)
    for b in range(3):
        g9[f"feat{b}"], g9[f"enc{b}"] = fb[b].numpy(), enc[b].numpy()
    for b in range(2):
        g9[f"vfeat{b}"], g9[f"venc{b}"] = vb[b].numpy(), enc_val[b].numpy()
    for name, p in ae.state_dict().items():
        g9["final." + name] = p.numpy().copy()
    g9["stdout"] = np.asarray(buf9.getvalue())
    np.savez_compressed(os.path.join(OUT, "g9_ae.npz"), **g9)
    print(buf9.getvalue())

    # ---------------- G10: token lists of the downstream models (implementations/SPARCphormer/utils.py) ----
    # the module imports dgl (absent) at load time and never uses it on this path: stubbed like metis
    import importlib.util
    dgl = types.ModuleType("dgl")
    dgl.DGLGraph = object
    sys.modules.setdefault("dgl", dgl)
    spec = importlib.util.spec_from_file_location("sparcphormer_utils", os.path.join(REF, "implementations", "SPARCphormer", "utils.py"))
    ph = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ph)
    rng = np.random.default_rng(55)
    n10, K10 = 150, 9
    enc10 = rng.standard_normal((n10, 5))
    feat10 = rng.standard_normal((n10, 7)).astype(np.float32)
    mask10 = np.ones(n10, dtype=bool)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        tok = ph.re_features_spectral_diffusion_distance_avarage_seq_spectral_encoding_coldstart(
            enc10, feat10, K10, mask10, mask10, mask10)
    np.savez_compressed(os.path.join(OUT, "g10_tokens.npz"), encoding=enc10, features=feat10, K=np.asarray(K10),
                        tokens=tok.numpy())
    print("golden vectors written to", OUT)
    return 0


if __name__ == "__main__":
    sys.exit(main())
