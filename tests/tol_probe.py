import re, sys, os, io, contextlib
import numpy as np, torch
sys.path.insert(0, os.getcwd())
dev = torch.device("cuda:0")
def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
g = np.load("tests/golden/g8_trainer.npz")
from sparc_amd.SpectralTrainer import SpectralTrainer
os.makedirs("/tmp/tolprobe", exist_ok=True); os.chdir("/tmp/tolprobe")
feats = [torch.tensor(g[f"feat{b}"]) for b in range(3)]
sups = [(g[f"coords{b}"], g[f"values{b}"], tuple(int(x) for x in g[f"shape{b}"])) for b in range(3)]
vf = [torch.tensor(g[f"vfeat{b}"]) for b in range(2)]
vs = [(g[f"vcoords{b}"], g[f"vvalues{b}"], tuple(int(x) for x in g[f"vshape{b}"])) for b in range(2)]
cfg = {"dataset": "golden", "spectral": {"architecture": {"n_layers": 4, "hidden_dim1": 32, "hidden_dim2": 32, "hidden_dim3": 16, "output_dim": 6},
       "epochs": 2, "lr": 1e-3, "lr_decay": 0.1, "min_lr": 1e-7, "batch_size": 1024, "n_clusters": 6, "patience": 10}}
torch.manual_seed(0); np.random.seed(0)
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    net = SpectralTrainer(cfg, dev).train(([np.arange(f.shape[0]) for f in feats], feats, sups, [None]*3, [None]*3), (list(range(2)), vf, vs, [None]*2, [None]*2))
nums = lambda l: [float(x) for x in re.findall(r"[-+]?\d+\.\d+", l)]
ours = [l for l in buf.getvalue().splitlines() if l.startswith("Epoch")]
ref = [l for l in str(g["stdout"]).splitlines() if l.startswith("Epoch")]
for a, b in zip(ours, ref):
    print("ours:", a); print("ref: ", b)
    na, nb = nums(a), nums(b)
    print("  rel err train", abs(na[0]-nb[0])/nb[0], "valid", abs(na[1]-nb[1])/nb[1])
for name, p in net.state_dict().items():
    print("final", name, rel(p.cpu().numpy(), g["final." + name]))
# how ill-conditioned are the orthonormalisation batches of this run?
from oracle import sparc_oracle as O
torch.manual_seed(0)
params = [torch.tensor(g["init." + n]) for n in ["layers.0.0.weight","layers.0.0.bias","layers.1.0.weight","layers.1.0.bias","layers.2.0.weight","layers.2.0.bias","layers.3.0.weight","layers.3.0.bias"]]
for b in range(3):
    Yt = O.mlp_forward(params, feats[b]).double().numpy()
    print("cond(Yt) of batch", b, "at the initial weights:", np.linalg.cond(Yt))

# ---- one optimisation step (g5) and the autoencoder stage (g9): observed errors
os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparc_amd.SpectralTrainer import SpectralNetLoss, SpectralNetModel
from sparc_amd.graph import ClusterBatch
g = np.load("tests/golden/g5_step.npz")
arch = {"n_layers": 4, "hidden_dim1": 32, "hidden_dim2": 32, "hidden_dim3": 16, "output_dim": 8}
net = SpectralNetModel(arch, 12, dev).to(dev)
net.load_state_dict({f"layers.{i}.0.{w}": torch.tensor(g[f"init.layers.{i}.0.{w}"]) for i in range(4) for w in ("weight", "bias")})
x1, x2 = torch.tensor(g["x1"], device=dev), torch.tensor(g["x2"], device=dev)
batch = ClusterBatch.from_support((g["sup1_coords"], g["sup1_values"], tuple(int(x) for x in g["sup1_shape"])), dev)
Y = net.forward_pair(x2, x1, top_idx=g["perm2"][:8])
loss = SpectralNetLoss()(batch, Y, is_normalized=True)
loss.backward()
print("step: loss", rel(loss.item(), g["loss"]), "Y", rel(Y.detach().cpu().numpy()[g["perm1"]], g["Y"]), "orth", rel(net.orthonorm_weights.detach().cpu().numpy(), g["orth"]))
for name, p in net.named_parameters():
    print("  grad", name, rel(p.grad.cpu().numpy(), g["grad." + name]))
print("cond of the orthonormalisation batch:", np.linalg.cond(O.mlp_forward([torch.tensor(g[f"init.layers.{i}.0.{w}"]) for i in range(4) for w in ("weight", "bias")], torch.tensor(g["x2"])).double().numpy()))
from sparc_amd.AETrainer import AETrainer
g9 = np.load("tests/golden/g9_ae.npz")
AE_CFG = {"dataset": "golden_ae", "should_use_ae": True, "should_use_siamese": False,
          "ae": {"architecture": {"hidden_dim1": 24, "hidden_dim2": 24, "hidden_dim3": 12, "output_dim": 5},
                 "epochs": 3, "n_samples": 1000, "lr": 1e-3, "lr_decay": 0.1, "min_lr": 1e-7, "patience": 10, "batch_size": 256}}
os.chdir("/tmp/tolprobe")
fb = [torch.tensor(g9[f"feat{b}"]) for b in range(3)]
vb = [torch.tensor(g9[f"vfeat{b}"]) for b in range(2)]
torch.manual_seed(1)
with contextlib.redirect_stdout(io.StringIO()):
    tr = AETrainer(AE_CFG, dev)
    ae = tr.train(fb, vb)
    enc, enc_val = tr.embed(fb, vb)
print("AE weights", max(rel(p.cpu().numpy(), g9["final." + n]) for n, p in ae.state_dict().items()),
      "codes", max(rel(enc[b].cpu().numpy(), g9[f"enc{b}"]) for b in range(3)), max(rel(enc_val[b].cpu().numpy(), g9[f"venc{b}"]) for b in range(2)))
