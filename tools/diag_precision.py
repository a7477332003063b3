"""Diagnostic (GPU box): per-stage error of the HIP path vs an fp64 evaluation, next to the reference's own fp32 error."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from oracle import ultr_oracle as O
from tests.hipref import HipRun, load_golden

name = sys.argv[1] if len(sys.argv) > 1 else "pairdebias_odd"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 0
d, m = load_golden(name)
Fs, hidden, L, B = m["F"], m["hidden"], m["L"], m["B"]
algo = {"na": "softmax", "ipw": "softmax", "dla": "dla", "pairdebias": "pairdebias", "lambdarank": "lambdarank"}[m["algo"]]
run = HipRun(Fs, hidden, B, L, algo=algo, learning_rate=m["lr"])
run.set_inputs(d["s%d_" % T + "features"], d["s%d_" % T + "docids"], d["s%d_" % T + "labels"])
scores = run.forward(d["s%d_" % T + "pre_params"])
p64 = torch.tensor(d["s%d_" % T + "pre_params"], dtype=torch.float64, requires_grad=True)
x64 = O.gather_rows(d["s%d_" % T + "features"], d["s%d_" % T + "docids"]).double()
s64 = O.dnn_forward(p64, Fs, hidden, x64).view(L, B).t()
print("scores: hip vs fp64 max abs %.3e ; golden(ref fp32) vs fp64 %.3e" % (np.abs(scores - s64.detach().numpy()).max(), np.abs(d["s%d_" % T + "scores"] - s64.detach().numpy()).max()))
aux = None
if m["algo"] in ("pairdebias", "lambdarank"):
    aux = np.concatenate([d["s%d_" % T + "pre_t_plus"].ravel(), d["s%d_" % T + "pre_t_minus"].ravel()])
ds, tail = run.loss(aux=aux, ipw_table=d["ipw_list"] if m["algo"] == "ipw" else None)
# fp64 dscores AT THE HIP SCORES
sh = torch.tensor(scores, dtype=torch.float64, requires_grad=True)
if m["algo"] == "pairdebias":
    c = torch.tensor(d["s%d_" % T + "labels"], dtype=torch.float64).t()
    mask = torch.minimum(torch.ones((), dtype=torch.float64), F.relu(c.unsqueeze(2) - c.unsqueeze(1)))
    pair = F.softplus(sh.unsqueeze(1) - sh.unsqueeze(2))
    tp64 = torch.tensor(aux[:L], dtype=torch.float64); tm64 = torch.tensor(aux[L:], dtype=torch.float64)
    PLw = lambda pair_: (float(B) * (mask * pair_).sum(0) * (1 - torch.eye(L, dtype=torch.float64)) / tp64.unsqueeze(1) / tm64.unsqueeze(0)).sum()
    loss = PLw(pair)
    (g,) = torch.autograd.grad(loss, sh)
    print("dscores: hip vs fp64(at hip scores) max abs %.3e (max |ds| %.3e)" % (np.abs(ds - g.numpy()).max(), np.abs(g.numpy()).max()))
# fp64 backward AT THE HIP dscores
(g64,) = torch.autograd.grad((s64 * torch.tensor(ds, dtype=torch.float64)).sum(), p64)
g64 = g64.numpy()
gh, _ = run.backward()
mx = np.abs(g64).max()
print("grads: hip vs fp64(at hip dscores): max abs/max %.3e" % (np.abs(gh - g64).max() / mx))
for n, s, o in O.param_layout(Fs, hidden):
    k = int(np.prod(s))
    print("  %-34s %-10s err/max %.3e" % (n, s, np.abs(gh[o:o + k] - g64[o:o + k]).max() / mx))
# end-to-end fp64 truth from the parameters
if m["algo"] == "pairdebias":
    p2 = torch.tensor(d["s%d_" % T + "pre_params"], dtype=torch.float64, requires_grad=True)
    s2 = O.dnn_forward(p2, Fs, hidden, x64).view(L, B).t()
    pair = F.softplus(s2.unsqueeze(1) - s2.unsqueeze(2))
    loss2 = PLw(pair)
    (ge,) = torch.autograd.grad(loss2, p2)
    ge = ge.numpy()
    mx = np.abs(ge).max()
    print("END-TO-END vs fp64: hip %.3e   reference(golden) %.3e   hip-vs-golden %.3e  (all /max|g|)" % (
        np.abs(gh - ge).max() / mx, np.abs(d["s%d_" % T + "grads"] - ge).max() / mx, np.abs(gh - d["s%d_" % T + "grads"]).max() / mx))
# ---- per-layer forward errors (saved activations / LayerNorm statistics) vs fp64 and vs torch fp32 ----
import ctypes
sv = run.eng.saved.cpu().numpy()
N = B * L
dims = O.layer_dims(Fs, hidden)
# saved layout: xs[j] (j>=1) each N*K_j rounded up to 4, then mean_j, rstd_j
off = 0; sx = {}
for j in range(1, len(dims)):
    sx[j] = off; off += N * dims[j][0]; off = (off + 3) // 4 * 4
sm, sr = {}, {}
for j in range(len(dims)):
    sm[j] = off; off += N; sr[j] = off; off += N
# our rows are list-major n = b*L + l ; oracle rows are position-major l*B + b
perm = (np.arange(N) % L) * B + np.arange(N) // L  # list-major row n -> position-major index
def fwd_layers(dtype):
    p = torch.tensor(d["s%d_" % T + "pre_params"], dtype=dtype)
    pp = O.unflatten(p, Fs, hidden)
    h = O.gather_rows(d["s%d_" % T + "features"], d["s%d_" % T + "docids"]).to(dtype)
    outs = []
    for j, (k, mm) in enumerate(dims):
        mu = h.mean(1); var = h.var(1, unbiased=False)
        outs.append((h.clone(), mu, 1.0 / torch.sqrt(var + 1e-5)))
        h = F.layer_norm(h, (k,), pp["sequential.layer_norm%d.weight" % j], pp["sequential.layer_norm%d.bias" % j], 1e-5)
        h = F.linear(h, pp["sequential.linear%d.weight" % j], pp["sequential.linear%d.bias" % j])
        if j != len(dims) - 1: h = F.elu(h)
    return outs
o64 = fwd_layers(torch.float64); o32 = fwd_layers(torch.float32)
for j in range(len(dims)):
    K = dims[j][0]
    mean_h = sv[sm[j]:sm[j] + N]; rstd_h = sv[sr[j]:sr[j] + N]
    m64 = o64[j][1].numpy()[perm]; r64 = o64[j][2].numpy()[perm]
    m32 = o32[j][1].numpy()[perm]; r32 = o32[j][2].numpy()[perm]
    line = "layer %d K=%d: mean err hip %.2e torch32 %.2e | rstd relerr hip %.2e torch32 %.2e (max rstd %.1f)" % (
        j, K, np.abs(mean_h - m64).max(), np.abs(m32 - m64).max(), np.abs(rstd_h / r64 - 1).max(), np.abs(r32 / r64 - 1).max(), r64.max())
    if j >= 1:
        xh = sv[sx[j]:sx[j] + N * K].reshape(N, K); x64 = o64[j][0].numpy()[perm]; x32 = o32[j][0].numpy()[perm]
        line += " | x err hip %.2e torch32 %.2e" % (np.abs(xh - x64).max(), np.abs(x32 - x64).max())
    print(line)
# final stage in fp64 fed with OUR saved x_k (fp32): isolates the accuracy of LN_k + dot from inherited error
j = len(dims) - 1
if j >= 1:
    K = dims[j][0]
    xk = torch.tensor(sv[sx[j]:sx[j] + N * K].reshape(N, K), dtype=torch.float64)
    pp = O.unflatten(torch.tensor(d["s%d_" % T + "pre_params"], dtype=torch.float64), Fs, hidden)
    u = F.layer_norm(xk, (K,), pp["sequential.layer_norm%d.weight" % j], pp["sequential.layer_norm%d.bias" % j], 1e-5)
    sc = F.linear(u, pp["sequential.linear%d.weight" % j], pp["sequential.linear%d.bias" % j]).view(-1).numpy()
    print("final stage alone (fp64 on hip x_k) vs hip scores: max abs %.3e" % np.abs(sc - scores.reshape(-1)).max())
    x32 = o32[j][0].double()
    u = F.layer_norm(x32, (K,), pp["sequential.layer_norm%d.weight" % j], pp["sequential.layer_norm%d.bias" % j], 1e-5)
    sc32 = F.linear(u, pp["sequential.linear%d.weight" % j], pp["sequential.linear%d.bias" % j]).view(-1).numpy()
    t32 = o32  # torch fp32 end result
    print("same for torch fp32 x_k -> fp64 final stage vs fp64 truth: %.3e (inherited error of torch's own x_k)" % np.abs(sc32[perm] - s64.detach().numpy().reshape(-1)).max())
    print("hip x_k -> fp64 final stage vs fp64 truth: %.3e (inherited error of hip x_k)" % np.abs(sc - s64.detach().numpy().reshape(-1)).max())
