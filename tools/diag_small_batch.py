import importlib, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from golden_util import rel_err
F_ = importlib.import_module("pytorch-kaldi_amd.functional")
nn_amd = importlib.import_module("pytorch-kaldi_amd.nn")
optim_ = importlib.import_module("pytorch-kaldi_amd.optim")
opts = {"dnn_lay": "1024,1024,1024,1024,200", "dnn_drop": "0.15,0.15,0.15,0.15,0.0", "dnn_use_laynorm_inp": "False",
        "dnn_use_batchnorm_inp": "False", "dnn_use_batchnorm": "True,True,True,True,False",
        "dnn_use_laynorm": "False,False,False,False,False", "dnn_act": "relu,relu,relu,relu,softmax"}
g = torch.Generator().manual_seed(8)
x = torch.randn(128, 440, generator=g).cuda()
lab = torch.randint(0, 200, (128,), generator=g).cuda()
F_.set_precision("bf16")
KEYS = ("mlp_fused", "mlp_fused_bwd", "direct_grads", "gemm_skinny")  # keys of PK_EXPERIMENT
def run(on):
    # (gemm_skinny is read once by the library: it takes the value of the FIRST run of this process)
    os.environ["PK_EXPERIMENT"] = ",".join("%s=%d" % (k, k in on) for k in KEYS)
    torch.manual_seed(3)
    net = nn_amd.MLP(opts, 440).cuda().train()
    flat = optim_.FlatParams(net)
    masks = [(torch.rand(128, 1024, generator=torch.Generator().manual_seed(50 + i)) > 0.15).float() for i in range(4)]
    F_.set_forced_dropout([m.cuda() for m in masks])
    flat.zero_grad()
    out = net(x)
    loss = torch.nn.functional.nll_loss(out, lab)
    loss.backward()
    torch.cuda.synchronize()
    F_.set_forced_dropout(None)
    return float(loss), {k: q.grad.detach().clone() for k, q in net.named_parameters()}
l0, ref = run(())
l1, got = run(("mlp_fused",))
for k in ref:
    if float(ref[k].abs().max()) > 1e-5:
        d = (got[k] - ref[k]).abs()
        print(k, "rel_err %.2e" % rel_err(got[k], ref[k]), "ratio of norms %.5f" % float(got[k].norm() / ref[k].norm()),
              "cos %.6f" % float((got[k] * ref[k]).sum() / (got[k].norm() * ref[k].norm())), "n>1e-2*max %d" % int((d > 1e-2 * ref[k].abs().max()).sum()))
# a torch reference of the same net (fp32 math on the bf16-rounded operands of each Linear)
def bf(t):
    return t.to(torch.bfloat16).float()
torch.manual_seed(3)
net = nn_amd.MLP(opts, 440).cuda().train()
ps = {k: q.detach().clone().requires_grad_(True) for k, q in net.named_parameters()}
masks = [(torch.rand(128, 1024, generator=torch.Generator().manual_seed(50 + i)) > 0.15).float().cuda() / 0.85 for i in range(4)]
class Rnd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return bf(t)
    @staticmethod
    def backward(ctx, g):
        return bf(g)
h = x
for i in range(4):
    zz = torch.nn.functional.linear(Rnd.apply(h), Rnd.apply(ps["wx.%d.weight" % i]), ps.get("wx.%d.bias" % i))
    mu, var = zz.mean(0), zz.var(0, unbiased=False)
    h = torch.relu((zz - mu) / torch.sqrt(var + 1e-5) * ps["bn.%d.weight" % i] + ps["bn.%d.bias" % i]) * masks[i]
out = torch.log_softmax(torch.nn.functional.linear(Rnd.apply(h), Rnd.apply(ps["wx.4.weight"]), ps["wx.4.bias"]), 1)
torch.nn.functional.nll_loss(out, lab).backward()
for name, gr in (("unfused", ref), ("fused", got)):
    print(name, "vs torch model:", ["%s %.1e" % (k, rel_err(gr[k], ps[k].grad)) for k in ("wx.0.weight", "wx.2.weight", "wx.4.weight", "bn.1.weight", "bn.3.bias")])
