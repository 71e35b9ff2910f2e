import argparse, sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
args = argparse.Namespace(recipe=sys.argv[1] if len(sys.argv) > 1 else "timit_mlp", T=500, B=128, prec="bf16", algo="auto", layers=None,
                          mask_rng="device", overlap=False, torch_optim=False)
tr = bench.Trainer(args, 0, 1)
for i in range(3):
    tr.step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(50):
    tr.step(i)
torch.cuda.synchronize()
print("eager ms/step", 1e3 * (time.perf_counter() - t0) / 50)
# capture one whole step on a static input
static_inp = tr.batches[0].clone()
tr.batches = [static_inp]
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(3):
        tr.step(0)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = tr.step(0)
torch.cuda.synchronize()
g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200):
    g.replay()
torch.cuda.synchronize()
print("graph ms/step", 1e3 * (time.perf_counter() - t0) / 200, "loss", float(loss))
