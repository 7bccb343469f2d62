"""Eager launches vs replaying the captured step (scalars baked in: timing only) at cfg2 (run on the GPU box)."""
import sys, os, argparse, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

args = argparse.Namespace(keyframes=16, height=480, width=640, rays=4096, log2_T=19, mlp='baseline', precision='fp16x3', finest=256)
torch.cuda.set_device(0)
runner, cfg = bench.build_runner(args, 0, 1, torch.device('cuda', 0))
fld = runner.field
for _ in range(20):
    runner.train_loop(); runner.global_step += 1
torch.cuda.synchronize()
R = 4096
ids = runner.data_loader.next_ids().clone()


def step():
    fld.train_step(runner.rays, ids, R, seed=1)


def timed(fn, n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f'eager: {timed(step):.4f} ms/step')
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print(f'graph replay: {timed(g.replay):.4f} ms/step')
t0 = time.perf_counter()
for _ in range(200):
    step()
cpu = (time.perf_counter() - t0) / 200 * 1e3
torch.cuda.synchronize()
print(f'host time to enqueue one eager step: {cpu:.4f} ms')
