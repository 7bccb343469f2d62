"""Is the eager step host-bound?  GPU time per step vs host time to enqueue a step, and where the host time goes (cProfile),
at cfg2 (run on the GPU box)."""
import sys, os, argparse, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import contextlib
import torch
import bench

args = argparse.Namespace(keyframes=16, height=480, width=640, rays=4096, log2_T=19, mlp='baseline', precision='fp16x3', finest=256)
torch.cuda.set_device(0)
with contextlib.redirect_stdout(sys.stderr):
    runner, cfg = bench.build_runner(args, 0, 1, torch.device('cuda', 0))
for _ in range(50):
    runner.train_loop(); runner.global_step += 1
torch.cuda.synchronize()


def loop(n):
    for _ in range(n):
        runner.train_loop(); runner.global_step += 1


for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loop(400)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'400 steps: host enqueue {(t1 - t0) / 400 * 1e3:.4f} ms/step, until the GPU is done {(t2 - t0) / 400 * 1e3:.4f} ms/step')
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
loop(200)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18)
print('\n'.join(l[:150] for l in s.getvalue().splitlines()[:40]))
