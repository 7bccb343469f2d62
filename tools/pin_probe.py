import torch, time
n_v, n_f = 380000, 760000
v = torch.randn(n_v, 3, dtype=torch.float64, device='cuda'); f = torch.randint(0, n_v, (n_f, 3), device='cuda')
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    a, b = v.cpu().numpy(), f.cpu().numpy()
    t1 = time.perf_counter()
    pv = torch.empty(v.shape, dtype=v.dtype, pin_memory=True); pf = torch.empty(f.shape, dtype=f.dtype, pin_memory=True)
    t2 = time.perf_counter()
    pv.copy_(v, non_blocking=True); pf.copy_(f, non_blocking=True); torch.cuda.synchronize()
    a2, b2 = pv.numpy(), pf.numpy()
    t3 = time.perf_counter()
    f32 = f.to(torch.int32); torch.cuda.synchronize(); t4 = time.perf_counter()
    b3 = f32.cpu().numpy().astype('int64'); t5 = time.perf_counter()
    print(f'pageable .cpu(): {(t1-t0)*1e3:.2f} ms | pinned alloc {(t2-t1)*1e3:.2f} + copy {(t3-t2)*1e3:.2f} ms | int32 faces: cast {(t4-t3)*1e3:.2f} + d2h+astype {(t5-t4)*1e3:.2f}')
