import time, torch, numpy as np
dev = torch.device('cuda')
x = torch.empty(64*1024*1024, dtype=torch.float64, device=dev)   # 512 MB
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); p = torch.empty(64*1024*1024, dtype=torch.float64, pin_memory=True); t1 = time.perf_counter()
    p.copy_(x, non_blocking=True); torch.cuda.synchronize(); t2 = time.perf_counter()
    p.copy_(x, non_blocking=True); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f'pinned alloc 512MB {1e3*(t1-t0):.1f} ms; first D2H {1e3*(t2-t1):.1f} ms; second D2H {1e3*(t3-t2):.1f} ms = {0.512/(t3-t2):.1f} GB/s')
    del p
for rep in range(3):
    t0 = time.perf_counter(); h = np.empty(64*1024*1024); t1 = time.perf_counter()
    ht = torch.from_numpy(h); ht.copy_(x); torch.cuda.synchronize(); t2 = time.perf_counter()
    ht.copy_(x); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f'pageable np.empty {1e3*(t1-t0):.1f} ms; first D2H {1e3*(t2-t1):.1f} ms; second D2H {1e3*(t3-t2):.1f} ms = {0.512/(t3-t2):.1f} GB/s')
    del h, ht
a = np.empty(64*1024*1024); b = np.ones(64*1024*1024)
t0 = time.perf_counter(); a[:] = b; t1 = time.perf_counter(); a[:] = b; t2 = time.perf_counter()
print(f'host memcpy 512MB: first {1e3*(t1-t0):.1f} ms, second {1e3*(t2-t1):.1f} ms = {0.512/(t2-t1):.1f} GB/s')
