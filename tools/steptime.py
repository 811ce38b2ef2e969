import sys, time, importlib, os
sys.path.insert(0, os.getcwd())
import torch
lp = importlib.import_module("vlite-fast_amd.libpb")
import bench
S=10
h = lp.PbHandle(device=0, nant=1, nbit=8, npol=1, rfi_mode=2, fft_backend=lp.FFT_LDS, rows_per_seg=1024, max_seg=S, nsets=2)
n = h.seg_samples
dev=torch.device("cuda",0)
sec = bench.synth_second(torch, dev, 42, n, S, rfi_frac=0.0)
torch.cuda.synchronize()
for st in range(2):
    h.select_set(st)
    for s in range(S):
        h.submit_planar_dev(0, s, sec[s][0].data_ptr(), sec[s][1].data_ptr(), n)
h.sync()
ts=[]
for k in range(80):
    t0=time.perf_counter()
    h.select_set(k%2)
    h.process(S)
    t1=time.perf_counter()
    if k>=1:
        h.select_set((k-1)%2)
        for st in (0,1):
            v=h.fetch_view(0,st,S); x=int(v[0])+int(v[-1])
    t2=time.perf_counter()
    ts.append(((t1-t0)*1e3,(t2-t1)*1e3))
h.sync()
for k,(a,b) in enumerate(ts):
    if a > 0.2 or b > 2.0: print(k, "process %.3f ms  collect %.3f ms"%(a,b))
print("total", sum(a+b for a,b in ts))
