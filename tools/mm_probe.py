import torch, time
a=torch.randn(8192,51200,dtype=torch.float16,device='cuda'); w=torch.randn(8192,51200,dtype=torch.float16,device='cuda')
for _ in range(3): c=a@w.t()
torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): c=a@w.t()
e1.record(); torch.cuda.synchronize()
ms=e0.elapsed_time(e1)/5
print(f'torch fp16 8192x8192x51200: {ms:.3f} ms  {2*8192*8192*51200/ms/1e9:.1f} TFLOP/s')
a=torch.randn(256,256,dtype=torch.float64,device='cuda'); w=torch.randn(147456,256,dtype=torch.float64,device='cuda'); d=torch.randn(256,147456,dtype=torch.float64,device='cuda')
for name,fn,fl in (('fp64 dW = dpre^T z [147456x256] K=256', lambda: d.t()@a, 2*147456*256*256), ('fp64 dz = dpre W K=147456', lambda: d@w, 2*256*256*147456), ('fp64 heads fwd z W^T', lambda: a@w.t(), 2*256*256*147456)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/10
    print(f'torch {name}: {ms:.3f} ms {fl/ms/1e9:.1f} TFLOP/s')
# the merged weight-gradient GEMMs of a train step (K = 9 x 256 chains) and the input layer's shapes
K9 = 2304
for name, A, B, fl in (
        ('fp64 heads dW, K=2304: dpre9^T z9 [147456x256]', torch.randn(K9, 147456, dtype=torch.float64, device='cuda'), torch.randn(K9, 256, dtype=torch.float64, device='cuda'), 2 * 147456 * 256 * K9),
        ('fp64 input dW, K=2304: dpi9^T xf9 [256x131072]', torch.randn(K9, 256, dtype=torch.float64, device='cuda'), torch.randn(K9, 131072, dtype=torch.float64, device='cuda'), 2 * 256 * 131072 * K9)):
    fn = lambda: A.t() @ B
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 5
    print(f'torch {name}: {ms:.3f} ms {fl / ms / 1e9:.1f} TFLOP/s')
