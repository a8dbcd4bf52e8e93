import importlib, sys, numpy as np, torch
sys.path.insert(0, '.')
att = importlib.import_module('1xgpt_amd.attention')
M, N, K = 128, 128, 16
x = torch.ones(M, K, device='cuda')
W = torch.zeros(N, K, device='cuda'); W[:, 0] = torch.arange(N, device='cuda').float()
y = att.hip_linear(x, W, None)
torch.cuda.synchronize()
print(y[:4, :8]); print(y[60:68, 30:36]); print(torch.isfinite(y).all())
x = torch.arange(M, device='cuda').float()[:, None] * torch.ones(1, K, device='cuda')
W = torch.zeros(N, K, device='cuda'); W[:, 3] = 1
y = att.hip_linear(x, W, None); print(y[:6, :4]); print(y[100:104, 100:104])
g = torch.Generator(device='cpu').manual_seed(0)
x = torch.randn(256, 64, generator=g).cuda(); W = torch.randn(192, 64, generator=g).cuda()
y = att.hip_linear(x, W, None); ref = x.double() @ W.double().T
print('maxerr', (y - ref).abs().max().item())
