import sys, time, torch
sys.path.insert(0, '/root/repo')
from flasht5_amd import AdamWScale
dev = 'cuda'
shapes = [(32768, 768)] * 2 + [(768, 768)] * (4 * 36) + [(2048, 768)] * (3 * 24) + [(768,)] * 62 + [(32, 12)] * 2
params = [torch.nn.Parameter((torch.randn(*sh, device=dev) * 0.02).bfloat16()) for sh in shapes]
for p_ in params: p_.grad = (torch.randn_like(p_) * 0.01)
opt = AdamWScale(params, lr=1e-3, weight_decay=0.01, kahan_sum=True)
for _ in range(3): opt.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): opt.step()
th = time.perf_counter() - t0
torch.cuda.synchronize()
t = time.perf_counter() - t0
print("host %.3f ms/step, total %.3f ms/step" % (th / 10 * 1e3, t / 10 * 1e3))
