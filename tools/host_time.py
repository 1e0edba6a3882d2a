import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vae_play_amd as V
from vae_play_amd import optim
from vae_play_amd.engine import FusedVAEStep
torch.manual_seed(0)
vae = V.VAE(128, 128, 3).cuda()
opt = optim.Adam(vae.parameters(), lr=1e-4)
st = FusedVAEStep(vae, opt, 32, 128, 3)
x, eps = torch.rand(32, 3, 128, 128, device="cuda"), torch.randn(32, 128, device="cuda")
for _ in range(10): st.step(x, eps)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20): st.step(x, eps)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0)/20:.3f} ms/step, total {1e3*(t2-t0)/20:.3f} ms/step")
st.capture()
for _ in range(5): st.step(x, eps)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20): st.step(x, eps)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"graph: enqueue {1e3*(t1-t0)/20:.3f} ms/step, total {1e3*(t2-t0)/20:.3f} ms/step")
