"""timing decomposition of pk_patch_embed_splitk (PK_PATCH_DBG bits: 1 no MFMAs, 2 no W stream, 4 no A conversion); run once per setting"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phenaki_pytorch_amd import _lib as L
B, C, F, H, W, N = 8, 3, 17, 256, 256, 512
video = torch.randn(B, C, F, H, W, device='cuda')
spec = []
for f0, nt, pt in ((1, 8, 2), (0, 1, 1)):
    P = C * pt * 32 * 32
    rows = B * nt * 64
    ns = L.patch_embed_slices(P)
    spec.append((torch.randn(N, P, device='cuda').to(torch.bfloat16), torch.empty(ns, rows, N, device='cuda'), torch.empty(ns, rows, 2, device='cuda'), f0, nt, pt))
for _ in range(3):
    L.patch_embed_splitk(video, 32, 32, N, spec)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    L.patch_embed_splitk(video, 32, 32, N, spec)
e1.record()
torch.cuda.synchronize()
print(f"PK_PATCH_DBG={os.environ.get('PK_PATCH_DBG', '0')}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch")
