"""developer aid: time srbh_trunk_wgrad alone (all 69 RDBs' weight + bias gradients as one launch + its reduce) on synthetic planes.
python tools/time_trunk_wgrad.py [B]   (SRBH_LIB_PATH selects a variant build, e.g. -DTW_ABL=1/2/4: phase ablations, timing only)"""
import sys, time, torch
sys.path.insert(0, '.')
from srbh_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = _lib.lib()
dev = 'cuda:0'
nblk, H, W = 23, 64, 64
n = nblk * 3
nb = (L.srbh_act16_bytes(B, 192, H, W) + 255) // 256 * 256
D = (torch.randn((n + 1) * nb // 2, device=dev) * 0.5).to(torch.float16).view(torch.uint8)
G = (torch.randn((n + 1) * nb // 2, device=dev) * 0.01).to(torch.bfloat16).view(torch.uint8)
ws = torch.empty(L.srbh_trunk_wgrad_ws_bytes(nblk, B, H, W), dtype=torch.uint8, device=dev)
dw = torch.empty(n * 9 * 26624, device=dev); db = torch.empty(n * 192, device=dev)
def run():
    _lib.check(L.srbh_trunk_wgrad(nblk, D.data_ptr(), nb, G.data_ptr(), nb, B, H, W, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), _lib.stream_ptr()), "trunk_wgrad")
for _ in range(3): run()
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 20
for _ in range(N): run()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / N * 1e3
gf = 2.0 * B * H * W * 9 * 26624 * n / 1e9
print(f"B={B}: {ms:.3f} ms per call (kernel + reduce), {gf / ms:.0f} TFLOP/s")
