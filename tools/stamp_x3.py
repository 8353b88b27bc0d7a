"""Development aid: phase stamps of the parity kernel's first tile (lib built with -DX3_TIMING): per unit
start / after the k loop / before the barrier / after the barrier, shader clocks of workgroup 0, thread 0."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dhaug_amd
from dhaug_amd import fused, _lib, ops
from dhaug_amd.function_aug.config import synth_args
from dhaug_amd.models_Fk_GAN import Fk_discriminator
B = 65536
args = synth_args(B, 256)
D3 = Fk_discriminator.Fk_3D_Discriminator("cuda", args).cuda()
x3 = torch.randn(B, 48, device="cuda") * 0.3
kcs = ops.kcs_forward(x3, True, f32=True)[0]
L = _lib.lib()
L.dhaug_debug_mlp_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_longlong * 256)()
with torch.no_grad():
    for _ in range(20):
        fused.critic3d(D3, x3, kcs=kcs, mode="f16x3")
    torch.cuda.synchronize()
    L.dhaug_debug_mlp_stamps(buf, 256)
st = [buf[i] for i in range(256)]
base = st[0]
for u in range(32):
    s0, s1, s2, s3, s4, s5 = st[8 * u:8 * u + 6]
    if not s0:
        continue
    if s1:
        print("unit %2d: start +%6d | k loop %6d | prefetch issue %5d | barrier A %5s | epilogue %6s | barrier B %5d | total %6d" %
              (u, s0 - base, s1 - s0, s4 - s1, (s5 - s4) if s5 else "-", (s2 - s5) if s5 else (s2 - s4), s3 - s2, s3 - s0))
    else:
        print("unit %2d: start +%6d | load %6d | barrier %5d" % (u, s0 - base, s2 - s0, s3 - s2))
print("tile total", max(st) - base)
