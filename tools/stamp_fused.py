"""Development aid: per-unit shader-clock stamps of the fused kernel (needs a lib built with -DDHAUG_MLP_TIMING)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dhaug_amd
from dhaug_amd import fused, _lib
from dhaug_amd.function_aug.config import synth_args
from dhaug_amd.models_Fk_GAN import Fk_discriminator, Fk_generator, forward_kinematics_DH_model as fkm
B = int(os.environ.get("STAMP_B", "65536"))
args = synth_args(B, 256)
fk = fkm.Forward_Kinematics_DH_Model(args, ["S1"], None)
G = Fk_generator.Fk_Generator(fk, args, "cuda").cuda()
D3 = Fk_discriminator.Fk_3D_Discriminator("cuda", args).cuda()
D2 = Fk_discriminator.Fk_2D_Discriminator(args, 16).cuda()
x2 = torch.randn(B, 16, 2, device="cuda") * 0.3
z = torch.randn(B, 128, device="cuda"); x3 = torch.randn(B, 16, 3, device="cuda") * 0.3
L = _lib.lib()
N = 6 * 32 + 69
buf = (ctypes.c_longlong * N)()
with torch.no_grad():
    for name, fn in (("G", lambda: fused.generator_head(G, z)), ("D3", lambda: fused.critic3d(D3, x3)), ("D2", lambda: fused.critic2d(D2, x2)), ("D3", lambda: fused.critic3d(D3, x3))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        L.dhaug_debug_mlp_stamps(buf, N)
        st = [buf[i] for i in range(N)]
        idx = [i for i, v in enumerate(st[:34]) if v]
        base = st[idx[0]]
        print(name, " ".join("%d:%d" % (i, st[i] - base) for i in idx))
        # stamps inside the last stack of the last tile (index - 32: see gemm_stack / stack_layer)
        idx2 = [i for i, v in enumerate(st[32:96], 32) if v]
        if idx2:
            b2 = min(st[i] for i in idx2)
            print("   stack:", " ".join("%d:%d" % (i - 32, st[i] - b2) for i in idx2))
        # last tile, per unit: start, [generic GEMM: after the k loop, after the epilogue], after the trailing barrier
        q = st[228:261]
        idq = [i for i, v in enumerate(q) if v]
        if idq:
            print("   later tile:", " ".join("%d:%d" % (i, q[i] - q[idq[0]]) for i in idq))
        us = [u for u in range(32) if st[96 + 4 * u]]
        if us:
            b3 = st[96 + 4 * us[0]]
            print("   units(last tile):", " | ".join("%d: %s" % (u, " ".join(str(st[96 + 4 * u + j] - b3) if st[96 + 4 * u + j] else "-" for j in range(4))) for u in us))
