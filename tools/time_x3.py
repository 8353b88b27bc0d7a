"""Parity-grade (f16x3) fused programs at B = 65 536, D = 256: timing of the three networks.  If the library also exports
dhaug_mlp_forward_x3_r3 (the round-3 kernel, kept in the tree while this round's was developed -- git history has it), the two
are run in interleaved rounds in one process and the difference of their outputs is printed."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import ops, fused, _lib
from dhaug_amd.function_aug.config import synth_args
from dhaug_amd.models_Fk_GAN import model_fk_gan_train as T
from dhaug_amd.models_Fk_GAN.forward_kinematics_DH_model import Forward_Kinematics_DH_Model

B = int(os.environ.get("B", 65536)); D = 256
args = synth_args(B, D)
d = T.my_get_poseFk_model(args, None, Forward_Kinematics_DH_Model(args, ["S1"], None))
G, D3, D2 = d["model_G"], d["model_d3d"], d["model_d2d"]
z = torch.randn(B, 128, device="cuda")
x3 = torch.randn(B, 48, device="cuda") * .3
x2 = torch.rand(B, 32, device="cuda") - .5
mac = dict(G=128 * D + 6 * D * D + 35 * D, D3=78 * D + 12 * D * D + 200 * D + 2 * 100 * 100 + 100, D2=32 * D + 4 * D * D + D)
fns = dict(G=lambda: fused.generator_head(G, z, mode="f16x3"), D3=lambda: fused.critic3d(D3, x3, mode="f16x3"),
           D2=lambda: fused.critic2d(D2, x2, mode="f16x3"))
L = _lib.lib()
have_r3 = hasattr(L, "dhaug_mlp_forward_x3_r3")
if have_r3:
    L.dhaug_mlp_forward_x3_r3.argtypes = _lib.SIGNATURES["dhaug_mlp_forward_x3"]
    L.dhaug_mlp_forward_x3_r3.restype = ctypes.c_int
new_fn = L.dhaug_mlp_forward_x3


def use(which):
    _lib._fn["dhaug_mlp_forward_x3"] = L.dhaug_mlp_forward_x3_r3 if which == "r3" else new_fn


def timeit(fn, n=100):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


with torch.no_grad():
    for name in ("G", "D3", "D2"):
        use("new")
        a = fns[name]()
        a = a[0] if isinstance(a, tuple) else a
        if have_r3:
            use("r3")
            b = fns[name]()
            b = b[0] if isinstance(b, tuple) else b
            print("%s: max |new - r3| = %.3e  (scale %.3e)  nan %d" % (name, (a - b).abs().max().item(), b.abs().mean().item(), int(torch.isnan(a).sum())))
        else:
            print("%s: mean |out| %.3e nan %d" % (name, a.abs().mean().item(), int(torch.isnan(a).sum())))
    use("new")
    for _ in range(200):
        fns["D3"]()                                        # clocks up
    for name in ("G", "D3", "D2"):
        res = {"new": [], "r3": []}
        for rnd in range(5):
            for which in (("new", "r3") if have_r3 else ("new",)):
                use(which)
                res[which].append(timeit(fns[name]))
        fl = 2.0 * mac[name] * B
        for which in res:
            if res[which]:
                t = sorted(res[which])
                print("%-3s %-3s median %7.1f us  min %7.1f us  %6.1f TFLOP/s algorithmic  executed frac %.3f" %
                      (name, which, t[len(t) // 2], t[0], fl / (t[len(t) // 2] * 1e-6) / 1e12, 3 * fl / (t[len(t) // 2] * 1e-6) / 2.5e15))
    use("new")
