"""Development aid: which split launches (dhaug_split_bf16) and layer products one bf16x6 critic step of the bench shape (B = 65 536, D = 256) still
makes -- by shape, with the planes path on (critic_step.PLANES / PLANES_OUT)."""
import os, sys, argparse, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import golden_util as GU
import dhaug_amd
from dhaug_amd import critic_step as cs, ops, _lib
import test_gpu_critic_step as T
M = argparse.Namespace(dis=__import__("dhaug_amd.models_Fk_GAN.Fk_discriminator", fromlist=["x"]),
                       train=__import__("dhaug_amd.models_Fk_GAN.model_fk_gan_train", fromlist=["x"]), cs=cs)
tag = sys.argv[1] if len(sys.argv) > 1 else "d3"
B, D = int(os.environ.get("B", 65536)), 256
args = T._args(B, D)
sd = GU.seeded_state_dict(GU.shapes_d3(D) if tag == "d3" else GU.shapes_d2(D), 43)
data = T._data(tag, B, 14)
T._run(M, tag, args, sd, "bf16x6", data, True)          # (weights' operand copies made)
log = collections.Counter()
real = _lib.call
def spy(name, *a):
    if name == "dhaug_split_bf16":
        log[("split rows %d cols %d pad %d mode %d" % (a[3], a[4], a[5], a[6]))] += 1
    elif name.startswith("dhaug_gemm"):
        if name == "dhaug_gemm_bf16x6_planes":
            log["%s M %d N %d kp %d planes_out %s" % (name, a[14], a[15], a[16], bool(a[12]))] += 1
        elif name == "dhaug_gemm_tn_group_bf16_phase":
            arr, n = a[0], a[1]
            for i in range(n):
                log["tn item M %d N1 %d N2 %d planes %d %d" % (arr[i].M, arr[i].N1, arr[i].N2, arr[i].planes_a, arr[i].planes_b)] += 1
        else:
            log[name] += 1
    return real(name, *a)
_lib.call = spy
net, opt = T._net(M, tag, args, sd, "bf16x6")
r, f, al = data
M.train.train_Fk_discriminator(net, r.cuda(), f.cuda(), argparse.Namespace(train_iter_num=0), None, tag, opt, args, alpha=al.cuda())
torch.cuda.synchronize()
_lib.call = real
for k, v in sorted(log.items()):
    print("%3d x %s" % (v, k))
