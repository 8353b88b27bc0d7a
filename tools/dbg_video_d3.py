"""Development aid: first D3 critic step of the video loop golden, GPU path vs the oracle, per parameter."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import golden_util as GU, loop_util as LU
from oracle import dhaug_oracle as O
import dhaug_amd
from dhaug_amd.models_Fk_GAN import Fk_discriminator as dis, model_fk_gan_train as T
from test_gpu_models import make_args

g = np.load(os.path.join(ROOT, "tests/golden/video_loop_D32.npz")); g = {k: torch.from_numpy(g[k]) for k in g.files}
B, R = 8, 9
sds = LU.video_state_dicts(g, R=R)
nets = dict(G=O.Net(sds["G"], lambda z, p, bl, sc: O.generator_forward(z, p, bl, sc, frames=R)[0]), d3=O.Net(sds["d3"], O.d3_forward))
bl = O.bone_lengths(g["real3d"][0].reshape(-1, 16, 3))
cp = g["cam_param"]
cR = cp[:, 9:13].unsqueeze(1).repeat(1, R, 1).reshape(-1, 4); cT = cp[:, 13:16].unsqueeze(1).repeat(1, R, 1).reshape(-1, 3)
rw = O.camera_to_world(g["real3d"][0].reshape(-1, 16, 3), cR, cT); real = (rw - rw[:, :1]).reshape(-1, 48)
with torch.no_grad():
    fw = nets["G"].fwd(g["noise"][0], nets["G"].p, bl, torch.as_tensor(g["scaler"][0])).reshape(-1, 16, 3)
fake = (fw - fw[:, :1]).reshape(-1, 48)
alpha = g["alpha_000"]
print("alpha", alpha.shape, "real", real.shape)
d3 = nets["d3"]; d3.zero_grad()
gp = O.gradient_penalty(d3, real.reshape(B * R, -1), fake.reshape(B * R, -1), alpha)
(d3(fake).mean() - d3(real).mean() + gp).backward()
ref_g = d3.grads()
args = make_args(batch_size=B, single_or_multi_train_mode="multi", architecture="3,3")
for prec in ("bf16x6", "bf16x3", "bf16"):
    net = dis.Fk_3D_Discriminator("cuda", args); net.load_state_dict(sds["d3"]); net.precision = prec; net = net.cuda()
    opt = T.FusedAdam(net.parameters(), lr=1e-4, betas=(0.5, 0.9))
    class S: train_iter_num = 0
    W, C = T.train_Fk_discriminator(net, real.cuda(), fake.cuda(), S(), None, "x", opt, args, alpha=alpha.cuda())
    print(prec, "W %.6f C %.6f" % (W.item(), C.item()))
    for k, p in net.named_parameters():
        r = ref_g[k]; e = (p.grad.cpu() - r).abs()
        print("  %-34s max|g| %.2e  err max %.2e  rel %.2e" % (k, r.abs().max(), e.max(), e.max() / (r.abs().max() + 1e-30)))
    net2 = dis.Fk_3D_Discriminator("cuda", args); net2.load_state_dict(sds["d3"]); net2.precision = prec; net2 = net2.cuda()
    gp2 = dis.calc_gradient_penalty(net2, real.cuda(), fake.cuda(), B * R, 10, "cuda", alpha=alpha.cuda())
    print("  gp build %.7f oracle %.7f" % (gp2.item(), gp.item()))
    with torch.no_grad():
        l = net2(real.cuda()); lo = d3(real)
        print("  logits err", (l.cpu() - lo).abs().max().item())
