"""Times the generator tail (critics variant) at B = 65 536 with HIP events: the bench's forward step calls it as
ops.gen_tail_forward_critics(head, bl, None, pre, camera, rng=..., inputs_bf16=True)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
head = torch.randn(B, 35, device="cuda")
bl = torch.rand(B, 15, device="cuda") * 0.4 + 0.1
cam = ([0.5, 0.5, -0.5, 0.5], [0.0, 0.0, 5.0], [2.3, 2.3, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
x = torch.randn(4096, 4096, device="cuda")
for _ in range(50):
    x @ x                                          # clocks up
for mode in ("critics_bf16_draw", "critics_f32_scaler", "plain"):
    def run():
        if mode == "critics_bf16_draw":
            ops.gen_tail_forward_critics(head, bl, None, True, cam, rng=(1, 0), inputs_bf16=True)
        elif mode == "critics_f32_scaler":
            ops.gen_tail_forward_critics(head, bl, torch.zeros(B, 8, device="cuda"), True, cam)
        else:
            ops.gen_tail_forward(head, bl, None, True)
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    # 50 launches in one hipGraph: the host is out of the timed region
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        run()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(50):
                run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(8):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
    print("%s B=%d: %.2f us per launch" % (mode, B, best))
