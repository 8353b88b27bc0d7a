"""CPU-only checks of the boundary and the host logic (no compute calls: there is no GPU here)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.build_lib(verbose=False)
    ge.build_hostcheck(verbose=False)
    return ge


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "dhaug.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(dhaug_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol(built):
    import dhaug_amd
    lib = ctypes.CDLL(dhaug_amd._lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libdhaug.so does not export %s" % n
    lib.dhaug_version.restype = ctypes.c_int
    lib.dhaug_arch.restype = ctypes.c_char_p
    assert lib.dhaug_version() == 100 and lib.dhaug_arch() == b"gfx950"
    # every declared compute entry point has a ctypes signature in the binding (and nothing else is bound)
    bound = set(dhaug_amd._lib.SIGNATURES)
    assert bound == set(names) - {"dhaug_version", "dhaug_arch"}


def test_scratch_sizes_mirror_the_header(built):
    """the binding's scratch sizes are the header's (a short scratch is an out-of-bounds write on the device)"""
    from dhaug_amd import ops
    hdr = open(os.path.join(ROOT, "include", "dhaug.h")).read()
    assert int(re.search(r"#define\s+DHAUG_CRITIC_SCALARS_SCRATCH\s+(\d+)", hdr).group(1)) == ops.CRITIC_SCALARS_SCRATCH
    from dhaug_amd import fused
    expr = re.search(r"#define\s+DHAUG_MLP_X3_WORKSPACE_BYTES\s+\(([0-9 *]+)\)", hdr).group(1)
    assert eval(expr) == fused.X3_WORKSPACE_BYTES


def test_ablation_switches_need_an_ablation_build(tmp_path):
    """a development switch ("timing only, results wrong") cannot reach a product library: every such -D is a compile error
    without -DDHAUG_ABLATION_BUILD (csrc/dhaug_common.h), and build_lib ignores DHAUG_EXTRA_HIPFLAGS unless the environment
    names an ablation build"""
    import subprocess
    import __graft_entry__ as G
    assert G.ablation_flags({"DHAUG_EXTRA_HIPFLAGS": "-DX3_ABL_NOSPLIT"}) == []
    assert G.ablation_flags({}) == []
    assert G.ablation_flags({"DHAUG_ABLATION_BUILD": "1", "DHAUG_EXTRA_HIPFLAGS": "-DX3_ABL_NOSPLIT"}) == ["-DDHAUG_ABLATION_BUILD", "-DX3_ABL_NOSPLIT"]
    src = tmp_path / "probe.cpp"
    src.write_text('#include "dhaug_common.h"\nint main() { return 0; }\n')
    base = [G.HIPCC, "-x", "hip", "--cuda-host-only", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), "-I" + G.CSRC]
    ok = subprocess.run(base + [str(src)], capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr[-2000:]
    for d in ("X3_ABL_NOSPLIT", "ABL_NOWRITE", "SAVE_ABL_NULLSTORES", "T4_ABL_NOCOMPUTE", "X3_NWAVES=4", "DHAUG_MLP_TIMING"):
        bad = subprocess.run(base + ["-D" + d, str(src)], capture_output=True, text=True)
        assert bad.returncode != 0 and "DHAUG_ABLATION_BUILD" in bad.stderr, d
        good = subprocess.run(base + ["-D" + d, "-DDHAUG_ABLATION_BUILD", str(src)], capture_output=True, text=True)
        assert good.returncode == 0, (d, good.stderr[-2000:])
    # the shipped sources name no switch the guard does not know: every *_ABL_* / X3_* / timing macro tested by an #if / #ifdef
    guard = open(os.path.join(G.CSRC, "dhaug_common.h")).read()
    known = set(re.findall(r"defined\((\w+)\)", guard))
    internal = {"X3_SHAPE16", "X3_NW", "X3_MT", "X3_BM", "X3_B16", "X3_THREADS", "X3_LDS_BYTES", "X3_MAX_UNITS", "X3_CASE", "X3_CASE_ADD",
                "X3_CASE_STASH", "X3_STAMP", "X3_WORKSPACE_BYTES"}                     # defined by the sources themselves, unconditionally
    for path in sorted(p for p in os.listdir(G.CSRC) if p.endswith((".hip", ".h"))):
        text = open(os.path.join(G.CSRC, path)).read()
        for mac in set(re.findall(r"^\s*#\s*(?:if|ifdef|ifndef|elif)[^\n]*?\b((?:\w*_ABL_\w+|ABL_\w+|X3_\w+|\w*TIMING\w*|\w+_OVERRIDE|W_NO_\w+))\b", text, re.M)):
            assert mac in known or mac in internal, (path, mac)


def test_argument_errors_are_returned_not_thrown(built):
    """host-side validation happens before any launch, so it can be exercised without a GPU"""
    import dhaug_amd
    L = dhaug_amd._lib.lib()
    assert L.dhaug_fk_forward(None, None, None, None, 4, 16, None) == -1             # null pointers
    assert L.dhaug_fk_forward(None, None, None, None, 0, 16, None) == 0              # empty batch is a no-op
    assert L.dhaug_fk_forward(None, None, None, None, 4, 17, None) == -1             # bad out_joints
    buf = (ctypes.c_float * 64)()
    mis = ctypes.c_void_p(ctypes.addressof(buf) + 4)
    assert L.dhaug_fk_forward(mis, mis, mis, mis, 1, 16, None) == -2                 # misaligned
    assert L.dhaug_gemm_bf16(mis, 8, mis, 8, None, None, 0, None, 0, None, 0, 0, None, 0, 4, 4, 24, 0, 0.0, None) in (-1, -3)
    # the round-6 entries: argument checks of the planes products and of the contraction's planes fields (no launch is reached)
    big = (ctypes.c_float * 4096)()
    a16 = ctypes.c_void_p((ctypes.addressof(big) + 15) & ~15)
    assert L.dhaug_gemm_bf16x6_planes(a16, 768, a16, 1536, None, None, 0, None, 0, 0, 0.0, a16, 256, None, 0, 4, 256, 256, 3, 0, 0.0, None) == -1      # x_order 3
    assert L.dhaug_gemm_bf16x6_planes(a16, 768, a16, 1536, None, None, 0, None, 0, 0, 0.0, a16, 256, None, 0, 4, 256, 192, 0, 0, 0.0, None) == -3      # piece width not 64 * 2^j
    assert L.dhaug_gemm_bf16x6_planes(a16, 512, a16, 1536, None, None, 0, None, 0, 0, 0.0, a16, 256, None, 0, 4, 256, 256, 0, 0, 0.0, None) == -2      # lda < 3 kp
    assert L.dhaug_gemm_bf16x6_planes(a16, 768, a16, 1536, None, None, 0, None, 0, 0, 0.0, a16, 256, None, 0, 0, 256, 256, 0, 0, 0.0, None) == 0       # empty batch
    assert L.dhaug_gemm_f16x3_planes(a16, 512, 1, a16, 768, None, None, 0, a16, 256, None, 0, 0, 4, 256, 200, 0, 0.0, None) == -3                        # planes need 64 * 2^j
    assert L.dhaug_gemm_f16x3_planes(a16, 512, 1, a16, 768, None, None, 0, a16, 256, a16, 256, 256, 4, 256, 256, 0, 0.0, None) == -2                     # ld_planes < 2 planes_kp
    assert L.dhaug_split_bf16(a16, 64, a16, 4, 64, 64, 2, 3, None) == -1                                                                                # mode 2 is a six-term layout
    lay = (dhaug_amd._lib.TnLayer * 1)()
    lay[0].A, lay[0].lda, lay[0].B, lay[0].ldb = a16.value, 64, a16.value, 64
    lay[0].C, lay[0].ldc, lay[0].M, lay[0].N1, lay[0].N2 = a16.value, 64, 64, 64, 64
    lay[0].planes_a = 3
    assert L.dhaug_gemm_tn_group_bf16(lay, 1, a16, None) == -1                                                                                          # planes_a out of range
    lay[0].planes_a, lay[0].M = 2, 64                                                                                                                    # planes: M must be 6 x rows
    assert L.dhaug_gemm_tn_group_bf16(lay, 1, a16, None) == -1
    with pytest.raises(RuntimeError):
        dhaug_amd._lib.check(-2, "x")


def test_parity_program_planner_refuses_what_one_image_cannot_hold(built):
    """dhaug_mlp_forward_x3 plans the three virtual buffers of a program onto ONE in-place LDS image (+ registers + a workspace)
    on the host, before any launch: a program that reads a value from where it no longer is comes back DHAUG_EUNSUPPORTED, mixed
    fragment layouts and a parked result without a workspace DHAUG_EINVAL -- checked here without a GPU."""
    import dhaug_amd
    from dhaug_amd import _lib
    L = _lib.lib()
    buf = (ctypes.c_float * 4096)()
    base = ctypes.addressof(buf)
    base += (-base) % 16
    ptr = ctypes.c_void_p(base)

    def unit(kind, **kw):
        u = _lib.MlpUnit()
        u.kind, u.src, u.dst, u.res, u.src2 = kind, -1, -1, -1, -1
        for k, v in kw.items():
            setattr(u, k, v)
        return u

    def run(units):
        arr = (_lib.MlpUnit * len(units))(*units)
        return L.dhaug_mlp_forward_x3(arr, len(units), 128, None)

    LOAD, GEMM, OUT, T16 = 0, 3, 4, 32
    load = lambda dst: unit(LOAD, dst=dst, cols=64, ld=64, g=ptr)
    gemm = lambda src, dst, **kw: unit(GEMM, src=src, dst=dst, ksteps=4, n=256, w=ptr, bias=ptr, **kw)
    # the source must be what the image holds: buffer 0 was never written
    assert run([load(1), gemm(0, 1)]) == -3
    # buffer 1 is read again as a SOURCE after the layer in between has overwritten the image
    assert run([load(1), gemm(1, 0), gemm(1, 0)]) == -3
    # one fragment layout per program
    assert run([load(1), gemm(1, 0, flags=T16), gemm(0, 1)]) == -1
    # a result that waits while another branch uses the image needs the workspace (g of the GEMM units that are not outputs)
    parked = [load(1), gemm(1, 0), unit(GEMM, src=0, dst=2, ksteps=16, n=100, w=ptr, bias=ptr), load(1), gemm(1, 0),
              unit(GEMM, src=0, dst=2, res=2, ksteps=16, n=100, w=ptr, bias=ptr)]
    assert run(parked) == -1
    # a second source (a concatenation in one unit) is not a unit of this kernel
    assert run([load(1), unit(GEMM, src=1, dst=0, src2=0, ksteps2=4, ksteps=4, n=256, w=ptr, w2=ptr, bias=ptr)]) == -3
    assert run([]) == -1


def test_ops_refuse_cpu_tensors(built):
    from dhaug_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.fk_forward(torch.zeros(2, 37), torch.zeros(2, 15), torch.zeros(2, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.kcs_forward(torch.zeros(2, 16, 3))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "dh-aug-dh-forward-kinematics-model-driven-augmentation-for-3d-human-pose-estimation_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                assert "oracle" not in src, os.path.join(d, f)


def test_module_state_dict_keys_and_host_logic(built):
    """module construction, state_dict key parity with the reference, config defaults: all host-side"""
    import golden_util as GU
    from dhaug_amd.function_aug.config import get_parse_args
    from dhaug_amd.models_Fk_GAN import Fk_discriminator as dis, Fk_generator as gen, forward_kinematics_DH_model as fkm
    from dhaug_amd.models_Fk_GAN.video_mode_operate import video_receptive_field
    args = get_parse_args([])
    assert args.batch_size == 1024 and args.GAN_OUTPUT_DIM == 35 and args.GAN_LAMBDA == 10 and args.Gen_DenseDim == 1000
    assert args.bone_len_scaler == "different" and args.flip_GAN_model_input is True and args.architecture == "3,3,3"
    a2 = get_parse_args(["--batch_size", "512", "--GAN_whether_use_preAngle", "False", "--architecture", "3,3",
                         "--single_or_multi_train_mode", "multi"])
    assert a2.batch_size == 512 and a2.GAN_whether_use_preAngle is False
    assert video_receptive_field([3, 3]) == 9 and video_receptive_field([3, 3, 3]) == 27
    args.Gen_DenseDim = args.Dis_DenseDim_3D = args.Dis_DenseDim_2D = 32
    fk = fkm.Forward_Kinematics_DH_Model(args, ["S1"], None)
    G = gen.Fk_Generator(fk, args, "cpu")
    assert list(G.state_dict().keys()) == list(GU.shapes_generator(32).keys())
    assert {k: tuple(v.shape) for k, v in G.state_dict().items()} == GU.shapes_generator(32)
    D3 = dis.Fk_3D_Discriminator("cpu", args)
    assert {k: tuple(v.shape) for k, v in D3.state_dict().items()} == GU.shapes_d3(32)
    D2 = dis.Fk_2D_Discriminator(args, 16)
    assert {k: tuple(v.shape) for k, v in D2.state_dict().items()} == GU.shapes_d2(32)
    assert sum(p.numel() for p in gen.Fk_Generator(fk, get_parse_args(["--Gen_DenseDim", "256"]), "cpu").parameters()) == 436771
    assert fkm.H36M_32_To_16_Table == [0, 1, 2, 3, 6, 7, 8, 12, 13, 15, 17, 18, 19, 25, 26, 27]


def test_hostcheck_fk_math_matches_oracle(built):
    """the per-pose arithmetic of csrc/dhaug_fk_math.h, compiled for the HOST by the test-only harness, against
    the oracle: forward <= 1e-5 abs, reverse mode <= 1e-4 relative (the device runs the same source)."""
    import numpy as np
    import golden_util as GU
    from oracle import dhaug_oracle as O
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "hostcheck", "_build", "libhostcheck.so"))
    P = ctypes.POINTER(ctypes.c_float)
    ptr = lambda a: a.ctypes.data_as(P)
    N = 2048
    a, bl, rt = GU.synth_fk_inputs(N, 3)
    an, bn, rn = a.numpy().copy(), bl.numpy().copy(), rt.numpy().copy()
    out = np.zeros((N, 48), np.float32)
    lib.hostcheck_fk_forward(ptr(an), ptr(bn), ptr(rn), ptr(out), ctypes.c_long(N))
    assert np.abs(out - O.fk_forward16(a, bl, rt).reshape(N, 48).numpy()).max() <= 1e-5
    g = torch.randn(N, 16, 3, generator=torch.Generator().manual_seed(1))
    ad, bd, rd = (t.double().requires_grad_(True) for t in (a, bl, rt))
    (O.fk_forward16(ad, bd, rd) * g.double()).sum().backward()
    ga, gb, gr = np.zeros((N, 37), np.float32), np.zeros((N, 15), np.float32), np.zeros((N, 3), np.float32)
    gn = g.numpy().reshape(N, 48).copy()
    lib.hostcheck_fk_backward(ptr(an), ptr(bn), ptr(gn), ptr(ga), ptr(gb), ptr(gr), ctypes.c_long(N))
    for got, ref in ((ga, ad.grad), (gb, bd.grad), (gr, rd.grad)):
        assert np.abs(got - ref.numpy()).max() <= 1e-4 * ref.abs().max().item()


def test_hostcheck_under_address_and_ub_sanitizers(built, tmp_path):
    """the same host build as an executable under -fsanitize=address,undefined (-fno-sanitize-recover): ordinary poses and
    the degenerate ones of tests/test_gpu_edge.py (angles of +-1e4 and +-1e7 degrees -- the library path of the range
    reduction --, zero bone lengths, inf / NaN angles) run clean, and the ordinary poses still match the oracle."""
    import subprocess
    import numpy as np
    import golden_util as GU
    from oracle import dhaug_oracle as O
    import __graft_entry__ as ge
    exe = ge.build_hostcheck_sanitized(verbose=False)
    N = 512
    a, bl, rt = GU.synth_fk_inputs(N, 17)
    a, bl = a.clone(), bl.clone()
    a[256:320] *= 55.0                      # +-1e4 deg
    a[320:384] *= 5.0e4                     # +-1e7 deg: beyond the Cody-Waite window
    bl[384:400] = 0.0
    a[400, 3] = float("inf"); a[401, 12] = float("nan"); a[402, 36] = -float("inf")
    g = torch.randn(N, 48, generator=torch.Generator().manual_seed(2))
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(np.int64(N).tobytes())
        for t in (a, bl, rt, g):
            f.write(t.numpy().astype(np.float32).tobytes())
    r = subprocess.run([exe, str(fin), str(fout)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0, r.stderr[-3000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    out = np.fromfile(fout, dtype=np.float32)[:N * 48].reshape(N, 48)
    ref = O.fk_forward16(a, bl, rt).reshape(N, 48).numpy()
    ok = np.r_[0:400]
    assert np.abs(out[ok] - ref[ok]).max() <= 1e-5
