"""CPU tests of host-side logic that never touches the GPU: weight re-packing, state_dict
compatibility, the C-ABI surface of libe4s_hip.so, sharding arithmetic."""
import ctypes
import os
import re

import pytest
import torch
import torch.nn.functional as F

from e4s_amd import synth
from oracle import e4s_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_polyphase_upconv_equals_transposed_conv_plus_blur():
    from e4s_amd.stylegan2 import polyphase_upconv_weights
    g = torch.Generator().manual_seed(5)
    cin, cout, h = 5, 7, 6
    x = torch.randn(2, cin, h, h + 3, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g)
    blur = orc.make_blur_kernel() * 4.0
    ref = F.conv_transpose2d(x, w.transpose(0, 1), stride=2)
    ref = orc.upfirdn2d(ref, blur, pad=(1, 1))
    pw = polyphase_upconv_weights(w, blur)                      # [4,9,Cout,Cin]
    out = torch.zeros_like(ref)
    for py in range(2):
        for px in range(2):
            k = pw[py * 2 + px].reshape(3, 3, cout, cin).permute(2, 3, 0, 1)
            out[:, :, py::2, px::2] = F.conv2d(x, k, padding=1)
    assert out.shape == ref.shape == (2, cout, 2 * h, 2 * (h + 3))
    assert float((out - ref).abs().max()) < 1e-5


def test_state_dict_keys_match_reference_layout():
    """Product modules expose exactly the reference Net3 state_dict; the same spec was
    loaded strict=True into the REAL reference when the golden fixtures were generated."""
    from e4s_amd.networks import Net3
    from e4s_amd.options import make_opts
    for size in (256, 1024):
        spec = {k: tuple(s) for k, s, _ in synth.net3_param_spec(size, 13)}
        with torch.device("meta"):
            net = Net3(make_opts(out_size=size))
        sd = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        assert sd == spec
    assert len(spec) == 344                                     # 344 tensors


def test_library_exports_every_declared_symbol():
    """include/e4s_hip.h <-> libe4s_hip.so <-> e4s_amd.lib.SIGNATURES stay in sync (no compute calls)."""
    from e4s_amd import lib
    hdr = open(os.path.join(ROOT, "include", "e4s_hip.h")).read()
    declared = set(re.findall(r"\b(e4s_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert os.path.isfile(lib.LIB_PATH), "build first: python -m e4s_amd.build"
    so = ctypes.CDLL(lib.LIB_PATH)
    for name in declared:
        assert hasattr(so, name), f"{name} declared in e4s_hip.h but not exported"
    assert set(lib.SIGNATURES) | {"e4s_abi_version", "e4s_build_arch"} == declared
    so.e4s_abi_version.restype = ctypes.c_int
    assert so.e4s_abi_version() == lib.ABI_VERSION
    so.e4s_build_arch.restype = ctypes.c_char_p
    assert so.e4s_build_arch() == b"gfx950"


def test_conv_params_struct_matches_header():
    from e4s_amd import lib
    hdr = open(os.path.join(ROOT, "include", "e4s_hip.h")).read()
    body = hdr[hdr.index("typedef struct {"):hdr.index("} e4s_conv_params;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for stmt in body.split(";"):
        stmt = stmt.replace("typedef struct {", "").strip()
        if not stmt:
            continue
        for part in stmt.split(","):
            names.append(re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*$", part.strip())[0])
    assert names == [f[0] for f in lib.ConvParams._fields_]


def test_no_cpu_fallback():
    """The product path must fail loudly off-GPU instead of silently computing in torch."""
    from e4s_amd.op import fused_leaky_relu, upfirdn2d
    with pytest.raises(RuntimeError):
        fused_leaky_relu(torch.zeros(1, 4, 2, 2), torch.zeros(4))
    with pytest.raises(RuntimeError):
        upfirdn2d(torch.zeros(1, 1, 4, 4), torch.ones(2, 2))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No silent fallback when libe4s_hip.so is absent: load() must raise, naming the build command."""
    from e4s_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "libe4s_hip.so"))
    with pytest.raises(RuntimeError, match="e4s_amd.build"):
        lib.load()


def test_src_shim_reexports_native_modules():
    import src.models.networks as n
    import src.models.stylegan2.model as m
    import src.models.stylegan2.op as op
    import e4s_amd.networks, e4s_amd.stylegan2, e4s_amd.op
    assert n.Net3 is e4s_amd.networks.Net3
    assert m.Generator is e4s_amd.stylegan2.Generator and m.Discriminator is e4s_amd.stylegan2.Discriminator
    assert op.upfirdn2d is e4s_amd.op.upfirdn2d and op.fused_leaky_relu is e4s_amd.op.fused_leaky_relu
    from src.utils.torch_utils import labelMap2OneHot
    lab = torch.randint(0, 12, (2, 1, 8, 8))
    oh = labelMap2OneHot(lab, 12)
    assert oh.shape == (2, 12, 8, 8) and torch.equal(oh.argmax(1, keepdim=True), lab) and float(oh.sum()) == 128.0


def test_discriminator_state_dict_matches_reference_layout():
    """The Discriminator module tree (config 5) has exactly the reference's keys/shapes: the same synthetic state
    dict was loaded strict=True into the REAL reference Discriminator when tests/golden/disc64.pt was produced."""
    from e4s_amd.stylegan2 import Discriminator
    spec = {k: tuple(s) for k, s, _ in synth.disc_param_spec(64)}
    mine = {k: tuple(v.shape) for k, v in Discriminator(64).state_dict().items()}
    assert mine == spec


def test_auto_precision_policy(monkeypatch):
    """E4S_PRECISION policy (host logic only): auto = split-bf16 where the kernel applies and the launch fills the
    chip (>= 128 tiles of 256x128), f32 = never, bf16x3 = wherever the kernel applies."""
    from e4s_amd import kernels as K
    monkeypatch.setattr(K, "PRECISION", "auto")
    assert K.want_bf16x3(16, 32, 32, 512, 512)            # bench batch: 2 x 8 images -> 256 tiles
    assert K.want_bf16x3(2, 32, 32, 512, 512)             # batch-1 latency run: 32 tiles x 8 K-splits
    assert K.want_bf16x3(1, 32, 32, 512, 512, masked=True)       # region-select kernel: 16 tiles x 8 K-splits
    assert not K.want_bf16x3(1, 4, 4, 512, 512, masked=True)     # 4 tiles x 8 splits: too few blocks -> exact fp32
    assert K.want_bf16x3(8, 512, 512, 64, 64)             # Cout 64 / 32: 128-pixel tiles, 64- / 32-wide column tiles
    assert K.want_bf16x3(8, 512, 512, 64, 32, ncls=4)
    assert not K.want_bf16x3(1, 64, 64, 64, 64)           # 16 tiles, 2 chunks: nothing to split -> exact fp32
    assert not K.want_bf16x3(8, 64, 64, 64, 64, masked=True)   # region-select kernel: 128-wide column tiles only
    assert K.want_bf16x3(8, 32, 32, 512, 512, ncls=4)     # polyphase up-conv: 4 phases count as tiles
    monkeypatch.setattr(K, "PRECISION", "f32")
    assert not K.want_bf16x3(16, 32, 32, 512, 512)
    monkeypatch.setattr(K, "PRECISION", "bf16x3")
    assert K.want_bf16x3(1, 16, 16, 512, 512)


def test_split_bf16_three_product_error_model():
    """The arithmetic of e4s_conv_bf16x3_f32, restated on the CPU: v = hi + lo (two round-to-nearest bf16), product =
    a_hi*b_hi + a_hi*b_lo + a_lo*b_hi.  Claims checked: the split loses < 2^-16 |v|; a 4608-term contraction (the
    512-channel 3x3 layer) lands within 2e-5 of its own scale of the exact result -- two orders of magnitude inside what
    one bf16 product gives, and the basis of the per-layer 1e-4 bound in tests/test_gpu_parity.py."""
    g = torch.Generator().manual_seed(5)
    a = torch.randn(256, 4608, generator=g, dtype=torch.float32) * 1.3 + 0.2
    b = torch.randn(4608, 64, generator=g, dtype=torch.float32) / 68.0

    def split(v):
        hi = v.to(torch.bfloat16).to(torch.float32)
        lo = (v - hi).to(torch.bfloat16).to(torch.float32)
        return hi, lo
    ah, al = split(a)
    bh, bl = split(b)
    assert float(((a - ah - al).abs() / a.abs().clamp_min(1e-30)).max()) < 2.0 ** -16
    exact = a.double() @ b.double()
    three = ah.double() @ bh.double() + ah.double() @ bl.double() + al.double() @ bh.double()
    one = ah.double() @ bh.double()
    scale = float(exact.abs().max())
    assert float((three - exact).abs().max()) < 2e-5 * scale
    assert float((one - exact).abs().max()) > 1e-3 * scale      # a single bf16 product is not enough for the 1e-3 path


def test_header_is_plain_c(tmp_path):
    """The drop-in boundary is a C ABI: include/e4s_hip.h must compile as C99 (what a cgo / ctypes-cffi / JNI binding
    would include) and as C++, with no torch / HIP types in the signatures."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "inc.c"
    src.write_text('#include "e4s_hip.h"\nint main(void) { return 0; }\n')
    for cc, flags in (("gcc", ["-std=c99", "-x", "c"]), ("g++", ["-std=c++17", "-x", "c++"])):
        if shutil.which(cc) is None:
            pytest.skip(f"{cc} not installed")
        res = subprocess.run([cc] + flags + ["-Wall", "-Werror", "-fsyntax-only", "-I" + os.path.join(root, "include"),
                              str(src)], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr
    text = open(os.path.join(root, "include", "e4s_hip.h")).read()
    assert "#include <torch" not in text and "#include <hip" not in text and "at::Tensor" not in text


def test_plain_c_host_program_links_against_the_abi(tmp_path):
    """examples/c_abi_demo.c -- a C99 host with no Python and no torch -- compiles with gcc and links against
    libe4s_hip.so + the HIP runtime (it is not executed here: no GPU)."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "e4s_amd", "libe4s_hip.so")
    if shutil.which("gcc") is None or not os.path.isdir("/opt/rocm/include") or not os.path.isfile(lib):
        pytest.skip("needs gcc, /opt/rocm and the built library")
    exe = tmp_path / "c_abi_demo"
    res = subprocess.run(["gcc", "-std=c99", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                          "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "c_abi_demo.c"),
                          "-L" + os.path.join(root, "e4s_amd"), "-le4s_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                          "-o", str(exe)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert exe.is_file()


def _load_bench():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_parser_defaults_follow_the_driver_contract():
    """`python bench.py` with no flags (the driver's N = 1 run): one GPU, a K / W that finish within minutes, the uint8
    gather and the overlapped collective as the N > 1 defaults, the graphed config-3 step (--opt-eager opts out), >= 10 timed
    config-5 steps; the driver's own flags parse."""
    bench = _load_bench()
    a = bench.build_parser().parse_args([])
    assert a.gpus == 1 and 1 <= a.steps <= 50 and 0 <= a.warmup <= 10 and a.batch == 8
    assert not a.gather_fp32 and not a.sync_gather and not a.no_graph and a.opt_graph and not a.stub_swap
    assert not bench.build_parser().parse_args(["--opt-eager"]).opt_graph and a.train_steps >= 10
    assert a.opt_steps == 200 and set(a.opt_modes.split(",")) == {"full", "mse"}
    d = bench.build_parser().parse_args(["--gpus", "8", "--steps", "20", "--warmup", "3"])
    assert (d.gpus, d.steps, d.warmup) == (8, 20, 3)
    assert bench.SIZE == 1024 and bench.KREM == 13                      # BASELINE.json's configuration


def test_bench_refuses_to_run_without_a_gpu(monkeypatch):
    """No CPU fallback at the top level either: without a ROCm device bench.py exits with a message, it does not time a
    torch / oracle path."""
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    bench = _load_bench()
    monkeypatch.setattr("sys.argv", ["bench.py"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "no CPU fallback" in str(e.value)


def test_loss_network_parameter_folds_match_torch():
    """Host-side algebra of e4s_amd/criteria.py (no kernels): BatchNorm(eval) as the {mean, rstd} operand, conv + BatchNorm
    folded into one conv, and BatchNorm2d -> Flatten -> Linear -> BatchNorm1d folded into one NHWC-ordered affine map --
    each against the torch modules they replace, in fp64-checked fp32."""
    import types
    import torch.nn.functional as F
    from e4s_amd import criteria as C
    g = torch.Generator().manual_seed(3)
    # (x - mu') * rho' == BatchNorm2d.eval()(x)
    bn = torch.nn.BatchNorm2d(8).eval()
    bn.load_state_dict(synth.synth_module_state_dict(bn, 1, "bn."))
    x = torch.randn(2, 8, 5, 5, generator=g)
    st = C._bn_stats(bn, 2)                                               # [B,C,2]
    mine = (x - st[:, :, 0, None, None]) * st[:, :, 1, None, None]
    assert float((mine - bn(x)).abs().max()) < 2e-6
    # conv -> BN folded (with zero channel padding) == BN(conv(x)) on the real channels, exact zeros on the padded ones
    stage = C.unetConv2(3, 16).eval()
    stage.load_state_dict(synth.synth_module_state_dict(stage, 2, "st."))
    t = C._folded(stage.conv1, 3, 32)
    xi = torch.randn(1, 3, 9, 9, generator=g)
    ref = stage.conv1[1](stage.conv1[0](xi))
    got = F.conv2d(xi, t.weight, t.bias, padding=1)
    assert tuple(got.shape) == (1, 32, 9, 9) and float((got[:, :16] - ref).abs().max()) < 2e-6
    assert float(got[:, 16:].abs().max()) == 0.0
    t2 = C._folded(stage.conv2, 32, 32)
    assert tuple(t2.weight.shape) == (32, 32, 3, 3) and float(t2.weight[:, 16:].abs().max()) == 0.0
    # the identity head: one packed affine map on the NHWC-flattened [B,7,7,512] feature
    bb = C.Backbone().eval()
    bb.load_state_dict(synth.synth_module_state_dict(bb, 3, "bb."))
    feat = torch.randn(2, 512, 7, 7, generator=g)
    ref = bb.output_layer(feat)
    w, b = bb._head()
    got = feat.permute(0, 2, 3, 1).reshape(2, -1) @ w[0].t() + b[0]
    assert float((got - ref).abs().max()) < 5e-5 * float(ref.abs().max())
    # a zero BatchNorm scale cannot be expressed as (x - mu) * rho and is refused
    bn.weight.data[3] = 0.0
    with pytest.raises(RuntimeError):
        C._bn_stats(bn, 3)


def test_loss_networks_refuse_to_run_without_weights(monkeypatch):
    """the reference always loads trained weights into IDLoss / LPIPS / FaceParsingLoss
    (id_loss.py:15, lpips/utils.py:11-20, face_parsing_loss.py:29); the native classes must not silently stay random."""
    import types
    from e4s_amd import criteria as C
    monkeypatch.setattr(C, "ALLOW_UNINITIALIZED", False)
    monkeypatch.delenv("E4S_LPIPS_WEIGHTS", raising=False)
    with pytest.raises(FileNotFoundError):
        C.IDLoss(types.SimpleNamespace(ir_se50_path="/nonexistent/irse50.pth"))
    with pytest.raises(FileNotFoundError):
        C.IDLoss(types.SimpleNamespace())
    with pytest.raises(FileNotFoundError):
        C.FaceParsingLoss(types.SimpleNamespace())
    with pytest.raises(FileNotFoundError):
        C.LPIPS()
    monkeypatch.setattr(C, "ALLOW_UNINITIALIZED", True)
    C.LPIPS()                                                   # synthetic-weight runs opt in explicitly


def test_target_feature_cache_holds_the_target_tensor():
    """the cache entry keeps the target alive, so a freed target's address cannot be handed to a new
    image that then hits the old entry; a different tensor object never hits, whatever its pointer/version."""
    from e4s_amd import criteria as C
    y1 = torch.zeros(1, 3, 8, 8)
    k1 = C._target_key(y1)
    entry = (k1, y1, "feats")
    y2 = torch.zeros(1, 3, 8, 8)
    assert entry[1] is not y2                                   # identity is part of the hit test (see _target_feats)
    y1.add_(1.0)
    assert C._target_key(y1) != k1                              # in-place edits miss as well
    src = open(os.path.join(ROOT, "e4s_amd", "criteria.py")).read()
    assert src.count("self._target[1] is not y") == 3           # all three loss classes use the guarded hit test


def test_stitch_host_side_restatements():
    """N4 stitching, CPU side: (1) the integer alpha-composite formula the device kernel implements (libImaging/AlphaComposite.c,
    PRECISION_BITS 7, opaque destination) against PIL itself for every alpha; (2) the 8.8 fixed-point Gaussian taps, product
    and oracle statements agree and sum to 256; (3) pyramid restatements preserve constants and sizes."""
    import numpy as np
    from PIL import Image
    from e4s_amd import postproc as PP
    rs = np.random.RandomState(0)
    face = rs.randint(0, 256, (256, 64, 3)).astype(np.uint8)
    tgt = rs.randint(0, 256, (256, 64, 3)).astype(np.uint8)
    alpha = np.repeat(np.arange(256, dtype=np.uint8)[:, None], 64, 1)
    src = Image.fromarray(face).convert("RGBA")
    src.putalpha(Image.fromarray(alpha))
    dst = Image.fromarray(tgt).convert("RGBA")
    dst.alpha_composite(src)
    ref = np.array(dst)
    a = alpha.astype(np.uint64)[..., None]
    outa255 = a * 255 + 255 * (255 - a)
    coef1 = a * 255 * 255 * 128 // np.maximum(outa255, 1)
    coef2 = 255 * 128 - coef1
    t = face.astype(np.uint64) * coef1 + tgt.astype(np.uint64) * coef2 + (0x80 << 7)
    mine = ((((t >> 8) + t) >> 8) >> 7).astype(np.uint8)
    mine = np.where(a == 0, tgt, mine)
    assert np.array_equal(mine, ref[:, :, :3]) and bool((ref[:, :, 3] == 255).all())
    for k, s in ((11, 0.0), (5, 0.0), (7, 0.0), (3, 0.0), (9, 1.5), (11, 3.0)):
        taps = PP.gaussian_kernel_fixed8(k, s)
        assert taps == orc.cv2_gaussian_taps_fixed8(k, s) and sum(taps) == 256 and taps == taps[::-1]
    assert PP.gaussian_kernel_fixed8(11, 0.0) == [2, 7, 17, 31, 45, 52, 45, 31, 17, 7, 2]
    c = np.full((12, 10, 3), 77, dtype=np.uint8)
    d = orc.cv2_pyrdown(c)
    assert d.shape == (6, 5, 3) and bool((d == 77).all())
    u = orc.cv2_pyrup(d.astype(np.float32))
    assert u.shape == (12, 10, 3) and bool((u == 77).all())
    one = orc.laplacian_blend_u8(c, c, np.random.RandomState(1).rand(12, 10, 3).astype(np.float32), num_levels=2)
    assert bool((one == 77).all())


def test_rowdot_job_record_matches_the_header(tmp_path):
    """kernels.rowdot_jobs packs e4s_rowdot_job records with struct format "qqqQQiiif": size and field offsets must be what a C
    compiler gives the struct declared in include/e4s_hip.h."""
    import struct
    import subprocess
    from e4s_amd import kernels as K
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "e4s_hip.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(e4s_rowdot_job),offsetof(e4s_rowdot_job,in_off),offsetof(e4s_rowdot_job,in_stride),'
                   'offsetof(e4s_rowdot_job,out_off),offsetof(e4s_rowdot_job,M),offsetof(e4s_rowdot_job,bias),'
                   'offsetof(e4s_rowdot_job,G),offsetof(e4s_rowdot_job,O),offsetof(e4s_rowdot_job,K),'
                   'offsetof(e4s_rowdot_job,scale));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert got == [struct.calcsize(K._JOB_FMT), 0, 8, 16, 24, 32, 40, 44, 48, 52] and got[0] == 56


def test_style_grad_job_record_matches_the_header(tmp_path):
    """lib.StyleGradJob (ctypes) mirrors e4s_style_grad_job of include/e4s_hip.h: same size, same field offsets, and the job limit the
    Python side enforces is the header's (the jobs travel by value in the kernel arguments: 32 of them stay under 4 KB)."""
    import ctypes
    import subprocess
    from e4s_amd import lib
    fields = [f[0] for f in lib.StyleGradJob._fields_]
    src = tmp_path / "sg.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "e4s_hip.h"\nint main(void){printf("%zu %d", sizeof(e4s_style_grad_job), '
                   'E4S_STYLE_GRAD_MAX_JOBS);' + "".join('printf(" %%zu", offsetof(e4s_style_grad_job, %s));' % f for f in fields) +
                   'printf("\\n");return 0;}\n')
    exe = tmp_path / "sg"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert got[0] == ctypes.sizeof(lib.StyleGradJob) and got[1] == lib.STYLE_GRAD_MAX_JOBS
    assert got[2:] == [getattr(lib.StyleGradJob, f).offset for f in fields]
    assert got[0] * got[1] + 4 * (got[1] + 1) + 4 + 32 <= 4096          # jobs + block offsets + count + the other arguments of stage 2


def test_train_iteration_cadence_follows_the_coach_loop():
    """e4s_amd.train.TrainIteration.iteration = the body of Coach.train() (coach.py:281-398): D step when global_step % d_every ==
    0, R1 only inside a D step and only when d_reg_every != -1 and batch_idx % d_reg_every == 0, one G step every iteration; D is
    frozen during the G step and trainable during its own.  CPU stand-ins for the networks (the cadence is host logic)."""
    from e4s_amd.train import LossOpts, TrainIteration, adv_d_loss, adv_g_loss, d_r1_loss

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.ones(1))

        def forward(self, img, onehot, **kw):
            return img * self.w, None

    class Disc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(3 * 4 * 4, 1)

        def forward(self, x):
            return self.lin(x.flatten(1) ** 2)
    net, disc = Net(), Disc()
    log = []
    it = TrainIteration(net, disc, {}, torch.optim.SGD(net.parameters(), lr=1e-3), torch.optim.SGD(disc.parameters(), lr=1e-3),
                        lo=LossOpts(d_every=3, d_reg_every=2))
    for name in ("d_step", "r1_step", "g_step"):
        real = getattr(it, name)

        def wrap(*a, _n=name, _f=real, **k):
            log.append((_n, it.global_step, all(p.requires_grad for p in disc.parameters())))
            return _f(*a, **k)
        setattr(it, name, wrap)
    img = torch.rand(2, 3, 4, 4)
    for step in range(7):
        out = it.iteration(img, None, batch_idx=step)
        assert "loss" in out and ("d_loss" in out) == (step % 3 == 0) and ("r1_loss" in out) == (step % 3 == 0 and step % 2 == 0)
    assert [n for n, _, _ in log] == ["d_step", "r1_step", "g_step", "g_step", "g_step", "d_step", "g_step", "g_step", "g_step",
                                      "d_step", "r1_step", "g_step"]
    assert it.global_step == 7
    assert not any(p.requires_grad for p in disc.parameters())            # the last thing that ran was a G step: D frozen
    it2 = TrainIteration(net, disc, {}, None, None, lo=LossOpts(d_reg_every=-1))
    assert it2.lo.d_every == 15 and it2.lo.g_adv_lambda == 0.01 and it2.lo.r1_lambda == 10.0     # train_options.py:37-38,53-54
    # the loss helpers are the reference's formulas (adv_loss.py:8-45)
    rp, fp = torch.tensor([[0.3], [-1.2]]), torch.tensor([[0.7], [0.1]])
    assert torch.allclose(adv_g_loss(fp), torch.nn.functional.softplus(-fp).mean())
    assert torch.allclose(adv_d_loss(rp, fp), torch.nn.functional.softplus(-rp).mean() + torch.nn.functional.softplus(fp).mean())
    x = torch.rand(2, 3, 4, 4, requires_grad=True)
    pen = d_r1_loss(disc(x), x)
    g, = torch.autograd.grad(disc(x).sum(), x, create_graph=True)
    assert torch.allclose(pen, g.pow(2).reshape(2, -1).sum(1).mean())


def test_train_iteration_forgets_targets_per_batch_and_refuses_steps_it_cannot_capture():
    """TrainIteration.forget_targets drops the loss networks' per-target feature caches (a training batch is a NEW target even when it
    arrives in the same tensor: the reference recomputes the target features in every calc_loss, id_loss.py:33-35), and
    graphed_g_step refuses what a stream capture cannot hold -- a net wrapped in torch's DistributedDataParallel (its reducer) and a
    non-capturable optimiser -- both before anything touches a GPU.  (A trainable generator is NOT refused any more: its style-prologue
    job tables are keyed on addresses, tests/test_gpu_train.py; nor is a gradient averager, tests/test_gpu_nccl_world1.py.)"""
    import types
    from e4s_amd.train import TrainIteration

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.ones(1))
            self.G = torch.nn.Linear(2, 2)

        def forward(self, img, onehot, **kw):
            return img * self.w, None

    class Wrapper(torch.nn.Module):                       # what nn.parallel.DistributedDataParallel looks like from outside
        def __init__(self, module):
            super().__init__()
            self.module = module

        def forward(self, *a, **kw):
            return self.module(*a, **kw)
    net = Net()
    crit = {"id": types.SimpleNamespace(_target=("key", "y", "feats")), "lpips": types.SimpleNamespace(_target=(1, 2, 3)),
            "other": types.SimpleNamespace()}
    it = TrainIteration(net, None, crit, torch.optim.SGD(net.parameters(), lr=1e-3), None, averager=object())
    it.forget_targets()
    assert crit["id"]._target is None and crit["lpips"]._target is None and not hasattr(crit["other"], "_target")
    assert it.core is net
    img = torch.zeros(1, 3, 4, 4)
    with pytest.raises(RuntimeError, match="capturable"):
        it.graphed_g_step(img, img)
    wrapped = TrainIteration(Wrapper(net), None, crit, torch.optim.SGD(net.parameters(), lr=1e-3), None)
    assert wrapped.core is net                            # EMA / latent_avg address the module inside, as coach.py does
    with pytest.raises(RuntimeError, match="DistributedDataParallel"):
        wrapped.graphed_g_step(img, img)


def test_winograd_f23_row_algebra_of_conv_wino_hip():
    """The arithmetic csrc/conv_wino.hip implements, restated in fp64 torch and checked against F.conv2d: pairs of output columns, the four
    positions V0..V3 / U0..U3, the three vertical taps as separate contractions, out = (M0+M1+M2, M1-M2-M3); and the InstanceNorm fold of
    its input transform (padded pixels take the MEAN so that they are zero in the normalised map; the mean cancels in V0, V2, V3)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(0)
    b, cin, cout, h, w = 2, 5, 7, 6, 8
    x = torch.randn(b, cin, h, w, generator=g, dtype=torch.float64) * 1.5 + 0.7
    wt = torch.randn(cout, cin, 3, 3, generator=g, dtype=torch.float64)
    mu = x.mean((2, 3), keepdim=True)
    rs = 1.0 / torch.sqrt(x.var((2, 3), unbiased=False, keepdim=True) + 1e-5)

    def wino(x, fold):
        # halo with the kernel's padding rule: 0, or the per-channel mean when the normalisation is folded in
        fill = mu if fold else torch.zeros_like(mu)
        xp = fill.expand(b, cin, h + 2, w + 2).clone()
        xp[:, :, 1:-1, 1:-1] = x
        out = torch.zeros(b, cout, h, w, dtype=torch.float64)
        for j in range(w // 2):
            d = [xp[:, :, :, 2 * j + i] for i in range(4)]                      # columns 2j-1 .. 2j+2 of every halo row: [b, cin, h+2]
            if fold:
                v = [(d[0] - d[2]) * rs[..., 0], ((d[1] - mu[..., 0]) + (d[2] - mu[..., 0])) * rs[..., 0], (d[2] - d[1]) * rs[..., 0],
                     (d[1] - d[3]) * rs[..., 0]]
            else:
                v = [d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]]
            m = [torch.zeros(b, cout, h, dtype=torch.float64) for _ in range(4)]
            for ky in range(3):
                g0, g1, g2 = wt[:, :, ky, 0], wt[:, :, ky, 1], wt[:, :, ky, 2]
                u = [g0, ((g0 + g2) + g1) * 0.5, ((g0 + g2) - g1) * 0.5, g2]
                for p in range(4):
                    m[p] += torch.einsum("bcy,kc->bky", v[p][:, :, ky:ky + h], u[p])        # V_p[y + ky - 1] . U_p[ky]
            out[:, :, :, 2 * j] = (m[0] + m[1]) + m[2]
            out[:, :, :, 2 * j + 1] = (m[1] - m[2]) - m[3]
        return out

    assert float((wino(x, False) - F.conv2d(x, wt, padding=1)).abs().max()) < 1e-12
    assert float((wino(x, True) - F.conv2d((x - mu) * rs, wt, padding=1)).abs().max()) < 1e-11
