"""Build-container tests of the Python boundary: the reference's own scripts are imported UNCHANGED
from /root/reference with this repo's `src` overlay first on sys.path and the absent third-party packages (cv2,
torchvision, skimage, matplotlib ...) stubbed, and the parts of them that run without a GPU are executed against the
e4s_amd implementations.  Skipped where /root/reference does not exist (the GPU box)."""
import types

import pytest
import torch

from e4s_amd import synth
from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="needs /root/reference")


@pytest.fixture(scope="module")
def face_swap_script():
    return ref_shim.import_reference_script("face_swap")


@pytest.fixture(scope="module")
def optimization_script():
    return ref_shim.import_reference_script("optimization")


def test_scripts_resolve_hot_path_to_e4s_amd_and_the_rest_to_the_reference(face_swap_script, optimization_script):
    import e4s_amd.networks
    import e4s_amd.op
    fs, opt = face_swap_script, optimization_script
    assert fs.Net3 is e4s_amd.networks.Net3 and opt.Net3 is e4s_amd.networks.Net3
    # everything off the hot path is the reference's own module, imported where it lies
    assert fs.SwapFacePipelineOptions.__module__ == "src.options.swap_options"
    assert fs.dilation.__module__ == "src.utils.morphology" and "/reference/" in fs.dilation.__code__.co_filename
    assert "/reference/" in fs.swap_head_mask_revisit_considerGlass.__code__.co_filename
    assert "/reference/" in opt.OptimOptions.__init__.__code__.co_filename
    # src.utils.torch_utils: the reference's helpers + the version-bumping accumulate of the overlay
    assert hasattr(fs.torch_utils, "tensor2im") and hasattr(fs.torch_utils, "labelMap2OneHot")
    assert "/reference/" not in fs.torch_utils.accumulate.__code__.co_filename
    import importlib
    op = importlib.import_module("src.models.stylegan2.op")
    assert op.fused_leaky_relu is e4s_amd.op.fused_leaky_relu and op.upfirdn2d is e4s_amd.op.upfirdn2d
    assert hasattr(op.conv2d_gradfix, "no_weight_gradients")                   # src/criteria/adv_loss.py:34
    helpers = importlib.import_module("src.models.encoders.helpers")
    assert hasattr(helpers, "l2_norm") and "/reference/" in helpers.bottleneck_IR.__init__.__code__.co_filename  # parsing UNet
    import e4s_amd.criteria
    import e4s_amd.encoders
    assert helpers.bottleneck_IR_SE_Ours is e4s_amd.encoders.bottleneck_IR_SE_Ours
    assert helpers.bottleneck_IR_SE is e4s_amd.criteria.bottleneck_IR_SE
    # the loss networks of the optimisation loop (scripts/optimization.py:23-24) are the native ones
    assert opt.IDLoss is e4s_amd.criteria.IDLoss and opt.LPIPS is e4s_amd.criteria.LPIPS
    assert opt.FaceParsingLoss is e4s_amd.criteria.FaceParsingLoss


def test_loss_network_state_dicts_match_the_reference_classes():
    """IDLoss / LPIPS: same keys and shapes as the reference's own modules, so its checkpoints load unchanged."""
    import tempfile
    import e4s_amd.criteria as C
    ns = ref_shim.reference_criteria(C.alexnet_features)
    mine = C.IDLoss(types.SimpleNamespace())
    tmp = tempfile.mkdtemp()
    torch.save(mine.facenet.state_dict(), tmp + "/irse50.pth")
    ref = ns.IDLoss(types.SimpleNamespace(ir_se50_path=tmp + "/irse50.pth", id_loss_multiscale=True))
    shapes = lambda m: {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert shapes(mine) == shapes(ref)
    assert shapes(C.LPIPS()) == shapes(ns.LPIPS(net_type="alex"))
    minep = C.FaceParsingLoss(types.SimpleNamespace())
    torch.save(minep.G.state_dict(), tmp + "/unet.pth")
    assert shapes(minep) == shapes(ns.FaceParsingLoss(types.SimpleNamespace(face_parsing_model_path=tmp + "/unet.pth")))


@pytest.mark.parametrize("case", ["plain", "no_ear", "no_teeth", "below_face"])
def test_swap_comp_style_vector_matches_the_scripts(face_swap_script, case):
    """scripts/face_swap.py:117-146 (reference code, executed) vs the batched, sync-free e4s_amd version."""
    from e4s_amd.networks import swap_comp_style_vector
    g = torch.Generator().manual_seed(3)
    t_sv = torch.randn(1, 12, 1280, generator=g)
    d_sv = torch.randn(1, 12, 1280, generator=g)
    if case == "no_ear":
        d_sv[:, 7] = 0
    if case == "no_teeth":
        d_sv[:, 9] = 0
    comp = sorted(set(range(12)) - {0, 4, 11, 10})
    bf = case == "below_face"
    want = face_swap_script.swap_comp_style_vector(t_sv, d_sv, comp, belowFace_interpolation=bf)
    got = swap_comp_style_vector(t_sv, d_sv, comp, belowFace_interpolation=bf)
    assert torch.equal(want, got)
    # batched: row i of a batch equals the single-sample call
    t2, d2 = torch.cat([t_sv, d_sv.flip(1)]), torch.cat([d_sv, t_sv])
    got2 = swap_comp_style_vector(t2, d2, comp, belowFace_interpolation=bf)
    assert torch.equal(got2[:1], want)


def test_optimizer_setup_against_native_net3(optimization_script):
    """scripts/optimization.py:125-161 run on the latent the native Net3 produces the shape of."""
    from e4s_amd.networks import Net3
    from e4s_amd.options import make_opts
    with torch.device("meta"):
        net = Net3(make_opts(out_size=1024))
    assert set(net.state_dict()) == {k for k, _, _ in synth.net3_param_spec(1024, 13)}
    fake_self = types.SimpleNamespace(opts=types.SimpleNamespace(opt_name="adam", lr=1e-2))
    w_init = torch.randn(1, 12, 1280)
    optimizer, latent = optimization_script.Optimizer.setup_W_optimizer(fake_self, w_init)
    assert isinstance(optimizer, torch.optim.Adam) and latent.requires_grad and torch.equal(latent, w_init)
    latent.grad = torch.ones_like(latent)
    optimizer.step()
    assert float((latent - w_init).abs().max()) > 0


def test_accumulate_invalidates_weight_packs():
    """the reference's EMA writes through `.data` (src/utils/torch_utils.py:189-194) which leaves `_version`
    alone; the overlay's accumulate must make every cached weight pack stale."""
    import importlib
    from e4s_amd import packs
    tu = importlib.import_module("src.utils.torch_utils")
    a, b = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4)
    k0 = packs.param_key(a.weight)
    a.weight.data.mul_(0.5)                      # the reference's form: invisible to the version counter ...
    assert packs.param_key(a.weight) == k0
    packs.invalidate_packs()                     # ... hence the explicit invalidation hook
    k1 = packs.param_key(a.weight)
    assert k1 != k0
    before = a.weight.detach().clone()
    tu.accumulate(a, b, decay=0.9)
    assert torch.allclose(a.weight, before * 0.9 + b.weight * 0.1)
    assert packs.param_key(a.weight) != k1
