"""Import the REAL reference modules (read-only, from /root/reference) on CPU.

TEST INFRASTRUCTURE ONLY; works only where /root/reference exists (the build
container), never on the GPU box.  Used by tests/golden/make_golden.py to
generate fixtures and by tests/test_oracle_vs_reference.py to validate the
restatement in oracle/e4s_oracle.py.  Nothing from the reference is copied:
the modules are executed where they lie.

Recipe: SURVEY.md 8(c).  ``src.models.stylegan2.op`` JIT-compiles CUDA at
import (op/fused_act.py:8-15), so it is pre-seeded in sys.modules with the
reference's own pure-PyTorch fallbacks from the GPEN copy
(src/pretrained/gpen/face_model/op/fused_act.py:92-96, upfirdn2d.py:149-193)
and the pure-Python conv2d_gradfix.
"""
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get("E4S_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REF_ROOT, "src", "models", "networks.py"))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_CACHE = {}


def reference_modules():
    """Returns a namespace with Net3, Generator, fused_leaky_relu, upfirdn2d (reference code)."""
    if "ns" in _CACHE:
        return _CACHE["ns"]
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    # our repo also has a top-level ``src`` shim package; make sure the reference's wins here
    for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        del sys.modules[k]
    # the reference's `src` is a namespace package (no __init__.py); a regular `src` package anywhere on
    # sys.path (this repo's shim) would win over it, so hide such entries while importing
    saved_path = list(sys.path)
    sys.path[:] = [REF_ROOT] + [p for p in saved_path
                                if not os.path.isfile(os.path.join(p or os.getcwd(), "src", "__init__.py"))]
    src = os.path.join(REF_ROOT, "src")
    fa = _load("ref_gpen_fused_act", os.path.join(src, "pretrained/gpen/face_model/op/fused_act.py"))
    up = _load("ref_gpen_upfirdn2d", os.path.join(src, "pretrained/gpen/face_model/op/upfirdn2d.py"))
    gf = _load("src.models.stylegan2.op.conv2d_gradfix",
               os.path.join(src, "models/stylegan2/op/conv2d_gradfix.py"))

    class FusedLeakyReLU(nn.Module):
        def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
            super().__init__()
            self.bias = nn.Parameter(torch.zeros(channel))
            self.negative_slope, self.scale = negative_slope, scale

        def forward(self, x):
            return fa.fused_leaky_relu(x, self.bias, self.negative_slope, self.scale, "cpu")

    op = types.ModuleType("src.models.stylegan2.op")
    op.__path__ = []
    op.FusedLeakyReLU = FusedLeakyReLU
    op.fused_leaky_relu = lambda x, b, ns=0.2, sc=2 ** 0.5: fa.fused_leaky_relu(x, b, ns, sc, "cpu")
    upmod = up

    def ref_upfirdn2d(x, k, up=1, down=1, pad=(0, 0)):
        return upmod.upfirdn2d(x, k, up, down, pad, "cpu")

    op.upfirdn2d = ref_upfirdn2d
    op.conv2d_gradfix = gf
    sys.modules["src.models.stylegan2.op"] = op
    sys.modules["src.models.stylegan2.op.conv2d_gradfix"] = gf
    from src.models.networks import Net3                # reference code, unmodified
    from src.models.stylegan2.model import Discriminator, Generator, ModulatedConv2d, StyledConv, ToRGB
    ns = types.SimpleNamespace(Net3=Net3, Generator=Generator, Discriminator=Discriminator,
                               ModulatedConv2d=ModulatedConv2d,
                               StyledConv=StyledConv, ToRGB=ToRGB,
                               fused_leaky_relu=op.fused_leaky_relu, upfirdn2d=op.upfirdn2d)
    sys.path[:] = saved_path
    # leave the reference modules reachable only through `ns`; later `import src...` gets this repo's shim
    for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        del sys.modules[k]
    _CACHE["ns"] = ns
    return ns


def reference_gpen():
    """The reference's GPEN model module (src/pretrained/gpen/face_model/gpen_model.py), imported in place; its op
    package falls back to pure PyTorch on CPU."""
    if "gpen" in _CACHE:
        return _CACHE["gpen"]
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    import importlib
    for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        del sys.modules[k]
    saved_path = list(sys.path)
    sys.path[:] = [REF_ROOT] + [p for p in saved_path
                                if not os.path.isfile(os.path.join(p or os.getcwd(), "src", "__init__.py"))]
    try:
        mod = importlib.import_module("src.pretrained.gpen.face_model.gpen_model")
    finally:
        sys.path[:] = saved_path
        for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
            del sys.modules[k]
    _CACHE["gpen"] = mod
    return mod


def reference_criteria(alexnet_features):
    """The reference's own IDLoss (src/criteria/id_loss.py), LPIPS (src/criteria/lpips/lpips.py) and FaceParsingLoss
    (src/criteria/face_parsing/face_parsing_loss.py), imported in place with
    the reference's `src` winning over this repo's overlay.  torchvision is absent here: `alexnet_features` (a callable
    returning torchvision's AlexNet `features` Sequential, restated) stands in for `models.alexnet(True).features`
    (lpips/networks.py:76), and the LPIPS weight download (lpips/utils.py:11-20) is replaced by the LinLayers'
    initialisation -- callers load seeded weights afterwards."""
    if "crit" in _CACHE:
        return _CACHE["crit"]
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    import importlib
    stub_third_party()
    for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        del sys.modules[k]
    saved_path = list(sys.path)
    sys.path[:] = [REF_ROOT] + [p for p in saved_path
                                if not os.path.isfile(os.path.join(p or os.getcwd(), "src", "__init__.py"))]
    try:
        idm = importlib.import_module("src.criteria.id_loss")
        lpm = importlib.import_module("src.criteria.lpips.lpips")
        netm = importlib.import_module("src.criteria.lpips.networks")
        netm.models = types.SimpleNamespace(alexnet=lambda *a, **k: types.SimpleNamespace(features=alexnet_features()))
        lpm.get_state_dict = lambda net_type="alex", version="0.1": netm.LinLayers([64, 192, 384, 256, 256]).state_dict()
        fpm = importlib.import_module("src.criteria.face_parsing.face_parsing_loss")
        ns = types.SimpleNamespace(IDLoss=idm.IDLoss, LPIPS=lpm.LPIPS, FaceParsingLoss=fpm.FaceParsingLoss)
    finally:
        sys.path[:] = saved_path
        for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
            del sys.modules[k]
    _CACHE["crit"] = ns
    return ns


def make_opts(out_size=1024, remaining_layer_idx=13, num_seg_cls=12):
    return types.SimpleNamespace(fsencoder_type="psp", remaining_layer_idx=remaining_layer_idx,
                                 num_seg_cls=num_seg_cls, out_size=out_size, train_G=False,
                                 start_from_latent_avg=True, learn_in_w=False)


def build_reference_net3(state_dict, latent_avg, out_size=1024, remaining_layer_idx=13):
    ns = reference_modules()
    net = ns.Net3(make_opts(out_size, remaining_layer_idx))
    net.load_state_dict(state_dict, strict=True)
    net.latent_avg = latent_avg
    return net.eval()


# ---- third-party stand-ins so the reference's SCRIPTS import in this container -----------------------------------
# scripts/face_swap.py / optimization.py import cv2, torchvision, skimage, matplotlib, ... at module level; none of
# them is installed here and none of them is on the hot path.  The stubs below make those imports succeed with inert
# objects, so that tests can import the scripts and exercise the functions that only touch torch / numpy / Net3.
_STUB_ROOTS = ("cv2", "torchvision", "skimage", "matplotlib", "imageio", "face_alignment", "dlib", "gradio", "kornia",
               "ffmpeg", "lpips", "facexlib", "basicsr", "insightface", "onnxruntime", "seaborn", "tensorboard",
               "tensorboardX", "wandb", "moviepy", "av", "albumentations", "timm", "ninja_stub_never")


class _Stub:
    """Inert stand-in: callable, subscriptable, attribute access returns more stubs."""

    def __init__(self, name="stub"):
        self.__dict__["_name"] = name

    def __call__(self, *a, **k):
        return _Stub(self._name + "()")

    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        return _Stub(self._name + "." + item)

    def __getitem__(self, item):
        return _Stub(self._name + "[]")

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):       # `class X(stub.Base)` -> plain object subclass
        return (object,)

    def __repr__(self):
        return "<stub %s>" % self._name


class _StubModule(types.ModuleType):
    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        return _Stub(self.__name__ + "." + item)


class _StubFinder:
    """meta-path finder of last resort for the roots above (only consulted when the real module is absent)."""

    @staticmethod
    def find_spec(fullname, path=None, target=None):
        if fullname.split(".")[0] not in _STUB_ROOTS:
            return None
        import importlib.machinery

        class _Loader:
            @staticmethod
            def create_module(spec):
                m = _StubModule(spec.name)
                m.__path__ = []
                return m

            @staticmethod
            def exec_module(module):
                return None

        return importlib.machinery.ModuleSpec(fullname, _Loader(), is_package=True)


def stub_third_party():
    """Install the stub finder (idempotent).  It sits at the END of sys.meta_path: a really installed package wins."""
    if not any(isinstance(f, type) and f is _StubFinder for f in sys.meta_path):
        sys.meta_path.append(_StubFinder)


def import_reference_script(name):
    """Import /root/reference/scripts/<name>.py AS THE REFERENCE SHIPS IT, with this repo's `src` overlay first on
    sys.path (so `from src.models.networks import Net3` resolves to e4s_amd and everything else to the reference) and
    the absent third-party packages stubbed.  Returns the module."""
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    stub_third_party()
    os.environ.setdefault("E4S_REFERENCE_ROOT", REF_ROOT)
    repo_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        del sys.modules[k]
    saved = list(sys.path)
    sys.path[:] = [repo_root] + [p for p in saved if p not in (repo_root, REF_ROOT)] + [REF_ROOT]
    # src/pretrained/face_parsing/model.py:15 calls .cuda() at import; in the GPU-less build container make it a no-op
    patched = None
    if not torch.cuda.is_available():
        patched = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        return _load("ref_script_" + name, os.path.join(REF_ROOT, "scripts", name + ".py"))
    finally:
        sys.path[:] = saved
        if patched is not None:
            torch.Tensor.cuda = patched
