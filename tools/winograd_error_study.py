"""CPU study (VERDICT r3 'next' 4, go / no-go BEFORE a kernel is written): how much error does a Winograd F(2x2,3x3) formulation of the
encoder's stride-1 3x3 convs add on top of the split-bf16 (hi+lo, 3 products) arithmetic the HIP kernels use?

Both arithmetics are emulated in fp32 torch on the CPU oracle's encoder (oracle/e4s_oracle.py:encoder_forward), by swapping F.conv2d:
  direct  : x, w split to bf16 hi/lo, y = conv(xh, wh) + conv(xh, wl) + conv(xl, wh)              (what conv_bf16x3.hip computes)
  winograd: V = B^T d B (fp32, from the UNSPLIT input), U = G g G^T (fp32), both split to bf16 hi/lo, M = Vh Uh + Vh Ul + Vl Uh per
            transform position (fp32 accumulate over Cin), Y = A^T M A in fp32
against the fp64 evaluation of the same encoder.  Prints per-arithmetic max-abs / relative errors of the [B,12,1280] style vectors at
out_size 256 inputs (the encoder always runs at 256^2), and of single layers of the dominant shape (512->512 @32^2).
Run: python tools/winograd_error_study.py [--which 512]   (winograd only on the 512-channel layers, the prototype's scope)"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from e4s_amd import synth  # noqa: E402
from oracle import e4s_oracle as orc  # noqa: E402

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
_CONV = F.conv2d          # the real one: orc.F IS torch.nn.functional, so the patch below is global


def split(t):
    hi = t.to(torch.bfloat16).to(torch.float32)
    lo = (t - hi).to(torch.bfloat16).to(torch.float32)
    return hi, lo


def conv_direct_bf16x3(x, w):
    xh, xl = split(x)
    wh, wl = split(w)
    return _CONV(xh, wh, padding=1) + _CONV(xh, wl, padding=1) + _CONV(xl, wh, padding=1)


def conv_winograd(x, w, split_ops=True):
    b, c, h, wd = x.shape
    k = w.shape[0]
    dt = x.dtype
    bt, g, at = BT.to(dt), G.to(dt), AT.to(dt)
    xp = F.pad(x, (1, 1, 1, 1))
    th, tw = h // 2, wd // 2
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                        # [b,c,th,tw,4,4]
    v = torch.einsum("ij,bcyxjk,lk->bcyxil", bt, d, bt)           # B^T d B
    u = torch.einsum("ij,kcjl,ml->kcim", g, w, g)                 # G g G^T  [k,c,4,4]
    if split_ops:
        vh, vl = split(v)
        uh, ul = split(u)
        m = (torch.einsum("bcyxij,kcij->bkyxij", vh, uh) + torch.einsum("bcyxij,kcij->bkyxij", vh, ul)
             + torch.einsum("bcyxij,kcij->bkyxij", vl, uh))
    else:
        m = torch.einsum("bcyxij,kcij->bkyxij", v, u)
    y = torch.einsum("ij,bkyxjl,ml->bkyxim", at, m, at)           # [b,k,th,tw,2,2]
    return y.permute(0, 1, 2, 4, 3, 5).reshape(b, k, h, wd)


# 1-D forms along the image rows (what conv_wino.hip computes: the three vertical taps stay direct, one GEMM per (tap row, position)).
# F(2,3): 4 positions per 2 outputs (1.5x fewer MACs than direct); F(4,3): 6 positions per 4 outputs (2x fewer), Lavin & Gray's points
# 0, +-1, +-2, inf.
W1D = {
    2: (torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64),
        torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64),
        torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)),
    4: (torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                      [0, 4, 0, -5, 0, 1]], dtype=torch.float64),
        torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                      [0, 0, 1]], dtype=torch.float64),
        torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)),
}


def conv_winograd_1d(x, w, m=2, split_ops=True):
    """1-D Winograd F(m,3) along the width, direct along the height; operands split AFTER the transforms (as the kernel stages them)."""
    b, c, h, wd = x.shape
    k = w.shape[0]
    dt = x.dtype
    bt, g, at = (t.to(dt) for t in W1D[m])
    n = m + 2
    tw = wd // m
    xp = F.pad(x, (1, 1, 1, 1))
    d = xp.unfold(3, n, m)                                        # [b,c,h+2,tw,n]
    v = torch.einsum("ij,bcyxj->bcyxi", bt, d)                    # B^T d per row
    u = torch.einsum("ij,kcrj->kcri", g, w)                       # G g per tap row  [k,c,3,n]
    ops = ((split(v), split(u)) if split_ops else None)
    mm = None
    for r in range(3):
        if split_ops:
            (vh, vl), (uh, ul) = ops
            t = (torch.einsum("bcyxi,kci->bkyxi", vh[:, :, r:r + h], uh[:, :, r]) + torch.einsum("bcyxi,kci->bkyxi", vh[:, :, r:r + h], ul[:, :, r])
                 + torch.einsum("bcyxi,kci->bkyxi", vl[:, :, r:r + h], uh[:, :, r]))
        else:
            t = torch.einsum("bcyxi,kci->bkyxi", v[:, :, r:r + h], u[:, :, r])
        mm = t if mm is None else mm + t
    y = torch.einsum("ij,bkyxj->bkyxi", at, mm)                   # [b,k,h,tw,m]
    return y.reshape(b, k, h, wd)


class Patch:
    def __init__(self, mode, which):
        self.mode, self.which, self.real = mode, which, _CONV

    def __call__(self, x, w, bias=None, stride=1, padding=0, *a, **kw):
        if (self.mode != "f64" and x.dtype == torch.float32 and w.shape[2:] == (3, 3) and stride == 1 and padding == 1 and bias is None
                and w.shape[1] >= 64):
            if self.mode == "winograd" and (self.which == 0 or w.shape[1] == self.which) and x.shape[2] % 2 == 0:
                return conv_winograd(x, w)
            if self.mode in ("wino1d_f2", "wino1d_f4") and (self.which == 0 or w.shape[1] == self.which) and x.shape[3] % 4 == 0:
                return conv_winograd_1d(x, w, 2 if self.mode == "wino1d_f2" else 4)
            return conv_direct_bf16x3(x, w)
        return self.real(x, w, bias, stride, padding, *a, **kw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", type=int, default=0, help="0: winograd on every stride-1 3x3 conv with Cin >= 64; N: only on Cin == N")
    ap.add_argument("--batch", type=int, default=2)
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    # single layer, dominant shape
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 512, 32, 32, generator=g)
    w = torch.randn(512, 512, 3, 3, generator=g) / (3 * 512 ** 0.5)
    ref = _CONV(x.double(), w.double(), padding=1)
    sc = float(ref.abs().max())
    for name, y in (("fp32 direct (ATen)", _CONV(x, w, padding=1)), ("direct bf16x3", conv_direct_bf16x3(x, w)),
                    ("winograd fp32 (no split)", conv_winograd(x, w, False)), ("winograd bf16x3", conv_winograd(x, w)),
                    ("1-D F(2,3) bf16x3 (conv_wino.hip)", conv_winograd_1d(x, w, 2)), ("1-D F(4,3) fp32 (no split)", conv_winograd_1d(x, w, 4, False)),
                    ("1-D F(4,3) bf16x3", conv_winograd_1d(x, w, 4))):
        e = (y.double() - ref).abs()
        print(f"layer 512->512@32^2  {name:34s} max-abs/scale {float(e.max()) / sc:.3e}  rms/scale {float(e.pow(2).mean().sqrt()) / sc:.3e}")
    # whole encoder
    sd = synth.synth_state_dict(256, 13)
    img = synth.synth_image(args.batch, 1024, tag="wino_img")
    mask = synth.onehot(synth.synth_labels_face(args.batch, 512, seed=5))
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        ref, _ = orc.get_style_vectors(sd64, img.double(), mask.double())
        sc = float(ref.abs().max())
        for mode in ("direct", "winograd", "wino1d_f2", "wino1d_f4"):
            orc.F.conv2d = Patch(mode, args.which)
            try:
                sv, _ = orc.get_style_vectors(sd, img, mask)
            finally:
                orc.F.conv2d = _CONV
            e = (sv.double() - ref).abs()
            print(f"encoder style vectors [B,12,1280], scale {sc:.3f}: {mode:9s} max-abs {float(e.max()):.3e}  rms {float(e.pow(2).mean().sqrt()):.3e}")
        sv32, _ = orc.get_style_vectors(sd, img, mask)
        e = (sv32.double() - ref).abs()
        print(f"encoder style vectors: plain fp32 ATen          max-abs {float(e.max()):.3e}  rms {float(e.pow(2).mean().sqrt()):.3e}")


if __name__ == "__main__":
    main()
