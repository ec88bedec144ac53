"""Which launches make up the weight-gradient time of the config-5 generator step: one eager G step with e4s_amd.kernels.conv_wgrad wrapped
(shape, masked or not, HIP-event time per call).  Prints a table sorted by time and the masked / unmasked totals."""
import collections
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from e4s_amd import kernels as K, synth  # noqa: E402

dev = "cuda"
rec = []
orig = K.conv_wgrad


def wrapped(gz, x, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig(gz, x, **kw)
    e1.record()
    rec.append((tuple(gz.shape), tuple(x.shape), kw.get("ntaps", 9), kw.get("istride", 1), kw.get("ostride", 1), kw.get("labels") is not None,
                kw.get("s") is not None, e0, e1))
    return out


def main():
    from e4s_amd.optim import FusedAdam
    from e4s_amd.train import LossOpts, TrainIteration
    from e4s_amd import criteria
    from e4s_amd.criteria import FaceParsingLoss, IDLoss, LPIPS
    from e4s_amd.stylegan2 import Discriminator
    lat = synth.synth_latent_avg(bench.SIZE) if hasattr(synth, "synth_latent_avg") else None
    net = bench.Net3(bench.make_opts(out_size=bench.SIZE, train_G=True))
    sd = synth.synth_state_dict(bench.SIZE, bench.KREM)
    net.load_state_dict(sd, strict=True)
    net.latent_avg = (lat if lat is not None else torch.zeros(18, 512)).to(dev)
    net = net.to(dev).train()
    img = synth.synth_image(2, bench.SIZE, seed=7, tag="train_img").to(dev)
    mask = synth.onehot(synth.synth_labels_face(2, 512, seed=21)).to(dev)
    opt = FusedAdam([p for p in net.parameters() if p.requires_grad], lr=1e-4, capturable=True)
    criteria.ALLOW_UNINITIALIZED = True
    lp, idl, fpl = LPIPS(), IDLoss(types.SimpleNamespace(id_loss_multiscale=True)), FaceParsingLoss(types.SimpleNamespace())
    for m, tag in ((lp, "lp."), (idl, "id."), (fpl, "fp.")):
        m.load_state_dict(synth.synth_module_state_dict(m, 0, tag))
    crit = {"lpips": lp.to(dev).eval(), "id": idl.to(dev).eval(), "parsing": fpl.to(dev).eval()}
    disc = Discriminator(bench.SIZE)
    disc.load_state_dict(synth.synth_disc_state_dict(bench.SIZE), strict=True)
    disc = disc.to(dev).train()
    opt_d = FusedAdam(disc.parameters(), lr=1e-4, capturable=True)
    it = TrainIteration(net, disc, crit, opt, opt_d, lo=LossOpts(d_reg_every=16), net_ema=None)
    for i in range(3):
        if i == 2:
            K.conv_wgrad = wrapped
        it.forget_targets()
        it.g_step(img, mask)
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for gs, xs, nt, ist, ost, masked, hs, e0, e1 in rec:
        k = (gs, xs, nt, ist, ost, masked, hs)
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
    tot = {True: 0.0, False: 0.0}
    for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        tot[k[5]] += ms
        print(f"{ms:8.3f} ms  x{n:<3d} gz{k[0]} x{k[1]} taps{k[2]} is{k[3]} os{k[4]} masked={k[5]} s={k[6]}")
    print("masked total %.3f ms, unmasked total %.3f ms, calls %d" % (tot[True], tot[False], len(rec)))


if __name__ == "__main__":
    main()
