"""Eager D steps (and R1 steps with argv[1] == r1) of the config-5 iteration, for `rocprofv3 --kernel-trace -- python tools/dstep_only.py`."""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from e4s_amd import synth  # noqa: E402

dev = "cuda"
from e4s_amd import criteria  # noqa: E402
from e4s_amd.criteria import FaceParsingLoss, IDLoss, LPIPS  # noqa: E402
from e4s_amd.optim import FusedAdam  # noqa: E402
from e4s_amd.stylegan2 import Discriminator  # noqa: E402
from e4s_amd.train import LossOpts, TrainIteration  # noqa: E402

net = bench.Net3(bench.make_opts(out_size=bench.SIZE, train_G=True))
net.load_state_dict(synth.synth_state_dict(bench.SIZE, bench.KREM), strict=True)
net.latent_avg = synth.synth_latent_avg(bench.SIZE).to(dev)
net = net.to(dev).train()
img = synth.synth_image(2, bench.SIZE, seed=7, tag="train_img").to(dev)
mask = synth.onehot(synth.synth_labels_face(2, 512, seed=21)).to(dev)
opt = FusedAdam([p for p in net.parameters() if p.requires_grad], lr=1e-4, capturable=True)
criteria.ALLOW_UNINITIALIZED = True
crit = {}
disc = Discriminator(bench.SIZE)
disc.load_state_dict(synth.synth_disc_state_dict(bench.SIZE), strict=True)
disc = disc.to(dev).train()
opt_d = FusedAdam(disc.parameters(), lr=1e-4, capturable=True)
lo = LossOpts(d_reg_every=16)
lo.face_parsing_lambda = lo.id_lambda = lo.lpips_lambda = 0.0
it = TrainIteration(net, disc, crit, opt, opt_d, lo=lo, net_ema=None)
r1 = len(sys.argv) > 1 and sys.argv[1] == "r1"
for i in range(6):
    if r1:
        it.r1_step(img)
    else:
        it.d_step(img, mask)
torch.cuda.synchronize()
print("done")
