"""Bisect harness (round 5): TrainIteration.graphed_g_step vs the eager loop with a trainable generator, per kind of generator parameter.
For each variant: 2 eager warm-up steps + 2 replays vs 4 eager steps at 256^2, l2 loss, batch 2; prints per variant the first tensors
that differ after the warm-ups (must be none) and after each replay."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("E4S_ALLOW_UNINITIALIZED_LOSS_NETS", "1")
import torch  # noqa: E402

from e4s_amd import kernels as K, synth  # noqa: E402
from e4s_amd.networks import Net3  # noqa: E402
from e4s_amd.optim import FusedAdam  # noqa: E402
from e4s_amd.options import make_opts  # noqa: E402
from e4s_amd.train import LossOpts, TrainIteration  # noqa: E402

dev = torch.device("cuda", 0)
size, b = 256, 2
K.PRECISION = os.environ.get("DBG_PRECISION", "f32")
tmpl = Net3(make_opts(out_size=size, train_G=True))
tmpl.load_state_dict(synth.synth_state_dict(size, 13), strict=True)
tmpl.latent_avg = synth.synth_latent_avg(size).to(dev)
tmpl = tmpl.to(dev)
img = synth.synth_image(b, size, seed=9, tag="w1_img").to(dev)
mask = synth.onehot(synth.synth_labels_face(b, 512, seed=50)).to(dev)


def build(only):
    n3 = copy.deepcopy(tmpl).train()
    if only is not None:
        keep = only if callable(only) else (lambda name: any(s in name for s in only))
        for name, p in n3.named_parameters():
            if name.startswith("G.") and p.requires_grad and not keep(name):
                p.requires_grad = False
    if os.environ.get("DBG_FREEZE_ENCODER") == "1":          # far fewer graph nodes: does the failure follow the size of the graph?
        for name, p in n3.named_parameters():
            if name.startswith("encoder."):
                p.requires_grad = False
    params = [p for p in n3.parameters() if p.requires_grad]
    opt = FusedAdam(params, lr=1e-4, capturable=True)
    lo = LossOpts(face_parsing_lambda=0.0, id_lambda=0.0, lpips_lambda=0.0)
    n3.stash = {}
    orig = n3.forward

    def fwd(*a, **k):                       # keep the (static, when captured) output image of the last forward for inspection
        out = orig(*a, **k)
        n3.stash["recon"] = out[0].detach()
        return out
    n3.forward = fwd
    return TrainIteration(n3, None, {}, opt, None, lo=lo), n3


def diff(na, nb):
    out = []
    for (name, p), (_, q) in zip(na.named_parameters(), nb.named_parameters()):
        if not torch.equal(p, q):
            out.append((name, float((p - q).abs().max())))
    return out


def is_convw(name, lo=0, hi=99):
    if name == "G.conv1.conv.weight":
        return lo <= -1
    if name.startswith("G.convs.") and name.endswith(".conv.weight"):
        return lo <= int(name.split(".")[2]) <= hi
    return False


variants = [("none (G frozen by filter)", ["@@"]), ("modulation", ["modulation"]),
            ("convw_low conv.weight of conv1, convs.0, convs.1", lambda n: is_convw(n, -1, 1)),
            ("convw_all conv.weight of every styled conv", lambda n: is_convw(n, -1)),
            ("convw_2_7", lambda n: is_convw(n, 2, 7)), ("convw_up even convs (up-convs)", lambda n: is_convw(n, 0) and int(n.split(".")[2]) % 2 == 0),
            ("convw_plain odd convs", lambda n: is_convw(n, 0) and int(n.split(".")[2]) % 2 == 1),
            ("cwA conv1 + 2..7", lambda n: is_convw(n, 2, 7) or n == "G.conv1.conv.weight"),
            ("cwB convs.0 + 2..7", lambda n: is_convw(n, 2, 7) or is_convw(n, 0, 0)),
            ("cwC convs.1 + 2..7", lambda n: is_convw(n, 2, 7) or is_convw(n, 1, 1)),
            ("cwD -1..2", lambda n: is_convw(n, -1, 2)), ("cwE -1..3", lambda n: is_convw(n, -1, 3)), ("cwF -1..5", lambda n: is_convw(n, -1, 5)),
            ("cwG 0..7", lambda n: is_convw(n, 0, 7)), ("cwH -1,0,1,6,7", lambda n: is_convw(n, -1, 1) or is_convw(n, 6, 7)),
            ("cwI -1..1 + 2,3", lambda n: is_convw(n, -1, 3)),
            ("noise.weight", ["noise.weight"]), ("activate.bias", ["activate.bias"]), ("to_rgb", ["to_rgb"]), ("input", ["input.input"]),
            ("allbut_convw", lambda n: not is_convw(n, -1)), ("allbut_rgb", lambda n: "to_rgb" not in n),
            ("all", None)]
sel = os.environ.get("DBG_VARIANTS")
for vname, only in variants:
    if sel and vname.split()[0] not in sel.split(","):
        continue
    ite, ne = build(only)
    losses_e = []
    for _ in range(4):
        l, _ = ite.g_step(img, mask, randomize_noise=False)
        losses_e.append(float(l))
    itg, ng = build(only)
    dot = os.path.join("/tmp", "e4s_graph_%s.dot" % vname.split()[0])
    os.environ["E4S_GRAPH_DEBUG_DUMP"] = dot
    gs = itg.graphed_g_step(img, mask, warmup=2, randomize_noise=False)
    node_kinds = None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from graph_node_kinds import kinds
        node_kinds = kinds(dot)
        os.remove(dot)
    except Exception as e:      # noqa: BLE001
        node_kinds = "dump failed: %s" % e
    # state after the 2 warm-ups must equal the eager twin after 2 steps: rebuild an eager twin for that
    it2, n2 = build(only)
    for _ in range(2):
        it2.g_step(img, mask, randomize_noise=False)
    d_w = diff(n2, ng)
    l3 = float(gs.step())
    it2.g_step(img, mask, randomize_noise=False)
    rg, re_ = ng.stash["recon"], n2.stash["recon"]
    recon_info = {"graph_recon_finite": bool(torch.isfinite(rg).all()), "recon_maxdiff_vs_eager": float((rg - re_).abs().max()),
                  "l2_from_graph_recon": float(((rg - img) ** 2).mean()), "loss_tensor": l3}
    gd = []
    for (name, p), (_, q) in zip(ng.named_parameters(), n2.named_parameters()):
        if p.grad is not None and q.grad is not None and not torch.equal(p.grad, q.grad):
            gd.append((name, float((p.grad - q.grad).abs().max()), float(q.grad.abs().max())))
    recon_info["n_grads_differ"] = len(gd)
    recon_info["first_grads_differ"] = gd[:5] + gd[-3:]
    d_1 = diff(n2, ng)
    l4 = float(gs.step())
    d_2 = diff(ne, ng)
    torch.cuda.synchronize()
    print(json.dumps({"variant": vname, "replay1": recon_info, "trainable_G": sum(1 for n, p in ng.named_parameters() if n.startswith("G.") and p.requires_grad),
                      "eager_losses": losses_e, "graphed_losses_3_4": [l3, l4], "diff_after_warmups": d_w[:4], "n_after_warmups": len(d_w),
                      "diff_after_replay1": d_1[:6], "n_after_replay1": len(d_1), "diff_after_replay2": d_2[:6], "n_after_replay2": len(d_2)}), flush=True)
    del gs, itg, ng, ite, ne, it2, n2
