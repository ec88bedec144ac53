"""GPU tests of the native optimisation-step pieces: transposed LocalMLP / style-prologue
contractions, weight-gradient outer products, fused Adam, and run-to-run bit-reproducibility of every gradient
(ordered split reductions instead of floating-point atomics)."""
import math

import pytest
import torch

from e4s_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def maxabs(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


@pytest.mark.parametrize("b,r,o,k", [(1, 12, 6656, 512), (3, 12, 512, 1280), (16, 1, 96, 64), (40, 2, 50, 260), (7, 1, 512, 512)])
def test_grouped_linear_t_and_outer_vs_fp64(b, r, o, k):
    from e4s_amd import kernels as K
    g = torch.Generator().manual_seed(5)
    x = torch.randn(b, r, o, generator=g)
    w = torch.randn(r, o, k, generator=g) / math.sqrt(o)
    h = torch.randn(b, r, k, generator=g)
    base, mul = torch.randn(b, r, k, generator=g), torch.randn(b, r, k, generator=g)
    want = torch.einsum("bro,rok->brk", x.double(), w.double())
    got = K.grouped_linear_t(x.to(DEV), w.to(DEV), 0.37)
    assert maxabs(got, want * 0.37) < 2e-5 * float(want.abs().max())
    got2 = K.grouped_linear_t(x.to(DEV), w.to(DEV), -1.5, base=base.to(DEV), mul=mul.to(DEV), ref=h.to(DEV), alpha=0.01)
    want2 = base.double() + mul.double() * (-1.5) * torch.where(h > 0, 1.0, 0.01).double() * want
    assert maxabs(got2, want2) < 2e-5 * float(want2.abs().max())
    assert torch.equal(got, K.grouped_linear_t(x.to(DEV), w.to(DEV), 0.37))          # ordered partial sums: reproducible
    dw = K.grouped_outer(x.to(DEV), h.to(DEV), 0.5)
    assert maxabs(dw, 0.5 * torch.einsum("bro,brk->rok", x.double(), h.double())) < 1e-5 * (1 + b)
    assert maxabs(K.batch_sum(x.to(DEV)), x.double().sum(0)) < 1e-5 * (1 + b)
    # forward kernel at any batch (one weight stream): y[b,r,k'] = x[b,r,:] . w2[r,k',:]
    if o % 4 == 0:                                                                    # (K > 4096 takes the long-K kernel)
        w2 = w.transpose(1, 2).contiguous()                                          # [r,k,o]
        y = K.grouped_linear(x.to(DEV), w2.to(DEV), None, None, 1.0)
        assert maxabs(y, want) < 2e-5 * float(want.abs().max())


def test_fused_adam_matches_torch_adam():
    from e4s_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(6)
    p0 = torch.randn(1, 12, 1280, generator=g)
    grads = [torch.randn(1, 12, 1280, generator=g) * (0.1 + i) for i in range(6)]
    for wd in (0.0, 0.01):
        a = p0.clone().to(DEV).requires_grad_(True)
        b = p0.clone().to(DEV).requires_grad_(True)
        oa = FusedAdam([a], lr=1e-2, weight_decay=wd)
        ob = torch.optim.Adam([b], lr=1e-2, weight_decay=wd)
        for gr in grads:
            a.grad = gr.to(DEV).clone()
            b.grad = gr.to(DEV).clone()
            oa.step()
            ob.step()
        assert maxabs(a, b) < 2e-6, maxabs(a, b)
        assert maxabs(oa.state[a]["exp_avg_sq"], ob.state[b]["exp_avg_sq"]) < 1e-6 * float(ob.state[b]["exp_avg_sq"].abs().max())
        assert float((a.detach().cpu() - p0).abs().max()) > 1e-2                      # it really moved


def test_multi_tensor_capturable_adam_and_ema_vs_torch():
    """FusedAdam(capturable=True): one step-advance launch per group + multi-tensor updates (48 tensors per launch, pointers in the
    kernel arguments) on 120 tensors of awkward sizes (1 element, not a multiple of 4, exactly / just over one 4096-element block, a
    view at an odd offset) == torch.optim.Adam; a parameter that gets no gradient in a step does not advance; the learning rate is
    read from device memory (a changed group['lr'] acts on the next step); state_dict round trip; the multi-tensor EMA ==
    `p.mul_(decay).add_(q, alpha=1-decay)` bit for bit."""
    import copy
    from e4s_amd import kernels as K
    from e4s_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(11)
    sizes = [1, 3, 4, 5, 255, 1023, 4095, 4096, 4097, 8192 + 7, 70001] + [17 + 13 * i for i in range(108)]
    vals = [torch.randn(n, generator=g).to(DEV) for n in sizes]
    odd = torch.zeros(8, device=DEV)
    odd[1:6] = vals[3]                              # parameter 3 lives 4 bytes into an allocation: the unaligned (scalar) path
    a = [v.clone().requires_grad_(True) for v in vals]
    a[3] = odd[1:6].detach().requires_grad_(True)
    assert a[3].data_ptr() % 16 == 4
    b = [v.clone().requires_grad_(True) for v in vals]
    oa = FusedAdam(a, lr=1e-2, weight_decay=0.01, capturable=True)
    ob = torch.optim.Adam(b, lr=1e-2, weight_decay=0.01)
    for step in range(5):
        if step == 3:
            oa.param_groups[0]["lr"] = ob.param_groups[0]["lr"] = 1e-3
        for i, (pa, pb) in enumerate(zip(a, b)):
            gr = torch.randn(pa.shape, generator=g).to(DEV) * (0.1 + step)
            if step == 2 and i == 7:
                pa.grad = pb.grad = None            # no gradient this step: skipped, its step count stays
                continue
            pa.grad, pb.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
    for i, (pa, pb) in enumerate(zip(a, b)):
        assert maxabs(pa, pb) < 3e-6, (i, sizes[i], maxabs(pa, pb))
        assert int(oa.state[pa]["step"].item()) == (4 if i == 7 else 5)
    flat = oa._dev[0]["flat"]
    assert flat.numel() == len(a) and oa.state[a[5]]["step"].data_ptr() == flat.data_ptr() + 8 * 5
    # state_dict round trip: a reloaded optimiser continues bit for bit
    a2 = [p.detach().clone().requires_grad_(True) for p in a]
    oc = FusedAdam(a2, lr=1e-3, weight_decay=0.01, capturable=True)
    oc.load_state_dict(copy.deepcopy(oa.state_dict()))
    for pa, pc in zip(a, a2):
        gr = torch.randn(pa.shape, generator=g).to(DEV)
        pa.grad, pc.grad = gr.clone(), gr.clone()
    oa.step()
    oc.step()
    for i, (pa, pc) in enumerate(zip(a, a2)):
        assert torch.equal(pa.detach(), pc.detach()), i
        assert int(oc.state[pc]["step"].item()) == (5 if i == 7 else 6)
    # multi-tensor EMA
    dst = [torch.randn(n, generator=g).to(DEV) for n in sizes]
    src = [torch.randn(n, generator=g).to(DEV) for n in sizes]
    want = [d.clone().mul_(0.999).add_(s_, alpha=1 - 0.999) for d, s_ in zip(dst, src)]
    v0 = dst[0]._version
    K.ema_multi_(dst, src, 0.999)
    assert dst[0]._version == v0 + 1
    for i, (d, w) in enumerate(zip(dst, want)):
        assert maxabs(d, w) < 1e-7 * (1 + float(w.abs().max())), i
        one = src[i].clone()
        ref1 = dst[i].clone()
        K.ema_(ref1, one, 0.5)
        many = [dst[i].clone()]
        K.ema_multi_(many, [one], 0.5)
        assert torch.equal(ref1, many[0])           # == the single-tensor kernel bit for bit


def test_graphed_step_reads_the_learning_rate_from_device_memory():
    """a captured step used to bake group['lr'] into its kernel arguments; coach.py:390-392 multiplies it by 0.1 at step
    100000.  Now the replayed Adam launches read it from a device double that GraphedStep.step() refreshes: graphed == eager across
    an lr change; changing betas under a captured step raises."""
    from e4s_amd.optim import FusedAdam, GraphedStep
    g = torch.Generator().manual_seed(3)
    w0 = torch.randn(300, generator=g)
    x = torch.randn(300, generator=g).to(DEV)

    def run_graphed():
        w = w0.clone().to(DEV).requires_grad_(True)
        opt = FusedAdam([w], lr=1e-2, capturable=True)

        def body():
            loss = ((w * x - 1.0) ** 2).sum()
            loss.backward()
            opt.step()
            return loss.detach()
        gs = GraphedStep(opt, body, warmup=2)         # 2 eager steps, then the captured one
        for i in range(6):
            if i == 3:
                opt.param_groups[0]["lr"] = 1e-3
            gs.step()
        return w.detach().clone(), opt, gs

    # eager twin: 2 warm-up steps at 1e-2, then 6 steps with the change before the 4th
    w = w0.clone().to(DEV).requires_grad_(True)
    opt = FusedAdam([w], lr=1e-2, capturable=True)
    for i in range(8):
        if i == 5:
            opt.param_groups[0]["lr"] = 1e-3
        opt.zero_grad(set_to_none=True)
        loss = ((w * x - 1.0) ** 2).sum()
        loss.backward()
        opt.step()
    wg, og, gs = run_graphed()
    assert torch.equal(wg, w.detach()), float((wg - w.detach()).abs().max())
    v = og.param_groups[0]["params"][0]._version
    gs.step()
    assert og.param_groups[0]["params"][0]._version == v + 1          # a replay advances the version of what it wrote
    og.param_groups[0]["betas"] = (0.5, 0.999)
    with pytest.raises(RuntimeError, match="re-capture"):
        gs.step()
    # a state reload after the capture swaps the step / exp_avg tensors the graph's kernel arguments point to -- refused, not
    # replayed on freed memory
    og.param_groups[0]["betas"] = (0.9, 0.999)
    gs.step()
    import copy
    og.load_state_dict(copy.deepcopy(og.state_dict()))
    with pytest.raises(RuntimeError, match="re-capture"):
        gs.step()
    # ... and the suggested re-capture WORKS with the same optimiser (the refusal used to be sticky inside FusedAdam: the new
    # GraphedStep's warm-up steps raised again, and so did plain eager stepping).  Eager twin: the same state, stepped eagerly.
    wg2 = og.param_groups[0]["params"][0]
    w2 = wg2.detach().clone().requires_grad_(True)
    opt2 = FusedAdam([w2], lr=og.param_groups[0]["lr"], capturable=True)
    sd2 = copy.deepcopy(og.state_dict())
    opt2.load_state_dict(sd2)

    def body2():
        loss = ((wg2 * x - 1.0) ** 2).sum()
        loss.backward()
        og.step()
        return loss.detach()
    gs2 = GraphedStep(og, body2, warmup=1)            # 1 eager step on the reloaded optimiser, then a fresh capture
    for _ in range(3):
        gs2.step()
    for _ in range(4):
        opt2.zero_grad(set_to_none=True)
        loss = ((w2 * x - 1.0) ** 2).sum()
        loss.backward()
        opt2.step()
    assert torch.equal(wg2.detach(), w2.detach()), float((wg2.detach() - w2.detach()).abs().max())
    with pytest.raises(RuntimeError, match="re-capture"):
        gs.step()                                     # the OLD graph stays refused


def test_optimisation_step_gradients_are_bit_reproducible():
    """Two identical fwd+bwd passes at 256^2 (masked + unmasked layers, fixed noise) give bitwise-identical latent
    gradients: ds / dd / ToRGB-weight reductions and the LocalMLP backward add their partial sums in a fixed order."""
    from e4s_amd import kernels as K
    from e4s_amd.networks import Net3
    from e4s_amd.options import make_opts
    saved = K.PRECISION
    K.PRECISION = "f32"
    try:
        net = Net3(make_opts(out_size=256))
        net.load_state_dict(synth.synth_state_dict(256, 13), strict=True)
        net.latent_avg = synth.synth_latent_avg(256).to(DEV)
        net = net.to(DEV).eval()
        for p in net.parameters():
            p.requires_grad = False
        target = synth.synth_image(2, 256, tag="det_t").to(DEV)
        mask = synth.onehot(torch.cat([synth.synth_labels_face(1, 512, seed=8), synth.synth_labels_blocks(1, 512, 32, seed=9)])).to(DEV)
        noise = [n.to(DEV) for n in synth.synth_noise(256, batch=2)]
        g = torch.Generator().manual_seed(7)
        sv = (torch.randn(2, 12, 1280, generator=g) * 0.2).to(DEV)
        grads = []
        for _ in range(2):
            latent = sv.clone().requires_grad_(True)
            img, _, _ = net.gen_img(None, net.cal_style_codes(latent), mask, noise=noise)
            torch.nn.functional.mse_loss(img, target).backward()
            grads.append(latent.grad.clone())
        assert bool(torch.isfinite(grads[0]).all()) and float(grads[0].abs().max()) > 0
        assert torch.equal(grads[0], grads[1])
    finally:
        K.PRECISION = saved


def test_capturable_adam_and_graphed_step_equal_eager():
    """FusedAdam(capturable=True) (step count on the device) tracks torch.optim.Adam, and a GraphedStep -- generator forward,
    LPIPS + identity + l2 losses, backward and the update captured in ONE HIP graph -- reproduces the eager loop bit for bit
    (fixed noise; every reduction on the path has a fixed order)."""
    import types
    from e4s_amd.criteria import IDLoss, LPIPS
    from e4s_amd.networks import Net3
    from e4s_amd.optim import FusedAdam, GraphedStep
    from e4s_amd.options import make_opts
    g = torch.Generator().manual_seed(6)
    p0 = torch.randn(1, 12, 1280, generator=g)
    grads = [torch.randn(1, 12, 1280, generator=g) * (0.1 + i) for i in range(5)]
    a = p0.clone().to(DEV).requires_grad_(True)
    b = p0.clone().to(DEV).requires_grad_(True)
    oa, ob = FusedAdam([a], lr=1e-2, capturable=True), torch.optim.Adam([b], lr=1e-2)
    for gr in grads:
        a.grad, b.grad = gr.to(DEV).clone(), gr.to(DEV).clone()
        oa.step()
        ob.step()
    assert maxabs(a, b) < 2e-6 and int(oa.state[a]["step"].item()) == 5
    # the step count is per parameter and travels with state_dict(): a reloaded optimiser continues at t = 6
    a2 = a.detach().clone().requires_grad_(True)
    oc = FusedAdam([a2], lr=1e-2, capturable=True)
    import copy
    oc.load_state_dict(copy.deepcopy(oa.state_dict()))     # (load_state_dict aliases same-device tensors of a live state dict)
    a.grad, a2.grad = grads[0].to(DEV).clone(), grads[0].to(DEV).clone()
    oa.step()
    oc.step()
    assert torch.equal(a.detach(), a2.detach()) and int(oc.state[a2]["step"].item()) == 6

    _graphed_equals_eager(256, (256, 128), steps=5)


def _graphed_equals_eager(size, lpips_sizes, steps, fork=False):
    """The script's default objective (optimization.py:88-122: l2 + 0.8 LPIPS x scales + 0.1 ID + 0.1 parsing) as ONE replayed
    HIP graph == the eager loop, bit for bit.  fork: the captured step runs the three loss networks on three forked streams
    (optim.forked_sum, as bench.py's config-3 leg does) -- and must still equal the ONE-stream eager loop bit for bit."""
    import types
    from e4s_amd.criteria import FaceParsingLoss, IDLoss, LPIPS
    from e4s_amd.networks import Net3
    from e4s_amd.optim import FusedAdam, GraphedStep
    from e4s_amd.options import make_opts
    g = torch.Generator().manual_seed(6)
    net = Net3(make_opts(out_size=size))
    net.load_state_dict(synth.synth_state_dict(size, 13), strict=True)
    net.latent_avg = synth.synth_latent_avg(size).to(DEV)
    net = net.to(DEV).eval()
    for p in net.parameters():
        p.requires_grad = False
    lp = LPIPS()
    lp.load_state_dict(synth.synth_module_state_dict(lp, 0, "lp."))
    idl = IDLoss(types.SimpleNamespace(id_loss_multiscale=True))
    idl.load_state_dict(synth.synth_module_state_dict(idl, 0, "id."))
    fpl = FaceParsingLoss(types.SimpleNamespace())
    fpl.load_state_dict(synth.synth_module_state_dict(fpl, 0, "fp."))
    lp, idl, fpl = lp.to(DEV).eval(), idl.to(DEV).eval(), fpl.to(DEV).eval()
    _, target = synth.synth_image_pair(1, size, seed=12)
    target = target.to(DEV)
    mask = synth.onehot(synth.synth_labels_face(1, 512, seed=6)).to(DEV)
    noise = [n.to(DEV) for n in synth.synth_noise(size)]
    sv = (torch.randn(1, 12, 1280, generator=g) * 0.1).to(DEV)

    from e4s_amd.train import mse_loss          # the l2 term as bench.py's captured config-3 step computes it (no ATen global reduce)

    from e4s_amd.optim import forked_sum

    def run(graphed):
        latent = sv.clone().requires_grad_(True)
        opt = FusedAdam([latent], lr=1e-2, capturable=True)

        def body():
            img, _, _ = net.gen_img(None, net.cal_style_codes(latent), mask, noise=noise)
            if fork and graphed:
                loss = forked_sum(mse_loss(img, target), [lambda: 0.8 * lp.forward_pooled(img, target, lpips_sizes),
                                                          lambda: 0.1 * idl(img, target)[0], lambda: 0.1 * fpl(img, target)[0]],
                                  inputs=(img, target))
            else:
                loss = mse_loss(img, target) + 0.8 * lp.forward_pooled(img, target, lpips_sizes) \
                    + 0.1 * idl(img, target)[0] + 0.1 * fpl(img, target)[0]
            loss.backward()
            opt.step()
            return loss.detach()
        losses = []
        if graphed:
            gs = GraphedStep(opt, body, warmup=2)
            for _ in range(steps - 2):
                losses.append(float(gs.step()))
            assert gs.steps_done == steps
            gs.validate()
        else:
            for _ in range(steps):
                opt.zero_grad(set_to_none=True)
                losses.append(float(body()))
        return latent.detach().clone(), losses
    lat_e, loss_e = run(False)
    lat_g, loss_g = run(True)
    assert torch.equal(lat_e, lat_g)
    assert loss_g == loss_e[2:]
    assert loss_e[-1] < loss_e[0]


@pytest.mark.parametrize("fork", [False, True])
def test_graphed_four_term_step_equals_eager_at_1024(fork):
    """the config-3 step bench.py times by default -- l2 + LPIPS(1024, 512, 256) + ID + parsing at 1024^2,
    captured (fork: with the three loss networks on three forked streams inside the capture, as bench.py runs it) -- against the
    one-stream eager loop, bitwise."""
    _graphed_equals_eager(1024, (1024, 512, 256), steps=6 if fork else 4, fork=fork)


def test_plan_path_replays_in_a_graph_with_other_masks_than_the_captured_one(monkeypatch):
    """the row-plan statement of region-select (StyledConv.forward(use_plan=True), and the automatic
    one-region plan of kernels.conv_mfma for per-sample scale rows on maps that are not a multiple of 128 rows) used to be
    suspect under HIP-graph replay.  Its tables are now initialised by a kernel (plan.hip used to hold the library's only
    hipMemsetAsync calls) and the consumer bound-checks what it reads.  Here: capture plan build + conv with mask A, replay
    with masks B, C (different region populations => different tile tables) several times: every replay == eager == oracle."""
    from e4s_amd import kernels as K
    from e4s_amd.stylegan2 import StyledConv
    monkeypatch.setattr(K, "PRECISION", "f32")
    torch.manual_seed(0)
    cin = cout = 512
    res = 16
    m = StyledConv(cin, cout, 3, 512, mask_op=True).to(DEV)
    with torch.no_grad():
        m.noise.weight.fill_(0.1)
        m.activate.bias.normal_(0, 0.1)
    g = torch.Generator().manual_seed(3)
    b = 2
    x = torch.randn(b, cin, res, res, generator=g).to(DEV)
    style = torch.randn(b, 12, 512, generator=g).to(DEV)
    noise = torch.randn(b, 1, res, res, generator=g).to(DEV)
    masks = [synth.onehot(synth.synth_labels_blocks(b, 512, cells, seed=sd)).to(DEV) for cells, sd in ((16, 7), (4, 8), (64, 9))]
    static_mask = masks[0].clone()
    with torch.no_grad():
        eager = [m(x, style, mk, noise=noise, use_plan=True) for mk in masks]
        sel = [m(x, style, mk, noise=noise, use_plan=False) for mk in masks]
    for e, s_ in zip(eager, sel):
        assert maxabs(e, s_) < 5e-5                                   # plan statement == in-GEMM region-select
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        m(x, style, static_mask, noise=noise, use_plan=True)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    flags = torch.zeros(1, device=DEV, dtype=torch.int32)
    with K.flag_sink(flags), torch.no_grad():
        with torch.cuda.graph(graph):
            out = m(x, style, static_mask, noise=noise, use_plan=True)
    for rep in range(4):
        for mk, e in zip(masks[::-1] if rep % 2 else masks, eager[::-1] if rep % 2 else eager):
            static_mask.copy_(mk)
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, e), f"plan-path graph replay {rep} differs from the eager launch"
    assert int(flags.item()) == 0


def test_native_mse_and_sum_all_vs_torch():
    """e4s_amd.train.mse_loss / kernels.sum_all (ordered native sums instead of ATen's global reduce, whose semaphore memset must stay out
    of captured steps): value and both gradients vs torch in fp64, bit-reproducible, odd sizes fall back to ATen."""
    from e4s_amd import kernels as K
    from e4s_amd.train import mse_loss
    g = torch.Generator().manual_seed(2)
    for shape in ((2, 3, 256, 256), (1, 3, 64, 64), (3, 5, 7)):
        a = torch.randn(shape, generator=g).to(DEV).requires_grad_(True)
        b = torch.randn(shape, generator=g).to(DEV).requires_grad_(True)
        l = mse_loss(a, b)
        (l * 3.0).backward()
        a64, b64 = a.detach().double().cpu().requires_grad_(True), b.detach().double().cpu().requires_grad_(True)
        l64 = torch.nn.functional.mse_loss(a64, b64)
        (l64 * 3.0).backward()
        assert abs(float(l) - float(l64)) < 2e-6 * float(l64)
        assert float((a.grad.cpu().double() - a64.grad).abs().max()) < 1e-6 * float(a64.grad.abs().max())
        assert float((b.grad.cpu().double() - b64.grad).abs().max()) < 1e-6 * float(b64.grad.abs().max())
        assert torch.equal(mse_loss(a.detach(), b.detach()), l.detach())
        x = a.detach()
        assert abs(float(K.sum_all(x)) - float(x.double().sum())) < 1e-3


@pytest.mark.parametrize("b,r", [(1, 12), (2, 12), (3, 5)])
def test_style_grad_multi_vs_fp64(b, r):
    """e4s_style_grad_multi_f32 -- the dL/ds -> dL/dlatent tail of every generator layer in two launches (the chain rule of
    model.py:242-320 through the demodulation coefficients and the modulation EqualLinear; ToRGB: ws = scale * w3 * s, model.py:422-440)
    -- against the per-layer fp64 statement: masked and unmasked StyledConv jobs, ToRGB jobs, two jobs sharing a latent slot, a slot no
    job writes (must come back zero), more than 16 rows (two row passes); bit-reproducible."""
    from e4s_amd import kernels as K
    g = torch.Generator().manual_seed(123)
    nl, sdim = 6, 512

    def rnd(*shape):
        return torch.randn(*shape, generator=g)

    specs = [("conv", True, 64, 128, 0), ("rgb", True, 128, 0, 2), ("conv", False, 128, 64, 2), ("conv", True, 512, 512, 3),
             ("rgb", False, 32, 0, 5), ("conv", True, 32, 32, 5)]
    jobs, ref = [], []
    for kind, masked, cin, cout, slot in specs:
        G = b * r if masked else b
        j = dict(G=G, Cin=cin, Cout=cout, slot=slot, masked=masked, wmod=rnd(cin, sdim) / 8, mod_scale=0.0442)
        if kind == "conv":
            j.update(ds_raw=rnd(G, cin), dd_d=rnd(G, cout), d=torch.rand(G, cout, generator=g) + 0.5, s=rnd(G, cin), wsq=torch.rand(cout, cin, generator=g))
        else:
            j.update(dws=rnd(G, 3, cin), w3=rnd(3, cin), conv_scale=0.177)
        ref.append(j)
        jobs.append({k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in j.items()})
    dlat, ds_totals = K.style_grad_multi(jobs, b, r, nl, sdim, DEV)
    want_lat = torch.zeros(b, r, nl, sdim, dtype=torch.float64)
    for j, got in zip(ref, ds_totals):
        f = {k: (v.double() if torch.is_tensor(v) else v) for k, v in j.items()}
        if "dws" in f:
            ds = f["conv_scale"] * (f["dws"] * f["w3"][None]).sum(1)
        else:
            ds = f["ds_raw"] - f["s"] * ((f["dd_d"] * f["d"] ** 2) @ f["wsq"])
        sc = float(ds.abs().max())
        assert float((got.cpu().double() - ds).abs().max()) < 2e-6 * sc, (j["Cin"], j["masked"])
        dstyle = f["mod_scale"] * ds @ f["wmod"]
        if j["masked"]:
            want_lat[:, :, j["slot"]] += dstyle.view(b, r, sdim)
        else:
            want_lat[:, 0, j["slot"]] += dstyle
    sc = float(want_lat.abs().max())
    assert float((dlat.cpu().double() - want_lat).abs().max()) < 2e-6 * sc
    assert float(dlat[:, :, 1].abs().max()) == 0.0 and float(dlat[:, :, 4].abs().max()) == 0.0      # slots nobody writes
    if r > 1:
        assert float(dlat[:, 1:, 2].abs().max()) > 0.0 and float(dlat[:, 1:, 5].abs().max()) > 0.0
    dlat2, ds2 = K.style_grad_multi(jobs, b, r, nl, sdim, DEV)
    assert torch.equal(dlat, dlat2) and all(torch.equal(a, c) for a, c in zip(ds_totals, ds2))
