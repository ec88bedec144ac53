"""Build container only (needs /root/reference): time ONE 1024^2 E4S-core swap on the CPU with (a) the REAL reference modules
(oracle/ref_shim.build_reference_net3: src/models/networks.py:Net3 on the reference's own pure-PyTorch op fallbacks) and (b) the oracle
port that bench.py's cpu_baseline leg times on the GPU box (where /root/reference does not exist), on the same seeded inputs: the
port / reference ratio qualifies `cpu_baseline.kind = "port"`.  Writes profiles/r04_cpu_port_vs_reference.json."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from e4s_amd import synth  # noqa: E402
from oracle import e4s_oracle as orc, ref_shim  # noqa: E402

SIZE, KREM = 1024, 13
torch.set_num_threads(os.cpu_count())
sd = synth.synth_state_dict(SIZE, KREM)
lat = synth.synth_latent_avg(SIZE)
drv = synth.synth_image(1, SIZE, seed=100, tag="bench_d")
tgt = synth.synth_image(1, SIZE, seed=100, tag="bench_t")
dm, tm, sm = (synth.onehot(synth.synth_labels_face(1, 512, seed=300 + i)) for i in (1, 2, 3))
noise = synth.synth_noise(SIZE, seed=100, batch=1)
net = ref_shim.build_reference_net3(sd, lat, SIZE, KREM)


def ref_swap():
    """scripts/face_swap.py:237-273 on the reference's own Net3."""
    d_sv, _ = net.get_style_vectors(drv, dm)
    t_sv, _ = net.get_style_vectors(tgt, tm)
    sv = orc.swap_style_vectors(t_sv, d_sv)
    codes = net.cal_style_codes(sv)
    img, _, _ = net.gen_img(torch.zeros(1, 512, 16, 16), codes, sm, randomize_noise=False, noise=noise)
    return img


def port_swap():
    return orc.face_swap_core(sd, drv, dm, tgt, tm, sm, lat, noise, SIZE, KREM)


res = {"threads": torch.get_num_threads(), "host": "build container (no GPU)"}
with torch.no_grad():
    outs = {}
    for name, fn in (("reference", ref_swap), ("port", port_swap)):
        fn()
        t0 = time.perf_counter()
        for _ in range(2):
            outs[name] = fn()
        res[name + "_s_per_swap"] = round((time.perf_counter() - t0) / 2, 2)
res["port_over_reference_time"] = round(res["port_s_per_swap"] / res["reference_s_per_swap"], 3)
res["max_abs_port_vs_reference"] = float((outs["port"] - outs["reference"]).abs().max())
print(json.dumps(res))
json.dump(res, open(os.path.join(ROOT, "profiles", "r04_cpu_port_vs_reference.json"), "w"), indent=1)
