"""The option fields Net3 reads (src/models/networks.py:44-82; flag classes in src/options/*.py)."""
import types


def make_opts(out_size=1024, remaining_layer_idx=13, num_seg_cls=12, train_G=False, start_from_latent_avg=True,
              learn_in_w=False, fsencoder_type="psp", **extra):
    return types.SimpleNamespace(fsencoder_type=fsencoder_type, remaining_layer_idx=remaining_layer_idx,
                                 num_seg_cls=num_seg_cls, out_size=out_size, train_G=train_G,
                                 start_from_latent_avg=start_from_latent_avg, learn_in_w=learn_in_w, **extra)
