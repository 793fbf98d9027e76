"""tedm_amd -- MI355X-native implementation of TEDM's diffusion-UNet denoising hot path.

Host side (this package): nn.Module facades with the reference's surface
(models/unet_model.py `Unet`, models/diffusion_model.py `DiffusionModel`,
models/datasetDM_model.py `DatasetDM`).  Compute: hand-written HIP kernels for gfx950 in
tedm_amd/csrc behind the C ABI of include/tedm_hip.h (libtedm_hip.so, loaded with ctypes).

Also here: `trainer` (the reference training loop / checkpoint format over the native train step), `dp` (one process per
GPU: sharding, flat-gradient all-reduce), `head` (pixel-classifier first layer fused into the decoder taps), `tedm` (the
shared-weights classifier over timesteps, its training loop, mean-over-steps ensembling, Dice / precision / recall).
"""
from .unet import Unet                      # noqa: F401
from .diffusion import DiffusionModel       # noqa: F401
from .dataset_dm import DatasetDM           # noqa: F401
from .weights import load_synth_weights, synth_state_dict   # noqa: F401



def set_tuning(key: str, value: int) -> None:
    """Library-wide tuning knob (include/tedm_hip_ops.h: tedm_set_tuning), e.g. `set_tuning('deep_fp16', 0)` keeps the fp4 cross
    terms on every convolution of the f16f4 mode.  Keys that change how weights are PACKED (deep_fp16) take effect at the next
    pack: call `Unet.invalidate_packed()` on live modules."""
    from . import _lib
    _lib.check(_lib.lib().tedm_set_tuning(key.encode(), int(value)), f'tedm_set_tuning({key!r}, {value})')


__all__ = ['Unet', 'DiffusionModel', 'DatasetDM', 'load_synth_weights', 'synth_state_dict', 'set_tuning']
